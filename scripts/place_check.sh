#!/bin/bash
# The store-placement choice (nk_place_store) on one box: plain bench twice, under rocprofv3, and across re-allocations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/place; mkdir -p $O; cd $R
for i in 0 1 2 3; do
  NK_VERBOSE=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/plain_$i.json 2> $O/plain_$i.err || { echo FAILED; tail -5 $O/plain_$i.err; exit 1; }
  grep -h "store placement" $O/plain_$i.err
  python3 -c "
import json; j=json.load(open('$O/plain_$i.json')); r=j['roofline']; print('plain: ms/step %.4f sweep %.4f frac %.3f'%(j['ms_per_step'], r['kernel_ms'], r['frac']), j['store_placement'])"
done
NK_PLACE_TRIES=0 timeout -k 10 300 python3 bench.py --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/off.json 2> $O/off.err && python3 -c "
import json; j=json.load(open('$O/off.json')); r=j['roofline']; print('choice off: ms/step %.4f sweep %.4f frac %.3f'%(j['ms_per_step'], r['kernel_ms'], r['frac']))"
NK_VERBOSE=1 bash scripts/kt_config.sh place c2 50 | head -3
grep -h "store placement" $O/kt_c2.log
NK_VERBOSE=1 timeout -k 10 300 python3 scripts/placement_probe.py c2 6 > $O/probe.log 2>&1; grep -h "store placement\|allocation" $O/probe.log
