#!/bin/bash
# The mode map's two weights against the box sweep: NK_EVENT_WEIGHT (cost of a boundary event in particle-steps) x NK_AGE_SKEW
# (share of work by dispatch age).  gpurun -- 'bash scripts/map_scan.sh tag'
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/$tag; mkdir -p $O
for w in 0.0 0.2 0.45 0.7 1.0; do for k in 0.1 0.2 0.3; do
  ( export NK_EVENT_WEIGHT=$w NK_AGE_SKEW=$k
    timeout -k 5 200 python3 $R/bench.py --steps 40 --warmup 20 --repeats 3 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/w${w}_k$k.json 2> $O/w${w}_k$k.err )
done; done
python3 - <<PY | tee $O/summary.txt
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(os.path.basename(f), 'no line'); continue
    print('%-22s ms/step %.4f  k_sweep %.4f  store copy %.0f GB/s' % (os.path.basename(f), j['ms_per_step'], j['roofline']['kernel_ms'], j['store_placement']['kept_copy_GBps']))
PY
