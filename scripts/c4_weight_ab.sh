#!/bin/bash
# A/B on one box: config 4 with the mode map's event weight at its default and at 0 (NK_EVENT_WEIGHT), two runs each, interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c4ab; mkdir -p $O; cd $R
for i in 0 1; do for w in default 0; do
  if [ $w = default ]; then unset NK_EVENT_WEIGHT; else export NK_EVENT_WEIGHT=$w; fi
  timeout -k 10 300 python3 bench.py --config c4 --no-cpu-baseline --steps 20 --warmup 20 --sustained 0 --per-call 0 --small 0 > $O/w${w}_$i.json 2> $O/w${w}_$i.err || { echo FAILED; tail -5 $O/w${w}_$i.err; exit 1; }
  python3 - $O/w${w}_$i.json $w <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('weight %-8s ms/step %.4f kernels %.4f emit %.4f'%(sys.argv[2], j['ms_per_step'], r['kernel_ms'], r['emit_count_kernel_ms']), {k:v for k,v in r.items() if k.endswith('_ms')})
PY
done; done
