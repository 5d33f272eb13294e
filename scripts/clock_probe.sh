#!/bin/bash
# In-kernel shader clock of the sweep (stamps build) at 1, 2, 3 workgroups per CU:  gpurun -- 'bash scripts/clock_probe.sh <tag> <config>'
tag=$1; c=$2
out=gpurun_out/$tag; mkdir -p $out
for n in 3 2 1; do
  NK_SWEEP_PER_CU=$n NK_LIBNAME=libnanokappa_hip_stamps.so NK_STAMPS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 200 --repeats 3 --config $c --sustained 0 --per-call 0 --small 0 > $out/clk$n.json 2> $out/clk$n.err
  echo "== per_cu $n"; grep stamps $out/clk$n.err | tail -2
  python - $out/clk$n.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('   sweep %.4f ms (stamps build)'%r['kernel_ms'])
PY
done
