#!/bin/bash
# Config 4: residency scan of the split sweep, then the stamps build's per-wave clocks
out=gpurun_out/c4p; mkdir -p $out
for n in 4 3 2; do
  NK_SWEEP_PER_CU=$n NK_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 20 --config c4 --sustained 0 --per-call 0 --small 0 > $out/occ$n.json 2> $out/occ$n.err || exit 1
  python - $out/occ$n.json $n <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('per_cu %s  ms/step %.4f kernels %.4f emit %.4f'%(sys.argv[2],j['ms_per_step'],r['kernel_ms'],r['emit_count_kernel_ms']))
PY
done
NK_LIBNAME=libnanokappa_hip_stamps.so NK_STAMPS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 10 --repeats 1 --config c4 --sustained 0 --per-call 0 --small 0 > $out/stamps.json 2> $out/stamps.err
grep -i "stamps\|events" $out/stamps.err | tail -30
