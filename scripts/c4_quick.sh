#!/bin/bash
# config 4, two bench runs of the current build (plus kernel stats under rocprofv3 with "kt" as first argument)
out=gpurun_out/c4q; mkdir -p $out
for i in 0 1; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 20 --config c4 --sustained 0 --per-call 0 --small 0 > $out/r$i.json 2> $out/r$i.err || { tail -5 $out/r$i.err; exit 1; }
  python - $out/r$i.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('ms/step %.4f kernels %.4f emit %.4f reduce %.4f'%(j['ms_per_step'],r['kernel_ms'],r['emit_count_kernel_ms'],r['reduce_update_ms']))
PY
done
if [ "$1" = kt ]; then bash scripts/kt_config.sh c4q c4 20 | head -8; fi
