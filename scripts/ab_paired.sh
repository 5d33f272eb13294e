#!/bin/bash
# A/B of two library builds on one box with the speed of the store's memory beside every run (pair runs of equal memory):
#   scripts/ab_paired.sh <tag> "<libA> <libB>" "<configs>" [rounds]
tag=$1; libs=$2; cfgs=$3; rounds=${4:-3}
out=gpurun_out/$tag; mkdir -p $out
for r in $(seq 1 $rounds); do for c in $cfgs; do for lib in $libs; do
  f=$out/${lib}_${c}_$r
  NK_LIBNAME=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 20 --config $c --sustained 0 --per-call 0 --small 0 > $f.json 2> $f.err || { echo "FAILED $lib $c"; tail -5 $f.err; exit 1; }
  python - "$f.json" "$lib" "$c" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('%-24s %-4s ms/step %.4f sweep %.4f tail %.4f  copy %4.0f GB/s'%(sys.argv[2],sys.argv[3],j['ms_per_step'],r['kernel_ms'],r['reduce_update_ms'],j['store_placement']['kept_copy_GBps']))
PY
done; done; done
