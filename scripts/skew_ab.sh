#!/bin/bash
# NK_AGE_SKEW scan on one box, the speed of the store's memory beside every run:  scripts/skew_ab.sh "<skews>" "<configs>" [rounds]
skews=$1; cfgs=$2; rounds=${3:-2}
out=gpurun_out/skew; mkdir -p $out
for r in $(seq 1 $rounds); do for c in $cfgs; do for s in $skews; do
  f=$out/s${s}_${c}_$r
  NK_AGE_SKEW=$s timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 20 --config $c --sustained 0 --per-call 0 --small 0 > $f.json 2> $f.err || { echo FAILED; tail -3 $f.err; exit 1; }
  python - "$f.json" "$s" "$c" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('skew %-5s %-4s ms/step %.4f sweep %.4f  copy %4.0f GB/s'%(sys.argv[2],sys.argv[3],j['ms_per_step'],r['kernel_ms'],j['store_placement']['kept_copy_GBps']))
PY
done; done; done
