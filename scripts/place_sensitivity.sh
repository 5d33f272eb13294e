#!/bin/bash
# How much does the sweep still depend on where the store's allocation lies?  NK_PLACE_TRIES=1: the store is timed
# (k_probe_place) but never moved.  Several processes = several placements.  gpurun -- 'bash scripts/place_sensitivity.sh tag [n] [env...]'
R=$GRAFT_REPO_ROOT; tag=$1; n=${2:-6}; shift 2
O=$R/gpurun_out/$tag; mkdir -p $O
for i in $(seq 1 $n); do
  ( export NK_PLACE_TRIES=1; for e in "$@"; do export $e; done
    timeout -k 5 200 python3 $R/bench.py --steps 40 --warmup 20 --repeats 3 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/run$i.json 2> $O/run$i.err )
  echo "run $i rc $?"
done
python3 - <<PY | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob('$O/run*.json')):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, e); continue
    r = j['roofline']
    print('%s  store copy %.0f GB/s  k_sweep %.4f ms  ms/step %.4f  tail %.4f' % (f.split('/')[-1], j['store_placement']['kept_copy_GBps'], r['kernel_ms'], j['ms_per_step'], r['reduce_update_ms']))
PY
