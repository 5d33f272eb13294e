#!/bin/bash
# A/B of library builds on ONE box: gpurun -- 'bash scripts/ab_bench.sh <tag> <config> <lib1> <lib2> ...'  (lib = NK_LIBNAME, "-" = default;
# "name:ENV=VAL" adds an environment switch).  Every build runs twice, interleaved; gpurun_out/<tag>/summary.txt lists ms per step,
# k_sweep, tail and what the placement search kept.
R=$GRAFT_REPO_ROOT; tag=$1; cfg=$2; shift 2
O=$R/gpurun_out/$tag; mkdir -p $O
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
    name=$(echo "$spec" | tr ':=/' '___')
    ( [[ "$lib" != "-" ]] && export NK_LIBNAME=$lib; [[ -n "$envs" ]] && export ${envs//,/ }
      timeout -k 5 300 python3 $R/bench.py --config $cfg --steps 40 --warmup 20 --repeats 5 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/$name.$rep.json 2> $O/$name.$rep.err )
    echo "$spec rep $rep rc $?"
  done
done
python3 - <<PY | tee $O/summary.txt
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'no line', e); continue
    r = j['roofline']
    print('%-44s ms/step %.4f  sweep %.4f  tail %.4f  stream %.4f  place %.0f GB/s (%d tried)  live %d' % (os.path.basename(f), j['ms_per_step'], r['kernel_ms'], r['reduce_update_ms'], r['stream_ms_per_step'], j['store_placement']['kept_copy_GBps'], j['store_placement']['allocations_timed'], j['config']['live_particles_end']))
PY
