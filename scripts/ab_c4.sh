#!/bin/bash
# config 4 (wire) with several library builds on one box: ms per step and k_sweep + k_events.  gpurun -- 'bash scripts/ab_c4.sh tag lib[:ENV=VAL]...'
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  name=$(echo "$spec" | tr ':=/' '___')
  ( [[ "$lib" != "-" ]] && export NK_LIBNAME=$lib; [[ -n "$envs" ]] && export ${envs//,/ }
    NK_VERBOSE=1 timeout -k 10 400 python3 $R/bench.py --config c4 --steps 20 --warmup 10 --repeats 3 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/$name.json 2> $O/$name.err )
  echo "$spec rc $?"
done
grep -h "k_events:" $O/*.err | sort | uniq -c
python3 - <<PY | tee $O/summary.txt
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(os.path.basename(f), 'no line'); continue
    print('%-44s ms/step %.4f  k_sweep + k_events %.4f' % (os.path.basename(f), j['ms_per_step'], j['roofline']['kernel_ms']))
PY
