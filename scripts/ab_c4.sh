#!/bin/bash
# config 4 (wire) with several library builds on one box: ms per step and k_sweep + k_events.  gpurun -- 'bash scripts/ab_c4.sh tag lib...'
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
for lib in "$@"; do
  ( [[ "$lib" != "-" ]] && export NK_LIBNAME=$lib
    timeout -k 10 400 python3 $R/bench.py --config c4 --steps 20 --warmup 10 --repeats 3 --no-cpu-baseline > $O/$(echo $lib | tr '/' '_').json 2> $O/$(echo $lib | tr '/' '_').err )
  echo "$lib rc $?"
done
python3 - <<PY | tee $O/summary.txt
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(os.path.basename(f), 'no line'); continue
    print('%-36s ms/step %.4f  k_sweep + k_events %.4f' % (os.path.basename(f), j['ms_per_step'], j['roofline']['kernel_ms']))
PY
