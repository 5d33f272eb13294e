#!/bin/bash
# The resident kernel of small ensembles: ms per step against the grid (workgroups) and the ensemble size, beside the
# launch-per-step path.  gpurun -- 'bash scripts/resident_scan.sh tag'
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/$tag; mkdir -p $O
run() { name=$1; shift; ( for e in "$@"; do [[ "$e" == *=* ]] && export $e; done
  timeout -k 10 200 python3 $R/bench.py --particles $N --mesh-n $MESH --steps 200 --warmup 100 --repeats 3 --no-cpu-baseline --sustained 0 --per-call 0 --ramp 0 > $O/$name.json 2> $O/$name.err ); echo "$name rc $?"; }
MESH=${2:-31}
for N in 100000 300000 1000000; do
  for g in 128 256 512; do run m${MESH}_n${N}_g$g NK_RESIDENT_GRID=$g NK_RESIDENT_MAX=100000000; done
  run m${MESH}_n${N}_launches NK_NO_RESIDENT=1
done
python3 - <<PY | tee $O/summary.txt
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'no line'); continue
    print('%-28s ms/step %.5f  phonon-steps/s %.3e  kernel %.5f' % (os.path.basename(f), j['ms_per_step'], j['value'], j['roofline']['kernel_ms']))
PY
