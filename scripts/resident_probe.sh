#!/bin/bash
# The resident kernel at 1e5 / 1e6 particles with its clock marks (NK_VERBOSE), beside the launch-per-step path.  gpurun -- 'bash scripts/resident_probe.sh tag [mesh-n]'
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/$tag; mkdir -p $O; MESH=${2:-31}
run() { name=$1; shift; ( for e in "$@"; do [[ "$e" == *=* ]] && export $e; done
  timeout -k 10 200 python3 $R/bench.py --particles $N --mesh-n $MESH --steps 200 --warmup 100 --repeats 3 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 --ramp 0 > $O/$name.json 2> $O/$name.err ); echo "$name rc $?"; }
for N in 100000 1000000; do
  run n${N}_resident NK_RESIDENT=1 NK_RESIDENT_MAX=100000000 NK_VERBOSE=1
  grep "resident step" $O/n${N}_resident.err | tail -2
  run n${N}_launches NK_NO_RESIDENT=1
done
python3 - <<PY | tee $O/summary.txt
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), 'no line'); continue
    print('%-28s ms/step %.5f  phonon-steps/s %.3e  kernel %.5f' % (os.path.basename(f), j['ms_per_step'], j['value'], j['roofline']['kernel_ms']))
PY
