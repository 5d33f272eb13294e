#!/bin/bash
# Small ensembles on the launch-per-step path: ms per step with and without the segments' mode records staged in LDS.
R=$GRAFT_REPO_ROOT; tag=$1; O=$R/gpurun_out/$tag; mkdir -p $O
for N in 100000 300000 1000000; do for v in lrec nolrec; do
  ( [[ $v == nolrec ]] && export NK_NO_LREC=1
    timeout -k 10 200 python3 $R/bench.py --particles $N --steps 200 --warmup 100 --repeats 3 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 --ramp 0 > $O/n${N}_$v.json 2> $O/n${N}_$v.err )
done; done
python3 - <<PY | tee $O/summary.txt
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print(os.path.basename(f), 'no line'); continue
    r = j['roofline']
    print('%-24s ms/step %.5f  phonon-steps/s %.3e  k_sweep %.5f  tail %.5f' % (os.path.basename(f), j['ms_per_step'], j['value'], r['kernel_ms'], r['reduce_update_ms']))
PY
