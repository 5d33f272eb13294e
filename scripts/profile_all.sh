#!/bin/bash
# The round's profiles in one go:  gpurun --timeout 1200 -- 'bash scripts/profile_all.sh r03'
#   config 2: kernel trace + SQ / FETCH / WRITE passes + the default bench line (scripts/profile_round.sh)
#   configs 1b, 3, 4, 5: kernel trace (scripts/kt_config.sh) + their bench lines with the CPU leg
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
bash $R/scripts/profile_round.sh $TAG | tail -4
for c in c1b c3 c5 c4; do
  bash $R/scripts/kt_config.sh prof_$TAG $c 20 | head -6
  cd $R
  timeout -k 10 400 python3 bench.py --config $c > gpurun_out/prof_$TAG/${c}_bench.json 2> gpurun_out/prof_$TAG/${c}_bench.err
  python3 - gpurun_out/prof_$TAG/${c}_bench.json $c <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('%s: ms/step %.4f value %.3e sweep %.4f frac %.3f cpu %.3e'%(sys.argv[2],j['ms_per_step'],j['value'],r['kernel_ms'],r['frac'],j.get('cpu_baseline',{}).get('value',0)))
PY
done
