#!/bin/bash
# rocprofv3 kernel statistics of one BASELINE configuration: gpurun -- 'bash scripts/kt_config.sh <tag> <config> [steps]'
tag=$1; c=$2; steps=${3:-20}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 500 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o kt --output-format csv -- python3 $R/bench.py --config $c --steps $steps --warmup 10 --repeats 3 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/${c}_bench_under_rocprof.json 2> $O/kt_$c.log
cp $O/kt_$c/kt_kernel_stats.csv $O/${c}_kernel_stats.csv
python3 - $O/${c}_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print('%-70s calls %5s  avg %10.1f us  %5s %%' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
