#!/bin/bash
# L2 (TCC) view of k_events (config 4): requests, hits, misses, reads that leave the XCD.  gpurun -- 'bash scripts/pmc_l2_events.sh <tag> [config] [lib]'
R=$GRAFT_REPO_ROOT; tag=$1; c=${2:-c4}; lib=${3:-}
[[ -n "$lib" ]] && export NK_LIBNAME=$lib
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for G in "TCC_REQ_sum TCC_READ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $G -d $O/g$i -o p --output-format csv -- python3 $R/bench.py --config $c --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > /dev/null 2> $O/g$i.log
  echo "pass $i rc $?"
done
python3 - <<PY | tee $O/l2.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for g in sorted(glob.glob('$O/g*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(g)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:10]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if not (k.startswith("k_sweep") or k.startswith("k_events<")): continue
    print(k)
    for cn in sorted(acc[k]):
        v = acc[k][cn][-4:]
        print('   %-40s %16.0f' % (cn, sum(v) / len(v)))
PY
