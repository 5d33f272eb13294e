#!/bin/bash
# Dynamic instruction mix of k_sweep / k_tail (config given, default c2): what the VALU is busy with.
#   gpurun -- 'bash scripts/pmc_valu_mix.sh <tag> [config]'  ->  gpurun_out/<tag>/mix.txt
R=$GRAFT_REPO_ROOT; tag=$1; c=${2:-c2}
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
G2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_BRANCH"
G3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $G -d $O/g$i -o p --output-format csv -- python3 $R/bench.py --config $c --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > /dev/null 2> $O/g$i.log
  echo "group $i rc $?"
done
python3 - <<PY | tee $O/mix.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for g in sorted(glob.glob('$O/g*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(g)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:10]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    if not (k.startswith("k_sweep") or k.startswith("k_tail") or k.startswith("k_emit<") or k.startswith("k_events")): continue
    print(k)
    for cn in sorted(acc[k]):
        v = acc[k][cn][-5:]
        print('   %-28s %14.0f' % (cn, sum(v) / len(v)))
PY
