#!/bin/bash
# machine LICM on / off: time AND dynamic VALU count of the plain sweep on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/licm; mkdir -p $O
cd $R
bash scripts/ab_libs.sh licm "libnanokappa_hip.so libnk_licm.so libnanokappa_hip.so libnk_licm.so" "c2" --sustained 0 --per-call 0 | grep -v "sweep:"
cd /tmp && export TMPDIR=/tmp
for lib in libnanokappa_hip.so libnk_licm.so; do
  NK_LIBNAME=$lib timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_WAIT_INST_ANY -d $O/$lib -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline --sustained 0 --per-call 0 > /dev/null 2> $O/$lib.log
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for g in glob.glob('$O/$lib/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(g)):
        if 'k_sweep' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$lib', {k: round(sum(v[-5:]) / len(v[-5:]) / 1e6, 2) for k, v in sorted(acc.items())})
PY
done
