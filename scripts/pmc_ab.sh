#!/bin/bash
# SQ counters of k_sweep / k_emit for several library builds on ONE box:
#   gpurun -- 'bash scripts/pmc_ab.sh <tag> "<libs>" "<configs>"'  ->  gpurun_out/<tag>/summary.txt
# One --pmc pass per counter group (SQ block: 8 slots), each under its own timeout; bench.py directly behind `--`.
R=$GRAFT_REPO_ROOT
tag=$1; libs=$2; cfgs=$3
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
G2="SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
for lib in $libs; do for c in $cfgs; do
  i=0
  for G in "$G1" "$G2"; do
    i=$((i+1))
    NK_LIBNAME=$lib timeout -k 5 200 rocprofv3 --kernel-trace --pmc $G -d $O/${lib}_${c}_g$i -o p --output-format csv -- python3 $R/bench.py --config $c --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline > /dev/null 2> $O/${lib}_${c}_g$i.log
    echo "$lib $c group $i rc $?" | tee -a $O/progress.txt
  done
done; done
python3 - <<PY > $O/summary.txt
import csv, glob, collections
rows = collections.defaultdict(dict)
for g in sorted(glob.glob('$O/*_g*/**/*counter_collection.csv', recursive=True)):
    tagname = g.split('$O/')[1].split('/')[0].rsplit('_g', 1)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(g)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:14]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        if k.startswith('k_sweep') or k.startswith('k_emit<'):
            for c, v in acc[k].items():
                rows[(tagname, k)][c] = sum(v[-5:]) / len(v[-5:])
for (t, k), cs in sorted(rows.items()):
    print(t, k, ' '.join('%s=%.4g' % (c.replace('SQ_', ''), v) for c, v in sorted(cs.items())))
PY
cat $O/summary.txt
