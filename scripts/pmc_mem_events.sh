#!/bin/bash
# What k_events (config 4) waits for: texture-address / L1 (TCP) / texture-data busy and stall cycles beside the instruction counters.
#   gpurun -- 'bash scripts/pmc_mem_events.sh <tag> [config] [lib]'  ->  gpurun_out/<tag>/mem.txt      (two counters per block and pass)
R=$GRAFT_REPO_ROOT; tag=$1; c=${2:-c4}; lib=${3:-}
[[ -n "$lib" ]] && export NK_LIBNAME=$lib
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
P2="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum SQ_WAVE_CYCLES"
P3="TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum SQ_WAIT_INST_ANY"
P4="TCP_GATE_EN1_sum TCP_GATE_EN2_sum SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
P5="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT"
i=0
for G in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $G -d $O/g$i -o p --output-format csv -- python3 $R/bench.py --config $c --steps 6 --warmup 3 --repeats 1 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > /dev/null 2> $O/g$i.log
  echo "pass $i rc $?"
done
python3 - <<PY | tee $O/mem.txt
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
