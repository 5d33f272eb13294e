# Memory-pipeline counters of k_events on the 5000-triangle wire (config 4 at 1e7 particles): one --pmc pass per counter group.
#   gpurun -- 'bash scripts/pmc_events.sh'  ->  gpurun_out/pmc_events/*.csv + summary.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_events
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config c4 --particles 1e7 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline"
i=0
# (a pass that asks a block for more counters than it has aborts inside rocprofv3 -- "Request exceeds the capabilities of the
# hardware to collect" -- and the aborted process then hangs in the tool's signal handler: two counters per block and pass,
# and every pass under its own timeout)
for G in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
         "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $G -d $O/g$i -o p --output-format csv -- $B > /dev/null 2> $O/g$i.log
  echo "group $i rc $?" | tee -a $O/progress.txt
done
python3 - <<PY > $O/summary.txt
import csv, glob, collections
for g in sorted(glob.glob('$O/g*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(g)):
        k = r['Kernel_Name'].split('(')[0][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k in acc:
        if 'k_events' in k or 'k_sweep' in k:
            print(g.split('/')[-2], k, {c: v for c, v in acc[k].items()})
PY
cat $O/summary.txt
