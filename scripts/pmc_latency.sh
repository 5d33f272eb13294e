# Memory-latency counters of k_sweep (developer probe): separate --pmc passes, kernel-trace only.
# Not run here: a pass with the TCC_EA0_*_LEVEL counters (queue-depth accumulators of the L2's memory side).  Round 1 tried one
# such pass once and the call ended at gpurun's time limit with nothing under gpurun_out/.  Round 2 met the same ending with
# evidence (gpurun_out/pmc_events/g1.log, scripts/pmc_events.sh): a pass that asks ONE block for more counters than it has
# (there: four TA counters) fails at once in rocprofiler_create_counter_config -- "Request exceeds the capabilities of the
# hardware to collect" --, rocprofv3 aborts, and the aborted process then sits in the tool's own signal handler
# ("rocprofv3 finalizing after signal 6...") until the box's silence limit kills it.  It is not a GPU hang and not specific
# to the _LEVEL counters; the round-1 pass asked the TCC block for more than its slots.  Rule: TA: two counters per pass; TCP takes at least four, SQ eight; per
# hardware block and pass (SQ takes more), and `timeout -k 5 150` around every rocprofv3 --pmc command.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/lat1 -o p --output-format csv -- $B > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $R/gpurun_out/lat2 -o p --output-format csv -- $B > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum -d $R/gpurun_out/lat3 -o p --output-format csv -- $B > /dev/null 2>&1
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $R/gpurun_out/lat5 -o p --output-format csv -- $B > /dev/null 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/lat?')):
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        if 'k_sweep' in k:
            print(d, {c: sum(v[-5:]) / len(v[-5:]) for c, v in cs.items()})
PY
