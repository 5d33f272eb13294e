# Memory-latency counters of k_sweep (developer probe): separate --pmc passes, kernel-trace only.
# Not run here: a pass with the TCC_EA0_*_LEVEL counters (TCC_EA0_RDREQ_LEVEL, _WRREQ_LEVEL: queue-depth accumulators of the
# L2's memory side).  Round 1 tried one such pass once; the call ended at gpurun's time limit with nothing under gpurun_out/ --
# no rocprofv3 log, no counter CSV (the box was killed before the merge), so the evidence ends there.  What is known: the
# counters are listed for gfx950 (rocprofv3 -L: gpurun_out/counters.txt:2357-2400), the TCC block has 4 slots per pass
# (MI355X_MICROARCH.md, PMC slots) and that pass asked for more than 4 TCC counters next to SQ ones, which makes rocprofv3
# replay the workload in several passes; bench.py re-enqueues persistent kernels back to back, and every other multi-pass
# request made since has been split by hand into passes of <= 4 TCC counters (scripts/profile_round.sh) and returned.  The
# hang was not reproduced on purpose: a hung profiling pass costs a GPU-box strike.  If those counters are needed: ONE
# counter per pass, `timeout -k 10 120` around the command, a 2-step bench (`--steps 2 --warmup 1 --repeats 1`).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 12 --warmup 4 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/lat1 -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $R/gpurun_out/lat2 -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum -d $R/gpurun_out/lat3 -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $R/gpurun_out/lat5 -o p --output-format csv -- $B > /dev/null 2>&1
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
