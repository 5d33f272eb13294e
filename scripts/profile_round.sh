# Round profile of the bench workload: kernel-trace stats + calibrated HBM traffic of k_sweep (separate --pmc passes).
#   gpurun -- 'bash scripts/profile_round.sh v4'   ->  gpurun_out/prof_<tag>/...  (copy the summaries into profiles/)
TAG=${1:-vX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# every rocprofv3 pass under its own timeout (an abort inside the tool otherwise holds the box until the silence limit); bench.py
# directly behind `--`; SQ takes 8 counters per pass, TCC: FETCH_SIZE and WRITE_SIZE each need a pass of their own
B="python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --calibrate --sustained 0 --per-call 0 --small 0"
T="timeout -k 5 240"
$T rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --sustained 0 --per-call 0 --small 0 > $O/bench_under_rocprof.json 2> $O/kt.log
echo "kernel trace done"
$T rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -o p --output-format csv -- $B > /dev/null 2> $O/sq.log
echo "sq pass done"
$T rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/sq2 -o p --output-format csv -- $B > /dev/null 2> $O/sq2.log
python3 - <<PY > $O/sq2.json
import sys
sys.path.insert(0, '$R/scripts')
import json, pmc_summary
try:
    print(json.dumps(pmc_summary.load('$O/sq2'), indent=1, sort_keys=True))
except Exception as e:
    print(json.dumps({'error': repr(e)}))
PY
echo "sq2 pass done"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fe -o p --output-format csv -- $B > /dev/null 2> $O/fe.log
echo "fetch pass done"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr -o p --output-format csv -- $B > /dev/null 2> $O/wr.log
echo "write pass done"
cd $R
CAL=$(grep -h "calibration:" $O/fe.log | tail -1 | sed 's/.*reads \([0-9]*\) B and writes \([0-9]*\) B.*/\1 \2/')
python3 scripts/pmc_summary.py $O/sq $O/fe $O/wr $CAL > $O/pmc.json && python3 -c "
import json; j=json.load(open('$O/pmc.json')); k=j['k_sweep']; print('k_sweep read %.3f GB write %.3f GB' % (k['read_bytes']/1e9, k['write_bytes']/1e9)); print(j['calibration'])"
python3 - $O/pmc.json $TAG > $O/pmc_traffic.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); k = j['k_sweep']; c = j['calibration']
print(json.dumps({'k_sweep_hbm_bytes_per_launch': k['read_bytes'] + k['write_bytes'], 'read_bytes': k['read_bytes'], 'write_bytes': k['write_bytes'],
                  'source': 'profiles/%s_c2_1e7_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 10 --warmup 5 (BASELINE config 2), '
                            'scaled by the k_cal_stream calibration (read factor %.4f, write factor %.4f); a constant of that profile run, not a measurement of this bench run'
                            % (sys.argv[2], c['read_factor'], c['write_factor']), 'calibration': c}, indent=1))
PY
cp $O/kt/kt_kernel_stats.csv $O/kernel_stats.csv
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json
