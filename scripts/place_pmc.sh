#!/bin/bash
# Counter passes over scripts/probes/place_pmc (the same copy kernel on the slowest / fastest / a middle allocation of one process):
#   gpurun -- 'bash scripts/place_pmc.sh tag'  ->  gpurun_out/<tag>/summary.txt
R=$GRAFT_REPO_ROOT; tag=$1
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
P2="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum"
P3="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum"
P4="GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o p --output-format csv -- $R/scripts/probes/place_pmc 520 40 > $O/p$i.txt 2> $O/p$i.log
  echo "pass $i rc $?"; tail -1 $O/p$i.txt
done
python3 - <<PY | tee $O/summary.txt
import csv, glob, collections
for g in sorted(glob.glob('$O/p*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(g)):
        k = r['Kernel_Name']
        if 'k_copy<' not in k: continue
        tag = k.split('k_copy<')[1][0]
        if tag == '3': continue
        acc[r['Counter_Name']][tag].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    kt = g.replace('counter_collection', 'kernel_trace')
    try:
        for r in csv.DictReader(open(kt)):
            if 'k_copy<' in r['Kernel_Name']:
                tag = r['Kernel_Name'].split('k_copy<')[1][0]
                dur[tag].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    except Exception as e:
        pass
    print(g.split('/')[-3], ' kernel us (under the counters): slow %.1f  middle %.1f  fast %.1f' % tuple(sum(dur[t][-4:]) / max(len(dur[t][-4:]), 1) for t in '021'))
    for c in sorted(acc):
        v = {t: sum(acc[c][t][-4:]) / max(len(acc[c][t][-4:]), 1) for t in '021'}
        print('   %-48s slow %16.0f   middle %16.0f   fast %16.0f   slow/fast %.3f' % (c, v['0'], v['2'], v['1'], v['0'] / v['1'] if v['1'] else 0))
PY
