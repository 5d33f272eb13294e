#!/bin/bash
# A/B of library builds on ONE box: scripts/ab_libs.sh <tag> "<libs...>" "<configs...>" [bench flags]
# Each (lib, config) pair is one bench.py run; prints one summary line per run and keeps the JSON under gpurun_out/<tag>/.
set -o pipefail
tag=$1; libs=$2; cfgs=$3; shift 3
out=gpurun_out/$tag; mkdir -p $out
for lib in $libs; do for c in $cfgs; do
  n=$(ls $out 2>/dev/null | grep -c "^${lib}_${c}_") 
  f=$out/${lib}_${c}_$n
  NK_VERBOSE=1 NK_LIBNAME=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 20 --config $c "$@" > $f.json 2> $f.err || { echo "FAILED $lib $c"; tail -5 $f.err; exit 1; }
  python - "$f.json" "$lib" "$c" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print('%-26s %-4s ms/step %.4f [%.4f-%.4f] sweep %.4f emit %.4f red %.4f frac %.3f live %d'%(sys.argv[2],sys.argv[3],j['ms_per_step'],j['ms_per_step_min'],j['ms_per_step_max'],r['kernel_ms'],r['emit_count_kernel_ms'],r['reduce_update_ms'],r['frac'],j['config']['live_particles_end']))
PY
done; done
grep -h "sweep:" $out/*.err | sort | uniq -c
