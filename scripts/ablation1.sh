# ablation at ONE workgroup per CU: k_sweep time ~ serial latency of a wave, so the parts add up (developer probe)
for m in ${MASKS:-0 1 4 5 24 25 26 28 30 31}; do
  NK_SWEEP_PER_CU=${PER_CU:-1} NK_LIBNAME=libnanokappa_hip_ablate.so NK_DEBUG=$m timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('NK_DEBUG=%3d k_sweep %.1f us'%($m, r['kernel_ms']*1e3))
    elif 'rror' in l: print('NK_DEBUG=$m', l.strip()[:100])
"
done
