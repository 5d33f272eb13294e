# short ablation runs (3 steps) for masks that change the population (developer probe)
for m in ${MASKS:-0 16}; do
  NK_SWEEP_PER_CU=${PER_CU:-1} NK_LIBNAME=libnanokappa_hip_ablate.so NK_DEBUG=$m timeout -k 10 200 python bench.py --steps 3 --warmup 0 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('NK_DEBUG=%3d k_sweep %.1f us live %d'%($m, r['kernel_ms']*1e3, j['config']['live_particles_end']))
    elif 'rror' in l: print('NK_DEBUG=$m', l.strip()[:100])
"
done
