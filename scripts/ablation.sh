# k_sweep with parts switched off (developer probe; needs `make -C nanokappa_amd/csrc ablate`)
for m in 0 1 2 4 8 16 24 3 7 15 31; do
  NK_LIBNAME=libnanokappa_hip_ablate.so NK_DEBUG=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('NK_DEBUG=%2d k_sweep %.1f us ms_per_step %.3f live %d'%($m, r['kernel_ms']*1e3, j['ms_per_step'], j['config']['live_particles_end']))
    elif 'rror' in l: print('NK_DEBUG=$m', l.strip())
"
done
