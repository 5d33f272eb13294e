# k_sweep time vs resident workgroups per CU (developer probe)
for m in 1 2 3 4; do
  NK_VERBOSE=1 NK_SWEEP_PER_CU=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('per_cu=%d k_sweep %.1f us ms_per_step %.3f'%($m, r['kernel_ms']*1e3, j['ms_per_step']))
    elif l.startswith('[nanokappa_hip]'): print(l.strip())
"
done
