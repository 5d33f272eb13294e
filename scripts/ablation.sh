for m in 0 1 2 4 8 16 3 7 31; do
  NK_DEBUG=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('NK_DEBUG=$m k_step %.1f us emit %.1f us ms_per_step %.3f'%(r['kernel_ms']*1e3, r['emit_kernel_ms']*1e3, j['ms_per_step']))
"
done
