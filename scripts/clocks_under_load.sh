#!/bin/bash
# GPU clocks and power while the bench runs: a sampler beside `bench.py --sustained` (one line per 0.5 s)
out=gpurun_out/clk; mkdir -p $out
( for i in $(seq 1 60); do echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'fclk|mclk|sclk|Power' | sed 's/GPU\[0\][^:]*: //' | tr '\n' ' ')"; sleep 0.5; done ) > $out/samples.log &
S=$!
NK_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline --per-call 0 --small 0 --sustained 30000 > $out/bench.json 2> $out/bench.err
kill $S 2>/dev/null
grep -h "store placement" $out/bench.err
python3 -c "
import json; j=json.load(open('$out/bench.json')); r=j['roofline']; print('ms/step %.4f sweep %.4f frac %.3f sustained %.4f'%(j['ms_per_step'], r['kernel_ms'], r['frac'], j['sustained']['ms_per_step']))"
sed -n '1,60p' $out/samples.log | awk 'NR%3==1' | cut -c1-200
