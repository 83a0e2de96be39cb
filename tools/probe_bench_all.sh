#!/bin/bash
# every model's default bench line (ms/step) into gpurun_out/bench_all.log
mkdir -p gpurun_out
: > gpurun_out/bench_all.log
for m in vrnn srnn lstm cwvae stcn wavenet; do
  timeout -k 10 200 python bench.py --model $m --steps 8 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$m', round(d['ms_per_step'], 2), 'ms', d['config'].get('workload'), 'roofline', d.get('roofline', {}).get('frac'))
except Exception as e:
    print('$m', 'FAILED', l[-300:])
" | tee -a gpurun_out/bench_all.log
done
