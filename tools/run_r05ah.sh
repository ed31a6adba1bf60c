#!/bin/bash
O=gpurun_out/r05ah; mkdir -p $O
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('run $i  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
