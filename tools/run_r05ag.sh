#!/bin/bash
O=gpurun_out/r05ag; mkdir -p $O
for npg in 4 2 4 2 4 2; do
  LMN_CONV_NPG=$npg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LMN_CONV_NPG=$npg  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
