#!/bin/bash
# round 5: distinct issue priorities for the co-resident waves of a SIMD in the 3x3 conv kernels (LMN_CONV_PRIO)
O=gpurun_out/r05y; mkdir -p $O
for p in 0 1 2 3 11; do
  echo "== LMN_CONV_PRIO=$p" | tee -a $O/conv_bench.log
  LMN_CONV_PRIO=$p timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "3x3\|sum" | tee -a $O/conv_bench.log
done
for p in 0 1 0 1 2; do
  LMN_CONV_PRIO=$p timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LMN_CONV_PRIO=$p  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
