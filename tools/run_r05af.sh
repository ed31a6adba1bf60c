#!/bin/bash
# round 5: 256-pixel tiles (four pixel groups per wave) in the N-split 3x3 conv kernel -- parity, micro-benchmark, step A/B (LMN_CONV_NPG)
O=gpurun_out/r05af; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > $O/pytest_conv.log 2>&1; echo "pytest conv rc $?"; tail -3 $O/pytest_conv.log
for npg in 4 2; do
  echo "== LMN_CONV_NPG=$npg" | tee -a $O/conv_bench.log
  LMN_CONV_NPG=$npg timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "3x3\|sum" | grep -v "L3\|L4" | tee -a $O/conv_bench.log
done
for npg in 4 2 4 2; do
  LMN_CONV_NPG=$npg timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LMN_CONV_NPG=$npg  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
