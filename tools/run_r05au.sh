#!/bin/bash
# round 5: K chunks of 64 / 128 channels in the M-split 3x3 conv kernel on the small maps (LMN_CONVM_CKB3) -- parity, micro-benchmark, step A/B
O=gpurun_out/r05au; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > $O/pytest_conv.log 2>&1; echo "pytest conv rc $?"; tail -3 $O/pytest_conv.log
for c in 8 4 2; do
  echo "== LMN_CONVM_CKB3=$c" | tee -a $O/conv_bench.log
  LMN_CONVM_CKB3=$c timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "L2 3x3\|L3 3x3\|L4 3x3" | tee -a $O/conv_bench.log
done
for c in 8 2 8 2; do
  LMN_CONVM_CKB3=$c timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LMN_CONVM_CKB3=$c  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
