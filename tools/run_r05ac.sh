#!/bin/bash
# round 5: window prefetch across the MFMA loop + epilogue in the NHWC 1x1 conv instances (-DLMN_CONV_PF=2)
O=gpurun_out/r05ac; mkdir -p $O
timeout 900 env LMNET_HIP_LIB=$PWD/lm_net_amd/csrc/liblmnet_hip_pf2.so python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > $O/pytest_conv.log 2>&1; echo "pytest conv rc $?"; tail -3 $O/pytest_conv.log
for lib in csrc/liblmnet_hip_pf2.so liblmnet_hip.so; do
  echo "== $lib" | tee -a $O/conv_bench.log
  LMNET_HIP_LIB=$PWD/lm_net_amd/$lib timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "1x1\|sum" | tee -a $O/conv_bench.log
done
for lib in csrc/liblmnet_hip_pf2.so liblmnet_hip.so csrc/liblmnet_hip_pf2.so liblmnet_hip.so; do
  LMNET_HIP_LIB=$PWD/lm_net_amd/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
