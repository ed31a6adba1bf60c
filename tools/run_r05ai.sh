#!/bin/bash
# A/B: conv kernels before / after the pixel-group generalisation (same box, alternating)
O=gpurun_out/r05ai; mkdir -p $O
for lib in liblmnet_hip.so csrc/liblmnet_hip_old.so liblmnet_hip.so csrc/liblmnet_hip_old.so liblmnet_hip.so csrc/liblmnet_hip_old.so; do
  LMNET_HIP_LIB=$PWD/lm_net_amd/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
for lib in liblmnet_hip.so csrc/liblmnet_hip_old.so; do
  echo "== $lib" | tee -a $O/conv_bench.log
  LMNET_HIP_LIB=$PWD/lm_net_amd/$lib timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "3x3\|sum" | grep -v "L3\|L4" | tee -a $O/conv_bench.log
done
