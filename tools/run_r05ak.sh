#!/bin/bash
# round 5: ordered asm FMAs in the forward-type depthwise passes (LMN_DWF_ASM) -- parity, probe, step A/B
O=gpurun_out/r05ak; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "dw or zpath or bf16_storage" > $O/pytest_dw.log 2>&1; echo "pytest dw rc $?"; tail -3 $O/pytest_dw.log
for lib in liblmnet_hip.so csrc/liblmnet_hip_asm0.so; do
  echo "== $lib" | tee -a $O/probe.log
  LMNET_HIP_LIB=$PWD/lm_net_amd/$lib timeout 300 python tools/gpu_dw_probe.py 2>&1 | grep -v amdgpu.ids | grep "level\|stats0\|fwd\|stats1" | tee -a $O/probe.log
done
for lib in liblmnet_hip.so csrc/liblmnet_hip_asm0.so liblmnet_hip.so csrc/liblmnet_hip_asm0.so; do
  LMNET_HIP_LIB=$PWD/lm_net_amd/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log
done
