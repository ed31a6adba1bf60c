#!/bin/bash
# round 5: soak of the final build (default mode fp32 + bf16; deterministic fp32: two runs bit-identical?)
O=gpurun_out/r05as; mkdir -p $O
timeout 600 python tools/gpu_soak.py 1500 f32 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/soak_f32.log
timeout 600 python tools/gpu_soak.py 300 bf16 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/soak_bf16.log
LMN_DETERMINISTIC=1 timeout 600 python tools/gpu_soak.py 400 f32 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/soak_det.log
