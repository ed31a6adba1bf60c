#!/bin/bash
O=gpurun_out/r05am; mkdir -p $O
timeout 300 python tools/gpu_power_step.py 2>&1 | grep -v amdgpu.ids | tee $O/power_step.log
LMN_DETERMINISTIC=1 timeout 300 python tools/gpu_power_step.py 2>&1 | grep -v amdgpu.ids | tee $O/power_step_det.log | head -4
