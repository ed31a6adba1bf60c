#!/bin/bash
O=gpurun_out/r05ab; mkdir -p $O
for mb in 1024 1280; do
  echo "== PF=1 LMN_CONV_MAXB=$mb" | tee -a $O/phases.log
  PHASES_3X3=1 LMN_CONV_MAXB=$mb timeout 300 python tools/gpu_conv_phases.py 2>&1 | grep -v amdgpu.ids | tee -a $O/phases.log
done
