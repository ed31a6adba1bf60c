#!/bin/bash
# round 5: 3x3 conv phase clocks against the number of co-resident blocks (LMN_CONV_MAXB)
O=gpurun_out/r05z; mkdir -p $O
for mb in 256 512 768 1024 1280 2560; do
  echo "== LMN_CONV_MAXB=$mb" | tee -a $O/phases.log
  PHASES_3X3=1 LMN_CONV_MAXB=$mb timeout 300 python tools/gpu_conv_phases.py 2>&1 | grep -v amdgpu.ids | tee -a $O/phases.log
done
