#!/bin/bash
O=gpurun_out/r05t; mkdir -p $O
for v in x2_transnop x2_expnop x2; do
  LMNET_HIP_LIB=$PWD/lm_net_amd/csrc/liblmnet_hip_$v.so timeout 300 python tools/gpu_x2_variants.py 20 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/variants.log
done
