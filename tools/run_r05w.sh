#!/bin/bash
O=gpurun_out/r05w; mkdir -p $O
echo "== x2 library, bf16 co-runner" > $O/canary_matrix.log
CANARY_REGS_ONLY=1 timeout 600 python tools/gpu_x2_canary_lds.py 30 2>&1 | grep -v amdgpu.ids >> $O/canary_matrix.log
echo "== PRODUCT library, bf16 co-runner (v_mfma_f32_16x16x16_bf16)" >> $O/canary_matrix.log
LMNET_HIP_LIB=$PWD/lm_net_amd/liblmnet_hip.so CANARY_REGS_ONLY=1 timeout 600 python tools/gpu_x2_canary_lds.py 30 2>&1 | grep -v amdgpu.ids >> $O/canary_matrix.log
echo "== PRODUCT library, fp32 co-runner (v_mfma_f32_16x16x4_f32)" >> $O/canary_matrix.log
LMNET_HIP_LIB=$PWD/lm_net_amd/liblmnet_hip.so CANARY_SIDE=f32 CANARY_REGS_ONLY=1 timeout 600 python tools/gpu_x2_canary_lds.py 30 2>&1 | grep -v amdgpu.ids >> $O/canary_matrix.log
cat $O/canary_matrix.log
