#!/bin/bash
# x2 library: the round-4 reproducer itself, one-pass and two-pass attention backward
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
export LMNET_HIP_LIB=$R/lm_net_amd/csrc/liblmnet_hip_x2.so
for fused in 1 0; do for side in wgrad conv; do for dt in bf16 f32; do
  echo "== LMN_NA_FUSED=$fused SIDE=$side SIDE_DT=bf16 na=$dt"
  LMN_NA_FUSED=$fused SIDE=$side SIDE_DT=bf16 TRUTH=1 timeout 120 python tools/gpu_na_stress2.py 30 $dt 2>&1 | grep -v amdgpu.ids | tail -4
done; done; done > $O/stress2_x2.log 2>&1
cat $O/stress2_x2.log
unset LMNET_HIP_LIB
echo "== product library, two-pass" >> $O/stress2_x2.log
LMN_NA_FUSED=0 SIDE=wgrad SIDE_DT=bf16 timeout 120 python tools/gpu_na_stress2.py 30 bf16 2>&1 | tail -2 | tee -a $O/stress2_x2.log
