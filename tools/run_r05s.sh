#!/bin/bash
# round 5: the x2 corruption -- LDS-exchange / packed-fp32 canaries + more build variants of na.hip
O=gpurun_out/r05s; mkdir -p $O
timeout 600 python tools/gpu_x2_canary_lds.py 20 > $O/canary_lds.log 2>&1; echo "canary rc $?"; grep -v amdgpu.ids $O/canary_lds.log | tail -12
for v in x2_o2 x2_noslp x2_nounroll x2_ilp x2_o1; do
  [ -f lm_net_amd/csrc/liblmnet_hip_$v.so ] && LMNET_HIP_LIB=$PWD/lm_net_amd/csrc/liblmnet_hip_$v.so timeout 300 python tools/gpu_x2_variants.py 20 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/variants.log
done
