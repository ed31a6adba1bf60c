#!/bin/bash
# round 5: the x2 corruption -- CU-mask placement experiment + build variants of na.hip
O=gpurun_out/r05r; mkdir -p $O
timeout 600 python tools/gpu_x2_cumask.py 20 > $O/cumask.log 2>&1; echo "cumask rc $?"; grep -v amdgpu.ids $O/cumask.log | tail -40
for v in x2 x2_wz x2_o1 x2_membar x2_wait0; do
  LMNET_HIP_LIB=$PWD/lm_net_amd/csrc/liblmnet_hip_$v.so timeout 300 python tools/gpu_x2_variants.py 20 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/variants.log
done
