#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
echo "== weights of the small kernels in VGPRs (product build)"; timeout 200 python tools/gpu_dw_probe.py 2>&1 | grep -E "level|bwd  "
echo "== DPP, branches, fewer fences"; LMNET_HIP_LIB=$R/lm_net_amd/csrc/liblmnet_hip_ws.so timeout 200 python tools/gpu_dw_probe.py 2>&1 | grep -E "level|bwd  "
