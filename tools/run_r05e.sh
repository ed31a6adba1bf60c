#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
export LMNET_HIP_LIB=$R/lm_net_amd/csrc/liblmnet_hip_x2.so
run() { echo "== $*"; env "$@" SIDE=conv SIDE_DT=bf16 timeout 100 python tools/gpu_na_stress2.py 30 f32 2>&1 | grep -v amdgpu.ids | tail -2; }
{ run A=0; run FRESH=1; run SYNC=1; run PRODUCER=copy; run PRODUCER=mul; run FRESH=1 SYNC=1 PRODUCER=copy; run LMN_NA_FUSED=0; run LMN_NA_FUSED=0 FRESH=1; } 2>&1 | tee $O/knobs.log
