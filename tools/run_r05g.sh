#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dw or zpath or bf16_storage" > $O/pytest_kernels.log 2>&1; echo "kernels rc $?"; tail -12 $O/pytest_kernels.log
timeout 200 python tools/gpu_dw_probe.py > $O/dw_probe.log 2>&1; cat $O/dw_probe.log | grep -v amdgpu
