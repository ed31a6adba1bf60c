#!/bin/bash
O=gpurun_out/r05v; mkdir -p $O
CANARY_REGS_ONLY=1 timeout 600 python tools/gpu_x2_canary_lds.py 30 > $O/canary_sel.log 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/canary_sel.log | tail -8
