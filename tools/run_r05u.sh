#!/bin/bash
O=gpurun_out/r05u; mkdir -p $O
timeout 600 python tools/gpu_x2_forensics.py > $O/forensics.log 2>&1; echo "rc $?"; grep -v amdgpu.ids $O/forensics.log | tail -50
