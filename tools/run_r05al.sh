#!/bin/bash
O=gpurun_out/r05al; mkdir -p $O
timeout 300 python tools/gpu_clock_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/clock.log
