#!/bin/bash
# round 5: M-split 1x1 tile pixels (LMN_CONVM_TP) on the small-map layers
O=gpurun_out/r05ar; mkdir -p $O
for tp in 0 128 64 32; do
  echo "== LMN_CONVM_TP=$tp" | tee -a $O/conv_bench.log
  LMN_CONVM_TP=$tp timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "L2 1x1\|L3 1x1\|L4\|L3 3x3" | tee -a $O/conv_bench.log
done
