#!/bin/bash
# round 5: M-split 3x3 tile height on the bottleneck's 372 -> 372 conv (LMN_CONVM_TH)
O=gpurun_out/r05at; mkdir -p $O
for th in 0 5 4 3 2 1; do
  echo "== LMN_CONVM_TH=$th" | tee -a $O/conv_bench.log
  LMN_CONVM_TH=$th timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "L4 3x3\|L3 3x3" | tee -a $O/conv_bench.log
done
