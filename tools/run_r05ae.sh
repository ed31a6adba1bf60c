#!/bin/bash
# round 5: tile height of the N-split 3x3 conv kernel against block-count quantisation (LMN_CONV_TH)
O=gpurun_out/r05ae; mkdir -p $O
for th in 8 7 6 5 4 3; do
  echo "== LMN_CONV_TH=$th" | tee -a $O/conv_bench.log
  LMN_CONV_TH=$th timeout 300 python tools/gpu_conv_bench.py 2>&1 | grep -v amdgpu.ids | grep "3x3\|sum" | grep -v "L3\|L4" | tee -a $O/conv_bench.log
done
