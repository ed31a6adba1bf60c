#!/bin/bash
O=gpurun_out/r05aj; mkdir -p $O
bash tools/pmc_dw.sh "" fwd "dw_" > $O/pmc_dw_all.log 2>&1
cat $O/pmc_dw_all.log | cut -c1-420
python tools/gpu_dw_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/probe.log | tail -12
