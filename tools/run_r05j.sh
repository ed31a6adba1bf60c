#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tools/collect_profiles.sh r05p > $R/gpurun_out/r05p_collect.log 2>&1
O=$R/gpurun_out/r05p
timeout 200 python tools/gpu_step_timeline.py --steps 3 > $O/step_timeline.txt 2>&1
timeout 200 python tools/gpu_prof_step.py --serial --top 200 > $O/serial_kernels.txt 2>&1
timeout 200 python tools/gpu_prof_step.py --serial --layers --top 80 > $O/serial_layers.txt 2>&1
tail -1 $O/bench_n1.json | cut -c1-400; head -30 $O/serial_kernels.txt
