#!/bin/bash
# round 5 final: full GPU suite, profile collection (bench + rocprofv3 stats + PMC passes), timeline and serial tables
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p $R/gpurun_out/r05ad_t
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r05ad_pytest.log 2>&1; echo "pytest rc $?"; tail -3 $R/gpurun_out/r05ad_pytest.log
bash tools/collect_profiles.sh r05ad > $R/gpurun_out/r05ad_collect.log 2>&1; tail -3 $R/gpurun_out/r05ad_collect.log
cd $R
timeout 300 python tools/gpu_step_timeline.py --steps 3 > $R/gpurun_out/r05ad_t/step_timeline.txt 2>&1
timeout 300 python tools/gpu_prof_step.py --serial --top 200 > $R/gpurun_out/r05ad_t/serial_kernels.txt 2>&1
timeout 300 python tools/gpu_prof_step.py --serial --layers --top 80 > $R/gpurun_out/r05ad_t/serial_layers.txt 2>&1
head -12 $R/gpurun_out/r05ad_t/step_timeline.txt
