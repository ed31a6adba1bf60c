#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p $R/gpurun_out/r05p
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/r05p_pytest.log 2>&1; echo "pytest rc $?"; tail -4 $R/gpurun_out/r05p_pytest.log
bash tools/run_r05j.sh 2>&1 | tail -3
cp $R/gpurun_out/r05p_pytest.log $R/gpurun_out/r05p/pytest.log
