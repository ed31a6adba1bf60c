#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
timeout 200 python tools/gpu_na_dbg.py 20 2>&1 | grep -v amdgpu.ids | tee $O/na_dbg.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q > $O/pytest_kernels.log 2>&1; echo "kernels rc $?"; tail -12 $O/pytest_kernels.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_kernels_gpu.py > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
  LMN_FUSE_LN=$1 LMN_FUSE_UP=$2 timeout 300 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_ln$1_up$2.json 2> $O/bench_ln$1_up$2.err
  echo "LN=$1 UP=$2: $(tail -1 $O/bench_ln$1_up$2.json | cut -c80-230)"
done
timeout 200 python tools/gpu_dw_probe.py > $O/dw_probe.log 2>&1; tail -12 $O/dw_probe.log
