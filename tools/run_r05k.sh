#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_up2 or conv_wgrad" > $O/pytest_k.log 2>&1; echo "kernel checks rc $?"; tail -6 $O/pytest_k.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_bench_size_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
for v in 1 0 1 0; do
  LMN_FUSE_UP_WGRAD=$v timeout 300 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_uw$v.json 2> $O/bench_uw$v.err
  echo "UP_WGRAD=$v: $(tail -1 $O/bench_uw$v.json | cut -c80-170)"
done
