#!/bin/bash
# round 5: LayerNorm-on-load + bilinear-on-load forward fusions -- kernel checks, full suite, canary experiment (longer overlap), bench A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "ln_linear or conv_up2 or test_ln or resample" > $O/pytest_new.log 2>&1; echo "new checks rc $?"; tail -15 $O/pytest_new.log
timeout 400 python tools/gpu_x2_canary.py x2 20 > $O/canary_x2.log 2>&1; echo "x2 rc $?"; grep -v amdgpu.ids $O/canary_x2.log | tail -30
timeout 300 python tools/gpu_x2_canary.py product 20 > $O/canary_product.log 2>&1; echo "product rc $?"; grep -v amdgpu.ids $O/canary_product.log | tail -22
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
for cfg in "1 1" "0 1" "1 0" "0 0"; do set -- $cfg
  LMN_FUSE_LN=$1 LMN_FUSE_UP=$2 timeout 300 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_ln$1_up$2.json 2> $O/bench_ln$1_up$2.err
  echo "LN=$1 UP=$2: $(tail -1 $O/bench_ln$1_up$2.json | cut -c1-210)"
done
