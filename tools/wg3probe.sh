cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python3 $R/tools/gpu_kernel_check.py conv_wgrad conv_large conv_bf16 2>&1 | tail -12
for m in f32 bf16; do
  LMN_WGRAD_V1=0 timeout 200 python3 $R/tools/gpu_wgrad3_bench.py $m 2>&1 | tail -30
  LMN_WGRAD_V1=1 timeout 200 python3 $R/tools/gpu_wgrad3_bench.py $m 2>&1 | tail -30
done
for m in f32 bf16s; do timeout 120 python3 $R/tools/gpu_wgrad_phases.py $m; done
