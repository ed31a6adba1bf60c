cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python3 $R/tools/gpu_kernel_check.py conv_wgrad conv_large conv_bf16 2>&1 | tail -3
for m in ${MODES:-f32 bf16}; do
  LMN_WGRAD_V1=0 timeout 200 python3 $R/tools/gpu_wgrad3_bench.py $m 2>&1 | tail -1
  LMN_WGRAD_V1=1 timeout 200 python3 $R/tools/gpu_wgrad3_bench.py $m 2>&1 | tail -${TAILN:-1}
done
