cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 python3 -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q -k "dw" 2>&1 | tail -2
rm -rf $R/gpurun_out/dwp; timeout 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dwp -- python3 $R/tools/gpu_pmc_probe.py dw > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $(find $R/gpurun_out/dwp -name "*.db" | head -1) 1 | grep -E "dw_|name"
