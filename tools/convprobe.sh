# per-launch durations of the conv probe cases (tools/gpu_pmc_probe.py conv): bash tools/convprobe.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/cvp; timeout 100 rocprofv3 --kernel-trace -d $R/gpurun_out/cvp -- python3 $R/tools/gpu_pmc_probe.py ${1:-conv} > /dev/null 2>&1
python3 $R/tools/rocprof_summary.py $(find $R/gpurun_out/cvp -name "*.db" | head -1) 1 | grep "conv_t" | cut -c1-120
