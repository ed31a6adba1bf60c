# per-stream timeline of a plan-mode training step: bash tools/timeline.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl3; timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/tl3 -- python3 $R/bench.py --steps 4 --warmup 5 --no-cpu-baseline --plans "$@" > /dev/null 2>&1
python3 $R/tools/rocprof_timeline.py $(find $R/gpurun_out/tl3 -name "*.db" | head -1)
find $R/gpurun_out/tl3 -name "*.db" -delete
