#!/bin/bash
# On the GPU box: `bash tools/collect_profiles.sh r01x` -> gpurun_out/r01x/{bench_n1.json,stats,fetch,write};
# back in the container: `python tools/refresh_profiles.py r01x` copies the summaries into profiles/.
D=${1:-r01x}
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/$D; mkdir -p $R/gpurun_out/$D
cd $R
timeout 400 python bench.py > $R/gpurun_out/$D/bench_n1.json 2> $R/gpurun_out/$D/bench_n1.err
tail -1 $R/gpurun_out/$D/bench_n1.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$D/stats -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$D/fetch -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$D/write -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
ls $R/gpurun_out/$D
