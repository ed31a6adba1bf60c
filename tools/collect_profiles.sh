#!/bin/bash
# On the GPU box: `bash tools/collect_profiles.sh r02a` -> gpurun_out/r02a/{bench_n1.json,stats,fetch,write,mfma};
# back in the container: `python tools/refresh_profiles.py r02a` copies the summaries into profiles/.
D=${1:-r02x}
shift
EXTRA="$@"
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/$D; mkdir -p $R/gpurun_out/$D
cd $R
timeout 900 python bench.py $EXTRA > $R/gpurun_out/$D/bench_n1.json 2> $R/gpurun_out/$D/bench_n1.err
tail -1 $R/gpurun_out/$D/bench_n1.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$D/stats -o r -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs $EXTRA > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$D/fetch -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $EXTRA > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/$D/write -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $EXTRA > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $R/gpurun_out/$D/mfma -o r -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $EXTRA > $R/gpurun_out/$D/mfma.log 2>&1
# keep the merge-back small: only the summaries
find $R/gpurun_out/$D -name "*.db" -delete
for d in fetch write mfma; do
  f=$R/gpurun_out/$D/$d/r_counter_collection.csv
  [ -f $f ] && python3 $R/tools/pmc_reduce.py $f > $R/gpurun_out/$D/$d.json && rm -f $f
  rm -f $R/gpurun_out/$D/$d/r_kernel_trace.csv
done
ls -la $R/gpurun_out/$D
