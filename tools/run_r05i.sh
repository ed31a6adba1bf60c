#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
for cfg in "1 1" "1 0" "0 0"; do set -- $cfg
  LMN_FUSE_LN=$1 LMN_FUSE_UP=$2 timeout 300 python bench.py --no-cpu-baseline --no-other-configs > $O/bench_ln$1_up$2.json 2> $O/bench_ln$1_up$2.err
  echo "LN=$1 UP=$2: $(tail -1 $O/bench_ln$1_up$2.json | cut -c80-200)"
done
