#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05an; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -1 $O/bench_n1.json | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
