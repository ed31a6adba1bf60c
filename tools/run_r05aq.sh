#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05aq; mkdir -p $O
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -1 $O/bench_n1.json | cut -c1-300
