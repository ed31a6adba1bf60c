#!/bin/bash
# round 5, first GPU call: the -m gpu suite after the translation-unit split, the x2 / canary experiment, a short bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout 400 python tools/gpu_x2_canary.py x2 30 > $O/canary_x2.log 2>&1; echo "x2 rc $?"; cat $O/canary_x2.log | tail -30
timeout 300 python tools/gpu_x2_canary.py product 30 > $O/canary_product.log 2>&1; echo "product rc $?"; cat $O/canary_product.log | tail -20
timeout 300 python bench.py --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300
