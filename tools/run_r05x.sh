#!/bin/bash
# round 5: SE parameter gradients inside reparam_wfin, BN finalize / coefficient launches inside the skip fusers' tails; canary test
O=gpurun_out/r05x; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for cfg in "1 1" "0 0" "1 1" "0 0"; do
  set -- $cfg
  LMN_FUSE_SE_WFIN=$1 LMN_FUSE_BN_TAIL=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('SE_WFIN=$1 BN_TAIL=$2  %.3f ms  %.1f img/s  launches %s' % (d['ms_per_step'], d['value'], d['config'].get('kernel_launches_per_step')))
" | tee -a $O/ab.log
done
