#!/bin/bash
# round 5: LayerNorm backward in the epilogue of the data-gradient conv (LMN_EP_LN_BWD) -- parity, suite, step A/B
O=gpurun_out/r05ao; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "ln" > $O/pytest_ln.log 2>&1; echo "pytest ln rc $?"; tail -12 $O/pytest_ln.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for v in 1 0 1 0; do
  LMN_FUSE_LN_BWD=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('LMN_FUSE_LN_BWD=$v  %.3f ms  %.1f img/s  launches %s' % (d['ms_per_step'], d['value'], d['config'].get('kernel_launches_per_step')))
" | tee -a $O/ab.log
done
