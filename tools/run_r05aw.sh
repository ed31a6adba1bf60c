#!/bin/bash
O=gpurun_out/r05aw; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > $O/pytest_conv.log 2>&1; echo "pytest conv rc $?"
LMN_CONVM_CKB3_UP=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > $O/pytest_conv_up.log 2>&1; echo "pytest conv (UP) rc $?"; tail -2 $O/pytest_conv_up.log
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*  %.3f ms  %.1f img/s' % (d['ms_per_step'], d['value']))
" | tee -a $O/ab.log; }
run X=0
run LMN_CONVM_CKB3_UP=1
run LMN_CONVM_CKB3_T3=900
run LMN_CONVM_CKB3_UP=1 LMN_CONVM_CKB3_T3=900
run LMN_CONVM_CKB3_T1=576 LMN_CONVM_CKB3_T2=1152
run X=0
run LMN_CONVM_CKB3_UP=1
run LMN_CONVM_CKB3_T3=900
