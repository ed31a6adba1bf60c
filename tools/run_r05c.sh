#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
export LMNET_HIP_LIB=$R/lm_net_amd/csrc/liblmnet_hip_x2.so
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/bisect.log
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from lm_net_amd import hip
import test_na_stress_gpu as S
hip.load(); hip.set_deterministic(True)
bf = torch.bfloat16
for nside in (4, 6, 10):
    for gr in (False, True):
        for kind in ("conv", "wgrad"):
            bad = S._stress(bf, kind, bf, 8, 176, 24, 20, nside=nside, gpu_rand=gr)
            print("nside %d gpu_rand %s %s: %d of 20 differ" % (nside, gr, kind, len(bad)), flush=True)
hip.set_deterministic(False)
PY
for n in 4 6; do echo "stress2 side launches: default 6 (SIDE=conv)"; done
SIDE=conv SIDE_DT=bf16 timeout 100 python tools/gpu_na_stress2.py 20 bf16 2>&1 | tail -1 | tee -a $O/bisect.log
