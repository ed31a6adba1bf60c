"""Stress of the attention backward beside x2 convs for ONE library build (LMNET_HIP_LIB), printing one line.
    LMNET_HIP_LIB=lm_net_amd/csrc/liblmnet_hip_x2_wz.so python tools/gpu_x2_variants.py [reps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
import torch  # noqa: E402
from lm_net_amd import hip  # noqa: E402
import test_na_stress_gpu as S  # noqa: E402
hip.load()
hip.set_deterministic(True)
res = []
for dt in (torch.bfloat16, torch.float32):
    for kind in ("conv", "wgrad"):
        bad = S._stress(dt, kind, torch.bfloat16, 8, 176, 24, reps, nside=12)
        res.append("%s/%s %d/%d" % (str(dt).split(".")[1][:4], kind, len(bad), reps))
print("%-34s %s" % (os.path.basename(hip.LIB_PATH), "   ".join(res)), flush=True)
