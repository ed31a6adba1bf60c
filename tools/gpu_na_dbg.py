"""Which operand moves?  The attention backward built with -DLMN_NA_DBG (make -C lm_net_amd/csrc nadbg: x2 convs + counters inside
na_bwd_fused_kernel's phase A: dO read twice, the v / k window in LDS compared with global memory, dp and dsum evaluated twice) beside
bf16 3x3 convs that issue v_mfma_f32_16x16x32_bf16, and alone.   python tools/gpu_na_dbg.py [reps]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["LMNET_HIP_LIB"] = os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_nadbg.so")
import torch  # noqa: E402
from lm_net_amd import hip  # noqa: E402
import test_na_stress_gpu as S  # noqa: E402
lib = hip.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
names = ["dO re-read differs", "v window != memory", "dp re-evaluation differs", "dsum rebuild differs", "k window != memory", "-", "-", "items"]
hip.set_deterministic(True)
for nside, label in ((12, "beside x2 bf16 convs"), (0, "alone")):
    lib.lmn_na_dbg(None, 1)
    bad = S._stress(torch.float32, "conv", torch.bfloat16, 8, 176, 24, reps, nside=nside)
    out = (C.c_uint * 8)()
    lib.lmn_na_dbg(out, 0)
    print("%s: %d of %d runs differ from the quiet re-run; counters: %s" % (label, len(bad), reps, ", ".join("%s %d" % (n, v) for n, v in zip(names, out) if n != "-")), flush=True)
hip.set_deterministic(False)
