"""Phase clocks of dw_bwd_strip_kernel (debug build with -DLMN_DW_TIMING, `make -C lm_net_amd/csrc timing`): per block, in shader
cycles: staging (barrier, window loads -> LDS, barrier) against the five row steps of a batch."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LMNET_HIP_LIB"] = os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_timing.so")
import numpy as np, torch
from lm_net_amd import hip
B = 8
for H, E in ((352, 24), (176, 48), (88, 96), (44, 192)):
    x1 = torch.randn(B, H, H, E, device="cuda"); dpre = torch.randn_like(x1); dx1 = torch.empty_like(x1)
    w5, w3, wv, wh = (torch.randn(E, k, device="cuda") for k in (25, 9, 3, 3))
    cA = torch.rand(4, E, device="cuda")
    dws = [torch.zeros_like(w) for w in (w5, w3, wv, wh)]
    f = lambda: hip.dw_bwd(x1, dpre, dx1, w5, w3, wv, wh, cA, cA, cA, *dws)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    n = 4096
    buf = (C.c_ulonglong * (n * 4))()
    hip.load().lmn_dw_timing(buf, n * 4)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.float64)
    a = a[a[:, 3] > 0]
    ns = a[:, 3].mean()
    print("H=%3d E=%3d  %6.1f us | blocks (first 4096) %d, steps %.0f, life %.0f cyc = %.0f per step | staging %.0f per batch, row steps %.0f per step" % (
        H, E, e0.elapsed_time(e1) * 1e3, len(a), ns, a[:, 2].mean(), a[:, 2].mean() / ns, a[:, 0].mean() / (ns / 5), a[:, 1].mean() / ns))
