"""Phase clocks of the LDS-staged 3x3 weight-gradient kernel (debug build: make -C lm_net_amd/csrc timing).
Per block: s1 = wait at the tile-top barrier, s2 = commit (wait for the prefetched loads + LDS stores), s3 = second barrier +
issue of the next tile's loads, k = K loops (MFMA) + everything else; in shader cycles, averaged over blocks."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip
hip.LIB_PATH = os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_timing.so")
import numpy as np
B = 8
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
if mode != "f32": hip._MMA[0] = hip.BF16
dt = torch.bfloat16 if mode == "bf16s" else torch.float32
for H, cin, cout in [(352, 12, 12), (352, 24, 12), (176, 24, 24), (176, 48, 24), (88, 96, 48), (44, 192, 96)]:
    x = torch.randn(B, H, H, cin, device="cuda").to(dt); dy = torch.randn(B, H, H, cout, device="cuda").to(dt)
    dW = torch.zeros(cout, cin, 3, 3, device="cuda"); db = torch.zeros(cout, device="cuda")
    f = lambda: hip.conv_wgrad([x], dy, dW, db, B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=cout, ksize=3)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    n = 768
    buf = (C.c_ulonglong * (n * 6))()
    rc = hip.load().lmn_wg_timing(buf, n * 6)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 6).astype(np.float64)
    a = a[a[:, 5] > 0]
    span = a[:, 5].max() - a[:, 4].min()
    tot = a[:, 5] - a[:, 4]
    print("%dx%d %d->%d: %.1f us | kernel span %.0f cyc, block life avg %.0f max %.0f | s1 %.0f  s2 %.0f  s3 %.0f  k %.0f | start spread %.0f" % (
        H, H, cin, cout, e0.elapsed_time(e1) * 1e3, span, tot.mean(), tot.max(), a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(),
        a[:, 4].max() - a[:, 4].min()))
