"""Phase clocks of conv_tileM_kernel (the M-split 1x1 / 3x3 kernel of the wide layers; debug build with -DLMN_CT_TIMING, see
lm_net_amd/csrc/Makefile `timing`), cold operands.  Per block, summed over its tiles and K chunks, in shader-clock cycles:
b1 = wait at the chunk-top barrier, st = staging (window loads -> LDS), b2 = second barrier, mm = MFMA loop, ep = epilogue."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LMNET_HIP_LIB"] = os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_timing.so")
import numpy as np, torch
from lm_net_amd import hip
B = 8
NSET = 6
for name, H, cin, cout, k in [("L2 1x1 48->96", 88, 48, 96, 1), ("L3 1x1 96->192", 44, 96, 192, 1), ("L3 1x1 192->96", 44, 192, 96, 1),
                              ("L4 1x1 372->1116", 22, 372, 1116, 1), ("L4 1x1 1116->372", 22, 1116, 372, 1), ("L4 3x3 372->372", 22, 372, 372, 3)]:
    xs = [torch.randn(B, H, H, cin, device="cuda") for _ in range(NSET)]
    outs = [torch.empty(B, H, H, cout, device="cuda") for _ in range(NSET)]
    w = torch.randn(cout, cin, k, k, device="cuda"); wp = hip.conv_pack(w, k, [cin])
    def f(i): hip.conv_fwd([xs[i % NSET]], wp, outs[i % NSET], B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=cout, ksize=k)
    n = 4096
    z = (C.c_ulonglong * (n * 8))()
    for i in range(8): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(9); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (n * 8))()
    hip.load().lmn_ct_timing(buf, n * 8)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
    a = a[a[:, 7] > 0]
    a = a[a[:, 7] >= a[:, 7].max() - 1e6]          # entries of this launch (earlier, larger grids leave stale rows behind)
    life = a[:, 5]
    span = a[:, 7].max() - a[:, 6].min()
    print("%-18s %6.1f us | blocks %d, span %.0f cyc, life avg %.0f max %.0f | b1 %.0f  st %.0f  b2 %.0f  mm %.0f  ep %.0f  (other %.0f)" % (
        name, e0.elapsed_time(e1) * 1e3, len(a), span, life.mean(), life.max(), a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(),
        a[:, 4].mean(), (life - a[:, :5].sum(1)).mean()))
