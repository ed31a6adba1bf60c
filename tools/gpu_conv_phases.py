"""Phase clocks of conv_tile_kernel (debug build with -DLMN_CT_TIMING, see lm_net_amd/csrc/Makefile `timing`), cold operands.
Per block, summed over its tiles, in shader-clock cycles: b1 = wait at the chunk-top barrier, st = staging (window loads -> LDS),
b2 = second barrier, mm = MFMA loop, ep = epilogue (+ stores)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LMNET_HIP_LIB"] = os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_timing.so")
import numpy as np, torch
from lm_net_amd import hip
B = 8
NSET = 6
SHAPES3 = [("3x3 12->12", 352, 12, 12, 3), ("3x3 24->12", 352, 24, 12, 3), ("3x3 24->24", 176, 24, 24, 3), ("3x3 48->24", 176, 48, 24, 3), ("3x3 48->48", 88, 48, 48, 3)]
for name, H, cin, cout, k in SHAPES3 if os.environ.get("PHASES_3X3") else [("1x1 12->24", 352, 12, 24, 1), ("L1 1x1 24->48", 176, 24, 48, 1), ("L1 1x1 48->24", 176, 48, 24, 1), ("1x1 24->12", 352, 24, 12, 1), ("3x3 12->12", 352, 12, 12, 3), ("3x3 24->12", 352, 24, 12, 3),
                              ("3x3 24->24", 176, 24, 24, 3), ("3x3 48->48", 88, 48, 48, 3)]:
    xs = [torch.randn(B, H, H, cin, device="cuda") for _ in range(NSET)]
    outs = [torch.empty(B, H, H, cout, device="cuda") for _ in range(NSET)]
    w = torch.randn(cout, cin, k, k, device="cuda"); wp = hip.conv_pack(w, k, [cin])
    def f(i): hip.conv_fwd([xs[i % NSET]], wp, outs[i % NSET], B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=cout, ksize=k)
    for i in range(8): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(9); e1.record(); torch.cuda.synchronize()
    n = 1280
    buf = (C.c_ulonglong * (n * 8))()
    hip.load().lmn_ct_timing(buf, n * 8)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
    a = a[a[:, 7] > 0]
    life = a[:, 5]
    print("%-12s %6.1f us | blocks %d, life avg %.0f max %.0f cyc | b1 %.0f  st %.0f  b2 %.0f  mm %.0f  ep %.0f  (other %.0f)" % (
        name, e0.elapsed_time(e1) * 1e3, len(a), life.mean(), life.max(), a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 3].mean(),
        a[:, 4].mean(), (life - a[:, :5].sum(1)).mean()))
    # the timing array keeps stale entries of earlier (larger) grids: clear it by relaunching is not possible, so mask by end time
