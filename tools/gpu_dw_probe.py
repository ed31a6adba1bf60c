"""The four depthwise passes alone on the GPU at the four level shapes of the batch-8 / 352x352 step (row-planar tensors,
three rotating sets).  Run it under rocprofv3 (--kernel-trace --stats, or --pmc ...) to read a pass's counters, or plain for times:
    python tools/gpu_dw_probe.py [level] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip

dev = torch.device("cuda", 0)
levels = [(8, 352, 352, 24), (8, 176, 176, 48), (8, 88, 88, 96), (8, 44, 44, 192)]
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
NS = 3


class BN:
    def __init__(self, E):
        self.weight, self.bias = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev) * 0.1
        self.running_mean, self.running_var = torch.zeros(E, device=dev), torch.ones(E, device=dev)
        self.eps, self.momentum = 1e-5, 0.1


def timed(fn):
    for i in range(3):
        fn(i % NS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % NS)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for li, (B, H, W, E) in enumerate(levels):
    if only >= 0 and li != only:
        continue
    mk = lambda: [hip.rp4(torch.randn(B, H, W, E, device=dev)) for _ in range(NS)]
    z, pre, u, dpre, dh = mk(), mk(), mk(), mk(), mk()
    ws = [torch.randn(E, 1, 5, 5, device=dev) * 0.2, torch.randn(E, 1, 3, 3, device=dev) * 0.3, torch.randn(E, 1, 3, 1, device=dev) * 0.5, torch.randn(E, 1, 1, 3, device=dev) * 0.5]
    A1, sh1 = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev) * 0.3
    zp = dict(A=A1, shift=sh1)
    st = torch.zeros(4, 2, E, device=dev)
    bns = [BN(E) for _ in range(4)]
    N = B * H * W
    mean, rstd, A = (torch.zeros(4, E, device=dev) for _ in range(3))
    gsum, sg, dm = torch.zeros(B, E, device=dev), torch.rand(B, E, device=dev), torch.randn(B, E, device=dev) * 0.01
    bst = torch.zeros(5, E, device=dev)
    hst = torch.zeros(2, E, device=dev)
    dgs, dbs = [torch.zeros(E, device=dev) for _ in range(4)], [torch.zeros(E, device=dev) for _ in range(4)]
    dws = [torch.zeros_like(w) for w in ws]
    hip.dw_stats(z[0], *ws, st, zpre=zp)
    passes = [
        ("stats0", 1, lambda i: hip.dw_stats(z[i], *ws, st, zpre=zp)),
        ("fwd", 2, lambda i: hip.dw_fwd_bn(z[i], pre[i], gsum, st, N, bns, ws, mean, rstd, A, zpre=zp)),
        ("stats1", 4, lambda i: hip.dw_bwd_stats(z[i], pre[i], u[i], sg, dm, dpre[i], *ws, bst, zpre=zp)),
        ("bwd", 3, lambda i: hip.dw_bwd_bn(z[i], dpre[i], dh[i], *ws, bst, mean, rstd, A, N, True, dgs, dbs, *dws, zpre=zp, hstats=hst)),
        ("bwd dx only", 3, lambda i: hip.dw_bwd_bn(z[i], dpre[i], dh[i], *ws, bst, mean, rstd, A, N, True, dgs, dbs, *dws, part=1, zpre=zp, hstats=hst)),
        ("bwd dW only", 2, lambda i: hip.dw_bwd_bn(z[i], dpre[i], dh[i], *ws, bst, mean, rstd, A, N, True, dgs, dbs, *dws, part=2)),
    ]
    print("level %d: %dx%d E=%d" % (li, H, W, E))
    for name, npass, fn in passes:
        us = timed(fn)
        print("  %-12s %8.1f us  %6.2f TB/s" % (name, us, npass * N * E * 4 / us / 1e6))
