"""dw_bwd_bn whole against its two halves (part 1: dx1 only, part 2: weight gradients only) at the four level shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip
from tools.gpu_microbench import timeit
dev = "cuda"; B = 8
for (H, E) in [(352, 24), (176, 48), (88, 96), (44, 192)]:
    x1 = torch.randn(B, H, H, E, device=dev); dpre = torch.randn_like(x1); dx1 = torch.empty_like(x1)
    w5, w3, wv, wh = (torch.randn(E, 1, a, b, device=dev) for a, b in ((5, 5), (3, 3), (3, 1), (1, 3)))
    bst = torch.randn(5, E, device=dev); mean = torch.randn(4, E, device=dev); rstd = torch.rand(4, E, device=dev) + 0.5; A = torch.rand(4, E, device=dev)
    dgs = [torch.zeros(E, device=dev) for _ in range(4)]; dbs = [torch.zeros(E, device=dev) for _ in range(4)]
    dws = [torch.zeros_like(w) for w in (w5, w3, wv, wh)]
    ts = []
    for part in (0, 1, 2):
        ts.append(timeit(lambda: hip.dw_bwd_bn(x1, dpre, dx1, w5, w3, wv, wh, bst, mean, rstd, A, B * H * H, True, dgs, dbs, *dws, part=part)) * 1e6)
    print("H=%3d E=%3d  whole %7.1f us   dx1 only %7.1f us   weight gradients only %7.1f us" % (H, E, *ts))
