"""Inference throughput of the eval path (running-stat BN folded into the conv epilogues) and of the deployed
(structural_reparam) model: batch 8, 352x352, fp32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net

torch.manual_seed(0)
m = LM_Net(3, 2).cuda().eval()
x = torch.randn(8, 3, 352, 352, device="cuda")


def t(model, n=20):
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


a = t(m)
with torch.no_grad():
    y0 = m(x)
m.structural_reparam()
b = t(m)
with torch.no_grad():
    y1 = m(x)
print("eval: %.2f ms/batch (%.0f img/s); deployed: %.2f ms/batch (%.0f img/s); deploy vs eval logits rel %.2e" % (
    a * 1e3, 8 / a, b * 1e3, 8 / b, float((y1 - y0).abs().max() / y0.abs().max())))
