"""Inference throughput of the eval path (running-stat BN folded into the conv epilogues) and of the deployed
(structural_reparam) model, eagerly and as one hipGraph replay: `python tools/gpu_infer_bench.py [batch]`, 352x352, fp32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net

torch.manual_seed(0)
m = LM_Net(3, 2).cuda().eval()
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(BS, 3, 352, 352, device="cuda")


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
m.enable_graphs()
ag = t(m)
with torch.no_grad():
    yg = m(x)
m.enable_graphs(False)
print("eval, one hipGraph replay per batch: %.2f ms/batch (%.0f img/s); logits vs eager rel %.2e" % (ag * 1e3, BS / ag, float((yg - y0).abs().max() / y0.abs().max())))
m.structural_reparam()
b = t(m)
with torch.no_grad():
    y1 = m(x)
m.enable_graphs()
bg = t(m)
with torch.no_grad():
    y1g = m(x)
print("deployed, graph replay: %.2f ms/batch (%.0f img/s); logits vs eager rel %.2e" % (bg * 1e3, BS / bg, float((y1g - y1).abs().max() / y1.abs().max())))
print("eval: %.2f ms/batch (%.0f img/s); deployed: %.2f ms/batch (%.0f img/s); deploy vs eval logits rel %.2e" % (
    a * 1e3, BS / a, b * 1e3, BS / b, float((y1 - y0).abs().max() / y0.abs().max())))
