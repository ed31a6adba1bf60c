"""Soak: N training steps (default mode: float atomics, fused SE gate, recorded plans, four streams) on a fixed batch; reports the loss
curve, non-finite values and, for two runs from the same weights, how far their losses drift apart.
    python tools/gpu_soak.py [steps] [f32|bf16] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_batch
from lm_net_amd import LM_Net
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW
from tools.detweights import fill_module

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dtype = "bf16" if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else "fp32"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
x, y = make_batch(B, 352, 352, dev, 1234)
curves = []
for run in range(2):
    m = LM_Net(3, 2)
    fill_module(m, 5)
    m = m.to(dev).train()
    m.compute_dtype = dtype
    m.enable_plans()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
    crit = SegLoss(label_smoothing=1e-3).to(dev)
    ls = []
    for s in range(steps):
        loss = crit(m(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if s % 25 == 0 or s == steps - 1:
            ls.append(float(loss.detach()))
    bad = sum(1 for p in m.parameters() if not torch.isfinite(p).all())
    print("%s run %d: loss %s ; parameters with non-finite values: %d" % (dtype, run, " ".join("%.4f" % v for v in ls[:4] + ls[-3:]), bad), flush=True)
    curves.append(ls)
d = max(abs(a - b) for a, b in zip(*curves))
print("%s: largest loss difference between the two runs over %d steps: %.3e (final %.5f / %.5f)" % (dtype, steps, d, curves[0][-1], curves[1][-1]))
