"""Graph-mode training (two hipGraph replays per step) against eager training on the same model / data."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW
from tools.detweights import det_input, disc_labels, fill_module

dev = torch.device("cuda", 0)


def make(drop):
    m = LM_Net(3, 2)
    fill_module(m, 11)
    if not drop:
        for mod in m.modules():
            if hasattr(mod, "p") and isinstance(getattr(mod, "p"), float):
                mod.p = 0.0
    return m.to(dev).train()


x = det_input((2, 3, 64, 96), "graph/x").to(dev)
y = disc_labels(2, 64, 96).to(dev)
crit = SegLoss(label_smoothing=1e-3).to(dev)


def run(m, n):
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
    losses = []
    for _ in range(n):
        out = m(x)
        loss = crit(out, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return losses


a, b = make(False), make(False)
b.enable_graphs()
la, lb = run(a, 6), run(b, 6)
print("eager ", ["%.6f" % v for v in la])
print("graph ", ["%.6f" % v for v in lb])
err = max(abs(u - v) / abs(u) for u, v in zip(la, lb))
pe = max(float((p1 - p2).abs().max() / (p1.abs().max() + 1e-12)) for p1, p2 in zip(a.parameters(), b.parameters()))
print("max rel loss diff %.2e, max rel param diff %.2e, graphs captured: %d" % (err, pe, sum(g.fwd is not None for g in b._graphs.values())))
rm = max(float((m1.running_mean - m2.running_mean).abs().max()) for m1, m2 in zip(
    [m for m in a.modules() if isinstance(m, torch.nn.BatchNorm2d)], [m for m in b.modules() if isinstance(m, torch.nn.BatchNorm2d)]))
print("max running_mean diff %.2e" % rm)
# dropout on: consecutive replays must draw different masks
c = make(True).enable_graphs()
outs = []
for i in range(5):
    o = c(x); (o.sum()).backward(); outs.append(o.detach().clone())
    for p in c.parameters(): p.grad = None
print("dropout: |out[3]-out[4]| = %.3e (replays draw new masks)" % float((outs[3] - outs[4]).abs().max()))
