"""Two fresh runs of training steps at batch 8 / 352x352: how many gradient tensors differ bitwise, default mode vs deterministic mode;
   and what deterministic mode costs per step.   python tools/gpu_determinism_check.py [batch] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net, hip
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW
from tools.detweights import det_input, disc_labels, fill_module
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 352
x = det_input((B, 3, S, S), "detchk/x").cuda()
y = disc_labels(B, S, S).cuda()


def run(det, steps=2, plans=False):
    m = LM_Net(3, 2)
    fill_module(m, 41)
    m = m.cuda().train()
    m.deterministic = det
    if plans:
        m.enable_plans(direct_grads=True)
    crit = SegLoss(label_smoothing=1e-3).cuda()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
    for _ in range(steps):
        loss = crit(m(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        g = [p.grad.detach().clone() for p in m.parameters()]
        opt.step()
    torch.cuda.synchronize()
    return float(loss), g, m, crit, opt


for det in (False, True):
    a, b = run(det), run(det)
    nd = sum(0 if torch.equal(u, v) else 1 for u, v in zip(a[1], b[1]))
    worst = max(float((u - v).abs().max() / (u.abs().max() + 1e-30)) for u, v in zip(a[1], b[1]))
    print("deterministic=%s: loss %.9g vs %.9g, %d of %d gradient tensors differ bitwise (largest difference %.2e of the tensor's max)" % (
        det, a[0], b[0], nd, len(a[1]), worst), flush=True)
for det in (False, True):
    _, _, m, crit, opt = run(det, steps=6, plans=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        loss = crit(m(x), y); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    torch.cuda.synchronize()
    print("deterministic=%s: %.2f ms/step (recorded plans)" % (det, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
hip.set_deterministic(False)
