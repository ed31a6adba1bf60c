"""Replays the gradient-equality part of tests/test_model_gpu.py::test_plan_mode_matches_host_launches and lists every parameter
whose eager / plan gradients differ by more than float-atomic noise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import rel_err, no_dropout
from tools.detweights import det_input, disc_labels, fill_module
from lm_net_amd import LM_Net
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW

def net(seed):
    m = LM_Net(3, 2); fill_module(m, seed); no_dropout(m)
    return m.cuda()

x = det_input((2, 3, 64, 96), "plan/x").cuda()
y = disc_labels(2, 64, 96).cuda()
crit = SegLoss(label_smoothing=1e-3).cuda()
def run(m, n):
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
    for _ in range(n):
        loss = crit(m(x), y); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
a, b = net(13).train(), net(13).train()
b.enable_plans()
run(a, 7); run(b, 7)
G = det_input((2, 2, 64, 96), "plan/G").cuda()
mode = sys.argv[1] if len(sys.argv) > 1 else "cd"
c, d = net(17).train(), net(17).train()
if mode == "cd": d.enable_plans()          # eager vs plans
elif mode == "dd": c.enable_plans(); d.enable_plans()   # plans vs plans
# "cc": eager vs eager
nbad = 0
for it in range(12):
    for m in (c, d):
        for p in m.parameters(): p.grad = None
        (m(x) * G).sum().backward()
    if os.environ.get('RACE_SYNC', '0') == '1': torch.cuda.synchronize()
    gmax = max(float(p.grad.abs().max()) for p in c.parameters())
    for (k, pc), (_, pd) in zip(c.named_parameters(), d.named_parameters()):
        e = rel_err(pd.grad, pc.grad)
        if e >= 2e-4 and float((pd.grad - pc.grad).abs().max()) >= 1e-5 * gmax:
            nbad += 1
            dd = (pd.grad - pc.grad).abs().flatten()
            print("it %d %-46s rel %.2e  nbad_el %d/%d argmax %d" % (it, k, e, int((dd > 1e-4 * float(pc.grad.abs().max())).sum()), dd.numel(), int(dd.argmax())))
print("mode %s: offenders %d" % (mode, nbad))
