"""Run-to-run spread of the gradients of two identical models (float-atomic summation order), per parameter; optional plan mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import rel_err
from tools.detweights import det_input, fill_module
from lm_net_amd import LM_Net

def net(seed):
    m = LM_Net(3, 2); fill_module(m, seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    return m.cuda().train()

x = det_input((2, 3, 64, 96), "plan/x").cuda()
G = det_input((2, 2, 64, 96), "plan/G").cuda()
c, d = net(17), net(17)
if len(sys.argv) > 1 and sys.argv[1] == "plans":
    d.enable_plans()
worst = {}
for it in range(8):
    for m in (c, d):
        for p in m.parameters(): p.grad = None
        (m(x) * G).sum().backward()
    gmax = max(float(p.grad.abs().max()) for p in c.parameters())
    for (k, pc), (_, pd) in zip(c.named_parameters(), d.named_parameters()):
        e = rel_err(pd.grad, pc.grad)
        if float((pd.grad - pc.grad).abs().max()) < 1e-5 * gmax: continue
        worst[k] = max(worst.get(k, 0.0), e)
for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]:
    print("%-50s %.2e" % (k, v))
