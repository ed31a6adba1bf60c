"""Deterministic-mode check of the data-parallel bucket hooks: gradients with / without hooks, host launches / recorded plans must be
bit-identical, and the reported buckets must tile the flat gradient buffer contiguously.   python tools/gpu_bucket_hook_check.py"""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from lm_net_amd import LM_Net, hip
from tools.detweights import det_input, fill_module
from tests.helpers import no_dropout
x = det_input((4, 3, 352, 352), "hk/x").cuda()
G = det_input((4, 2, 352, 352), "hk/G").cuda()
def run(hooks, plans, steps=5):
    m = LM_Net(3, 2); fill_module(m, 51); no_dropout(m); m = m.cuda().train()
    m.deterministic = True
    calls = []
    if hooks:
        m.grad_begin_hook = lambda flat: calls.append(("begin", flat.numel()))
        m.grad_ready_hook = lambda lo, hi, streams=(): calls.append((lo, hi, len(streams)))
        m.grad_finish_hook = lambda: calls.append(("finish",))
    if plans:
        m.enable_plans()
    for _ in range(steps):
        for p in m.parameters(): p.grad = None
        calls.clear()
        y = m(x); (y * G).sum().backward(); torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in m.parameters()], list(calls)
ref, _ = run(False, False)
for hooks, plans in ((True, False), (False, True), (True, True)):
    got, calls = run(hooks, plans)
    bad = sum(0 if torch.equal(u, v) else 1 for u, v in zip(ref, got))
    cov = [c for c in calls if isinstance(c[0], int)]
    contiguous = all(cov[i][1] == cov[i + 1][0] for i in range(len(cov) - 1)) and (not cov or (cov[0][0] == 0))
    print("hooks=%s plans=%s: %d of %d gradient tensors differ from host launches without hooks; %d buckets, contiguous=%s, end=%s" % (hooks, plans, bad, len(ref), len(cov), contiguous, cov[-1][1] if cov else None))
hip.set_deterministic(False)
