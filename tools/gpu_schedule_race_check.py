"""Deterministic mode as a race detector: the same step on the four-stream schedule (branch chains, weight-gradient streams, late
weight-gradient issue) and launched serially on one stream must agree BIT FOR BIT -- the kernels and their grids are the same, only
the streams and the interleaving differ.  Any difference is a missing dependency.   python tools/gpu_schedule_race_check.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net, hip
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW
from tools.detweights import det_input, disc_labels, fill_module
from tests.helpers import no_dropout

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, S = int(os.environ.get('BATCH', '8')), int(os.environ.get('SIZE', '352'))
x = det_input((B, 3, S, S), "race/x").cuda()
y = disc_labels(B, S, S).cuda()


def run(cfg, steps=int(os.environ.get('STEPS', '2'))):
    m = LM_Net(3, 2)
    fill_module(m, 43)
    no_dropout(m)
    m = m.cuda().train()
    m.deterministic = True
    m.compute_dtype = os.environ.get('DTYPE', 'fp32')
    e = m._engine
    for k, v in cfg.items():
        setattr(e, k, v)
    crit = SegLoss(label_smoothing=1e-3).cuda()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
    for _ in range(steps):
        loss = crit(m(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        g = [p.grad.detach().clone() for p in m.parameters()]
        opt.step()
    torch.cuda.synchronize()
    return float(loss), g


ref = run(dict(branch_overlap=False, overlap_wgrad=False))
names = [n for n, _ in LM_Net(3, 2).named_parameters()]
bad = 0
for r in range(reps):
    for label, cfg in (("four streams", {}), ("four streams, weight gradients where they arise", dict(lazy_wgrad=False)),
                       ("branch stream only", dict(overlap_wgrad=False)), ("weight-gradient streams only", dict(branch_overlap=False))):
        got = run(cfg)
        diff = [(names[i], float((u - v).abs().max() / (u.abs().max() + 1e-30))) for i, (u, v) in enumerate(zip(ref[1], got[1])) if not torch.equal(u, v)]
        diff.sort(key=lambda t: -t[1])
        print("rep %d  %-50s loss %.9g (serial %.9g)  %d of %d gradient tensors differ %s" % (r, label, got[0], ref[0], len(diff), len(names), diff[:4]), flush=True)
        bad += len(diff)
hip.set_deterministic(False)
print("RACE-FREE" if bad == 0 else "MISMATCH")
