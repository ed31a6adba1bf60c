"""How far ahead of the GPU does the host run?  Wall time to ENQUEUE n training steps vs. to FINISH them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW
from bench import make_batch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = LM_Net(3, 2).to(dev).train()
opt = FusedAdamW(net, lr=1e-3, weight_decay=1e-4)
crit = SegLoss(label_smoothing=0.001).to(dev)
import sys as _s
BS, HW = (1, 64) if len(_s.argv) > 1 and _s.argv[1] == "tiny" else (8, 352)
x, y = make_batch(BS, HW, HW, dev, 1234)
if "serial" in _s.argv:
    net._engine.branch_overlap = net._engine.overlap_wgrad = False


def step():
    out = net(x)
    loss = crit(out, y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, finish %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
if os.environ.get("LMN_CPROFILE") is None: raise SystemExit(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
pstats.Stats(pr).sort_stats("cumtime").print_stats(30)
