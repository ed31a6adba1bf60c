import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.lmnet_ref import LM_Net
m = LM_Net(3, 2); m.train()
x = torch.randn(1, 3, 352, 352)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    t0 = time.time(); y = m(x); y.square().mean().backward(); t1 = time.time()
    y = m(x); y.square().mean().backward(); t2 = time.time()
    print("threads %d: first %.2fs second %.2fs" % (nt, t1 - t0, t2 - t1), flush=True)
