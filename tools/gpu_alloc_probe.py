"""Does the caching allocator call hipMalloc inside steady-state training steps?  Per-step wall time + allocator counters."""
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
x, y = make_batch(8, 352, 352, dev, 1234)


def step():
    out = net(x)
    loss = crit(out, y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
import gc
mode = sys.argv[1] if len(sys.argv) > 1 else 'default'
if mode == 'freeze':
    gc.collect(); gc.freeze()
elif mode == 'disable':
    gc.disable()
print('gc mode', mode, gc.get_count(), gc.get_threshold())
rows = []
for i in range(40):
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    s1 = torch.cuda.memory_stats()
    rows.append((dt, s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"],
                 s1["reserved_bytes.all.current"] >> 20))
print(" ".join("%.1f/%d/%d" % r[:3] for r in rows))
print("reserved MB", rows[0][3], "->", rows[-1][3], " peak allocated MB", torch.cuda.max_memory_allocated() >> 20)
