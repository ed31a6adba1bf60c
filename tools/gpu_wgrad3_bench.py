"""3x3 weight-gradient micro-benchmark: records the 3x3 lmn_conv_wgrad calls of one LM-Net training step (B=8, 352x352)
and replays each of them alone on the GPU with the step's own tensors.

    python tools/gpu_wgrad3_bench.py [f32|bf16]       (LMN_WGRAD_V1=0 selects the kernel without cross-tile prefetch and tile-split waves)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_batch
from lm_net_amd import LM_Net, hip
from lm_net_amd.loss import SegLoss

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda", 0)
net = LM_Net(3, 2).to(dev).train()
net.compute_dtype = "bf16" if mode == "bf16" else "fp32"
net._engine.branch_overlap = net._engine.overlap_wgrad = False
crit = SegLoss(label_smoothing=1e-3).to(dev)
x, y = make_batch(8, 352, 352, dev, 1234)
calls = []
orig = hip.conv_wgrad
def rec(srcs, dy, dW, db, **kw):
    if kw.get("ksize", 1) == 3:
        calls.append((srcs, dy, dW, db, kw))
    return orig(srcs, dy, dW, db, **kw)
hip.conv_wgrad = rec
import lm_net_amd.engine as E
if hasattr(E, "hip"): E.hip.conv_wgrad = rec
crit(net(x), y).backward()
torch.cuda.synchronize()
hip.conv_wgrad = orig
hip._MMA[0] = hip.BF16 if mode == "bf16" else hip.F32

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

tot = 0.0
for srcs, dy, dW, db, kw in calls:
    cin = []
    for s in srcs:
        v = s["view"] if isinstance(s, dict) else s
        if not isinstance(v, torch.Tensor): v = v.t
        cin.append(int(v.shape[-1]))
    t = timeit(lambda: orig(srcs, dy, dW, db, **kw))
    fl = 2.0 * kw["B"] * kw["Hout"] * kw["Wout"] * kw["Cout"] * sum(cin) * 9
    print("3x3 s%d %4dx%-4d %-14s -> %-4d %8.1f us  %5.1f TF" % (kw.get("stride", 1), kw["Hin"], kw["Win"], "+".join(map(str, cin)), kw["Cout"],
                                                             t * 1e6, fl / t / 1e12))
    tot += t
print("sum %.1f us over %d calls (%s, v1=%s)" % (tot * 1e6, len(calls), mode, os.environ.get("LMN_WGRAD_V1", "1")))
