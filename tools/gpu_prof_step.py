"""Per-kernel time table of one training step from the in-library timer (HIP events around every launch).

    python tools/gpu_prof_step.py [--batch 8] [--size 352] [--dtype f32|bf16] [--serial] [--steps 3] [--top 40]
--serial: branch / weight-gradient streams off (every kernel alone on the GPU: the low-noise A/B metric)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_batch
from lm_net_amd import LM_Net, hip
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--size", type=int, default=352)
ap.add_argument("--dtype", default="f32"); ap.add_argument("--serial", action="store_true")
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--top", type=int, default=45)
ap.add_argument("--plans", action="store_true")
ap.add_argument("--layers", action="store_true", help="one line per (kernel, declared cost) = layer shape, sorted by time above the roofline")
a = ap.parse_args()
dev = torch.device("cuda", 0)
net = LM_Net(3, 2).to(dev).train()
net.compute_dtype = "bf16" if a.dtype == "bf16" else "fp32"
if a.serial:
    net._engine.branch_overlap = net._engine.overlap_wgrad = False
if a.plans:
    net.enable_plans()
opt = FusedAdamW(net, lr=1e-3, weight_decay=1e-4)
crit = SegLoss(label_smoothing=1e-3).to(dev)
x, y = make_batch(a.batch, a.size, a.size, dev, 1234)
def step():
    loss = crit(net(x), y); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps * 1e3
hip.prof_begin("@" if a.layers else None)
for _ in range(a.steps): step()
r = hip.prof_end()
if a.layers:
    # roofline time of a launch: max(bytes / 5 TB/s achievable HBM, flops / 137 TF fp32 MFMA (733 TF bf16 16x16x16) measured peaks)
    pk = 733e12 if a.dtype == "bf16" else 137e12
    rows = []
    for k, v in r.items():
        n = v["launches"]; us = v["total_us"] / n
        fl, by = v["flops"] / n, v["bytes"] / n
        roof = max(by / 5e12, fl / pk) * 1e6
        rows.append(((us - roof) * n / a.steps, k, n // a.steps, us, roof, fl, by))
    rows.sort(reverse=True)
    print("%-42s %3s %8s %8s %9s %8s %8s" % ("kernel#flops/bytes", "n", "avg_us", "roof_us", "excess/st", "GB/s", "TF/s"))
    for ex, k, n, us, roof, fl, by in rows[:a.top]:
        nm, _, cost = k.partition("#")
        print("%-42s %3d %8.1f %8.1f %9.1f %8.0f %8.1f   %s" % (nm[:42], n, us, roof, ex, by / us / 1e3, fl / us / 1e6, cost))
    print("sum of excess over all lines: %.2f ms/step; kernel-time sum %.2f ms/step" % (sum(r_[0] for r_ in rows) / 1e3, sum(v["total_us"] for v in r.values()) / a.steps / 1e3))
    sys.exit(0)
tot = sum(v["total_us"] for v in r.values())
print("wall %.2f ms/step (untimed-kernel run); kernel-time sum %.2f ms/step over %d kernels names; %d launches/step" % (
    wall, tot / a.steps / 1e3, len(r), sum(v["launches"] for v in r.values()) // a.steps))
print("%-44s %6s %9s %8s %7s %8s %7s" % ("kernel", "n/step", "us/step", "avg_us", "share", "GB/s", "TF/s"))
for k, v in sorted(r.items(), key=lambda kv: -kv[1]["total_us"])[:a.top]:
    t = v["total_us"] * 1e-6
    print("%-76s %6d %9.1f %8.1f %6.1f%% %8.0f %7.1f" % (k[:76], v["launches"] // a.steps, v["total_us"] / a.steps, v["total_us"] / v["launches"],
          100 * v["total_us"] / tot, v["bytes"] / t / 1e9 if t else 0, v["flops"] / t / 1e12 if t else 0))
