"""Overlapped timeline of training steps from the in-library timer (HIP events per launch on every stream; unlike rocprofv3's
kernel trace this does not serialise the streams).

    python tools/gpu_step_timeline.py [--batch 8] [--size 352] [--dtype f32|bf16] [--steps 2] [--no-plans]
Prints: wall per step, time with 0 / 1 / 2+ kernels in flight, the largest idle gaps (kernels on either side) and the kernels
that most often run alone."""
import argparse, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_batch
from lm_net_amd import LM_Net, hip
from lm_net_amd.loss import SegLoss
from lm_net_amd.optim import FusedAdamW

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--size", type=int, default=352)
ap.add_argument("--dtype", default="f32"); ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--no-plans", action="store_true"); ap.add_argument("--tail", type=int, default=0); ap.add_argument("--s0gaps", type=int, default=0); ap.add_argument("--top", type=int, default=28); ap.add_argument("--buckets", type=float, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
net = LM_Net(3, 2).to(dev).train()
net.compute_dtype = "bf16" if a.dtype == "bf16" else "fp32"
if not a.no_plans:
    net.enable_plans()
opt = FusedAdamW(net, lr=1e-3, weight_decay=1e-4)
crit = SegLoss(label_smoothing=1e-3).to(dev)
x, y = make_batch(a.batch, a.size, a.size, dev, 1234)
def step():
    loss = crit(net(x), y); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps * 1e3
hip.prof_begin("!")
for _ in range(a.steps): step()
tl = hip.prof_timeline()
streams = sorted(set(r[1] for r in tl))
sname = {s: "S%d" % i for i, s in enumerate(streams)}
T0, T1 = min(r[2] for r in tl), max(r[3] for r in tl)
print("untimed wall %.2f ms/step; timed %d launches over %.2f ms (%d steps: %.2f ms/step); %d streams (torch ops between the "
      "library's launches are not timed: they show up as idle)" % (wall, len(tl), (T1 - T0) / 1e3, a.steps, (T1 - T0) / 1e3 / a.steps, len(streams)))
ev = sorted([(r[2], 1, i) for i, r in enumerate(tl)] + [(r[3], -1, i) for i, r in enumerate(tl)])
active, last, by_n, solo = set(), T0, defaultdict(float), defaultdict(float)
gaps = []
prev_end = None
for t, d, i in ev:
    if t > last:
        by_n[min(len(active), 3)] += t - last
        if len(active) == 1:
            solo[tl[next(iter(active))][0]] += t - last
        if len(active) == 0:
            gaps.append((t - last, last, prev_end, i))
    last = t
    if d > 0: active.add(i)
    else:
        active.discard(i); prev_end = i
print("kernels in flight (per step): " + ", ".join("%s: %.2f ms" % ("3+" if n == 3 else n, v / 1e3 / a.steps) for n, v in sorted(by_n.items())))
for s in streams:
    rs = [r for r in tl if r[1] == s]
    print("  %s: %5d launches, busy %.2f ms/step" % (sname[s], len(rs) // a.steps, sum(r[3] - r[2] for r in rs) / 1e3 / a.steps))
print("largest idle gaps:")
for g, at, pe, nx in sorted(gaps, reverse=True)[:12]:
    print("  %7.1f us at %8.2f ms: %s -> %s" % (g, (at - T0) / 1e3, tl[pe][0][:50] if pe is not None else "-", tl[nx][0][:50]))
print("idle in gaps < 20 us: %.2f ms/step (%d gaps/step)" % (sum(g[0] for g in gaps if g[0] < 20) / 1e3 / a.steps, sum(1 for g in gaps if g[0] < 20) // a.steps))
print("alone on the GPU (per step):")
for k, v in sorted(solo.items(), key=lambda kv: -kv[1])[:25]:
    print("  %-60s %8.1f us" % (k[:60], v / a.steps))
# per stream: kernel time by name, and the stream's own idle time between its first and last launch of a step
for s in streams:
    rs = sorted([r for r in tl if r[1] == s], key=lambda r: r[2])
    by = defaultdict(lambda: [0, 0.0])
    for r in rs:
        by[r[0]][0] += 1; by[r[0]][1] += r[3] - r[2]
    gap = sum(max(0.0, b[2] - a_[3]) for a_, b in zip(rs, rs[1:]) if b[2] - a_[3] < 200.0)
    print("stream %s: %d launches/step, busy %.2f ms/step, gaps < 200 us between its launches %.2f ms/step" % (
        sname[s], len(rs) // a.steps, sum(r[3] - r[2] for r in rs) / 1e3 / a.steps, gap / 1e3 / a.steps))
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print("    %-52s %4d x %7.1f us = %8.1f us/step" % (k[:52], n // a.steps, t / n, t / a.steps))

# --tail N: the last N launches before each AdamW launch (what runs at the end of the backward), with stream and times relative to it
if "--tail" in sys.argv:
    n = int(sys.argv[sys.argv.index("--tail") + 1])
    order = sorted(range(len(tl)), key=lambda i: tl[i][2])
    for pos, i in enumerate(order):
        if tl[i][0].startswith("adamw"):
            t_ad = tl[i][2]
            print("---- before adamw at %.2f ms" % ((t_ad - T0) / 1e3))
            for j in order[max(0, pos - n):pos + 1]:
                r = tl[j]
                print("  %s  start %8.1f us  end %8.1f us  (%6.1f us)  %s" % (sname[r[1]], r[2] - t_ad, r[3] - t_ad, r[3] - r[2], r[0][:70]))
            break

# --s0gaps N: the N largest idle gaps of the main stream (the one with most launches) inside a step: where it waits for another stream
if a.s0gaps:
    main = max(streams, key=lambda s: sum(1 for r in tl if r[1] == s))
    rs = sorted([r for r in tl if r[1] == main], key=lambda r: r[2])
    gs = sorted(((b[2] - a_[3], a_, b) for a_, b in zip(rs, rs[1:]) if not b[0].startswith("nchw_to") and not a_[0].startswith("adamw")), key=lambda g: -g[0])[:a.s0gaps]
    print("largest gaps of %s (main stream), total %.2f ms/step in gaps > 15 us:" % (sname[main], sum(max(0, b[2] - a_[3]) for a_, b in zip(rs, rs[1:]) if 15 < b[2] - a_[3] < 2000) / 1e3 / a.steps))
    for g, a_, b in gs:
        others = [r for r in tl if r[1] != main and r[3] > a_[3] and r[2] < b[2]]
        print("  %7.1f us at %8.2f ms: %s -> %s" % (g, (a_[3] - T0) / 1e3, a_[0][:44], b[0][:44]))
        for r in sorted(others, key=lambda r: r[2])[:8]:
            print("        %s %7.1f..%7.1f  %s" % (sname[r[1]], r[2] - a_[3], r[3] - a_[3], r[0][:60]))

# --buckets US: busy fraction of every stream per time bucket of the LAST timed step (where is the slack?), with the main stream's
# first kernel of each bucket as a landmark
if "--buckets" in sys.argv:
    bw = float(sys.argv[sys.argv.index("--buckets") + 1])
    ad = sorted(r[2] for r in tl if r[0].startswith("adamw"))
    t_lo = ad[-2] if len(ad) > 1 else T0
    t_hi = ad[-1]
    main = max(streams, key=lambda s: sum(1 for r in tl if r[1] == s))
    nb = int((t_hi - t_lo) / bw) + 1
    print("---- busy fraction per %.0f us bucket, streams %s (last step, %.2f ms)" % (bw, " ".join(sname[s] for s in streams), (t_hi - t_lo) / 1e3))
    for b in range(nb):
        lo, hi = t_lo + b * bw, t_lo + (b + 1) * bw
        fr = []
        for s in streams:
            fr.append(sum(max(0.0, min(r[3], hi) - max(r[2], lo)) for r in tl if r[1] == s and r[3] > lo and r[2] < hi) / bw)
        first = [r for r in tl if r[1] == main and lo <= r[2] < hi]
        first.sort(key=lambda r: r[2])
        print("  %6.2f ms  %s   %s" % ((lo - t_lo) / 1e3, " ".join("%4.2f" % f for f in fr), first[0][0][:48] if first else ""))
