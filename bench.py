"""Headline benchmark: LM-Net training throughput (images/sec) at 352x352 on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one full training step of BASELINE.json configs[1] ("LM-Net fp32 training, batch 8, 352x352
synthetic masks") per GPU: forward, CE(weight [1,4], label_smoothing 1e-3) + Dice(weight [1,4]) loss
(train_eval_utils.py:141), backward, AdamW(lr 1e-3, wd 1e-4) step (train.py:156).  Dropout and
batch-stat BatchNorm are live.  Inputs are resident in HBM before the timed region.  For N > 1 the
mini-batch is sharded 8 images per GPU (weak scaling) with bucketed RCCL gradient all-reduce
overlapped with backward (lm_net_amd/ddp.py).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); 6.29 TB/s measured copy ceiling


def dice_loss(logits, target, weight=(1.0, 4.0), smooth=1e-5):
    """utils/loss.py:170-206 (softmax, one-hot, per-class 1-(2*sum(p*t)+s)/(sum(p^2)+sum(t^2)+s), /n_classes)."""
    p = torch.softmax(logits, dim=1)
    loss = 0.0
    n = logits.shape[1]
    for i in range(n):
        t = (target == i).float()
        pi = p[:, i]
        loss = loss + (1 - (2 * (pi * t).sum() + smooth) / ((pi * pi).sum() + (t * t).sum() + smooth)) * weight[i]
    return loss / n


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel`, measured by separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    very command and committed under profiles/ (a PMC pass cannot run inside the timed loop)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")) as f:
            return round(json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def make_batch(B, H, W, device, seed):
    from tools.detweights import disc_labels
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, H, W, generator=g)
    y = disc_labels(B, H, W, seed)
    return x.to(device), y.to(device)


def cpu_baseline(H, W, threads=16, budget_s=25.0):
    """The CPU restatement of the same path (oracle, graph-identical to core/LM_Net.py) timed on this
    box's host cores: a bounded sample of the same workload -- train steps at batch 2.  Runs in a child
    process (its own thread pool; a hard wall-clock bound) so the GPU line is printed no matter what.
    16 threads: measured fastest on the 256-core GPU box (8: 0.90 s, 16: 0.54 s, 32: 0.89 s, 64: 2.3 s per
    batch-1 forward+backward) -- the graph is ~1200 small ops, more threads only add sync cost."""
    import subprocess
    code = (
        "import sys, time, json, torch; sys.path.insert(0, %r)\n"
        "import torch.nn.functional as F\n"
        "from bench import dice_loss, make_batch\n"
        "from oracle.lmnet_ref import LM_Net as Oracle\n"
        "torch.manual_seed(0); torch.set_num_threads(%d)\n"
        "m = Oracle(3, 2); m.train(); opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)\n"
        "B = 2; x, y = make_batch(B, %d, %d, 'cpu', 99); w = torch.tensor([1.0, 4.0])\n"
        "def step():\n"
        "    out = m(x); loss = F.cross_entropy(out, y, weight=w, label_smoothing=0.001) + dice_loss(out, y)\n"
        "    opt.zero_grad(); loss.backward(); opt.step()\n"
        "t0 = time.time(); step(); warm = time.time() - t0\n"
        "n, t0 = 0, time.time()\n"
        "while n < 1 or (time.time() - t0 + warm < %f and n < 8):\n"
        "    step(); n += 1\n"
        "print(json.dumps({'n': n, 'dt': time.time() - t0, 'B': B}))\n"
    ) % (ROOT, threads, H, W, budget_s * 0.6)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s * 4, cwd=ROOT)
        r = json.loads(out.stdout.strip().splitlines()[-1])
        return {"value": round(r["B"] * r["n"] / r["dt"], 3), "unit": "images/sec", "cores": threads, "kind": "port",
                "sample": "%d train steps (fwd + CE/Dice + bwd + AdamW), batch %d, %dx%d, fp32, PyTorch-CPU oracle "
                          "(oracle/lmnet_ref.py), %d threads of %d host cores" % (r["n"], r["B"], H, W, threads, os.cpu_count() or 0)}
    except Exception as e:  # never lose the GPU measurement because the CPU leg misbehaved
        return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port", "sample": "failed: %s" % str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=352)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graphs", action="store_true",
                    help="replay the step as two hipGraphs (wins when the host is the bottleneck, e.g. batch 1; at batch 8 "
                         "the step is GPU-bound and host launches measured 7 %% faster than the replay)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") over xGMI; LMNET_BENCH_BACKEND=gloo only to exercise this code path with several ranks on ONE GPU
        dist.init_process_group(os.environ.get("LMNET_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from lm_net_amd import LM_Net
    from lm_net_amd.ddp import DistributedLMNet
    torch.manual_seed(1234)
    net = LM_Net(3, 2).to(dev)
    model = DistributedLMNet(net) if world > 1 else net
    model.train()
    if args.graphs:
        net.enable_graphs()     # forward / backward as two hipGraph replays per step (captured during the warm-up)
    from lm_net_amd.optim import FusedAdamW
    opt = FusedAdamW(net, lr=1e-3, weight_decay=1e-4)     # torch.optim.AdamW semantics, one launch per step
    B, H, W = args.batch, args.size, args.size
    x, y = make_batch(B, H, W, dev, 1234 + rank)          # rank-offset data seed (train.py:42)
    from lm_net_amd.loss import SegLoss
    crit = SegLoss(ce_weight=(1.0, 4.0), dice_weight=(1.0, 4.0), label_smoothing=0.001).to(dev)   # fused CE + Dice

    def step():
        out = model(x)
        loss = crit(out, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(max(args.warmup, 3 if args.graphs else 0)):   # graph capture happens on the 3rd step of a shape
        step()
    # dominant-kernel timing: HIP events on the launch stream around every dw_fwd launch -- inside the timed region
    # when kernels are launched from the host; in graph mode (no host code runs during a replay) over 5 extra
    # host-launched steps right after the timed region, same process, same buffers
    net._engine.kernel_events = {"dw_fwd": []}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if net.use_graphs:
        net.use_graphs = False
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        net.use_graphs = True
    ev = net._engine.kernel_events["dw_fwd"]
    # the same kernel with nothing else on the GPU: 5 more steps with the side / branch streams switched off (inside
    # the timed region the kernel shares the CUs with the concurrently running skip / attention chains)
    eng = net._engine
    saved = (eng.branch_overlap, eng.overlap_wgrad, net.use_graphs)
    eng.branch_overlap, eng.overlap_wgrad, net.use_graphs = False, False, False
    eng.kernel_events = {"dw_fwd": []}
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ev_serial = eng.kernel_events["dw_fwd"]
    eng.branch_overlap, eng.overlap_wgrad, net.use_graphs = saved
    net._engine.kernel_events = None
    if rank == 0:
        kt = sum(e0.elapsed_time(e1) for e0, e1, _ in ev) * 1e-3
        kb = sum(b for _, _, b in ev)
        achieved = kb / kt / 1e9 if kt > 0 else 0.0
        kts = sum(e0.elapsed_time(e1) for e0, e1, _ in ev_serial) * 1e-3
        achieved_serial = sum(b for _, _, b in ev_serial) / kts / 1e9 if kts > 0 else 0.0
        res = {
            "metric": "train images/sec at 352x352, 1/2/4/8 MI355X; Dice vs ref",
            "value": round(world * B * args.steps / dt, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "LM-Net fp32 training step (fwd + CE/Dice loss + bwd + AdamW), batch %d/GPU, %dx%d "
                                   "synthetic disc masks (BASELINE configs[1])" % (B, H, W),
                       "global_batch": world * B, "image": [3, H, W], "parallelism": "dp%d" % world,
                       "launch": "hipGraph replay (fwd + bwd graphs per step)" if args.graphs else "host",
                       "final_loss": round(float(loss.detach()), 5)},
            "roofline": {"bound": "hbm", "kernel": "dw_fwd_strip_kernel (row A2 forward, 5x5 merged depthwise stencil + GELU-sum)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_copy_ceiling": round(achieved / 6290.0, 4),
                         "launches": len(ev), "avg_us": round(kt / max(len(ev), 1) * 1e6, 2),
                         "achieved_alone": round(achieved_serial, 1), "frac_alone": round(achieved_serial / HBM_PEAK_GBS, 4),
                         "note": "achieved: HIP events around every launch inside the timed region, where the kernel shares "
                                 "the GPU with the concurrent branch / weight-gradient streams; achieved_alone: same events "
                                 "over 5 further steps of this process with those streams switched off",
                         "traffic": pmc_traffic("dw_fwd_strip_kernel") if (B, H, W) == (8, 352, 352) else None,
                         "algorithmic_bytes_per_launch": round(kb / max(len(ev), 1)),
                         "algorithmic_bytes": "2*E*H*W*B*4 per launch (read x1 once, write pre once), averaged over "
                                              "the 16 launches per step (four resolutions); traffic = HBM bytes per launch "
                                              "from the committed rocprofv3 PMC passes (profiles/r01_pmc_hbm_traffic.json)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(H, W)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
