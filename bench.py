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


def _pmc():
    """The committed PMC reduction of this very command (profiles/rNN_pmc.json, newest round first), or None."""
    for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.isfile(path):
            try:
                with open(path) as f:
                    d = json.load(f)
                d["_file"] = "profiles/" + name
                return d
            except (OSError, ValueError):
                pass
    return None


def _targs(name):
    """('base', [template arguments]) of a kernel name; nested <> kept inside an argument."""
    base, _, rest = name.partition("<")
    args, cur, depth = [], "", 0
    for ch in rest.rsplit(">", 1)[0] if rest else "":
        if ch == "," and depth == 0:
            args.append(cur.strip()); cur = ""
        else:
            depth += (ch == "<") - (ch == ">")
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return base.strip(), args


def _pmc_match(kernel):
    """PMC records of `kernel`.  The in-library timer and rocprofv3 file a launch under the same key -- the instantiated kernel name,
    template arguments resolved (`dw_bwd_kernel<float, 0, true, true, 2>`; lmn_kname in csrc/runtime.hip) -- so this is a lookup.
    (A key without resolved arguments, as the round-3 library produced, still matches by base name + literal arguments.)"""
    d = _pmc()
    if not d:
        return []
    ks = d["kernels"]
    if kernel in ks:
        return [ks[kernel]]
    base, args = _targs(kernel)
    out = []
    for name, rec in ks.items():
        b2, a2 = _targs(name)
        if b2 != base or len(a2) != len(args) or not rec:
            continue
        lit = lambda a: a in ("true", "false") or a.lstrip("-").isdigit()
        if any(not lit(a) and a not in ("float", "__bf16") for a in args) and all((not lit(a)) or a == b for a, b in zip(args, a2)):
            out.append(rec)
    return out


def rocprof_avg_us(kernel):
    """AverageNs of `kernel` in the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/rNN_bench_kernel_stats.csv, newest round first; its Name column = `void ` + the instantiated name + the parameter
    list), in microseconds, with the file it came from -- or (None, None).  The profiled run mixes four-stream and serial steps,
    so this sits between avg_us (inside the step) and avg_us_alone."""
    import csv
    for name in ("r06_bench_kernel_stats.csv", "r05_bench_kernel_stats.csv", "r04_bench_kernel_stats.csv"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.isfile(path):
            continue
        try:
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    nm = row.get("Name", "")
                    nm = nm[5:] if nm.startswith("void ") else nm
                    if nm.split("(")[0].strip() == kernel:
                        return float(row["AverageNs"]) / 1e3, "profiles/" + name
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def pmc_field(kernel, field):
    """Launch-weighted mean of a per-launch PMC figure over the instantiations behind `kernel` (profiles/rNN_pmc.json), or None."""
    recs = [r for r in _pmc_match(kernel) if r.get(field) is not None and r.get("launches_sampled")]
    n = sum(r["launches_sampled"] for r in recs)
    return sum(r[field] * r["launches_sampled"] for r in recs) / n if n else None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel`, measured by separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    very command and committed under profiles/ (a PMC pass cannot run inside the timed loop)."""
    v = pmc_field(kernel, "hbm_bytes_per_launch")
    return None if v is None else round(v)


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32) = fp32 vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (~2.5 PF; the 2:1-sparsity figure is never used)
_PEAK = [MFMA_F32_PEAK_TFLOPS]  # matrix-core peak of the arithmetic of the run being reported (set by run_config)
# SURVEY.md 8d, per image at 352x352 fp32 (scaled by pixels for other sizes): algorithmic bytes / FLOPs of one training step
STEP_MB_352, STEP_GFLOP_352 = 2085.54, 59.7


COPY_CEILING_GBS = 6290.0       # measured device copy ceiling (the spec figure HBM_PEAK_GBS is the `peak` of the contract)
HBM_ROWS = ("row", "dw_", "na_", "ln_", "affine", "pool", "up2_", "adamw", "loss")   # kernels of the byte-bound rows (8d "which roofline")


def _entry(name, rec, extra_us=0.0):
    """One kernel's achieved rates from the in-library timer record {launches, total_us, flops, bytes}."""
    t = (rec["total_us"] + extra_us) * 1e-6
    n = max(rec["launches"], 1)
    gbs, tfs = rec["bytes"] / t / 1e9 if t > 0 else 0.0, rec["flops"] / t / 1e12 if t > 0 else 0.0
    pk = _PEAK[0]
    t_hbm, t_mfma = rec["bytes"] / (HBM_PEAK_GBS * 1e9), rec["flops"] / (pk * 1e12)
    bound = "mfma" if t_mfma > t_hbm and not name.startswith(HBM_ROWS) else "hbm"
    e = {"kernel": name, "bound": bound, "launches": rec["launches"], "avg_us": round(rec["total_us"] / n, 2),
         "achieved": round(tfs if bound == "mfma" else gbs, 2), "peak": pk if bound == "mfma" else HBM_PEAK_GBS,
         "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
         "frac": round((tfs / pk) if bound == "mfma" else (gbs / HBM_PEAK_GBS), 4),
         "algorithmic_flops_per_launch": round(rec["flops"] / n), "algorithmic_bytes_per_launch": round(rec["bytes"] / n)}
    e["frac_vs_copy_ceiling"] = round(gbs / COPY_CEILING_GBS, 4) if bound == "hbm" else None
    return e


def _traffic_ratio(recs, _unused=None):
    """PMC HBM bytes / algorithmic bytes of a set of timer records {name: rec}: per-launch PMC bytes of each instantiation
    (profiles/rNN_pmc.json) x its launches, over the algorithmic bytes the same launches declared.  None without PMC data."""
    phys, alg = 0.0, 0.0
    for k, v in recs.items():
        t = pmc_field(k, "hbm_bytes_per_launch")
        if t is None:
            return None
        phys += t * v["launches"]
        alg += v["bytes"]
    return round(phys / alg, 3) if alg > 0 else None


def _group(live, names):
    tot = {"launches": 0, "total_us": 0.0, "flops": 0.0, "bytes": 0.0}
    for k, v in live.items():
        if k.split("<")[0] in names:
            for f in tot:
                tot[f] += v[f]
    return tot


def roofline_block(dominant, live, survey, tot_us, B, H, W, step_s, alone, esz=4):
    """`roofline` of the bench line: the kernel that takes the most GPU time (found by timing EVERY launch over two untimed
    steps), measured live with HIP events on its launch stream inside the timed region; algorithmic FLOPs / bytes from the
    layer shapes of each launch (SURVEY 8d convention).  Secondary entries: rows A2 / A7 (north_star's >= 70 % HBM targets)
    and the whole step against both rooflines."""
    rec = live.get(dominant) or survey[dominant]
    at_headline = (B, H, W) == (8, 352, 352) and esz == 4         # the committed PMC passes are of the default command
    r = _entry(dominant, rec)
    r["share_of_gpu_time"] = round(survey[dominant]["total_us"] / tot_us, 4)
    if dominant in alone:
        ea = _entry(dominant, alone[dominant])
        r["achieved_alone"], r["frac_alone"], r["avg_us_alone"] = ea["achieved"], ea["frac"], ea["avg_us"]
    # split-K weight gradients: the fixed-order reduction launches that finish the dominant kernel's partial sums
    red = [k for k in live if k.startswith("wgrad_reduce_kernel<%s," % ("1" if "1x1" in dominant else "9"))] if dominant.startswith("wgrad") else []
    if red:
        r["frac_with_reduce_launches"] = _entry(dominant, rec, sum(live[k]["total_us"] for k in red))["frac"]
    #  the same fraction from the committed rocprofv3 summary: algorithmic FLOPs (or bytes) per launch / the CSV's
    # AverageNs of the same kernel name -- the line and profiles/ agree without a footnote
    if at_headline:
        us, src = rocprof_avg_us(dominant)
        if us:
            per = r["algorithmic_flops_per_launch"] if r["bound"] == "mfma" else r["algorithmic_bytes_per_launch"]
            r["avg_us_rocprof"] = round(us, 2)
            r["frac_rocprof"] = round(per / (us * 1e-6) / ((r["peak"] * 1e12) if r["bound"] == "mfma" else (r["peak"] * 1e9)), 4)
            r["rocprof_file"] = src
    r["traffic"] = pmc_traffic(dominant) if at_headline else None
    r["traffic_ratio"] = (round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)
                          if r["traffic"] and r["algorithmic_bytes_per_launch"] else None)
    r["mfma_util_pmc"] = pmc_field(dominant, "mfma_busy_frac") if at_headline else None
    r["top5_by_time"] = [{"kernel": k, "share": round(v["total_us"] / tot_us, 4), "avg_us": round(v["total_us"] / max(v["launches"], 1), 1)}
                         for k, v in sorted(survey.items(), key=lambda kv: -kv[1]["total_us"])[:5]]
    pm = (_pmc() or {}).get("whole_step")
    # rows A2 / A7 under the 8d convention.  A2 train = 5*E*HW*B*esz: the forward (2 tensor passes) and the backward (3) declare
    # them; the two BatchNorm statistics passes declare ZERO bytes (extra passes of this implementation: they add time, and PMC
    # traffic, not algorithmic bytes).  A7 train = 11*C*HW*B*esz over the forward and the backward kernels.
    ROWS = (("row_A2", "row A2 (depthwise branches: fwd + 2 statistics passes + bwd)",
             ("dw_fwd_kernel", "dw_bwd_kernel", "dw_stats0_kernel", "dw_stats1_kernel")),
            ("row_A7", "row A7 (fused neighborhood attention: fwd + bwd)",
             ("na_fwd_kernel", "na_bwd_fused_kernel", "na_bwd_q_kernel", "na_bwd_kv_kernel", "na_bwd_q_tile_kernel", "na_bwd_kv_tile_kernel")))
    for key, label, names in ROWS:
        grp = _group(live, names)
        if grp["total_us"] <= 0:
            continue
        e = _entry(label, grp)
        row = {k: e[k] for k in ("kernel", "launches", "achieved", "peak", "unit", "frac", "frac_vs_copy_ceiling")}
        al = _group(alone, names)
        if al["total_us"] > 0:
            row["frac_alone"] = _entry(label, al)["frac"]
        members = {k: v for k, v in live.items() if k.split("<")[0] in names}
        row["traffic_ratio"] = _traffic_ratio(members, None) if at_headline else None
        for k, v in sorted(members.items()):
            ek = _entry(k, v)
            row[k] = {f: ek[f] for f in ("launches", "avg_us", "achieved", "frac")}
            if k in alone:
                row[k]["frac_alone"] = _entry(k, alone[k])["frac"]
            if at_headline:
                row[k]["traffic"] = pmc_traffic(k)
        r[key] = row
    # the conv families, from the survey steps (every launch timed, four streams live): algorithmic bytes / FLOPs of all launches
    # of a family / the sum of their event durations, against the roofline that bounds the family as a whole
    fam = {"conv 1x1 fwd + dgrad (rows A1, A3, A6/A8 linears)": ("conv_tile_kernel<1,", "conv_tileM_kernel<1,", "conv_dma1_kernel<"),
           "conv 3x3 fwd + dgrad (rows A4, A9-A11)": ("conv_tile_kernel<9,", "conv_tileM_kernel<9,", "conv_dma3_kernel<"),
           "weight gradients 1x1": ("wgrad_1x1w_kernel", "wgrad_1x1_kernel", "wgrad_reduce_kernel<1,"),
           "weight gradients 3x3": ("wgrad3_kernel", "wgrad_lds_kernel", "wgrad_reduce_kernel<9,")}
    r["families_survey"] = {}
    for label, prefixes in fam.items():
        tot = {"launches": 0, "total_us": 0.0, "flops": 0.0, "bytes": 0.0}
        for k, v in survey.items():
            if k.startswith(prefixes):
                for f in tot:
                    tot[f] += v[f]
        if tot["total_us"] > 0:
            e = _entry(label, tot)
            r["families_survey"][label] = {k: e[k] for k in ("bound", "launches", "achieved", "peak", "unit", "frac")}
            r["families_survey"][label]["share_of_gpu_time"] = round(tot["total_us"] / tot_us, 4)
    scale = (H * W) / (352.0 * 352.0)
    # activation elements x 4 B (fp32) or x 2 B (bf16 storage); the matrix-core peak is that of the operand type
    mb, gf = STEP_MB_352 * scale * B * (esz / 4.0), STEP_GFLOP_352 * scale * B
    r["whole_step"] = {"algorithmic_MB": round(mb, 1), "algorithmic_GFLOP": round(gf, 1),
                       "hbm_GBps": round(mb / 1e3 / step_s, 1), "hbm_frac": round(mb / 1e3 / step_s / HBM_PEAK_GBS, 4),
                       "TFLOPs": round(gf / 1e3 / step_s, 2), "mfma_peak_TFLOPs": _PEAK[0],
                       "mfma_frac": round(gf / 1e3 / step_s / _PEAK[0], 4)}
    if pm and (B, H, W) == (8, 352, 352) and esz == 4:
        r["whole_step"]["hbm_traffic_MB_pmc"] = pm["hbm_total_MB"]            # FETCH_SIZE x2 + WRITE_SIZE over one step
        r["whole_step"]["mfma_busy_ms_per_simd_pmc"] = pm["mfma_busy_ms_per_simd_at_2p4GHz"]
    r["note"] = ("dominant kernel = largest share of GPU kernel time over two untimed steps with every launch timed; its "
                 "achieved rate = sum of algorithmic FLOPs (2*MACs of each launch's layer shape) / sum of HIP-event durations "
                 "on the launch stream INSIDE the timed region, where it shares the GPU with the other streams of the step "
                 "(*_alone: the same events over 3 further steps of this process with the extra streams switched off); "
                 "row_A2 / row_A7: their other kernels are timed over the same number of steps right after the timed region (the "
                 "events of ~90 more launches cost the headline 0.45 ms per step); "
                 "kernel names are instantiated names = the Name column of profiles/*_kernel_stats.csv minus `void ` and the "
                 "parameter list; frac_vs_copy_ceiling = achieved / 6.29 TB/s (measured copy ceiling, SURVEY 8d); "
                 "traffic / traffic_ratio / mfma_util_pmc from the committed rocprofv3 PMC passes (profiles/): HBM bytes per "
                 "launch (FETCH_SIZE x2 + WRITE_SIZE) and their ratio to the algorithmic bytes")
    return r


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
        "r = {'n': n, 'dt': time.time() - t0, 'B': B}\n"
        "# SURVEY 8d's batch (8) once, beside the bounded batch-2 sample (VERDICT r4 item 7): one untimed + one timed step\n"
        "B8 = 8; x, y = make_batch(B8, %d, %d, 'cpu', 98)\n"
        "if time.time() - t0 + warm < %f:\n"
        "    step(); t1 = time.time(); step(); r['b8_dt'] = time.time() - t1\n"
        "print(json.dumps(r))\n"
    ) % (ROOT, threads, H, W, budget_s * 0.6, H, W, budget_s * 1.2)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s * 4, cwd=ROOT)
        r = json.loads(out.stdout.strip().splitlines()[-1])
        res = {"value": round(r["B"] * r["n"] / r["dt"], 3), "unit": "images/sec", "cores": threads, "kind": "port",
               "sample": "%d train steps (fwd + CE/Dice + bwd + AdamW), batch %d, %dx%d, fp32, PyTorch-CPU oracle "
                         "(oracle/lmnet_ref.py), %d threads of %d host cores" % (r["n"], r["B"], H, W, threads, os.cpu_count() or 0)}
        if r.get("b8_dt"):
            res["value_batch8"] = round(8 / r["b8_dt"], 3)      # one train step at SURVEY 8d's batch 8, same threads
        return res
    except Exception as e:  # never lose the GPU measurement because the CPU leg misbehaved
        return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port", "sample": "failed: %s" % str(e)[:200]}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) BEFORE this process has touched the GPU, relay their output (rank 0 prints the JSON line) and
    exit with the launcher's code.  Fewer than N visible GPUs is an error, never a silent 1-GPU run."""
    import subprocess
    have = torch.cuda.device_count()        # (counting devices does not initialise the GPU on this image)
    if have < n and not (have >= 1 and os.environ.get("LMNET_BENCH_BACKEND") == "gloo"):   # (gloo: several ranks may share a GPU)
        print("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to report a smaller run as n_gpus=%d"
              % (n, have, n), file=sys.stderr)
        sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.call(cmd, env=env))


class Run:
    """One benchmark configuration: model, optimizer, fused loss and a resident synthetic batch on `dev`."""

    def __init__(self, dev, world, rank, dtype, B, size, plans=True, graphs=False):
        from lm_net_amd import LM_Net
        from lm_net_amd.ddp import DistributedLMNet
        from lm_net_amd.loss import SegLoss
        from lm_net_amd.optim import FusedAdamW
        torch.manual_seed(1234)
        self.net = LM_Net(3, 2).to(dev)
        self.net.compute_dtype = "bf16" if dtype == "bf16" else "fp32"
        self.model = DistributedLMNet(self.net) if world > 1 else self.net
        self.model.train()
        if graphs:
            self.net.enable_graphs()    # forward / backward as two hipGraph replays per step (captured during the warm-up)
        elif plans:
            # forward / backward as one lmn_plan_run each (recorded on the 3rd step of the shape); the backward node assigns .grad itself
            self.net.enable_plans(direct_grads=True)
        self.opt = FusedAdamW(self.net, lr=1e-3, weight_decay=1e-4)     # torch.optim.AdamW semantics, one launch per step
        self.x, self.y = make_batch(B, size, size, dev, 1234 + rank)    # rank-offset data seed (train.py:42)
        self.crit = SegLoss(ce_weight=(1.0, 4.0), dice_weight=(1.0, 4.0), label_smoothing=0.001).to(dev)   # fused CE + Dice
        self.B, self.size, self.dtype = B, size, dtype

    def step(self):
        out = self.model(self.x)
        loss = self.crit(out, self.y)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        return loss

    def timed(self, steps, world, dev):
        """EXACTLY `steps` steps between barrier + synchronize pairs; max over ranks."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = self.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, loss


def deterministic_check(dev, B, size, steps=5):
    """Deterministic mode (LM_Net.deterministic / lmn_set_deterministic) on the headline shape: two fresh runs of two training steps,
    all gradient tensors compared bit for bit, then the step time of the mode over `steps` replayed steps."""
    from lm_net_amd import hip
    out = {"mode": "fixed-order reductions (lmn_set_deterministic)"}
    try:
        grads = []
        for _ in range(2):
            r = Run(dev, 1, 0, "f32", B, size, plans=False)
            r.net.deterministic = True
            for _ in range(2):
                loss = r.crit(r.model(r.x), r.y)
                r.opt.zero_grad(set_to_none=True)
                loss.backward()
                g = [p.grad.detach().clone() for p in r.net.parameters()]
                r.opt.step()
            torch.cuda.synchronize()
            grads.append((float(loss.detach()), g))
            del r
        (la, ga), (lb, gb) = grads
        out["gradient_tensors"] = len(ga)
        out["differing_tensors"] = sum(0 if torch.equal(u, v) else 1 for u, v in zip(ga, gb))
        out["bit_identical"] = bool(out["differing_tensors"] == 0 and la == lb)
        del grads, ga, gb
        r = Run(dev, 1, 0, "f32", B, size)
        r.net.deterministic = True
        for _ in range(5):
            r.step()
        dt, _ = r.timed(steps, 1, dev)
        out["ms_per_step"] = round(dt / steps * 1e3, 3)
        del r
    except Exception as e:                      # (reported, never fatal for the headline)
        out["error"] = str(e)[:300]
    finally:
        hip.set_deterministic(False)
        torch.cuda.empty_cache()
    return out


def other_config(dev, dtype, B, size, steps=10, warmup=5):
    """A further BASELINE configuration timed in the same process (never the headline): warm-up (incl. plan recording), then
    `steps` steps between synchronisations.  Returns the entry for `other_configs`."""
    _PEAK[0] = MFMA_BF16_PEAK_TFLOPS if dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    try:
        r = Run(dev, 1, 0, dtype, B, size)
        for _ in range(max(warmup, 4)):
            r.step()
        dt, loss = r.timed(steps, 1, dev)
        step_s = dt / steps
        esz = 2 if dtype == "bf16" else 4
        scale = (size * size) / (352.0 * 352.0)
        mb, gf = STEP_MB_352 * scale * B * (esz / 4.0), STEP_GFLOP_352 * scale * B
        ent = {"value": round(B * steps / dt, 2), "unit": "images/sec", "ms_per_step": round(step_s * 1e3, 3), "steps": steps,
               "warmup": max(warmup, 4), "dtype": dtype, "batch": B, "image": [3, size, size],
               "final_loss": round(float(loss.detach()), 5),
               "whole_step": {"algorithmic_MB": round(mb, 1), "hbm_frac": round(mb / 1e3 / step_s / HBM_PEAK_GBS, 4),
                              "algorithmic_GFLOP": round(gf, 1), "mfma_peak_TFLOPs": _PEAK[0],
                              "mfma_frac": round(gf / 1e3 / step_s / _PEAK[0], 4)}}
        # rows A2 / A7 of this configuration (8d bytes at this storage width), timed over 3 more steps outside the timed region
        from lm_net_amd import hip
        torch.cuda.synchronize()
        hip.prof_begin("dw_|na_")
        for _ in range(3):
            r.step()
        rows = hip.prof_end()
        for key, names in (("row_A2", ("dw_fwd_kernel", "dw_bwd_kernel", "dw_stats0_kernel", "dw_stats1_kernel")),
                           ("row_A7", ("na_fwd_kernel", "na_bwd_fused_kernel", "na_bwd_q_kernel", "na_bwd_kv_kernel", "na_bwd_q_tile_kernel",
                                       "na_bwd_kv_tile_kernel"))):
            g = _group(rows, names)
            if g["total_us"] > 0:
                e = _entry(key, g)
                ent[key] = {"ms_per_step": round(g["total_us"] / 3e3, 3), "achieved": e["achieved"], "unit": "GB/s", "frac": e["frac"],
                            "frac_vs_copy_ceiling": round(e["achieved"] / COPY_CEILING_GBS, 4)}
    except Exception as e:       # never lose the headline because a side configuration misbehaved
        ent = {"value": None, "error": str(e)[:300], "dtype": dtype, "batch": B, "image": [3, size, size]}
    finally:
        _PEAK[0] = MFMA_F32_PEAK_TFLOPS
    r = None
    torch.cuda.empty_cache()
    return ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=352)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra lines for BASELINE configs[2] (bf16, batch 64) and configs[4] at one GPU (512x512, batch 32)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the dense contractions: f32 (BASELINE configs[1], the headline) or bf16 matrix-core operands "
                         "with fp32 accumulation / statistics / master weights (BASELINE configs[2]; run with --batch 64)")
    ap.add_argument("--plans", dest="plans", action="store_true", default=True,
                    help="(default) replay the step as recorded C-side schedules (LM_Net.enable_plans(): one lmn_plan_run per "
                         "pass on the same four streams; host cost ~2 ms per step instead of 16-20)")
    ap.add_argument("--no-plans", dest="plans", action="store_false", help="launch every kernel from the host (Python)")
    ap.add_argument("--graphs", action="store_true",
                    help="replay the step as two hipGraphs (wins when the host is the bottleneck, e.g. batch 1; at batch 8 "
                         "the step is GPU-bound and host launches measured 7 %% faster than the replay)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:])            # never returns; nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("LMNET_BENCH_BACKEND", "nccl")   # "gloo" only to exercise this path with several ranks on ONE GPU
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a mislabelled run"
                  % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    ngpu = torch.cuda.device_count()
    if ngpu < 1 or (world > ngpu and backend == "nccl"):
        if rank == 0:
            print("bench.py: %d rank(s) but %d GPU(s) visible (one rank per GPU over RCCL)" % (world, ngpu), file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)     # "nccl" = RCCL over xGMI
    local = local % ngpu
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.graphs:
        args.plans = False
    esz = 2 if args.dtype == "bf16" else 4
    _PEAK[0] = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
    run = Run(dev, world, rank, args.dtype, args.batch, args.size, plans=args.plans, graphs=args.graphs)
    net, step = run.net, run.step
    B, H, W = args.batch, args.size, args.size

    for _ in range(max(args.warmup, 4 if (args.graphs or args.plans) else 0)):   # graph capture happens on the 3rd step of a shape
        step()
    # ---- which kernel dominates?  Two extra untimed steps with the in-library timer on EVERY kernel launch (HIP events on
    # the launch stream, lm_net_amd/csrc/runtime.hip); graph replays run no host code, so this survey uses host launches
    from lm_net_amd import hip
    eng = net._engine
    saved_graphs = net.use_graphs
    net.use_graphs = False
    torch.cuda.synchronize()
    hip.prof_begin(None)
    for _ in range(2):
        step()
    survey = hip.prof_end()
    net.use_graphs = saved_graphs
    launches_per_step = sum(v["launches"] for v in survey.values()) // 2
    tot_us = sum(v["total_us"] for v in survey.values()) or 1.0
    ranked = sorted(survey.items(), key=lambda kv: -kv[1]["total_us"])
    # the `roofline` kernel = the largest share of GPU time among the kernels that DECLARE algorithmic cost.  The two BatchNorm
    # statistics passes of row A2 declare zero bytes (extra passes add time, not algorithmic bytes), so a roofline
    # fraction of theirs would be 0 by construction; when one of them has the largest share it is named in
    # roofline.largest_share_without_algorithmic_cost, and its time counts fully against row_A2.
    costed = [kv for kv in ranked if (kv[1]["bytes"] > 0 if kv[0].startswith(HBM_ROWS) else (kv[1]["flops"] > 0 or kv[1]["bytes"] > 0))]   # (byte-bound rows: declared BYTES)
    dominant = (costed or ranked)[0][0]
    zero_cost_top = ranked[0][0] if ranked[0][0] != dominant else None
    watch = [dominant] + [k for k in survey if k.startswith(("dw_", "na_")) or "wgrad_reduce" in k]
    flt = lambda names: "|".join(sorted(set(w.split("<")[0] for w in names)))
    # ---- timed region: ONLY the dominant kernel is timed live (HIP events around its launches: the `roofline` contract).  Timing
    # the A2 / A7 kernels there as well (~90 launches, two events each) cost the step 0.45 ms (15.2 -> 15.65 ms, LMN_BENCH_NOLIVE=1
    # for the figure without any timer): they are timed over the same number of steps right after, outside the headline.
    torch.cuda.synchronize()
    nolive = os.environ.get("LMN_BENCH_NOLIVE") == "1"      # (measurement of the timers' own cost: no kernel timed in the region)
    head = [dominant] + [k for k in survey if "wgrad_reduce" in k and "wgrad" in dominant]
    if not net.use_graphs and not nolive:
        hip.prof_begin(flt(head))
    dt, loss = run.timed(args.steps, world, dev)
    live = hip.prof_end() if not net.use_graphs and not nolive else {}
    if not net.use_graphs:
        rows = [k for k in watch if k.split("<")[0] not in set(h.split("<")[0] for h in head)] if not nolive else watch
        if rows:
            torch.cuda.synchronize()
            hip.prof_begin(flt(rows))
            for _ in range(args.steps):
                step()
            live = {**hip.prof_end(), **live}
    if net.use_graphs:      # replays run no host code: time the same kernels over 5 host-launched steps right after
        net.use_graphs = False
        hip.prof_begin("|".join(sorted(set(w.split("<")[0] for w in watch))))
        for _ in range(5):
            step()
        live = hip.prof_end()
        net.use_graphs = True
    comm = None
    if world > 1:
        red = run.model.reducer
        comm = {"backend": "RCCL (torch.distributed 'nccl')" if backend == "nccl" else backend, "ranks": dist.get_world_size(),
                "buckets_per_step": len(red.launched), "launched_before_finish": red.launched_before_finish,
                "bucket_bounds_floats": [list(b) for b in red.launched],
                "gradient_MB": round(net._grad_flat.numel() * 4 / 1e6, 2) if net._grad_flat is not None else None}
    # the same kernels with nothing else on the GPU: 3 more steps launched from the host with the branch / weight-gradient
    # streams switched off (inside the timed region a kernel shares the CUs with the other streams of the step)
    saved = (eng.branch_overlap, eng.overlap_wgrad, net.use_graphs, net.use_plans)
    eng.branch_overlap, eng.overlap_wgrad, net.use_graphs, net.use_plans = False, False, False, False
    for _ in range(2):
        step()
    hip.prof_begin("|".join(sorted(set(w.split("<")[0] for w in watch))))
    for _ in range(3):
        step()
    alone = hip.prof_end()
    eng.branch_overlap, eng.overlap_wgrad, net.use_graphs, net.use_plans = saved
    if rank == 0:
        step_s = dt / args.steps
        cfg_idx = 1 if args.dtype == "f32" else 2
        res = {
            "metric": "train images/sec at 352x352, 1/2/4/8 MI355X; Dice vs ref",
            "value": round(world * B * args.steps / dt, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "LM-Net %s training step (fwd + CE/Dice loss + bwd + AdamW), batch %d/GPU, %dx%d "
                                   "synthetic disc masks (BASELINE configs[%d])" % (
                                       "fp32" if args.dtype == "f32" else "bf16 mixed-precision (bf16 MFMA operands, fp32 accumulate / "
                                       "statistics / master weights)", B, H, W, cfg_idx if world == 1 else 3),
                       "global_batch": world * B, "image": [3, H, W], "parallelism": "dp%d" % world,
                       "launch": "hipGraph replay (fwd + bwd graphs per step)" if args.graphs else
                                 ("lmn_plan_run (recorded C-side schedule, one crossing per pass)" if args.plans else "host"),
                       "kernel_launches_per_step": launches_per_step,
                       "final_loss": round(float(loss.detach()), 5)},
            "roofline": dict(roofline_block(dominant, live, survey, tot_us, B, H, W, step_s, alone, esz),
                             largest_share_without_algorithmic_cost=(None if zero_cost_top is None else
                                                                     {"kernel": zero_cost_top, "share_of_gpu_time": round(survey[zero_cost_top]["total_us"] / tot_us, 4),
                                                                      "avg_us": round(survey[zero_cost_top]["total_us"] / max(survey[zero_cost_top]["launches"], 1), 1)})),
        }
        if comm is not None:
            res["allreduce"] = comm
    run = net = step = None
    if rank == 0:
        default_headline = world == 1 and (args.dtype, B, H) == ("f32", 8, 352)
        if default_headline and not args.no_other_configs:
            torch.cuda.empty_cache()
            # the two other single-GPU BASELINE configurations, timed in this process AFTER the headline (value / config / dtype of
            # the line stay configs[1]): configs[2] = bf16 storage + bf16 MFMA operands at batch 64; configs[4] at one GPU =
            # 512x512 inputs, batch 32
            res["other_configs"] = {"configs[2] bf16 mixed precision, batch 64, 352x352": other_config(dev, "bf16", 64, 352),
                                    "configs[4] on 1 GPU: fp32, 512x512, batch 32": other_config(dev, "f32", 32, 512)}
            res["deterministic_mode"] = deterministic_check(dev, 8, 352)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(H, W)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
