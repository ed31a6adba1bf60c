"""GPU: the schedule / launch-path / A-B-switch checks at the size `bench.py` measures (batch 8, 352x352: 480 launches on four streams,
grids that fill the chip, K-split weight gradients with deferred reductions).  Races and lost reductions found in earlier rounds
showed only at this size -- at the 2 x 96 x 128 of test_model_gpu.py the kernels are too short to overlap.

All comparisons run in deterministic mode (fixed-order reductions, lmn_set_deterministic), where a different stream interleaving or
launch path must not change one bit; the A/B switches of INTEGRATION.md section 5 change the arithmetic's grouping, so they are held
to the gradient tolerance of the oracle tests instead.  Also here: `bench.py --gpus 2` through its own launcher (two ranks on the one
test GPU over gloo) and its refusal (exit 2) of a rank count the box cannot hold."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import no_dropout, rel_err
from tools.detweights import det_input, disc_labels, fill_module

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, S = 8, 352


def _inputs():
    return det_input((B, 3, S, S), "bsz/x").cuda(), disc_labels(B, S, S).cuda()


def _train(cfg=None, steps=2, plans=False, hooks=None, det=True, dtype="fp32"):
    """`steps` training steps (fused loss, AdamW) of a freshly filled model -> (loss of the last step, its gradients, loss of the
    first step, its gradients)."""
    from lm_net_amd import LM_Net
    from lm_net_amd.loss import SegLoss
    from lm_net_amd.optim import FusedAdamW
    x, y = _inputs()
    m = LM_Net(3, 2)
    fill_module(m, 43)
    no_dropout(m)
    m = m.cuda().train()
    m.deterministic = det
    m.compute_dtype = dtype
    for k, v in (cfg or {}).items():
        assert hasattr(m._engine, k), k
        setattr(m._engine, k, v)
    if hooks is not None:
        m.grad_begin_hook = lambda flat: hooks.append(("begin", flat.numel()))
        m.grad_ready_hook = lambda lo, hi, streams=(): hooks.append((lo, hi))
        m.grad_finish_hook = lambda: hooks.append(("finish",))
    if plans:
        m.enable_plans()
    crit = SegLoss(label_smoothing=1e-3).cuda()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
    first = None
    for _ in range(steps):
        if hooks is not None:
            hooks.clear()
        loss = crit(m(x), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        grads = [p.grad.detach().clone() for p in m.parameters()]
        opt.step()
        first = first or (float(loss.detach()), grads)
    torch.cuda.synchronize()
    return (float(loss.detach()), grads) + first


@pytest.fixture(scope="module")
def names():
    from lm_net_amd import LM_Net
    return [n for n, _ in LM_Net(3, 2).named_parameters()]


@pytest.fixture(scope="module")
def serial():
    """The step launched serially on one stream (no branch chains, no weight-gradient streams), deterministic mode."""
    from lm_net_amd import hip
    try:
        return _train(dict(branch_overlap=False, overlap_wgrad=False))
    finally:
        hip.set_deterministic(False)


def _same_bits(ref, got, names, what):
    assert ref[0] == got[0], (what, ref[0], got[0])
    bad = [names[i] for i, (u, v) in enumerate(zip(ref[1], got[1])) if not torch.equal(u, v)]
    assert not bad, (what, len(bad), bad[:6])


@pytest.mark.parametrize("cfg", [{}, dict(lazy_wgrad=False), dict(overlap_wgrad=False), dict(branch_overlap=False)],
                         ids=["four-streams", "eager-wgrad", "branch-stream-only", "wgrad-streams-only"])
def test_multi_stream_equals_serial_at_bench_size(cfg, serial, names):
    """Four-stream schedule == serial schedule, bit for bit, second training step at batch 8 / 352x352 (tools/gpu_schedule_race_check.py
    as a test).  A missing cross-stream dependency, or a deferred reduction one path never flushes, is a difference."""
    from lm_net_amd import hip
    try:
        _same_bits(serial, _train(cfg), names, cfg)
    finally:
        hip.set_deterministic(False)


def test_multi_stream_equals_serial_in_bf16(names):
    """The same check in the bf16 mode (BASELINE configs[2] arithmetic: bf16 storage + bf16 MFMA operands), four repetitions of the
    four-stream schedule against one serial run.  This is the check that found the cross-kernel corruption by
    v_mfma_f32_16x16x32_bf16 (csrc/conv_common.h, mfma_bf16x2): with that instruction in the 3x3 conv / weight-gradient kernels about
    every second repetition came out with ~1 % errors in the gradients of one attention block and of the encoder behind it -- inside
    the bf16 tolerances of the oracle comparisons, visible only bit for bit."""
    from lm_net_amd import hip
    try:
        ref = _train(dict(branch_overlap=False, overlap_wgrad=False), dtype="bf16")
        for rep in range(4):
            _same_bits(ref, _train({}, dtype="bf16"), names, ("bf16 four streams", rep))
    finally:
        hip.set_deterministic(False)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_hooks_plans_and_host_launches_agree_at_bench_size(names, dtype):
    """Gradient-bucket hooks on / off x host launches / recorded plans: identical bits after five steps (the plan is recorded on the
    third and replayed after), and the reported buckets tile the flat gradient buffer contiguously from 0
    (tools/gpu_bucket_hook_check.py as a test); in the fp32 and in the bf16 mode."""
    from lm_net_amd import hip
    try:
        ref = _train(steps=5, dtype=dtype)
        for hooks, plans in ((True, False), (False, True), (True, True)):
            calls = [] if hooks else None
            got = _train(steps=5, plans=plans, hooks=calls, dtype=dtype)
            _same_bits(ref, got, names, (hooks, plans))
            if hooks:
                cov = [c for c in calls if isinstance(c[0], int)]
                assert calls[0][0] == "begin" and calls[-1] == ("finish",), calls[:2]
                assert len(cov) >= 2 and cov[0][0] == 0 and cov[-1][1] == calls[0][1], (cov[:2], cov[-1], calls[0])
                assert all(cov[i][1] == cov[i + 1][0] for i in range(len(cov) - 1)), cov
    finally:
        hip.set_deterministic(False)


@pytest.mark.parametrize("switch", [dict(fuse_bn=False), dict(split_se=False), dict(split_dw=True), dict(fuse_se=0), dict(zpath=False),
                                    dict(zpath_m=False), dict(defer_reduce=False), dict(unpad_side=False), dict(prio_main=0), dict(conv_dma=0),
                                    dict(conv_dma=3)],
                         ids=lambda d: "%s=%s" % next(iter(d.items())))
def test_ab_switches_agree_at_bench_size(switch, serial, names):
    """Every engine-level A/B switch of INTEGRATION.md section 5 computes the same step as the default configuration at batch 8 /
    352x352 (first training step): same loss to 1e-5 rel, every gradient tensor to 5e-4 of the largest gradient (the fp32 tolerance of the oracle tests;
    the switches regroup sums, they do not change what is summed)."""
    from lm_net_amd import hip
    switch = dict(switch)
    dma = switch.pop("conv_dma", None)       # (library-level switch: the kernel FORMS of lmn_conv_fwd, 0 = LDS-tiled kernels everywhere)
    prev = hip.conv_dma_config(dma, -1) if dma is not None else None
    try:
        got = _train(switch, steps=1)
    finally:
        hip.set_deterministic(False)
        if dma is not None:
            hip.conv_dma_config(prev if prev >= 0 else 7, -1)
    # (the FIRST step of both: after an AdamW update -- m / sqrt(v) = +-1 on the first step whatever the gradient's size -- rounding
    # differences of near-zero gradients become lr-sized parameter differences)
    assert abs(got[2] - serial[2]) <= 1e-5 * abs(serial[2]), (switch, got[2], serial[2])
    gmax = max(float(g.abs().max()) for g in serial[3])
    for n, u, v in zip(names, serial[3], got[3]):
        assert rel_err(u, v) < 5e-4 or float((u - v).abs().max()) < 5e-5 * gmax, (switch, n, rel_err(u, v))


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_two_ranks_through_its_own_launcher():
    """`python bench.py --gpus 2` starts its ranks itself (torch.distributed.run, 127.0.0.1) before anything touches the GPU; with
    LMNET_BENCH_BACKEND=gloo both ranks share the one test GPU.  The line must say two ranks, carry the all-reduce record, and show
    gradient buckets launched while the backward was still running."""
    p = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"], dict(LMNET_BENCH_BACKEND="gloo"))
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak", {k: r[k] for k in ("n_gpus", "steps", "warmup")}
    assert r["config"]["global_batch"] == 16 and r["config"]["parallelism"] == "dp2", r["config"]
    assert r["allreduce"]["ranks"] == 2 and r["allreduce"]["backend"] == "gloo", r["allreduce"]
    assert r["allreduce"]["launched_before_finish"] >= 2, r["allreduce"]
    #  at least three buckets per step, and their boundaries tile the 15.87 MB flat gradient buffer from 0 without
    # gaps or overlap -- what the first real 8-GPU run relies on (reference gesture: /root/reference/utils/distributed_utils.py:60-70)
    bounds = r["allreduce"]["bucket_bounds_floats"]
    assert r["allreduce"]["buckets_per_step"] >= 3 and len(bounds) == r["allreduce"]["buckets_per_step"], r["allreduce"]
    assert bounds[0][0] == 0 and all(b0[1] == b1[0] for b0, b1 in zip(bounds, bounds[1:])) and all(hi > lo for lo, hi in bounds), bounds
    assert abs(bounds[-1][1] * 4 / 1e6 - r["allreduce"]["gradient_MB"]) < 0.01 and 15.8 < r["allreduce"]["gradient_MB"] < 15.95, r["allreduce"]
    assert r["value"] > 0 and abs(r["value"] - 16 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-2 * r["value"]
    assert "cpu_baseline" not in r and r["roofline"]["kernel"]


def test_bench_refuses_more_rccl_ranks_than_gpus():
    """One rank per GPU over RCCL: `--gpus N` on a box with fewer GPUs exits 2 with a message instead of reporting a mislabelled run."""
    n = torch.cuda.device_count() + 1
    p = _bench(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-other-configs"], {}, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert "GPU(s) visible" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]
