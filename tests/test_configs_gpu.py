"""GPU: the BASELINE.json configurations beyond configs[0]/[1]-at-batch-2, each against the CPU oracle
(oracle/lmnet_ref.py, pinned to the real reference by tests/golden) or through size-independent properties:

  configs[1]  fp32 training, batch 8, 352x352          -> full step (logits + all 514 gradients + BN statistics) against
                                                          the REAL reference run in float64 (tests/golden/train_f64_*)
  configs[3]  DDP, 8 images per GPU, RCCL all-reduce   -> one rank per GPU over backend "nccl" (= RCCL); self-skips
                                                          when the box has fewer than 2 GPUs (tests/test_ddp_gpu.py
                                                          covers the same reducer with 2 ranks on ONE GPU over gloo)
  configs[4]  512x512 inputs, batch 32 per GPU         -> training step at batch 2 vs the float64 reference incl. every gradient,
                                                          and at batch 32 the properties the domain offers
                                                          (sample independence in eval, finite training step, BatchNorm
                                                          statistics advance, loss consistent with the eval shards)
"""
import os
import socket

import pytest
import torch

from helpers import no_dropout, rel_err
from tools.detweights import det_input, disc_labels, fill_module

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _golden_step(name, tol_g=4e-3, tol_affine=2.5e-2):
    """One training step (batch-stat BatchNorm, dropout off) against tests/golden/<name>.npz: the REAL reference run in
    float64 on the same name-keyed weights / inputs (tools/make_golden_f64.py).  Tolerances: logits 1e-4 (north_star);
    gradients, on the sampled entries relative to the tensor's largest element: 4e-3 for weight tensors (measured worst
    1.4e-3, tools/gpu_model_check.py ... f64) and 2.5e-2 for the 1-D parameters (BatchNorm / LayerNorm affine, biases:
    dgamma = sum dh*zhat over 10^5..10^6 pixels cancels to ~1e-3 of its terms, so fp32 rounding of the TERMS already shows:
    5e-3 .. 1.3e-2 from run to run (the statistics are summed with float atomics in arrival order; 12 runs of the batch-2
    case), and the reference's own fp32 CPU autograd is at 1.3e-2 on the same entries); 2e-3 on every tensor's L2 norm."""
    import numpy as np
    from lm_net_amd import LM_Net
    from helpers import load_golden
    from tools.make_golden_f64 import sample_index
    g = load_golden(name + ".npz")
    size, B, seed = (int(v) for v in g["meta"])
    m = LM_Net(3, 2)
    fill_module(m, seed)
    no_dropout(m)
    m = m.cuda().train()
    key = "f64_%d_b%d" % (size, B)
    x = det_input((B, 3, size, size), key + "/x").cuda().requires_grad_(True)
    y = m(x)
    yf = y.detach().flatten().cpu().double()
    ys = yf[torch.from_numpy(sample_index(yf.numel(), 32768))].numpy()
    assert float(np.abs(ys - g["logits/sample"]).max()) < TOL * float(g["logits/stat"][0])
    assert abs(float(yf.norm()) - float(g["logits/stat"][1])) < TOL * float(g["logits/stat"][1])
    (y * det_input(tuple(y.shape), key + "/G").cuda()).sum().backward()
    torch.cuda.synchronize()
    gmax = max(float(g["gstat/" + k][0]) for k, _ in m.named_parameters())
    worst = 0.0

    def check(tag, grad, stat, samp):
        nonlocal worst
        gf = grad.detach().flatten().cpu().double()
        gs = gf[torch.from_numpy(sample_index(gf.numel()))].numpy()
        err = float(np.abs(gs - samp).max())
        if err < 2e-5 * gmax:          # pre-BatchNorm biases (exact gradient 0) and other tiny tensors: absolute scale
            return
        worst = max(worst, err / float(stat[0]))
        assert err < (tol_affine if grad.dim() == 1 else tol_g) * float(stat[0]), (tag, err, float(stat[0]))
        assert abs(float(gf.norm()) - float(stat[1])) < 2e-3 * float(stat[1]), (tag, float(gf.norm()), float(stat[1]))

    check("input", x.grad, g["gx/stat"], g["gx/sample"])
    for k, p in m.named_parameters():
        check(k, p.grad, g["gstat/" + k], g["gsamp/" + k])
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_err(v, g["state/" + k]) < 1e-4, k
    return worst


def test_config1_train_step_352_batch8_vs_reference_f64():
    """BASELINE configs[1] at its real batch size (fp32 training, batch 8, 352x352): logits, all 514 gradients, the input
    gradient and the BatchNorm running statistics against the reference's float64 step."""
    _golden_step("train_f64_352_b8")


def test_config1_batch8_deterministic_mode_bit_identical_gradients():
    """BASELINE configs[1] shape in deterministic mode (LM_Net.deterministic / lmn_set_deterministic: fixed-order reductions instead
    of arrival-order float atomics): two fresh runs of the batch-8 training step give bit-identical logits and gradients -- all 514
    tensors (the default mode differs in ~510 of them, tools/gpu_determinism_check.py) -- and the step still matches the float64
    reference within the usual tolerances."""
    from lm_net_amd import LM_Net, hip

    def run():
        m = LM_Net(3, 2)
        fill_module(m, 8)
        m = m.cuda().train()             # dropout on, batch-statistics BatchNorm
        m.deterministic = True
        x = det_input((8, 3, 352, 352), "det8/x").cuda()
        y = m(x)
        (y * det_input(tuple(y.shape), "det8/G").cuda()).sum().backward()
        torch.cuda.synchronize()
        return y.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]

    try:
        (ya, ga), (yb, gb) = run(), run()
        assert torch.equal(ya, yb)
        assert all(torch.equal(u, v) for u, v in zip(ga, gb)), [i for i, (u, v) in enumerate(zip(ga, gb)) if not torch.equal(u, v)][:8]
        # (the process-wide switch is still on: the golden step runs in deterministic mode, where the statistics are summed in a fixed
        #  order -- no arrival-order spread on the 1-D parameters, so the affine tolerance is the 1.2e-2 DESIGN section 2 derives from
        #  the measured 5-6e-3, not the 2.5e-2 the default mode needs for its 1.3e-2 run-to-run worst case)
        worst = _golden_step("train_f64_352_b8", tol_affine=1.2e-2)
        print("deterministic golden step: worst sampled gradient error %.3e of the tensor's largest element" % worst)
    finally:
        hip.set_deterministic(False)


def test_train_step_352_batch2_vs_reference_f64():
    _golden_step("train_f64_352_b2")


def test_config4_train_step_512_batch2_vs_reference_f64():
    """BASELINE configs[4] resolution in TRAINING mode (GFT on 1024 tokens, every level's tile remainders differ from
    352x352)."""
    _golden_step("train_f64_512_b2")


def test_config4_batch32_512_properties():
    """BASELINE configs[4] at its per-GPU batch (32 x 3 x 512 x 512; the level-0 E-wide tensors are 805 MB each):
    eval logits of the full batch equal the logits of its shards (samples are independent in eval mode), the fused loss of the
    full batch equals its fp64 restatement, a training step
    (fused CE+Dice loss, dropout on) is finite, every parameter receives a gradient, BatchNorm statistics advance once."""
    from lm_net_amd import LM_Net
    from lm_net_amd.loss import SegLoss
    m = LM_Net(3, 2)
    fill_module(m, 25)
    m = m.cuda()
    B = 32
    x = det_input((B, 3, 512, 512), "c4b32/x").cuda()
    y = disc_labels(B, 512, 512).cuda()
    m.eval()
    with torch.no_grad():
        full = m(x)
        for lo in (0, 13, 30):
            part = m(x[lo:lo + 2].contiguous())
            assert rel_err(full[lo:lo + 2], part) < 1e-5, lo
    crit = SegLoss(label_smoothing=1e-3).cuda()
    with torch.no_grad():
        loss_eval = float(crit(full, y))
        # CE(weight [1,4], label smoothing) + Dice(weight [1,4]) of utils/train_eval_utils.py:141 restated in fp64
        ref = float(_ref_loss(full.double(), y))
    assert abs(loss_eval - ref) < 1e-5 * max(1.0, abs(ref))
    del full
    m.train()
    rm0 = m.conv1[0].expand_conv[1].running_mean.clone()
    out = m(x)
    loss = crit(out, y)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and bool(torch.isfinite(loss))
    for k, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    assert sum(float(p.grad.abs().sum()) > 0 for p in m.parameters()) >= 510
    bn = m.conv1[0].expand_conv[1]
    assert int(bn.num_batches_tracked) == 1 and float((bn.running_mean - rm0).abs().max()) > 0


def _ref_loss(logits, target, eps=1e-3):
    import torch.nn.functional as F
    ce = F.cross_entropy(logits, target, weight=torch.tensor([1.0, 4.0], device=logits.device, dtype=logits.dtype),
                         label_smoothing=eps)
    p = torch.softmax(logits, dim=1)
    dl = 0.0
    for i, w in enumerate((1.0, 4.0)):
        t = (target == i).to(logits.dtype)
        dl = dl + (1 - (2 * (p[:, i] * t).sum() + 1e-5) / ((p[:, i] ** 2).sum() + (t * t).sum() + 1e-5)) * w
    return ce + dl / 2


# ------------------------------------------------------------------------------------------------ configs[2]: bf16, batch 64
def test_config2_bf16_batch64_352_full_size_properties():
    """BASELINE configs[2] at its real size (bf16 storage + bf16 MFMA operands, batch 64, 352x352): one training step.  The
    loss equals the fp32 path's (itself pinned to the reference's float64 goldens) to bf16 tolerance, every parameter gets a
    finite gradient, the gradients agree with the fp32 path's in L2 (median / worst over the 514 tensors), BatchNorm statistics
    advance once.  (A float64 run of the reference at batch 64 takes minutes of CPU time and 40 GB: the fp32 path stands in.)"""
    from lm_net_amd import LM_Net
    from lm_net_amd.loss import SegLoss
    B = 64
    x = det_input((B, 3, 352, 352), "c2b64/x").cuda()
    y = disc_labels(B, 352, 352).cuda()
    crit = SegLoss(label_smoothing=1e-3).cuda()
    res = {}
    for mode in ("fp32", "bf16"):
        m = LM_Net(3, 2)
        fill_module(m, 31)
        no_dropout(m)
        m = m.cuda().train()
        m.compute_dtype = mode
        rm0 = m.conv1[0].expand_conv[1].running_mean.clone()
        out = m(x)
        loss = crit(out, y)
        loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all() and bool(torch.isfinite(loss)), mode
        bn = m.conv1[0].expand_conv[1]
        assert int(bn.num_batches_tracked) == 1 and float((bn.running_mean - rm0).abs().max()) > 0, mode
        res[mode] = (float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, bn.running_mean.clone())
        del m, out, loss
        torch.cuda.empty_cache()
    l32, g32, rm32 = res["fp32"]
    l16, g16, rm16 = res["bf16"]
    assert abs(l16 - l32) < 2e-2 * abs(l32), (l16, l32)
    assert rel_err(rm16, rm32) < 1e-2
    errs = []
    gmax = max(float(g.abs().max()) for g in g32.values())
    for k, g in g16.items():
        assert bool(torch.isfinite(g).all()), k
        ref = g32[k]
        if float(ref.abs().max()) < 2e-5 * gmax:          # pre-BatchNorm biases: exact gradient 0
            continue
        errs.append(float((g - ref).norm() / (ref.norm() + 1e-30)))
    assert sum(float(g.abs().sum()) > 0 for g in g16.values()) >= 510
    errs.sort()
    assert errs[len(errs) // 2] < 0.1 and errs[-1] < 0.5, (errs[len(errs) // 2], errs[-1])


# ------------------------------------------------------------------------------------------------ configs[3]: RCCL
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rccl_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
        from lm_net_amd import LM_Net
        from lm_net_amd.ddp import DistributedLMNet
        net = LM_Net(3, 2)                                      # the real (default) configuration
        fill_module(net, seed=rank)                             # different weights per rank before wrapping ...
        net = net.cuda().train()
        no_dropout(net)
        x = det_input((8, 3, 96, 128), "rccl/x%d" % rank).cuda()    # configs[3]: 8 images per GPU, rank-specific shard
        model = DistributedLMNet(net)                           # ... replicated from rank 0 here (RCCL broadcast)
        hooks = (net.grad_begin_hook, net.grad_ready_hook, net.grad_finish_hook)
        net.grad_begin_hook = net.grad_ready_hook = net.grad_finish_hook = None
        net(x).square().mean().backward()                       # local (un-reduced) gradients
        local = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        net.zero_grad(set_to_none=True)
        net.grad_begin_hook, net.grad_ready_hook, net.grad_finish_hook = hooks
        model(x).square().mean().backward()
        torch.cuda.synchronize()
        reduced = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect = sum(gathered) / world
        err = float((reduced - expect).abs().max() / (expect.abs().max() + 1e-30))
        q.put((rank, err, len(model.reducer.launched), model.reducer.launched_before_finish))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        q.put((rank, repr(e), None, None))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="configs[3] needs >= 2 GPUs: RCCL wants one GPU per rank "
                    "(this box has %d); the 2-ranks-on-one-GPU gloo variant is tests/test_ddp_gpu.py" % torch.cuda.device_count())
def test_config3_rccl_allreduce_overlapped_with_backward():
    """One process per GPU, backend "nccl" (RCCL over xGMI), real LM_Net backward at 8 images per GPU: every rank ends
    with the MEAN of the per-rank gradients, and at least two buckets were handed to RCCL BEFORE the backward schedule
    finished (overlap by construction: the collective stream only waits for the producers of its bucket)."""
    import torch.multiprocessing as mp
    world, port = min(torch.cuda.device_count(), 8), _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=600) for _ in range(world)]
    [p.join(60) for p in ps]
    for rank, err, nb, early in res:
        assert isinstance(err, float), "rank %d failed: %s" % (rank, err)
        assert err < 1e-4, "rank %d: reduced gradient != mean of local gradients (rel %.3e)" % (rank, err)
        assert nb >= 2 and early >= 2, "rank %d: %s buckets, %s launched before the end of backward" % (rank, nb, early)
