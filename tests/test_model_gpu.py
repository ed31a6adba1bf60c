"""GPU: lm_net_amd.LM_Net (HIP kernels, called through the C-ABI) against
  (a) the golden vectors produced by the REAL reference (tests/golden, made by tools/make_golden.py), and
  (b) the CPU oracle on the same seeded inputs.
Tolerances (north_star): logits <= 1e-4 rel fp32; Dice/IoU identical to 4 dp; gradients 5e-4 rel (fp32
atomically-reduced sums vs the reference's fp32 CPU sums)."""
import numpy as np
import pytest
import torch

from helpers import digest, is_pre_bn_bias, load_golden, no_dropout, rel_err
from tools.detweights import det_input, disc_labels, fill_module
from tools.metrics_ref import dice_iou

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _net(filters=None, seed=0):
    from lm_net_amd import LM_Net
    m = LM_Net(3, 2) if filters is None else LM_Net(3, 2, filters=filters)
    fill_module(m, seed)
    no_dropout(m)
    return m.cuda()


def test_native_library_is_loaded_and_no_cpu_path():
    from lm_net_amd import LM_Net, hip
    hip.load()
    m = LM_Net(3, 2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32))                       # CPU tensors are refused: no fallback path
    import ctypes  # the in-tree .so must be the thing that is mapped into this process
    assert "liblmnet_hip.so" in open("/proc/self/maps").read()


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_tiny_config_matches_reference_golden(mode):
    g = load_golden("tiny_%s.npz" % mode)
    m = _net([12] * 5)
    m.train(mode == "train")
    x = det_input((1, 3, 32, 48), "tiny/x").cuda().requires_grad_(True)
    y = m(x)
    assert rel_err(y, g["logits"]) < TOL
    (y * det_input(tuple(y.shape), "tiny/G").cuda()).sum().backward()
    assert rel_err(x.grad, g["grad_input"]) < 5e-4
    gmax = max(float(np.abs(g["grad/" + k]).max()) for k, _ in m.named_parameters())
    for k, p in m.named_parameters():
        ref = g["grad/" + k]
        err = float(np.abs(p.grad.cpu().numpy() - ref).max())
        assert err < 5e-4 * float(np.abs(ref).max()) or err < 1e-5 * gmax, (k, err)
    if mode == "train":
        for k, v in m.state_dict().items():
            if "running_" in k:
                assert rel_err(v, g["state/" + k]) < 1e-5, k
            if "num_batches" in k:
                assert int(v) == 1


def test_default_config_64x96_eval_stages_deploy_train():
    g = load_golden("default_64x96.npz")
    m = _net()
    m.eval()
    m._keep_taps = True
    x = det_input((2, 3, 64, 96), "d64/x").cuda()
    with torch.no_grad():
        y = m(x)
    assert rel_err(y, g["logits"]) < TOL
    for k, v in m._taps.items():
        assert rel_err(v.permute(0, 3, 1, 2), g["stage/" + k]) < TOL, k
    m.train()
    xg = x.clone().requires_grad_(True)
    yt = m(xg)
    assert rel_err(yt, g["train_logits"]) < TOL
    (yt * det_input(tuple(yt.shape), "d64/G").cuda()).sum().backward()
    assert rel_err(xg.grad, g["train_grad_input"]) < 5e-4
    for k, p in m.named_parameters():
        if is_pre_bn_bias(k):
            continue
        d, r = digest(p.grad), g["gdig/" + k]
        assert abs(d[2] - r[2]) <= 5e-4 * r[2] + 1e-9, k
        if "grad/" + k in g:
            assert rel_err(p.grad, g["grad/" + k]) < 1e-3, k
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_err(v, g["state/" + k]) < 1e-5, k
    # structural_reparam(): deploy form == eval form (the reference's own invariant)
    m2 = _net()
    m2.eval()
    m2.structural_reparam()
    assert len(m2.state_dict()) == 510
    with torch.no_grad():
        yd = m2(x)
    assert rel_err(yd, g["deploy_logits"]) < TOL and rel_err(yd, g["logits"]) < TOL


def test_default_config_352_logits_and_dice():
    """BASELINE.json configs[0] shape (1x3x352x352): logits <= 1e-4 rel, Dice/IoU identical to 4 dp."""
    g = load_golden("default_352.npz")
    m = _net()
    m.eval()
    x = det_input((1, 3, 352, 352), "d352/x").cuda()
    with torch.no_grad():
        y = m(x)
    assert rel_err(y, g["logits"]) < TOL
    pred = y.argmax(1).cpu()
    dice, iou = dice_iou(pred, disc_labels(1, 352, 352))
    assert round(dice, 4) == round(float(g["dice"][0]), 4) and round(iou, 4) == round(float(g["iou"][0]), 4)
    assert abs(int(pred.sum()) - int(g["pred_sum"][0])) <= 2          # argmax ties at the decision boundary


def test_train_step_352_batch2_vs_oracle():
    """Full-size images, batch-stat BN: forward + every gradient against the CPU (fp32) oracle.  Gradient
    tolerance 5e-3 (L2 per tensor; 1e-2 for the single worst element): at 352x352 both sides sum ~250k fp32 terms per weight through BatchNorm cancellations and
    kinked activation derivatives; against an fp64 oracle (tools/gpu_model_check.py ... f64) the HIP path is
    within ~1e-3 and the fp32 CPU oracle itself is no closer."""
    from oracle.lmnet_ref import LM_Net as Oracle
    ora = Oracle(3, 2)
    fill_module(ora, 5)
    no_dropout(ora)
    m = _net(seed=5)
    x = det_input((2, 3, 352, 352), "t352/x")
    ora.train(); m.train()
    yo = ora(x)
    yg = m(x.cuda())
    assert rel_err(yg, yo) < TOL
    G = det_input(tuple(yo.shape), "t352/G")
    (yo * G).sum().backward()
    (yg * G.cuda()).sum().backward()
    gmax = max(float(p.grad.abs().max()) for p in ora.parameters())
    for (k, po), (_, pg) in zip(ora.named_parameters(), m.named_parameters()):
        d = (pg.grad.cpu() - po.grad).double()
        err, l2 = float(d.abs().max()), float(d.norm() / (po.grad.double().norm() + 1e-30))
        # max-abs within 1e-2 of the tensor's largest gradient (one fp32-noise outlier must not fail the run: both
        # sides use float atomics / different summation orders) AND the whole tensor within 5e-3 in L2
        small = err < 2e-4 * gmax
        assert (err < 1e-2 * float(po.grad.abs().max()) and l2 < 5e-3) or small, (k, err, l2)


def test_dropout_train_mode_is_active_and_consistent():
    """Dropout (p=0.1, Mlp) is live in train mode: outputs differ between steps, stay finite, and the
    backward regenerates the same masks (finite gradients of the right magnitude)."""
    from lm_net_amd import LM_Net
    m = LM_Net(3, 2, filters=[12] * 5)
    fill_module(m)
    m.cuda().train()
    x = det_input((2, 3, 32, 48), "drop/x").cuda()
    y1 = m(x)
    y2 = m(x)
    assert torch.isfinite(y1).all() and (y1 - y2).abs().max() > 1e-4
    y1.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.eval()
    with torch.no_grad():
        assert (m(x) - m(x)).abs().max() < 1e-5      # eval: no dropout (float-atomic SE sums may differ in the last bits)


def test_eval_512x512_vs_oracle():
    """BASELINE configs[4] resolution (512x512; GFT runs on 1024 tokens): eval forward against the CPU oracle."""
    from oracle.lmnet_ref import LM_Net as Oracle
    ora = Oracle(3, 2)
    fill_module(ora, 7)
    ora.eval()
    m = _net(seed=7)
    m.eval()
    x = det_input((1, 3, 512, 512), "t512/x")
    with torch.no_grad():
        yo = ora(x)
        yg = m(x.cuda())
    assert rel_err(yg, yo) < TOL
    assert (yg.argmax(1).cpu() != yo.argmax(1)).float().mean() < 1e-4


def test_non_square_and_odd_tile_sizes_vs_oracle():
    """H, W only need to be multiples of 16: 48x80 exercises partial tiles at every level (3x5 .. 48x80)."""
    from oracle.lmnet_ref import LM_Net as Oracle
    ora = Oracle(3, 2)
    fill_module(ora, 9)
    no_dropout(ora)
    m = _net(seed=9)
    x = det_input((3, 3, 48, 80), "odd/x")
    ora.train(); m.train()
    xo = x.clone().requires_grad_(True)
    xg = x.cuda().requires_grad_(True)
    yo, yg = ora(xo), m(xg)
    assert rel_err(yg, yo) < TOL
    G = det_input(tuple(yo.shape), "odd/G")
    (yo * G).sum().backward()
    (yg * G.cuda()).sum().backward()
    assert rel_err(xg.grad, xo.grad) < 2e-3
    gmax = max(float(p.grad.abs().max()) for p in ora.parameters())
    for (k, po), (_, pg) in zip(ora.named_parameters(), m.named_parameters()):
        err = float((pg.grad.cpu() - po.grad).abs().max())
        assert err < 2e-3 * float(po.grad.abs().max()) or err < 5e-5 * gmax, (k, err)


def test_zpath_gate_per_block_vs_oracle():
    """`lmn_reparam_fold` stages two W_e panels of 35 E floats in 64 KB of LDS, so engine.reparam_fwd takes the z-path per block only
    while 35 * E * 4 <= engine.zpath_lds and the two-pass BatchNorm form beyond (E > 468; the model's other kernels stop at 512
    channels, so no LM_Net reaches it: here the limit is lowered to E = 48, which puts levels 0-1 on the z-path and levels 2-3 on
    the other form IN ONE STEP).  Logits and every gradient of a training step against the CPU oracle."""
    from oracle.lmnet_ref import LM_Net as Oracle
    from lm_net_amd import hip
    ora = Oracle(3, 2)
    fill_module(ora, 13)
    no_dropout(ora)
    m = _net(seed=13)
    m._engine.zpath_lds = 35 * 48 * 4
    x = det_input((2, 3, 32, 48), "gate/x")
    ora.train(); m.train()
    xo = x.clone().requires_grad_(True)
    xg = x.cuda().requires_grad_(True)
    hip.prof_begin("reparam_fold|dw_stats0")
    yo, yg = ora(xo), m(xg)
    assert rel_err(yg, yo) < TOL
    G = det_input(tuple(yo.shape), "gate/G")
    (yo * G).sum().backward()
    (yg * G.cuda()).sum().backward()
    prof = hip.prof_end()
    folds = sum(v["launches"] for k, v in prof.items() if k.startswith("reparam_fold"))
    assert folds == 8, prof          # the 2 x 2 encoder + 2 x 2 decoder blocks of levels 0-1; the other 8 blocks took the other path
    assert rel_err(xg.grad, xo.grad) < 2e-3
    gmax = max(float(p.grad.abs().max()) for p in ora.parameters())
    for (k, po), (_, pg) in zip(ora.named_parameters(), m.named_parameters()):
        err = float((pg.grad.cpu() - po.grad).abs().max())
        assert err < 4e-3 * float(po.grad.abs().max()) or err < 5e-5 * gmax, (k, err)


@pytest.mark.parametrize("K", [5, 7])
def test_larger_neighborhood_window_vs_oracle(K):
    """natten kernel_size 5 / 7 in the four NAT blocks (the reference's LM_Net signature carries [3, 5], core/LM_Net.py:81-84; it
    constructs 3, core/modules.py:509): whole model, training mode, logits and every gradient against the CPU oracle.  112x128:
    the coarsest NAT map is 14x16 >= K."""
    from oracle.lmnet_ref import LM_Net as Oracle
    from lm_net_amd import LM_Net
    ora = Oracle(3, 2, na_kernel_size=K)
    fill_module(ora, 21)
    no_dropout(ora)
    m = LM_Net(3, 2, na_kernel_size=K)
    fill_module(m, 21)
    no_dropout(m)
    m = m.cuda()
    assert m.natt4.att1.rpb.shape == (12, 2 * K - 1, 2 * K - 1)
    x = det_input((2, 3, 112, 128), "nak/x")
    ora.train(); m.train()
    yo, yg = ora(x), m(x.cuda())
    assert rel_err(yg, yo) < TOL
    G = det_input(tuple(yo.shape), "nak/G")
    (yo * G).sum().backward()
    (yg * G.cuda()).sum().backward()
    gmax = max(float(p.grad.abs().max()) for p in ora.parameters())
    for (k, po), (_, pg) in zip(ora.named_parameters(), m.named_parameters()):
        err = float((pg.grad.cpu() - po.grad).abs().max())
        assert err < 2e-3 * float(po.grad.abs().max()) or err < 5e-5 * gmax, (k, err)


def test_deterministic_mode_is_bit_reproducible():
    """model.deterministic = True (lmn_set_deterministic): fixed-order reductions everywhere -- two fresh runs of three training steps
    (batch-statistics BatchNorm, dropout on, fused loss, AdamW; launch by launch and as recorded plans) give BIT-IDENTICAL losses,
    gradients, parameters and running statistics; the default mode agrees with it to the float-atomic noise level."""
    from lm_net_amd import LM_Net, hip
    from lm_net_amd.loss import SegLoss
    from lm_net_amd.optim import FusedAdamW
    x = det_input((2, 3, 96, 128), "det/x").cuda()
    y = disc_labels(2, 96, 128).cuda()

    def run(det, plans, steps, drop=True):
        m = LM_Net(3, 2)
        fill_module(m, 31)
        if not drop:
            no_dropout(m)
        m = m.cuda().train()
        m.deterministic = det
        if plans:
            m.enable_plans()
        crit = SegLoss(label_smoothing=1e-3).cuda()
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
        losses = []
        for _ in range(steps):
            loss = crit(m(x), y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            grads = [p.grad.detach().clone() for p in m.parameters()]
            opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        return losses, grads, [p.detach().clone() for p in m.parameters()], [b.detach().clone() for b in m.buffers()]

    try:
        for plans, steps in ((False, 3), (True, 5)):
            a, b = run(True, plans, steps), run(True, plans, steps)
            for u, v in zip(a[0], b[0]):
                assert torch.equal(u, v), (plans, float(u), float(v))
            for part in (1, 2, 3):
                for i, (u, v) in enumerate(zip(a[part], b[part])):
                    assert torch.equal(u, v), (plans, part, i, float((u - v).abs().max()))
        assert hip.get_deterministic()
        # recorded plans replay the SAME launches: in deterministic mode their results equal the host-launched ones bit for bit (the
        # strict form of test_plan_mode_matches_host_launches, whose tolerance has to cover the arrival-order noise of the default mode)
        # (dropout off here: plans draw their masks from a device-side counter, host launches from host seeds)
        e5, p5 = run(True, False, 5, drop=False), run(True, True, 5, drop=False)
        for u, v in zip(e5[0], p5[0]):
            assert torch.equal(u, v), ("eager vs plans", float(u), float(v))
        for part in (1, 2, 3):
            for i, (u, v) in enumerate(zip(e5[part], p5[part])):
                assert torch.equal(u, v), ("eager vs plans", part, i, float((u - v).abs().max()))
        ref = run(False, False, 1)
        one = run(True, False, 1)
        assert abs(float(ref[0][0]) - float(one[0][0])) < 1e-4 * abs(float(ref[0][0]))
        gmax = max(float(g.abs().max()) for g in ref[1])
        for i, (u, v) in enumerate(zip(ref[1], one[1])):
            assert rel_err(u, v) < 2e-2 or float((u - v).abs().max()) < 1e-5 * gmax, (i, rel_err(u, v))
    finally:
        hip.set_deterministic(False)
    assert not hip.get_deterministic()


def test_multi_stream_schedule_equals_serial_schedule_bit_for_bit():
    """Deterministic mode as a race detector: the training step on the four-stream schedule (branch chains, weight-gradient streams,
    late weight-gradient issue) and the same step launched serially on one stream run the same kernels on the same grids -- only the
    streams and the interleaving differ -- so their losses and all gradients must be bit-identical; a missing dependency between
    streams, or a reduction that is never flushed on one of the paths (the serial path once lost its deferred weight-gradient
    reductions on the default stream), shows as a difference."""
    from lm_net_amd import LM_Net, hip
    x = det_input((2, 3, 96, 128), "race/x").cuda()
    G = det_input((2, 2, 96, 128), "race/G").cuda()

    def run(**cfg):
        m = LM_Net(3, 2)
        fill_module(m, 37)
        no_dropout(m)
        m = m.cuda().train()
        m.deterministic = True
        for k, v in cfg.items():
            setattr(m._engine, k, v)
        out = []
        for _ in range(2):
            for p in m.parameters():
                p.grad = None
            y = m(x)
            (y * G).sum().backward()
            torch.cuda.synchronize()
            out.append((y.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]))
        return out

    try:
        ref = run(branch_overlap=False, overlap_wgrad=False)
        nz = sum(1 for g in ref[1][1] if float(g.abs().max()) > 0)
        assert nz >= len(ref[1][1]) - 40, nz          # (only the pre-BatchNorm biases have an exactly zero gradient)
        names = [n for n, _ in LM_Net(3, 2).named_parameters()]
        for cfg in ({}, dict(lazy_wgrad=False), dict(overlap_wgrad=False), dict(branch_overlap=False)):
            got = run(**cfg)
            for (yr, gr), (yg, gg) in zip(ref, got):
                assert torch.equal(yr, yg), cfg
                bad = [names[i] for i, (u, v) in enumerate(zip(gr, gg)) if not torch.equal(u, v)]
                assert not bad, (cfg, len(bad), bad[:6])
    finally:
        hip.set_deterministic(False)


@pytest.mark.parametrize("shape", [(1, 32, 32), (2, 32, 48), (5, 48, 48)])
def test_minimum_and_ragged_sizes_vs_oracle(shape):
    """Edge of the domain: the smallest input the five-level pyramid admits (32x32: a 2x2 bottleneck map, BatchNorm batch
    statistics over 4 .. 1024 pixels, 4x4 neighborhood-attention maps with every window clamped), batch 1, an odd batch, strips and
    tiles that are mostly padding -- eval and training mode, logits and every gradient against the CPU oracle."""
    from oracle.lmnet_ref import LM_Net as Oracle
    B, H, W = shape
    ora = Oracle(3, 2)
    fill_module(ora, 9)
    no_dropout(ora)
    m = _net(seed=9)
    x = det_input((B, 3, H, W), "edge/x")
    G = det_input((B, 2, H, W), "edge/G")
    for train in (False, True):
        ora.train(train); m.train(train)
        for p in list(ora.parameters()) + list(m.parameters()):
            p.grad = None
        xo = x.clone().requires_grad_(True)
        xg = x.cuda().requires_grad_(True)
        yo, yg = ora(xo), m(xg)
        assert rel_err(yg, yo) < TOL, (shape, train)
        (yo * G).sum().backward()
        (yg * G.cuda()).sum().backward()
        assert rel_err(xg.grad, xo.grad) < 2e-3, (shape, train)
        gmax = max(float(p.grad.abs().max()) for p in ora.parameters())
        for (k, po), (_, pg) in zip(ora.named_parameters(), m.named_parameters()):
            err = float((pg.grad.cpu() - po.grad).abs().max())
            assert err < 4e-3 * float(po.grad.abs().max()) or err < 5e-5 * gmax, (shape, train, k, err)


def test_fused_adamw_matches_torch_adamw_and_exchanges_state():
    """lm_net_amd.optim.FusedAdamW (one kernel over the flat buffers) against torch.optim.AdamW -- the reference's
    optimizer (train.py:156) -- on the same model, data and loss: parameters after 3 steps, then a state_dict
    hand-over from the torch optimizer and one more step."""
    from lm_net_amd.optim import FusedAdamW
    a, b = _net(seed=3), _net(seed=3)       # dropout off (helpers.no_dropout): both models see identical graphs
    a.train(); b.train()
    oa = torch.optim.AdamW(a.parameters(), lr=1e-3, weight_decay=1e-4)
    ob = FusedAdamW(b, lr=1e-3, weight_decay=1e-4)
    x = det_input((2, 3, 64, 64), "adamw/x").cuda()
    G = det_input((2, 2, 64, 64), "adamw/G").cuda()

    def run(m, o, n, ref=None, oref=None):
        """n training steps of model m; the torch optimizer `oref` of model `ref` is fed the SAME gradient values
        (Adam's m/sqrt(v) turns the last-bit noise of two separate backward passes into O(lr) differences)."""
        for _ in range(n):
            o.zero_grad(set_to_none=True)
            (m(x) * G).sum().backward()
            if ref is not None:
                for pr, pm in zip(ref.parameters(), m.parameters()):
                    pr.grad = pm.grad.detach().clone()
                oref.step()
            o.step()

    run(b, ob, 3, a, oa)
    for (n1, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_err(p2, p1) < 2e-6, n1
    # hand the torch optimizer's state to a fresh fused optimizer on a copy of model a, continue both
    c = _net(seed=3)
    c.load_state_dict(a.state_dict())
    c.train()
    oc = FusedAdamW(c, lr=1e-3, weight_decay=1e-4)
    oc.load_state_dict(oa.state_dict())
    assert oc.step_count == 3
    run(c, oc, 1, a, oa)
    for (n1, p1), (_, p2) in zip(a.named_parameters(), c.named_parameters()):
        assert rel_err(p2, p1) < 2e-6, n1
    sd = oc.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0


def _ref_loss(logits, target, eps=1e-3):
    """torch restatement of utils/train_eval_utils.py:141 (CE weight [1,4] + label smoothing, DiceLoss weight [1,4])."""
    import torch.nn.functional as F
    ce = F.cross_entropy(logits, target, weight=torch.tensor([1.0, 4.0], device=logits.device, dtype=logits.dtype),
                         label_smoothing=eps)
    p = torch.softmax(logits, dim=1)
    dl = 0.0
    for i, w in enumerate((1.0, 4.0)):
        t = (target == i).to(logits.dtype)
        dl = dl + (1 - (2 * (p[:, i] * t).sum() + 1e-5) / ((p[:, i] ** 2).sum() + (t * t).sum() + 1e-5)) * w
    return ce + dl / 2


def test_fused_seg_loss_matches_torch_ce_plus_dice():
    from lm_net_amd.loss import SegLoss
    lg = (det_input((3, 2, 40, 56), "loss/lg") * 3).cuda()
    y = disc_labels(3, 40, 56).cuda()
    l64 = lg.double().requires_grad_(True)
    ref = _ref_loss(l64, y)
    ref.backward()
    l32 = lg.clone().requires_grad_(True)
    out = SegLoss(label_smoothing=1e-3).cuda()(l32, y)
    (out * 1.5).backward()
    assert abs(float(out) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    assert rel_err(l32.grad, 1.5 * l64.grad) < 1e-5
    # the reference's call shape: labels.unsqueeze(1).float() for the Dice term
    out2 = SegLoss(label_smoothing=1e-3).cuda()(lg, y.unsqueeze(1).float())
    assert abs(float(out2) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))


def test_on_device_confusion_dice_iou():
    from lm_net_amd.metrics import ConfusionMeter
    lg = det_input((2, 2, 64, 48), "metric/lg").cuda()
    y = disc_labels(2, 64, 48).cuda()
    m = ConfusionMeter(2)
    m.update(lg[:1], y[:1])
    m.update(lg[1:], y[1:])
    r = m.compute()
    d, i = dice_iou(lg.argmax(1).cpu(), y.cpu())
    assert abs(r["dice"][1] - d) < 1e-12 and abs(r["iou"][1] - i) < 1e-12
    assert sum(map(sum, r["confusion"])) == 2 * 64 * 48


def test_graph_mode_matches_host_launches_and_redraws_dropout():
    """LM_Net.enable_graphs(): the training step as two hipGraph replays (forward, backward) per input shape.
    Same losses as host launches over 6 steps (steps 3+ are replays), BatchNorm running statistics advance once
    per step, and with dropout on two consecutive replays draw different masks (device-side stream counter)."""
    from lm_net_amd.loss import SegLoss
    from lm_net_amd.optim import FusedAdamW
    x = det_input((2, 3, 64, 96), "graph/x").cuda()
    y = disc_labels(2, 64, 96).cuda()
    crit = SegLoss(label_smoothing=1e-3).cuda()

    def run(m, n):
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
        out = []
        for _ in range(n):
            loss = crit(m(x), y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        return out

    a, b = _net(seed=11).train(), _net(seed=11).train()
    b.enable_graphs()
    la, lb = run(a, 6), run(b, 6)
    assert sum(g.fwd is not None for g in b._graphs.values()) == 1
    # (Adam turns the last-bit noise of two runs into O(lr) parameter differences: 2e-3 on the loss after six steps;
    #  a wrong gradient or a stale buffer in the replay shows up at the 1e-1 level)
    # (default mode: float-atomic noise grows over the Adam steps; the strict form -- bit-identical losses, gradients and parameters in
    #  deterministic mode -- is test_deterministic_mode_is_bit_reproducible)
    assert max(abs(u - v) / abs(u) for u, v in zip(la, lb)) < 5e-3, (la, lb)
    nb = [m for m in b.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
    na_ = [m for m in a.modules() if isinstance(m, torch.nn.BatchNorm2d)][0]
    assert int(nb.num_batches_tracked) == int(na_.num_batches_tracked) == 6
    assert rel_err(nb.running_mean, na_.running_mean) < 5e-2     # six Adam steps apart in the last bits of the gradients
    # dropout on (a fresh model keeps nn.Dropout(0.1) of the Mlp blocks): replays must not repeat the mask
    from lm_net_amd import LM_Net
    c = LM_Net(3, 2)
    fill_module(c, 11)
    c = c.cuda().train().enable_graphs()
    outs = []
    for _ in range(5):
        o = c(x)
        o.sum().backward()
        outs.append(o.detach().clone())
        for p in c.parameters():
            p.grad = None
    assert float((outs[3] - outs[4]).abs().max()) > 1e-3


def test_inference_graph_replay_matches_eager_and_follows_the_weights():
    """Eval forward as one hipGraph replay per input shape: same logits as host launches, a
    parameter update between calls is honoured by the replay (packing / BN folding are kernels inside the graph),
    and structural_reparam() drops the captured graphs."""
    x = det_input((2, 3, 64, 96), "igraph/x").cuda()
    m = _net(seed=5).eval()
    with torch.no_grad():
        y0 = m(x)
        m.enable_graphs()
        ys = [m(x) for _ in range(4)]                       # 2 eager warm-ups, capture, replay
        assert sum(g.fwd is not None for g in m._graphs.values()) == 1
        for y in ys:
            assert rel_err(y, y0) < 1e-5
        x2 = det_input((2, 3, 64, 96), "igraph/x2").cuda()
        m.enable_graphs(False)
        y2 = m(x2)
        m.enable_graphs()
        m._graphs = {}
        for _ in range(3):
            m(x)
        assert rel_err(m(x2), y2) < 1e-5                     # new input through the static buffer
        m.output_layer.weight.mul_(2.0)                      # in-place update of a parameter the graph reads
        m.output_layer.bias.mul_(2.0)
        assert rel_err(m(x2), 2.0 * y2) < 1e-5
        m.structural_reparam()
        assert m._graphs == {}
        yd = m(x2)
        assert rel_err(yd, 2.0 * y2) < 1e-4


def test_plan_mode_matches_host_launches():
    """LM_Net.enable_plans(): every pass of a repeated shape as ONE lmn_plan_run (include/lmnet_hip.h) on the same four
    streams as the host-launched schedule.  Same losses as host launches over 7 steps (steps 3+ are replays), BatchNorm
    statistics advance once per step, gradients land in one static buffer that FusedAdamW consumes without a copy,
    dropout masks are redrawn per replay, an eval forward in between and gradient accumulation behave as eager."""
    from lm_net_amd.loss import SegLoss
    from lm_net_amd.optim import FusedAdamW
    x = det_input((2, 3, 64, 96), "plan/x").cuda()
    y = disc_labels(2, 64, 96).cuda()
    crit = SegLoss(label_smoothing=1e-3).cuda()

    def run(m, n):
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-4)
        out = []
        for _ in range(n):
            loss = crit(m(x), y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            out.append(float(loss.detach()))
        return out, opt

    a, b = _net(seed=13).train(), _net(seed=13).train()
    b.enable_plans()
    (la, _), (lb, ob) = run(a, 7), run(b, 7)
    ps = [p for p in b._plans.values() if p.fwd is not None]
    assert len(ps) == 1 and ps[0].bwd is not None and ps[0].fwd.size() > 150 and ps[0].bwd.size() > 300
    # (default mode: float-atomic noise grows over the Adam steps; the strict form -- bit-identical losses, gradients and parameters in
    #  deterministic mode -- is test_deterministic_mode_is_bit_reproducible)
    assert max(abs(u - v) / abs(u) for u, v in zip(la, lb)) < 5e-3, (la, lb)
    assert ob._flat_grad().data_ptr() == ps[0].flat.data_ptr()          # the optimizer reads the static buffer in place
    na_, nb = [[m for m in n.modules() if isinstance(m, torch.nn.BatchNorm2d)][0] for n in (a, b)]
    assert int(nb.num_batches_tracked) == int(na_.num_batches_tracked) == 7
    assert rel_err(nb.running_mean, na_.running_mean) < 5e-2
    # exactness of a replay: same weights, same input -> the replayed step reproduces the host-launched gradients
    c, d = _net(seed=17).train(), _net(seed=17).train()
    d.enable_plans()
    G = det_input((2, 2, 64, 96), "plan/G").cuda()
    for it in range(5):
        for m in (c, d):
            for p in m.parameters():
                p.grad = None
            (m(x) * G).sum().backward()
    gmax = max(float(p.grad.abs().max()) for p in c.parameters())

    def same(u, v, tol=2e-3):
        return rel_err(u, v) < tol or float((u - v).abs().max()) < 1e-5 * gmax

    def agree(pairs):
        # Two runs of the SAME schedule already differ by more than float-atomic noise now and then: Hardswish' jumps by 1/2 at
        # |h| = 3 (ATen's hardswish_backward likewise) and among the ~10^7 pre-activations of a pass one or two lie within 1e-6
        # of the kink (for these weights: pixel 5012, channel 5 of dconv4.0, tools/gpu_glitch_locate.py), where the last bits of
        # the batch statistics (summed in arrival order) decide the branch; every gradient upstream of that element then moves
        # by 4e-4 .. 3e-3.  So: each parameter within 2e-2, the gradient as a whole within 3e-3 (pre-BatchNorm biases have an
        # exact gradient of 0: absolute scale).  A wrong coefficient or a stale buffer in a replay shows at the 1e-1 level.
        pairs = list(pairs)
        for k, u, v in pairs:
            assert same(u, v, 2e-2), (k, rel_err(u, v))
        fu, fv = (torch.cat([t[i].reshape(-1) for t in pairs]) for i in (1, 2))
        assert rel_err(fu, fv) < 3e-3, rel_err(fu, fv)

    agree((k, pd.grad, pc.grad) for (k, pc), (_, pd) in zip(c.named_parameters(), d.named_parameters()))
    # gradient accumulation into a .grad that aliases the static buffer: second backward adds
    g1 = [p.grad.clone() for p in d.parameters()]
    (d(x) * G).sum().backward()
    agree((i, p.grad, 2 * g) for i, (g, p) in enumerate(zip(g1, d.parameters())))
    # direct_grads: the node assigns .grad itself (no AccumulateGrad per parameter) -- same gradients, same accumulation
    dd = _net(seed=17).train()
    dd.enable_plans(direct_grads=True)
    for it in range(5):
        for p in dd.parameters():
            p.grad = None
        (dd(x) * G).sum().backward()
    assert all(p.grad is not None for p in dd.parameters())
    pl = [q for q in dd._plans.values() if q.fwd is not None][0]
    lo = pl.flat.data_ptr()
    assert all(lo <= p.grad.data_ptr() < lo + 4 * pl.flat.numel() for p in dd.parameters())     # views of the static buffer
    agree((k, pd.grad, pc.grad) for (k, pc), (_, pd) in zip(c.named_parameters(), dd.named_parameters()))
    (dd(x) * G).sum().backward()                                            # second backward: accumulated in place
    agree((k, pd.grad, 2 * pc.grad) for (k, pc), (_, pd) in zip(c.named_parameters(), dd.named_parameters()))
    # an eval forward between a training forward and its backward neither disturbs the tape nor the BatchNorm mode
    for p in d.parameters():
        p.grad = None
    out = d(x)
    d.eval()
    with torch.no_grad():
        ye = [d(x) for _ in range(4)]                # eval plans: 2 eager, record, replay
    d.train()
    (out * G).sum().backward()
    agree((k, pd.grad, pc.grad) for (k, pc), (_, pd) in zip(c.named_parameters(), d.named_parameters()))
    assert rel_err(ye[3], ye[0]) < 1e-5
    # dropout on: replays must not repeat the mask
    from lm_net_amd import LM_Net
    e = LM_Net(3, 2)
    fill_module(e, 11)
    e = e.cuda().train().enable_plans()
    outs = []
    for _ in range(5):
        o = e(x)
        o.sum().backward()
        outs.append(o.detach().clone())
        for p in e.parameters():
            p.grad = None
    assert float((outs[3] - outs[4]).abs().max()) > 1e-3


@pytest.mark.parametrize("mode", ["bf16", "bf16-mma"])
def test_bf16_mixed_precision_path(mode):
    """BASELINE configs[2] arithmetic (row X1).  mode "bf16": every activation tensor STORED as bf16 (half the HBM bytes of
    every row) and every dense contraction (conv / linear forward, data gradient, weight gradient) on
    v_mfma_f32_16x16x16_bf16; accumulators, epilogues, BatchNorm / LayerNorm / softmax statistics, master weights and
    weight gradients stay fp32.  mode "bf16-mma": fp32 storage, bf16 matrix-core operands only.  "bf16" is what running
    the module under torch.autocast selects (the reference's AMP branch, utils/train_eval_utils.py:130-138).
    Stated bf16 tolerances against the fp32 goldens of the REAL reference = 2x the measured distances
    (tools/gpu_bf16_probe.py, "bf16" / "bf16-mma": logits 2.0e-2 / 1.7e-2 of their range, 8.0e-3 / 6.9e-3 in L2; input
    gradient 4.9e-2 / 4.6e-2 in L2; parameter gradients 4.9e-2 / 4.4e-2 median, 0.19 / 0.16 worst in L2 -- the BatchNorm
    gammas of the expand convs, sums of dh*zhat with heavy cancellation; running statistics 4e-3; Dice / IoU of the 352x352
    fixture +8e-4 / +5e-4: ~30 of 123,904 argmax decisions flip, so "identical to 4 decimals" holds for fp32 only)."""
    def l2(a, b):
        a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
        return float((a - b).norm() / (b.norm() + 1e-30))

    g = load_golden("default_64x96.npz")
    m = _net()
    m.compute_dtype = mode
    m.eval()
    x = det_input((2, 3, 64, 96), "d64/x").cuda()
    with torch.no_grad():
        y = m(x)
    assert y.dtype == torch.float32                      # logits leave the module as fp32 NCHW in every mode
    assert 2e-4 < rel_err(y, g["logits"]) < 4e-2 and l2(y, g["logits"]) < 1.6e-2      # bf16 arithmetic is really in use
    if mode == "bf16":
        m.compute_dtype = None                           # follow autocast
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            ya = m(x)
        with torch.no_grad():
            yf = m(x)
        # autocast selects the same path (float-atomic SE sums differ in the last bits between two runs, which flips a
        # few bf16 roundings downstream: 4e-3, against the 2e-2 distance to the fp32 path) ...
        assert float((ya - y).abs().max()) < 4e-3 * float(y.abs().max()) and rel_err(ya, g["logits"]) > 2e-4
        assert rel_err(yf, g["logits"]) < TOL                                  # ... and outside of it the fp32 one
        m.compute_dtype = mode
    m.train()
    xg = x.clone().requires_grad_(True)
    yt = m(xg)
    assert rel_err(yt, g["train_logits"]) < 5e-2 and l2(yt, g["train_logits"]) < 3e-2
    (yt * det_input(tuple(yt.shape), "d64/G").cuda()).sum().backward()
    assert l2(xg.grad, g["train_grad_input"]) < 0.1
    errs = []
    for k, p in m.named_parameters():
        assert p.grad.dtype == torch.float32
        if is_pre_bn_bias(k) or "grad/" + k not in g:
            continue
        errs.append((l2(p.grad, g["grad/" + k]), k))
    errs.sort()
    assert errs[len(errs) // 2][0] < 0.1 and errs[-1][0] < 0.4, (errs[len(errs) // 2], errs[-1])
    for k, v in m.state_dict().items():
        if "running_" in k:
            assert rel_err(v, g["state/" + k]) < 1e-2, k
    # stage activations are stored as bf16 in "bf16" mode
    m.eval()
    m._keep_taps = True
    with torch.no_grad():
        m(x)
    assert m._taps["x1"].dtype == (torch.bfloat16 if mode == "bf16" else torch.float32)
    # Dice / IoU on the 352x352 fixture (BASELINE configs[0] shape)
    g3 = load_golden("default_352.npz")
    m = _net()                                           # (fresh BatchNorm running statistics)
    m.compute_dtype = mode
    m.eval()
    x3 = det_input((1, 3, 352, 352), "d352/x").cuda()
    with torch.no_grad():
        y3 = m(x3)
    assert rel_err(y3, g3["logits"]) < 5e-2
    pred = y3.argmax(1).cpu()
    dice, iou = dice_iou(pred, disc_labels(1, 352, 352))
    assert abs(dice - float(g3["dice"][0])) < 2e-3 and abs(iou - float(g3["iou"][0])) < 2e-3
    assert abs(int(pred.sum()) - int(g3["pred_sum"][0])) < 124         # < 0.1 % of the argmax decisions flip


def test_loss_and_metrics_vs_reference_classes():
    """Rows N1 / N2 pinned to the REAL reference: tests/golden/loss_metrics.npz holds the value and logits-gradient of
    nn.CrossEntropyLoss(weight=[1,4], label_smoothing=1e-3) + utils/loss.py::DiceLoss(2)(..., weight=[1,4]) exactly as
    utils/train_eval_utils.py:141 calls them (float64), and the confusion matrix / Dice / mean Dice / mIoU / accuracy of
    utils/train_eval_utils.py::Evaluator on argmax(logits) (tools/make_golden_loss.py imports both classes)."""
    from lm_net_amd.loss import SegLoss
    from lm_net_amd.metrics import ConfusionMeter
    g = load_golden("loss_metrics.npz")
    for tag in ("a", "b"):
        B, H, W, scale = g[tag + "/meta"]
        B, H, W = int(B), int(H), int(W)
        lg = (det_input((B, 2, H, W), "loss/%s" % tag) * float(scale)).cuda().requires_grad_(True)
        y = disc_labels(B, H, W, seed=77).cuda()
        loss = SegLoss(ce_weight=(1.0, 4.0), dice_weight=(1.0, 4.0), label_smoothing=0.001).cuda()(lg, y)
        loss.backward()
        ref = float(g[tag + "/loss"][0])
        assert abs(float(loss.detach()) - ref) < 2e-6 * max(1.0, abs(ref)), (tag, float(loss.detach()), ref)
        assert rel_err(lg.grad, g[tag + "/dlogits"]) < 1e-5, tag
        m = ConfusionMeter(2)
        m.update(lg.detach(), y)
        r = m.compute()
        conf = torch.tensor(r["confusion"], dtype=torch.float64)
        assert torch.equal(conf, torch.from_numpy(g[tag + "/confusion"]))          # integer counts: exact
        assert abs(r["dice"][1] - float(g[tag + "/dice_fg"][0])) < 1e-12           # Evaluator.Dice (foreground class)
        assert abs(sum(r["dice"]) / 2 - float(g[tag + "/mean_dice"][0])) < 1e-12   # Evaluator.Mean_Dice
        assert abs(sum(r["iou"]) / 2 - float(g[tag + "/miou"][0])) < 1e-12         # Mean_Intersection_over_Union
        assert abs(r["accuracy"] - float(g[tag + "/acc"][0])) < 1e-12
