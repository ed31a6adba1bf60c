"""Per-kernel parity checks of the C-ABI (lm_net_amd/hip.py) against plain PyTorch fp64 CPU references
(and oracle/natten_ref.py for neighborhood attention).  Each check returns a list of
(name, rel_err, tol) rows; ``tests/test_kernels_gpu.py`` asserts them, ``tools/gpu_kernel_check.py``
prints them all in one GPU run.  Tolerance for fp32 device arithmetic vs fp64: 1e-4 relative
(north_star), atomically-reduced sums 2e-4.
"""
import math

import torch
import torch.nn.functional as F

from lm_net_amd import hip
from oracle import natten_ref

DEV = "cuda"
TOL = 1e-4


_ROUND16 = [False]   # check_conv_bf16: draw bf16-representable test data (every bf16 product is then exact in fp32)


def R(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + 1000 * len(shape) + sum(shape))
    t = torch.randn(*shape, generator=g, dtype=torch.float64) * scale
    return t.to(torch.bfloat16).double() if _ROUND16[0] else t


def dev(t):
    return t.to(torch.float32).to(DEV).contiguous()


def rel(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    if got.shape != ref.shape:
        return float("inf")
    if not torch.isfinite(got).all():
        return float("inf")
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def nhwc(t):  # NCHW fp64 cpu -> NHWC fp32 device
    return dev(t.permute(0, 2, 3, 1))


def nchw(t):  # NHWC device -> NCHW fp64 cpu
    return t.detach().double().cpu().permute(0, 3, 1, 2)


def nhwcE(t):  # NCHW fp64 cpu -> ROW-PLANAR fp32 device (the E-wide tensors of a ReparamConv block, hip.rp4)
    return hip.nhwc_to_rp4(nhwc(t))


def nchwE(t):  # row-planar device -> NCHW fp64 cpu
    return nchw(hip.rp4_to_nhwc(t))


def nanE(B, H, W, E, dtype=torch.float32):  # NaN-filled ROW-PLANAR output buffer of a depthwise pass (the wrappers refuse unmarked tensors)
    return hip.rp4(torch.full((B, H, W, E), float("nan"), device=DEV, dtype=dtype))


def gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


# ------------------------------------------------------------------------------------------------ conv forward
def check_conv_fwd():
    rows = []
    cases = [
        # name, B, H, W, [Cin...], Cout, k, s
        ("1x1 12->24", 2, 9, 13, [12], 24, 1, 1),
        ("1x1 4->24 (padded rgb)", 1, 8, 8, [4], 24, 1, 1),
        ("1x1 96->192", 1, 7, 6, [96], 192, 1, 1),
        ("1x1 372->1116", 1, 5, 5, [372], 1116, 1, 1),
        ("1x1 two-src 24+12->12", 2, 6, 7, [24, 12], 12, 1, 1),
        ("3x3 s1 12->12", 2, 10, 11, [12], 12, 3, 1),
        ("3x3 s1 48->24", 1, 9, 9, [48], 24, 3, 1),
        ("3x3 s2 12->24", 2, 12, 10, [12], 24, 3, 2),
        ("3x3 s2 96->192", 1, 8, 8, [96], 192, 3, 2),
        ("3x3 s1 cat 24+24+24->24", 1, 8, 9, [24, 24, 24], 24, 3, 1),
        ("3x3 s1 100->112 (chunks)", 1, 6, 6, [100], 112, 3, 1),
        ("3x3 s1 372->372", 1, 6, 6, [372], 372, 3, 1),
        ("1x1 64->64 on 2x3 map", 1, 2, 3, [64], 64, 1, 1),
        ("1x1 32->72 on 1x1 map", 2, 1, 1, [32], 72, 1, 1),
        ("3x3 s1 64->80 on 2x3 map", 1, 2, 3, [64], 80, 3, 1),
        ("3x3 s1 48->144 22x22", 2, 22, 22, [48], 144, 3, 1),
        ("1x1 96->288 44x44", 1, 44, 44, [96], 288, 1, 1),
    ]
    for name, B, H, W, cins, cout, k, s in cases:
        cin = sum(cins)
        x = R(B, cin, H, W, seed=1)
        w = R(cout, cin, k, k, seed=2, scale=1.0 / math.sqrt(cin * k * k))
        b = R(cout, seed=3)
        ref = F.conv2d(x, w, b, stride=s, padding=k // 2)
        Ho, Wo = ref.shape[2:]
        xs, off = [], 0
        for c in cins:
            xs.append(nhwc(x[:, off:off + c]))
            off += c
        wp = hip.conv_pack(dev(w), k, cins)
        out = torch.full((B, Ho, Wo, cout), float("nan"), device=DEV)
        hip.conv_fwd(xs, wp, out, B=B, Hin=H, Win=W, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s, bias=dev(b))
        rows.append(("conv_fwd " + name, rel(nchw(out), ref), TOL))

    # sources that are slices of a wider buffer; output into a slice; residual; affine+hardswish; stats
    B, H, W = 2, 7, 9
    x = R(B, 36, H, W, seed=5)
    w = R(24, 24, 3, 3, seed=6, scale=0.1)
    b = R(24, seed=7)
    res = R(B, 24, H, W, seed=8)
    a0, a1 = R(24, seed=9), R(24, seed=10)
    z = F.conv2d(x[:, 8:32], w, b, padding=1)
    ref = F.hardswish((z * a0.view(1, -1, 1, 1) + a1.view(1, -1, 1, 1)).float()).double() + res
    xb = nhwc(x)
    outb = torch.zeros(B, H, W, 40, device=DEV)
    stats = torch.zeros(2, 24, device=DEV)
    wp = hip.conv_pack(dev(w), 3, [24])
    hip.conv_fwd([hip.V(xb, 8, 24)], wp, hip.V(outb, 12, 24), B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=24, ksize=3,
                 bias=dev(b), epilogue=hip.EP_AFFINE_ACT, act=hip.ACT_HSWISH, p=(dev(a0), dev(a1)), residual=nhwc(res),
                 stats=stats, stats_mode=hip.STATS_SUM_SQ)
    rows.append(("conv_fwd slice/affine/hswish/residual", rel(nchw(outb[..., 12:36]), ref), TOL))
    rows.append(("conv_fwd untouched slice stays 0", float(outb[..., :12].abs().max() + outb[..., 36:].abs().max()), 1e-30))
    sref = torch.stack([z.sum((0, 2, 3)), (z * z).sum((0, 2, 3))])
    rows.append(("conv_fwd stats sum/sumsq", rel(stats, sref), 2e-4))
    srep = torch.zeros(4, 2, 24, device=DEV)        # statistics in 4 slices (block % 4), summed by the consumer
    hip.conv_fwd([hip.V(xb, 8, 24)], wp, hip.V(outb, 12, 24), B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=24, ksize=3,
                 bias=dev(b), epilogue=hip.EP_AFFINE_ACT, act=hip.ACT_HSWISH, p=(dev(a0), dev(a1)), residual=nhwc(res),
                 stats=srep, stats_mode=hip.STATS_SUM_SQ, stats_rep=4)
    rows.append(("conv_fwd stats in 4 slices", rel(srep.sum(0), sref), 2e-4))
    mean_r, rstd_r = torch.empty(24, device=DEV), torch.empty(24, device=DEV)
    hip.bn_finalize(srep, B * H * W, torch.ones(24, device=DEV), torch.zeros(24, device=DEV), 1e-5, 0.1, mean_r, rstd_r,
                    None, None, None, None)
    rows.append(("bn_finalize over slices: mean", rel(mean_r, sref[0] / (B * H * W)), 2e-4))

    # A3 form: GELU + per-(image,channel) scale on source 0, plain source 1
    B, H, W = 3, 6, 5
    pre, xin = R(B, 24, H, W, seed=11), R(B, 12, H, W, seed=12)
    sc = R(B, 24, seed=13).abs()
    w = R(12, 36, 1, 1, seed=14, scale=0.2)
    b = R(12, seed=15)
    ref = F.conv2d(torch.cat([gelu(pre) * sc.view(B, 24, 1, 1), xin], 1), w, b)
    wp = hip.conv_pack(dev(w), 1, [24, 12])
    out = torch.empty(B, H, W, 12, device=DEV)
    hip.conv_fwd([dict(view=nhwc(pre), scale=dev(sc), flags=hip.SRC_GELU), nhwc(xin)], wp, out, B=B, Hin=H, Win=W,
                 Hout=H, Wout=W, Cout=12, bias=dev(b))
    rows.append(("conv_fwd gelu*scale source + plain source", rel(nchw(out), ref), TOL))
    b2 = R(12, seed=77)
    hip.conv_fwd([dict(view=nhwc(pre), scale=dev(sc), flags=hip.SRC_GELU), nhwc(xin)], wp, out, B=B, Hin=H, Win=W,
                 Hout=H, Wout=W, Cout=12, bias=dev(b), bias2=dev(b2))
    rows.append(("conv_fwd two biases (pointwise + shortcut)", rel(nchw(out), ref + b2.view(1, -1, 1, 1)), TOL))
    return rows


def check_conv_bwd_data():
    rows = []
    cases = [("1x1 24->12", 2, 7, 6, 24, 12, 1, 1), ("3x3 s1 24->36", 2, 8, 9, 24, 36, 3, 1),
             ("3x3 s2 12->24", 2, 12, 10, 12, 24, 3, 2), ("3x3 s2 48->96 odd", 1, 9, 7, 48, 96, 3, 2),
             ("3x3 s1 96->96", 1, 6, 6, 96, 96, 3, 1), ("3x3 s2 12->24 multi-tile", 2, 64, 80, 12, 24, 3, 2),
             ("3x3 s2 96->192 two cout chunks", 2, 22, 38, 96, 192, 3, 2), ("3x3 s2 24->48 odd wide", 1, 35, 71, 24, 48, 3, 2)]
    for name, B, H, W, cin, cout, k, s in cases:
        x = R(B, cin, H, W, seed=21).requires_grad_(True)
        w = R(cout, cin, k, k, seed=22, scale=0.1)
        y = F.conv2d(x, w, None, stride=s, padding=k // 2)
        dy = R(*y.shape, seed=23)
        y.backward(dy)
        Ho, Wo = y.shape[2:]
        wpt = hip.conv_pack_t(dev(w), k)
        dx = torch.full((B, H, W, cin), float("nan"), device=DEV)
        hip.conv_fwd([nhwc(dy)], wpt, dx, B=B, Hin=Ho, Win=Wo, Hout=H, Wout=W, Cout=cin, ksize=k, stride=s, transposed=1)
        rows.append(("conv_bwd_data " + name, rel(nchw(dx), x.grad), TOL))
    # gradient of one source slice of a cat conv (rows [12, 36) of 48 input channels)
    B, H, W = 1, 6, 7
    x = R(B, 48, H, W, seed=24).requires_grad_(True)
    w = R(24, 48, 3, 3, seed=25, scale=0.1)
    y = F.conv2d(x, w, None, padding=1)
    dy = R(*y.shape, seed=26)
    y.backward(dy)
    wpt = hip.conv_pack_t(dev(w), 3, row_off=12, rows=24)
    dx = torch.empty(B, H, W, 24, device=DEV)
    hip.conv_fwd([nhwc(dy)], wpt, dx, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=24, ksize=3, transposed=1)
    rows.append(("conv_bwd_data source slice of cat", rel(nchw(dx), x.grad[:, 12:36]), TOL))
    return rows


def check_conv_wgrad():
    rows = []
    cases = [("1x1 12->24", 2, 9, 7, [12], 24, 1, 1), ("1x1 two-src 24+12->12", 2, 6, 7, [24, 12], 12, 1, 1),
             ("3x3 s1 12->12", 2, 10, 11, [12], 12, 3, 1), ("3x3 s2 24->48", 2, 12, 10, [24], 48, 3, 2),
             ("3x3 s1 cat 24+24->24", 1, 8, 9, [24, 24], 24, 3, 1), ("3x3 s1 96->96", 1, 7, 7, [96], 96, 3, 1),
             ("1x1 372->744", 1, 6, 6, [372], 744, 1, 1)]
    for name, B, H, W, cins, cout, k, s in cases:
        cin = sum(cins)
        x = R(B, cin, H, W, seed=31)
        w = R(cout, cin, k, k, seed=32, scale=0.1).requires_grad_(True)
        b = R(cout, seed=33).requires_grad_(True)
        y = F.conv2d(x, w, b, stride=s, padding=k // 2)
        dy = R(*y.shape, seed=34)
        y.backward(dy)
        Ho, Wo = y.shape[2:]
        xs, off = [], 0
        for c in cins:
            xs.append(nhwc(x[:, off:off + c]))
            off += c
        dW = torch.zeros(cout, cin, k, k, device=DEV)
        db = torch.zeros(cout, device=DEV)
        hip.conv_wgrad(xs, nhwc(dy), dW, db, B=B, Hin=H, Win=W, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s)
        rows.append(("conv_wgrad dW " + name, rel(dW, w.grad), 2e-4))
        rows.append(("conv_wgrad db " + name, rel(db, b.grad), 2e-4))
        if len(cins) > 1:  # per-source gradient tensors + second bias gradient (pointwise + shortcut in one pass)
            parts = [torch.zeros(cout, c, k, k, device=DEV) for c in cins]
            db1, db2 = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
            hip.conv_wgrad(xs, nhwc(dy), None, db1, B=B, Hin=H, Win=W, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s,
                           dW_src=parts, db2=db2)
            off = 0
            for i, c in enumerate(cins):
                rows.append(("conv_wgrad per-source dW[%d] %s" % (i, name), rel(parts[i], w.grad[:, off:off + c]), 2e-4))
                off += c
            rows.append(("conv_wgrad db2 " + name, rel(db2, b.grad), 2e-4))
    # large two-source 1x1 (two-stage reduction path) with per-source outputs
    B, H, W, cins, cout = 2, 64, 48, [24, 12], 12
    x = R(B, 36, H, W, seed=41)
    w = R(cout, 36, 1, 1, seed=42, scale=0.1).requires_grad_(True)
    y = F.conv2d(x, w)
    dy = R(*y.shape, seed=43)
    y.backward(dy)
    parts = [torch.zeros(cout, c, 1, 1, device=DEV) for c in cins]
    db1, db2 = torch.zeros(cout, device=DEV), torch.zeros(cout, device=DEV)
    hip.conv_wgrad([nhwc(x[:, :24]), nhwc(x[:, 24:])], nhwc(dy), None, db1, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=cout,
                   dW_src=parts, db2=db2)
    rows.append(("conv_wgrad per-source large dW[0]", rel(parts[0], w.grad[:, :24]), 2e-4))
    rows.append(("conv_wgrad per-source large dW[1]", rel(parts[1], w.grad[:, 24:]), 2e-4))
    rows.append(("conv_wgrad per-source large db/db2", rel(db1 + db2, 2 * dy.sum((0, 2, 3))), 2e-4))
    # deferred K-split reductions that share a destination (a weight used by two calls): the batched reduce adds with plain
    # read-modify-writes, so hip.wgrad_reduce_flush must put such jobs into consecutive launches -- dW = sum of both calls
    B, H, W, cin, cout = 2, 64, 64, 24, 24
    x1, x2 = R(B, cin, H, W, seed=51), R(B, cin, H, W, seed=52)
    w = R(cout, cin, 3, 3, seed=53, scale=0.1).requires_grad_(True)
    b = R(cout, seed=54).requires_grad_(True)
    y1, y2 = F.conv2d(x1, w, b, padding=1), F.conv2d(x2, w, b, padding=1)
    d1, d2 = R(*y1.shape, seed=55), R(*y2.shape, seed=56)
    (y1 * d1).sum().backward()
    (y2 * d2).sum().backward()
    dW, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
    keep = [hip.conv_wgrad([nhwc(xx)], nhwc(dd), dW, db, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=cout, ksize=3, defer=True)
            for xx, dd in ((x1, d1), (x2, d2))]
    ntab = len(hip.wgrad_reduce_flush())
    torch.cuda.synchronize()
    if keep[0] is not None:   # (both calls really deferred a K-split reduction)
        rows.append(("conv_wgrad deferred, shared destination: launches", abs(ntab - 2), 0))
    rows.append(("conv_wgrad deferred, shared destination dW", rel(dW, w.grad), 2e-4))
    rows.append(("conv_wgrad deferred, shared destination db", rel(db, b.grad), 2e-4))
    return rows


def check_conv_dropout():
    """Epilogue dropout and its two backward uses regenerate the SAME mask; keep-rate ~ 1-p."""
    rows = []
    B, H, W, C = 2, 16, 16, 24
    x = R(B, C, H, W, seed=41)
    eye = torch.eye(C, dtype=torch.float64).view(C, C, 1, 1)
    wp = hip.conv_pack(dev(eye), 1, [C])
    out = torch.empty(B, H, W, C, device=DEV)
    hip.conv_fwd([nhwc(x)], wp, out, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=C, drop_p=0.1, drop_seed=77)
    o = nchw(out)
    keep = (o != 0)
    rows.append(("dropout keep-rate", abs(float(keep.double().mean()) - 0.9), 0.02))
    rows.append(("dropout kept values scaled 1/(1-p)", rel(o[keep], (x / 0.9)[keep]), TOL))
    # the same mask applied on load (LMN_SRC_DROP) to a tensor of ones
    out2 = torch.empty(B, H, W, C, device=DEV)
    ones = torch.ones(B, H, W, C, device=DEV)
    hip.conv_fwd([dict(view=ones, flags=hip.SRC_DROP, drop_seed=77, drop_p=0.1)], wp, out2, B=B, Hin=H, Win=W, Hout=H,
                 Wout=W, Cout=C)
    rows.append(("dropout mask identical on load", rel(nchw(out2), keep.double() / 0.9), TOL))
    return rows


# ------------------------------------------------------------------------------------------------ depthwise block
def _dw_ref(x1, ws, gam, bet, train, rm=None, rv=None):
    """torch reference of sum_b BN_b(dw_b(x1)) (fp64).  Returns pre and the 4 branch outputs."""
    E = x1.shape[1]
    pads = [(2, 2), (1, 1), (1, 0), (0, 1)]
    ys = [F.conv2d(x1, w, None, padding=p, groups=E) for w, p in zip(ws, pads)]
    pre = 0
    for i, y in enumerate(ys):
        if train:
            pre = pre + F.batch_norm(y, None, None, gam[i], bet[i], True, 0.1, 1e-5)
        else:
            pre = pre + F.batch_norm(y, rm[i], rv[i], gam[i], bet[i], False, 0.1, 1e-5)
    return pre, ys


def check_dw():
    rows = []
    for (B, H, W, E) in [(2, 20, 19, 24), (1, 33, 17, 48), (2, 18, 21, 12), (1, 70, 130, 24), (1, 9, 61, 8),
                         (3, 5, 120, 16), (1, 64, 57, 24), (1, 4, 4, 4), (2, 130, 60, 8)]:   # two whole strips / the halo instance (57 = 56 + 1) / the smallest map / many row segments of one strip
        x1 = R(B, E, H, W, seed=51)
        ws = [R(E, 1, 5, 5, seed=52, scale=0.2), R(E, 1, 3, 3, seed=53, scale=0.3), R(E, 1, 3, 1, seed=54, scale=0.5),
              R(E, 1, 1, 3, seed=55, scale=0.5)]
        gam = [R(E, seed=56 + i).abs() + 0.5 for i in range(4)]
        bet = [R(E, seed=60 + i) * 0.1 for i in range(4)]
        x1r = x1.clone().requires_grad_(True)
        wsr = [w.clone().requires_grad_(True) for w in ws]
        gr = [g.clone().requires_grad_(True) for g in gam]
        br = [b.clone().requires_grad_(True) for b in bet]
        pre_ref, ys = _dw_ref(x1r, wsr, gr, br, True)
        tag = " E=%d %dx%d" % (E, H, W)
        # --- stats
        x1d = nhwcE(x1)
        wd = [dev(w) for w in ws]
        stats = torch.zeros(4, 2, E, device=DEV)
        hip.dw_stats(x1d, *wd, stats)
        sref = torch.stack([torch.stack([y.sum((0, 2, 3)), (y * y).sum((0, 2, 3))]) for y in ys]).detach()
        rows.append(("dw_stats" + tag, rel(stats, sref), 2e-4))
        # --- merge (with the exact fp64 batch stats) + forward
        N = B * H * W
        mean = [y.mean((0, 2, 3)).detach() for y in ys]
        rstd = [1.0 / torch.sqrt(y.var((0, 2, 3), unbiased=False) + 1e-5).detach() for y in ys]
        A = torch.stack([gam[i] * rstd[i] for i in range(4)])
        shift = torch.stack([bet[i] - mean[i] * A[i] for i in range(4)])
        keff, beff = torch.empty(E, 25, device=DEV), torch.empty(E, device=DEV)
        hip.dw_merge(*wd, dev(A), dev(shift), keff, beff)
        pre = nanE(B, H, W, E)
        gsum = torch.zeros(B, E, device=DEV)
        hip.dw_fwd(x1d, pre, gsum, keff, beff)
        rows.append(("dw_fwd pre" + tag, rel(nchwE(pre), pre_ref), TOL))
        rows.append(("dw_fwd gsum" + tag, rel(gsum, gelu(pre_ref).sum((2, 3)).detach()), 2e-4))
        # --- squeeze-excite gate formed inside the pass (arrival counter per image) == lmn_se_fwd on the finished sums
        Rr = max(E // 4, 1)
        fw1, fb1 = dev(R(Rr, E, seed=101, scale=0.4)), dev(R(Rr, seed=102))
        fw2, fb2 = dev(R(E, Rr, seed=103, scale=0.8)), dev(R(E, seed=104))
        s_sep, h_sep = torch.empty(B, E, device=DEV), torch.empty(B, Rr, device=DEV)
        hip.se_fwd(gsum, 1.0 / (H * W), fw1, fb1, fw2, fb2, s_sep, h_sep)
        for rep in range(3):      # (repeated: the hand-off must hold whichever block arrives last)
            pre_f = nanE(B, H, W, E); gs_f = torch.zeros(B, E, device=DEV)
            s_f, h_f = torch.full((B, E), float("nan"), device=DEV), torch.full((B, Rr), float("nan"), device=DEV)
            hip.dw_fwd(x1d, pre_f, gs_f, keff, beff, se=dict(ticket=torch.zeros(B, device=DEV), fc1w=fw1, fc1b=fb1, fc2w=fw2,
                                                             fc2b=fb2, s=s_f, hidden=h_f, inv_hw=1.0 / (H * W)))
            rows.append(("dw_fwd + fused SE gate (run %d)" % rep + tag, max(rel(s_f, s_sep), rel(h_f, h_sep), rel(pre_f, pre)), 2e-5))
        # --- backward: dpre = (u*s + dm) * gelu'(pre)
        u = R(B, E, H, W, seed=70)
        s = R(B, E, seed=71).abs()
        dm = R(B, E, seed=72) * 0.01
        pre_l = pre_ref.detach().clone().requires_grad_(True)
        g = gelu(pre_l)
        (g * (u * s.view(B, E, 1, 1) + dm.view(B, E, 1, 1))).sum().backward()
        dpre_ref = pre_l.grad
        pre_ref.backward(dpre_ref)
        dpre = nanE(B, H, W, E)
        bstats = torch.zeros(5, E, device=DEV)
        hip.dw_bwd_stats(x1d, nhwcE(pre_ref.detach()), nhwcE(u), dev(s), dev(dm), dpre, *wd, bstats)
        rows.append(("dw_bwd_stats dpre" + tag, rel(nchwE(dpre), dpre_ref), TOL))
        bref = torch.stack([dpre_ref.sum((0, 2, 3))] + [(dpre_ref * y.detach()).sum((0, 2, 3)) for y in ys])
        rows.append(("dw_bwd_stats sums" + tag, rel(bstats, bref), 2e-4))
        # --- squeeze-excite backward formed inside the pass == lmn_se_bwd_dm feeding the plain pass
        ds_t = dev(R(B, E, seed=105))
        dm_sep, dvec_sep = torch.empty(B, E, device=DEV), torch.empty(B, E + Rr, device=DEV)
        hip.se_bwd_dm(ds_t, s_sep, 1.0 / (H * W), fw1, fw2, h_sep, dm_sep, dvec_sep)
        dpre_s, bst_s = nanE(B, H, W, E), torch.zeros(5, E, device=DEV)
        hip.dw_bwd_stats(x1d, nhwcE(pre_ref.detach()), nhwcE(u), s_sep, dm_sep, dpre_s, *wd, bst_s)
        dpre_f, bst_f, dvec_f = nanE(B, H, W, E), torch.zeros(5, E, device=DEV), torch.full((B, E + Rr), float("nan"), device=DEV)
        hip.dw_bwd_stats(x1d, nhwcE(pre_ref.detach()), nhwcE(u), s_sep, None, dpre_f, *wd, bst_f,
                         seb=dict(ds=ds_t, fc1w=fw1, fc2w=fw2, hidden=h_sep, dvec=dvec_f, inv_hw=1.0 / (H * W)))
        rows.append(("dw_bwd_stats + fused SE backward" + tag, max(rel(dpre_f, dpre_s), rel(bst_f, bst_s), rel(dvec_f, dvec_sep)), 2e-5))
        # coefficients from the exact fp64 sums: f_b = A_b dpre + C_b y_b + D_b ; dgamma_b, dbeta_b
        cA, cC, cD = (torch.empty(4, E, device=DEV) for _ in range(3))
        dgs = [torch.zeros(E, device=DEV) for _ in range(4)]
        dbs = [torch.zeros(E, device=DEV) for _ in range(4)]
        hip.dw_bwd_coef(dev(bref), dev(torch.stack(mean)), dev(torch.stack(rstd)), dev(A), N, True, cA, cC, cD, dgs, dbs)
        for i in range(4):
            rows.append(("dw_bwd_coef dgamma[%d]" % i + tag, rel(dgs[i], gr[i].grad), 2e-4))
            rows.append(("dw_bwd_coef dbeta[%d]" % i + tag, rel(dbs[i], br[i].grad), 2e-4))
        dx1 = nanE(B, H, W, E)
        dws = [torch.zeros_like(dev(w)) for w in ws]
        hip.dw_bwd(x1d, nhwcE(dpre_ref), dx1, *wd, cA, cC, cD, *dws)
        rows.append(("dw_bwd dx1" + tag, rel(nchwE(dx1), x1r.grad), TOL))
        for nm, got, ref in zip(("dW5", "dW3", "dWv", "dWh"), dws, wsr):
            rows.append(("dw_bwd " + nm + tag, rel(got, ref.grad), 2e-4))
        # --- the one-launch forms (BatchNorm bookkeeping inside the depthwise passes): same results as the launch pairs
        class _BN:   # the four attributes hip.dw_fwd_bn / dw_finalize_merge read of a BatchNorm2d
            def __init__(self, i, rm, rv):
                self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum = dev(gam[i]), dev(bet[i]), rm, rv, 1e-5, 0.1
        rm0 = [dev(R(E, seed=90 + i) * 0.2) for i in range(4)]
        rv0 = [dev(R(E, seed=94 + i).abs() + 0.5) for i in range(4)]
        bn_a = [_BN(i, rm0[i].clone(), rv0[i].clone()) for i in range(4)]
        bn_b = [_BN(i, rm0[i].clone(), rv0[i].clone()) for i in range(4)]
        sref_d = dev(sref)
        m_a, r_a, A_a = (torch.zeros(4, E, device=DEV) for _ in range(3))
        m_b, r_b, A_b = (torch.zeros(4, E, device=DEV) for _ in range(3))
        keff2, beff2 = torch.empty(E, 25, device=DEV), torch.empty(E, device=DEV)
        hip.dw_finalize_merge(sref_d, N, bn_b, wd, m_b, r_b, A_b, keff2, beff2)
        pre_b = nanE(B, H, W, E); gs_b = torch.zeros(B, E, device=DEV)
        hip.dw_fwd(x1d, pre_b, gs_b, keff2, beff2)
        pre_a = nanE(B, H, W, E); gs_a = torch.zeros(B, E, device=DEV)
        hip.dw_fwd_bn(x1d, pre_a, gs_a, sref_d, N, bn_a, wd, m_a, r_a, A_a)
        rows.append(("dw_fwd_bn pre vs finalize_merge + fwd" + tag, rel(pre_a, pre_b), 1e-6))
        rows.append(("dw_fwd_bn pre vs reference" + tag, rel(nchwE(pre_a), pre_ref), TOL))
        rows.append(("dw_fwd_bn gsum" + tag, rel(gs_a, gs_b), 1e-5))
        for nm, ta, tb in (("mean", m_a, m_b), ("rstd", r_a, r_b), ("A", A_a, A_b)):
            rows.append(("dw_fwd_bn %s" % nm + tag, rel(ta, tb), 1e-6))
        for i in range(4):
            rows.append(("dw_fwd_bn running stats[%d]" % i + tag, max(rel(bn_a[i].running_mean, bn_b[i].running_mean),
                                                                     rel(bn_a[i].running_var, bn_b[i].running_var)), 1e-6))
        dx1b = nanE(B, H, W, E)
        dws2 = [torch.zeros_like(dev(w)) for w in ws]
        dgs2 = [torch.zeros(E, device=DEV) for _ in range(4)]
        dbs2 = [torch.zeros(E, device=DEV) for _ in range(4)]
        hip.dw_bwd_bn(x1d, nhwcE(dpre_ref), dx1b, *wd, dev(bref), dev(torch.stack(mean)), dev(torch.stack(rstd)), dev(A), N, True,
                      dgs2, dbs2, *dws2)
        rows.append(("dw_bwd_bn dx1 vs coef + bwd" + tag, rel(dx1b, dx1), 1e-6))
        for i in range(4):
            rows.append(("dw_bwd_bn dgamma/dbeta[%d]" % i + tag, max(rel(dgs2[i], dgs[i]), rel(dbs2[i], dbs[i])), 1e-6))
            rows.append(("dw_bwd_bn dW[%d]" % i + tag, rel(dws2[i], dws[i]), 1e-5))
    return rows


def check_zpath():
    """z-path of ReparamConv (lmn_dw_pre_t, lmn_reparam_fold, lmn_affine2): the depthwise kernels fed z + (A, shift) must reproduce
    what they compute from x1 = Hardswish(A z + shift); the depthwise backward's drain must deliver dh and its sums; the folded
    three-source conv must equal the BatchNorm backward + both data gradients of the reference graph (fp64)."""
    rows = []
    for (B, H, W, Cin, E, Cout) in [(2, 20, 19, 12, 24, 12), (1, 33, 17, 24, 48, 24), (2, 9, 61, 4, 8, 12), (1, 70, 130, 12, 24, 12)]:
        tag = " Cin=%d E=%d %dx%d" % (Cin, E, H, W)
        N = B * H * W
        x = R(B, Cin, H, W, seed=201)
        we, be = R(E, Cin, seed=202, scale=0.5), R(E, seed=203, scale=0.2)
        ga, bt = R(E, seed=204).abs() + 0.5, R(E, seed=205) * 0.3
        wsc = R(Cout, Cin, seed=206, scale=0.5)
        ws = [R(E, 1, 5, 5, seed=52, scale=0.2), R(E, 1, 3, 3, seed=53, scale=0.3), R(E, 1, 3, 1, seed=54, scale=0.5), R(E, 1, 1, 3, seed=55, scale=0.5)]
        # ---- fp64 reference graph: z -> BN (batch stats) -> hswish -> x1 ; loss through x1 with a given dx1 ; shortcut through dy
        xr = x.clone().requires_grad_(True)
        wer, ber, gar, btr = (t.clone().requires_grad_(True) for t in (we, be, ga, bt))
        z = F.conv2d(xr, wer.view(E, Cin, 1, 1), ber)
        mu, var = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
        rstd = 1.0 / torch.sqrt(var + 1e-5)
        A1, sh1 = (gar * rstd), (btr - mu * gar * rstd)
        h = z * A1.view(1, E, 1, 1) + sh1.view(1, E, 1, 1)
        x1 = F.hardswish(h)
        dx1 = R(B, E, H, W, seed=207)
        dy = R(B, Cout, H, W, seed=208)
        ((x1 * dx1).sum() + (F.conv2d(xr, wsc.view(Cout, Cin, 1, 1)) * dy).sum()).backward()
        zd, x1d = nhwcE(z.detach()), nhwcE(x1.detach())
        A1d, sh1d = dev(A1.detach()), dev(sh1.detach())
        wd = [dev(w) for w in ws]
        zp = dict(A=A1d, shift=sh1d)
        # ---- forward-side kernels: z + transform == x1
        st_a, st_b = torch.zeros(4, 2, E, device=DEV), torch.zeros(4, 2, E, device=DEV)
        hip.dw_stats(x1d, *wd, st_a)
        hip.dw_stats(zd, *wd, st_b, zpre=zp)
        rows.append(("dw_stats z-path" + tag, rel(st_b, st_a), 2e-5))
        # ... with the BatchNorm finalised inside (fin): sums of z about a shift, 3 slices
        about = dev(R(E, seed=209) * 0.1)
        zc = z.detach() - about.double().cpu().view(1, E, 1, 1)
        sl = torch.zeros(3, 2, E, dtype=torch.float64)
        sl[0, 0], sl[0, 1] = zc.sum((0, 2, 3)) * 0.25, (zc * zc).sum((0, 2, 3)) * 0.5
        sl[1, 0], sl[1, 1] = zc.sum((0, 2, 3)) * 0.75, (zc * zc).sum((0, 2, 3)) * 0.25
        sl[2, 1] = (zc * zc).sum((0, 2, 3)) * 0.25
        rm, rv = dev(R(E, seed=210) * 0.2), dev(R(E, seed=211).abs() + 0.5)
        rm0, rv0 = rm.clone(), rv.clone()
        mo, ro, Ao, so = (torch.full((E,), float("nan"), device=DEV) for _ in range(4))
        st_c = torch.zeros(4, 2, E, device=DEV)
        fin = dict(mode=hip.FIN_BN, sums=dev(sl), nrep=3, count=N, gamma=dev(ga), beta=dev(bt), eps=1e-5, momentum=0.1, about=about,
                   mean=mo, rstd=ro, A=Ao, shift=so, rmean=rm, rvar=rv)
        hip.dw_stats(zd, *wd, st_c, zpre=dict(fin=fin))
        rows.append(("dw_stats z-path + BN finalise: sums" + tag, rel(st_c, st_a), 1e-4))
        rows.append(("dw_stats z-path + BN finalise: A / shift / mean / rstd" + tag,
                     max(rel(Ao, A1), rel(so, sh1), rel(mo, mu), rel(ro, rstd)), 1e-5))
        rows.append(("dw_stats z-path + BN finalise: running stats" + tag,
                     max(rel(rm, 0.9 * rm0.double().cpu() + 0.1 * mu.detach()), rel(rv, 0.9 * rv0.double().cpu() + 0.1 * var.detach() * N / (N - 1))), 1e-5))
        keff, beff = dev(R(E, 25, seed=212, scale=0.2)), dev(R(E, seed=213) * 0.1)
        pa, pb = nanE(B, H, W, E), nanE(B, H, W, E)
        ga_, gb_ = torch.zeros(B, E, device=DEV), torch.zeros(B, E, device=DEV)
        hip.dw_fwd(x1d, pa, ga_, keff, beff)
        hip.dw_fwd(zd, pb, gb_, keff, beff, zpre=zp)
        rows.append(("dw_fwd z-path" + tag, max(rel(pb, pa), rel(gb_, ga_)), 2e-5))
        u, sg, dm = nhwcE(R(B, E, H, W, seed=214)), dev(R(B, E, seed=215).abs()), dev(R(B, E, seed=216) * 0.01)
        da_, db_ = nanE(B, H, W, E), nanE(B, H, W, E)
        ba_, bb_ = torch.zeros(5, E, device=DEV), torch.zeros(5, E, device=DEV)
        hip.dw_bwd_stats(x1d, pa, u, sg, dm, da_, *wd, ba_)
        hip.dw_bwd_stats(zd, pa, u, sg, dm, db_, *wd, bb_, zpre=zp)
        rows.append(("dw_bwd_stats z-path" + tag, max(rel(db_, da_), rel(bb_, ba_)), 2e-5))
        # ---- depthwise backward: dx1 from x1 (old) vs dh + sums from z (new)
        bm, br_, bA = dev(R(4, E, seed=217) * 0.1), dev(R(4, E, seed=218).abs() + 0.5), dev(R(4, E, seed=219).abs() + 0.3)
        bst = dev(R(5, E, seed=220))
        dgs, dbs = [torch.zeros(E, device=DEV) for _ in range(4)], [torch.zeros(E, device=DEV) for _ in range(4)]
        dws = [torch.zeros_like(w) for w in wd]
        dxo = nanE(B, H, W, E)
        hip.dw_bwd_bn(x1d, da_, dxo, *wd, bst, bm, br_, bA, N, True, dgs, dbs, *dws)
        dgs2, dbs2 = [torch.zeros(E, device=DEV) for _ in range(4)], [torch.zeros(E, device=DEV) for _ in range(4)]
        dws2 = [torch.zeros_like(w) for w in wd]
        dhz, hst = nanE(B, H, W, E), torch.zeros(2, E, device=DEV)
        hip.dw_bwd_bn(zd, da_, dhz, *wd, bst, bm, br_, bA, N, True, dgs2, dbs2, *dws2, zpre=zp, hstats=hst)
        hd = nhwc(h.detach()).double().cpu()
        dhs = torch.where(hd < -3, torch.zeros_like(hd), torch.where(hd <= 3, hd / 3 + 0.5, torch.ones_like(hd)))
        dh_ref = hip.rp4_to_nhwc(dxo).double().cpu() * dhs
        rows.append(("dw_bwd_bn z-path dh" + tag, rel(hip.rp4_to_nhwc(dhz), dh_ref), 2e-5))
        rows.append(("dw_bwd_bn z-path sums (dh, dh z)" + tag,
                     rel(hst, torch.stack([dh_ref.sum((0, 1, 2)), (dh_ref * nhwc(z.detach()).double().cpu()).sum((0, 1, 2))])), 2e-4))
        rows.append(("dw_bwd_bn z-path weight gradients" + tag, max(rel(a_, b_) for a_, b_ in zip(dws2 + dgs2, dws + dgs)), 2e-5))
        # ---- fold: dh := the reference's gradient w.r.t. the BatchNorm output, sums from fp64
        dh64 = (dx1 * torch.where(h.detach() < -3, torch.zeros_like(h), torch.where(h.detach() <= 3, h.detach() / 3 + 0.5, torch.ones_like(h))).detach())
        hst64 = torch.stack([dh64.sum((0, 2, 3)), (dh64 * z.detach()).sum((0, 2, 3))])
        rows_c = (Cin + 3) // 4 * 4
        n3 = hip.conv_pack_size(1, rows_c, [E, rows_c, Cout])
        wp3, kb, coef = torch.full((n3,), float("nan"), device=DEV), torch.full((rows_c,), float("nan"), device=DEV), torch.full((3, E), float("nan"), device=DEV)
        dg, dbt = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
        hip.reparam_fold(dev(hst64), dev(mu.detach()), dev(rstd.detach()), A1d, N, True, dev(we), dev(be), dev(wsc), rows_c, Cout, wp3, kb, coef, dg, dbt)
        rows.append(("reparam_fold dgamma / dbeta" + tag, max(rel(dg, gar.grad), rel(dbt, btr.grad)), 2e-4))
        xin = x if rows_c == Cin else torch.cat([x, torch.zeros(B, rows_c - Cin, H, W, dtype=x.dtype)], 1)
        dxo3 = torch.full((B, H, W, rows_c), float("nan"), device=DEV)
        hip.conv_fwd([nhwcE(dh64), nhwc(xin), nhwc(dy)], wp3, dxo3, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=rows_c, bias=kb)
        rows.append(("reparam_fold + three-source conv: dx" + tag, rel(nchw(dxo3)[:, :Cin], xr.grad), TOL))
        dzo = hip.rp4(torch.full((B, H, W, E), float("nan"), device=DEV))
        hip.affine2(nhwcE(dh64), zd, coef, dzo)
        # dz of the reference: BatchNorm backward in closed form (fp64)
        T = ((dh64 * ((z.detach() - mu.detach().view(1, E, 1, 1)) * rstd.detach().view(1, E, 1, 1))).sum((0, 2, 3)))
        dz64 = A1.detach().view(1, E, 1, 1) * (dh64 - hst64[0].view(1, E, 1, 1) / N
                                              - (z.detach() - mu.detach().view(1, E, 1, 1)) * rstd.detach().view(1, E, 1, 1) * T.view(1, E, 1, 1) / N)
        rows.append(("affine2 dz" + tag, rel(nchwE(dzo), dz64), TOL))
        # weight / bias gradient without dz: raw gradient R = sum dh x^T and the moments of x (lmn_reparam_wfin)
        xin64 = xin.double()
        Rr = torch.einsum("behw,bchw->ec", dh64, xin64)
        Mr, mr = torch.einsum("bjhw,bchw->jc", xin64, xin64), xin64.sum((0, 2, 3))
        dWf, dbf = torch.zeros(E, Cin, device=DEV), torch.zeros(E, device=DEV)
        hip.reparam_wfin(dev(Rr), dev(Mr), dev(mr), coef, dev(hst64), dev(we), dev(be), N, dWf, dbf)
        rows.append(("reparam_wfin dW_e" + tag, rel(dWf, wer.grad), 2e-4))
        rows.append(("reparam_wfin db_e (exact 0 under batch statistics)" + tag, float(dbf.abs().max()) / float(Rr.abs().max()), 1e-5))
        rows.append(("dz -> expand weight gradient (reference)" + tag,
                     rel(torch.einsum("behw,bchw->ec", dz64, x), wer.grad), 1e-9))
    return rows


def check_conv_rp():
    """Row-planar (RP4) operands of the 1x1 conv family: every role a ReparamConv block uses -- out (expand conv), source with
    on-load GELU x gate (pointwise conv), out + aux (SE-gradient epilogue), source / dy of the weight gradients, first source of the
    three-source data gradient -- against the SAME call on NHWC tensors holding the same values (equal up to the order of the
    atomics), at map sizes from 4x4 to 20x52, both conv kernels (Cout <= 80 / > 80) and both storage types."""
    rows = []
    to_rp, to_nhwc = hip.nhwc_to_rp4, hip.rp4_to_nhwc
    for (B, H, W, Cin, E, Cout, dt) in [(2, 20, 52, 12, 24, 12, torch.float32), (2, 4, 4, 96, 192, 96, torch.float32), (3, 9, 13, 24, 48, 24, torch.float32),
                                        (1, 12, 16, 48, 96, 48, torch.float32), (2, 8, 12, 12, 24, 12, torch.bfloat16)]:
        tag = " Cin=%d E=%d %dx%d %s" % (Cin, E, H, W, "bf16" if dt == torch.bfloat16 else "fp32")
        tol = 1e-6 if dt == torch.float32 else 2e-2
        prev = hip._MMA[0]
        hip._MMA[0] = hip.BF16 if dt == torch.bfloat16 else hip.F32
        try:
            cast = lambda t: t.to(dt).contiguous()
            x, pre, dy = cast(nhwc(R(B, Cin, H, W, seed=301))), cast(nhwc(R(B, E, H, W, seed=302))), cast(nhwc(R(B, Cout, H, W, seed=303)))
            dh = cast(nhwc(R(B, E, H, W, seed=304)))
            we, be = dev(R(E, Cin, seed=305, scale=0.3)), dev(R(E, seed=306))
            wpw, wsc = dev(R(Cout, E, seed=307, scale=0.2)), dev(R(Cout, Cin, seed=308, scale=0.3))
            gate = dev(R(B, E, seed=309).abs())
            pre_r, dh_r = to_rp(pre), to_rp(dh)
            mk = lambda *sh: torch.full(sh, float("nan"), device=DEV, dtype=dt)
            # expand conv: out row-planar (+ batch sums)
            wpe = hip.conv_pack(we, 1, [Cin])
            za, zb = mk(B, H, W, E), hip.rp4(mk(B, H, W, E))
            sa, sb = torch.zeros(2, E, device=DEV), torch.zeros(2, E, device=DEV)
            kw = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W)
            hip.conv_fwd([x], wpe, za, Cout=E, bias=be, stats=sa, stats_mode=hip.STATS_SUM_SQ, **kw)
            hip.conv_fwd([x], wpe, zb, Cout=E, bias=be, stats=sb, stats_mode=hip.STATS_SUM_SQ, **kw)
            rows.append(("conv out row-planar" + tag, max(rel(to_nhwc(zb), za), rel(sb, sa) * 1e-2), tol))
            # pointwise + shortcut: first source row-planar with GELU x gate
            n0, n1 = hip.conv_pack_size(1, Cout, [E]), hip.conv_pack_size(1, Cout, [Cin])
            h2 = 2 if dt == torch.bfloat16 else 1
            wp2 = torch.empty(n0 + n1, device=DEV)
            hip.conv_pack(wpw, 1, [E], out=wp2[:n0 // h2])
            hip.conv_pack(wsc, 1, [Cin], out=wp2[n0 // h2:(n0 + n1) // h2])
            ya, yb = mk(B, H, W, Cout), mk(B, H, W, Cout)
            hip.conv_fwd([dict(view=pre, scale=gate, flags=hip.SRC_GELU), x], wp2, ya, Cout=Cout, **kw)
            hip.conv_fwd([dict(view=pre_r, scale=gate, flags=hip.SRC_GELU), x], wp2, yb, Cout=Cout, **kw)
            rows.append(("conv source row-planar (GELU x gate)" + tag, rel(yb, ya), tol))
            # data gradient of the pointwise conv with the SE-gradient epilogue: out and aux row-planar
            wpt = hip.conv_pack_t(wpw, 1, 0, E, cred=Cout)
            ua, ub = mk(B, H, W, E), hip.rp4(mk(B, H, W, E))
            da, db_ = torch.zeros(B, E, device=DEV), torch.zeros(B, E, device=DEV)
            hip.conv_fwd([dy], wpt, ua, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=pre, stats=da, stats_mode=hip.STATS_EP, **kw)
            hip.conv_fwd([dy], wpt, ub, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=pre_r, stats=db_, stats_mode=hip.STATS_EP, **kw)
            rows.append(("conv out + aux row-planar (SE gradient)" + tag, max(rel(to_nhwc(ub), ua), rel(db_, da) * (1e-2 if dt == torch.float32 else 1.0)), tol))
            # weight gradients: row-planar source (GELU x gate) / row-planar dy
            for nm, srcs_a, srcs_b, dya, dyb, co, ci in [
                    ("wgrad source row-planar", [dict(view=pre, scale=gate, flags=hip.SRC_GELU), x], [dict(view=pre_r, scale=gate, flags=hip.SRC_GELU), x], dy, dy, Cout, E + Cin),
                    ("wgrad dy row-planar", [x], [x], dh, dh_r, E, Cin)]:
                ga_, gb_ = torch.zeros(co, ci, device=DEV), torch.zeros(co, ci, device=DEV)
                ba_, bb_ = torch.zeros(co, device=DEV), torch.zeros(co, device=DEV)
                hip.conv_wgrad(srcs_a, dya, ga_, ba_, Cout=co, **kw)
                hip.conv_wgrad(srcs_b, dyb, gb_, bb_, Cout=co, **kw)
                rows.append((nm + tag, max(rel(gb_, ga_), rel(bb_, ba_)), 2e-5 if dt == torch.float32 else tol))
        finally:
            hip._MMA[0] = prev
    return rows


def check_se():
    rows = []
    for cfg in ((3, 24, 6, 35), (8, 192, 48, 121), (30, 192, 48, 64)):   # the last one: the per-image kernel (batch too large for one block)
        rows += _check_se(*cfg)
    return rows


def _check_se(B, E, Rr, HW):
    rows = []
    tag = " B=%d E=%d" % (B, E)
    gsum = R(B, E, seed=81) * HW * 0.3
    w1, b1 = R(Rr, E, seed=82, scale=0.4).requires_grad_(True), R(Rr, seed=83).requires_grad_(True)
    w2, b2 = R(E, Rr, seed=84, scale=0.8).requires_grad_(True), R(E, seed=85).requires_grad_(True)
    gs = gsum.clone().requires_grad_(True)
    m = gs / HW
    h = F.relu(m @ w1.t() + b1)
    s_ref = F.hardsigmoid((h @ w2.t() + b2).float()).double()
    # hardsigmoid through float loses autograd in fp64; redo in pure fp64
    t = h @ w2.t() + b2
    s_ref = torch.clamp(t + 3, 0, 6) / 6
    ds = R(B, E, seed=86)
    (s_ref * ds).sum().backward()
    s, hid = torch.empty(B, E, device=DEV), torch.empty(B, Rr, device=DEV)
    hip.se_fwd(dev(gsum), 1.0 / HW, dev(w1), dev(b1), dev(w2), dev(b2), s, hid)
    rows.append(("se_fwd s" + tag, rel(s, s_ref), TOL))
    dm = torch.empty(B, E, device=DEV)
    dw1, db1 = torch.zeros(Rr, E, device=DEV), torch.zeros(Rr, device=DEV)
    dw2, db2 = torch.zeros(E, Rr, device=DEV), torch.zeros(E, device=DEV)
    hip.se_bwd(dev(ds), dev(gsum), 1.0 / HW, dev(w1), dev(b1), dev(w2), dev(b2), hid, dm, dw1, db1, dw2, db2)
    rows.append(("se_bwd dm (=d gsum)" + tag, rel(dm, gs.grad), TOL))
    for nm, got, ref in (("dw1", dw1, w1.grad), ("db1", db1, b1.grad), ("dw2", dw2, w2.grad), ("db2", db2, b2.grad)):
        rows.append(("se_bwd " + nm + tag, rel(got, ref), 2e-4))
    # the two-launch form (per-image vectors, then the parameter gradients as a batch reduction without atomics)
    dm2, dvec = torch.empty(B, E, device=DEV), torch.empty(B, E + Rr, device=DEV)
    g1 = [torch.full_like(t, 0.5) for t in (dw1, db1, dw2, db2)]   # (+= semantics)
    hip.se_bwd_dm(dev(ds), s, 1.0 / HW, dev(w1), dev(w2), hid, dm2, dvec)
    hip.se_bwd_params(dvec, dev(gsum), 1.0 / HW, hid, *g1)
    rows.append(("se_bwd_dm dm" + tag, rel(dm2, gs.grad), TOL))
    for nm, got, ref in (("dw1", g1[0], w1.grad), ("db1", g1[1], b1.grad), ("dw2", g1[2], w2.grad), ("db2", g1[3], b2.grad)):
        rows.append(("se_bwd_params " + nm + tag, rel(got - 0.5, ref), 2e-4))
    # ... and riding along in lmn_reparam_wfin's launch (extra blocks; a dummy 8 x 4 expand conv in front)
    g2 = [torch.full_like(t, 0.5) for t in (dw1, db1, dw2, db2)]
    E0, C0 = 8, 4
    z_ = lambda *sh: torch.zeros(*sh, device=DEV)
    dW0, db0 = z_(E0, C0), z_(E0)
    hip.reparam_wfin(z_(E0, C0), z_(C0, C0), z_(C0), z_(3, E0), z_(2, E0), z_(E0, C0), z_(E0), 7.0, dW0, db0,
                     se=dict(dvec=dvec, gsum=dev(gsum), inv_hw=1.0 / HW, hidden=hid, dw1=g2[0], db1=g2[1], dw2=g2[2], db2=g2[3]))
    same = all(torch.equal(a, b) for a, b in zip(g1, g2)) and float(dW0.abs().max()) == 0.0
    rows.append(("se_bwd_params inside reparam_wfin == the separate launch (bitwise)" + tag, 0.0 if same else 1.0, 0.5))
    return rows


# ------------------------------------------------------------------------------------------------ attention
def check_na():
    """K = 3: the LDS-tiled kernels and (direct=True) the run-time-K direct kernels; K = 5, 7: the direct kernels (natten kernel_size;
    core/LM_Net.py:81-84 carries [3, 5]).  Oracle: oracle/natten_ref.py for every K (test_oracle_na.py pins its K = 5 form)."""
    rows = []
    cases = [(3, s) for s in [(2, 7, 9, 1), (1, 6, 5, 2), (2, 5, 8, 4), (1, 9, 6, 8), (1, 3, 3, 2), (1, 3, 17, 1),
                              (2, 37, 41, 1), (1, 16, 52, 2), (2, 31, 18, 2), (1, 48, 33, 1),   # LDS-tiled / one-pass backward (C <= 24, maps >= 16)
                              (1, 17, 19, 1), (2, 47, 19, 2), (1, 32, 36, 2), (1, 62, 70, 1)]]  # tiles ending two short of the border (H - 2, W - 2 multiples of 15 / 17), several tiles per block
    cases += [(5, s) for s in [(2, 7, 9, 1), (1, 5, 5, 2), (1, 9, 6, 4), (2, 23, 18, 2), (1, 11, 12, 8), (1, 5, 21, 16)]]
    cases += [(7, s) for s in [(1, 7, 7, 1), (2, 9, 12, 2), (1, 22, 15, 4), (1, 8, 13, 8)]]
    for K, (B, H, W, hd) in cases:
        heads, Cn = 12, 12 * hd
        qkv = R(B, H, W, 3 * Cn, seed=91).requires_grad_(True)
        rpb = (R(heads, 2 * K - 1, 2 * K - 1, seed=92) * 0.5).requires_grad_(True)
        q, k, v = qkv.reshape(B, H, W, 3, heads, hd).permute(3, 0, 4, 1, 2, 5).unbind(0)
        attn = torch.softmax(natten_ref.na2d_qkrpb(q * hd ** -0.5, k, rpb, K), -1)
        o_ref = natten_ref.na2d_av(attn, v, K).permute(0, 2, 3, 1, 4).reshape(B, H, W, Cn)
        do = R(B, H, W, Cn, seed=93)
        o_ref.backward(do)
        tag = " K=%d hd=%d %dx%d" % (K, hd, H, W)
        if H * W <= 64:
            bf = natten_ref.na2d_bruteforce(q.detach() * hd ** -0.5, k.detach(), v.detach(), rpb.detach(), K)
            rows.append(("na oracle vec==bruteforce" + tag, rel(o_ref, bf.permute(0, 2, 3, 1, 4).reshape(B, H, W, Cn)), 1e-9))
        for direct in ((False, True) if K == 3 else (False,)):
            t2 = tag + (" direct" if direct else "")
            out = torch.full((B, H, W, Cn), float("nan"), device=DEV)
            hip.na_fwd(dev(qkv), dev(rpb), out, heads, direct=direct)
            rows.append(("na_fwd" + t2, rel(out, o_ref), TOL))
            dqkv = torch.full((B, H, W, 3 * Cn), float("nan"), device=DEV)
            drpb = torch.zeros(heads, 2 * K - 1, 2 * K - 1, device=DEV)
            hip.na_bwd(dev(qkv), dev(rpb), dev(do), dqkv, drpb, heads, direct=direct)
            rows.append(("na_bwd dqkv" + t2, rel(dqkv, qkv.grad), 2e-4))
            rows.append(("na_bwd drpb" + t2, rel(drpb, rpb.grad), 2e-4))
    return rows


def check_gattn():
    rows = []
    for (B, N, heads, hd) in [(2, 70, 12, 31), (1, 484, 12, 31), (1, 130, 3, 8), (1, 1024, 12, 31), (1, 2500, 2, 8)]:   # (512x512 / 800x800 inputs)
        Cn = heads * hd
        qkv = (R(B, N, 3 * Cn, seed=101) * 0.7).requires_grad_(True)
        q, k, v = qkv.view(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
        p = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, -1)
        o_ref = (p @ v).transpose(1, 2).reshape(B, N, Cn)
        do = R(B, N, Cn, seed=102)
        o_ref.backward(do)
        out = torch.full((B, N, Cn), float("nan"), device=DEV)
        lse = torch.empty(B, heads, N, device=DEV)
        hip.gattn_fwd(dev(qkv), out, lse, heads)
        tag = " N=%d hd=%d" % (N, hd)
        rows.append(("gattn_fwd" + tag, rel(out, o_ref), TOL))
        dqkv = torch.full((B, N, 3 * Cn), float("nan"), device=DEV)
        delta = torch.empty(B, heads, N, device=DEV)
        hip.gattn_bwd(dev(qkv), out, dev(do), lse, dqkv, delta, heads)
        rows.append(("gattn_bwd dqkv" + tag, rel(dqkv, qkv.grad), 2e-4))
    return rows


# ------------------------------------------------------------------------------------------------ norms
def check_ln():
    rows = []
    for Cn in (12, 24, 48, 96, 372):
        n = 531 if Cn < 100 else 77
        x = R(n, Cn, seed=111).requires_grad_(True)
        g, b = (R(Cn, seed=112).abs() + 0.5).requires_grad_(True), R(Cn, seed=113).requires_grad_(True)
        y_ref = F.layer_norm(x, (Cn,), g, b, 1e-5)
        dy, dres = R(n, Cn, seed=114), R(n, Cn, seed=115)
        y_ref.backward(dy)
        y = torch.full((n, Cn), float("nan"), device=DEV)
        hip.ln_fwd(dev(x), dev(g), dev(b), y)
        rows.append(("ln_fwd C=%d" % Cn, rel(y, y_ref), TOL))
        dx = torch.full((n, Cn), float("nan"), device=DEV)
        dg, db = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
        hip.ln_bwd(dev(x), dev(g), dev(dy), dev(dres), dx, dg, db)
        rows.append(("ln_bwd dx(+res) C=%d" % Cn, rel(dx, x.grad + dres), TOL))
        rows.append(("ln_bwd dgamma C=%d" % Cn, rel(dg, g.grad), 2e-4))
        rows.append(("ln_bwd dbeta C=%d" % Cn, rel(db, b.grad), 2e-4))
    return rows


def check_ln_linear():
    """LMN_SRC_LN: LayerNorm fused into the consuming Linear (norm1 -> qkv, norm2 -> fc1; /root/reference/core/modules.py:516-518,
    343-344, 50-56): y = W . LN(x) + b (+ residual) in ONE conv launch vs fp64 F.layer_norm + F.linear, the (mean, rstd) table the
    conv leaves for the backward, and the weight / bias gradient with the same transform on load.  Channel counts of the model
    (12 / 24: in-register statistics of the N-split kernel; 48 / 96 / 372: the M-split kernel's pre-pass) plus 36 / 72 -> narrow
    outputs, which reach the pre-pass through the `wide LayerNorm source` rule; pixel counts that leave partial tiles / chunks."""
    rows = []
    for Cn, Co, n in ((12, 36, 1043), (12, 24, 300), (24, 72, 777), (24, 48, 130), (48, 144, 333), (48, 96, 257), (96, 288, 203),
                      (96, 192, 64), (372, 1116, 99), (372, 744, 50), (36, 24, 145), (72, 12, 77), (20, 8, 35), (12, 36, 24), (24, 48, 7)):
        x = (R(n, Cn, seed=131) * 1.7 + 0.6).requires_grad_(True)
        g, b = (R(Cn, seed=132).abs() + 0.5).requires_grad_(True), (R(Cn, seed=133) * 0.3).requires_grad_(True)
        w, bias = R(Co, Cn, seed=134, scale=0.2).requires_grad_(True), R(Co, seed=135).requires_grad_(True)
        res = R(n, Co, seed=136)
        nrm = F.layer_norm(x, (Cn,), g, b, 1e-5)
        nrm.retain_grad()
        y_ref = F.linear(nrm, w, bias) + res
        dy = R(n, Co, seed=137)
        y_ref.backward(dy)
        tag = " C=%d->%d n=%d" % (Cn, Co, n)
        xd, dyd = dev(x).view(1, 1, n, Cn), dev(dy).view(1, 1, n, Co)
        stats = torch.full((n, 2), float("nan"), device=DEV)
        src = dict(view=xd, ln=(dev(g), dev(b), 1e-5, stats))
        wp = hip.conv_pack(dev(w).view(Co, Cn, 1, 1), 1, [Cn])
        y = torch.full((1, 1, n, Co), float("nan"), device=DEV)
        hip.conv_fwd([src], wp, y, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co, bias=dev(bias), residual=dev(res).view(1, 1, n, Co))
        rows.append(("LN+linear fwd" + tag, rel(y.view(n, Co), y_ref), TOL))
        mu = x.detach().mean(1)
        rs = 1.0 / torch.sqrt(x.detach().var(1, unbiased=False) + 1e-5)
        rows.append(("LN+linear stats table" + tag, max(rel(stats[:, 0], mu), rel(stats[:, 1], rs)), 2e-5))
        dW, db = torch.zeros(Co, Cn, 1, 1, device=DEV), torch.zeros(Co, device=DEV)
        hip.conv_wgrad([src], dyd, dW, db, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co)
        rows.append(("LN+linear wgrad dW" + tag, rel(dW.view(Co, Cn), w.grad), 2e-4))
        rows.append(("LN+linear wgrad db" + tag, rel(db, bias.grad), 2e-4))
        # the data gradient of the pair = Linear^T then lmn_ln_bwd on the un-normalised input (unchanged kernels): dx of the reference
        dn = torch.full((1, 1, n, Cn), float("nan"), device=DEV)
        wpt = hip.conv_pack_t(dev(w).view(Co, Cn, 1, 1), 1, 0, Cn, cred=Co)
        hip.conv_fwd([dyd], wpt, dn, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Cn, transposed=1)
        dx, dg, dbt = torch.full((n, Cn), float("nan"), device=DEV), torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
        hip.ln_bwd(dev(x), dev(g), dn.view(n, Cn), None, dx, dg, dbt)
        rows.append(("LN+linear dx (linear^T, ln_bwd)" + tag, rel(dx, x.grad), 2e-4))
        rows.append(("LN+linear dgamma / dbeta" + tag, max(rel(dg, g.grad), rel(dbt, b.grad)), 2e-4))
        # ... and the LayerNorm backward inside the data-gradient conv (LMN_EP_LN_BWD, C <= 48): dx (+ a residual gradient), d gamma /
        # d beta accumulated (+=) into a [2][C] pair in either row order
        if Cn <= 48:
            dres = R(n, Cn, seed=138)
            for swap in (0, 1):
                gpair = torch.full((2, Cn), 0.25, device=DEV)
                dx2 = torch.full((1, 1, n, Cn), float("nan"), device=DEV)
                hip.conv_fwd([dyd], wpt, dx2, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Cn, transposed=1, epilogue=hip.EP_LN_BWD, act=swap,
                             aux=xd, p=(dev(g), None, None, None, None, None, stats), residual=dev(dres).view(1, 1, n, Cn), stats=gpair,
                             stats_mode=hip.STATS_EP)
                rows.append(("LN backward in the conv epilogue: dx" + tag + " swap=%d" % swap, rel(dx2.view(n, Cn), x.grad + dres), 2e-4))
                dgf, dbf = (gpair[0], gpair[1]) if swap else (gpair[1], gpair[0])
                rows.append(("LN backward in the conv epilogue: dgamma / dbeta (+=)" + tag + " swap=%d" % swap,
                             max(rel(dgf - 0.25, g.grad), rel(dbf - 0.25, b.grad)), 2e-4))
    return rows


def check_conv_up2():
    """LMN_SRC_UP2: bilinear x2 (align_corners=True) sampled where the 3x3 conv stages its window (up1..4, the `convs` branch of
    the skip fusers; /root/reference/core/LM_Net.py:58-74,117-120, core/modules.py:94,129) vs fp64 F.interpolate + F.conv2d, and
    against the materialised form (lmn_up2_fwd + plain conv) of rounds 1-4.  Shapes: the four decoder levels' channel pairs at small
    sizes (N-split with one / two / three cout tiles, LDS-staged weights or not, the M-split kernel at Cout 96), odd tile remainders,
    a source map as small as 2x2."""
    rows = []
    for (B, h, w, Cin, Cout) in ((2, 9, 11, 24, 12), (1, 22, 13, 48, 24), (1, 12, 12, 96, 48), (1, 8, 7, 192, 96), (2, 2, 2, 12, 12),
                                 (1, 40, 37, 24, 12), (1, 5, 33, 36, 20)):
        x = R(B, Cin, h, w, seed=141)
        wt, bias = R(Cout, Cin, 3, 3, seed=142, scale=0.15), R(Cout, seed=143)
        res = R(B, Cout, 2 * h, 2 * w, seed=144)
        up_ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        y_ref = F.conv2d(up_ref, wt, bias, padding=1) + res
        tag = " %dx%d %d->%d" % (h, w, Cin, Cout)
        xd = nhwc(x)
        wp = hip.conv_pack(dev(wt), 3, [Cin])
        y = torch.full((B, 2 * h, 2 * w, Cout), float("nan"), device=DEV)
        hip.conv_fwd([dict(view=xd, flags=hip.SRC_UP2)], wp, y, B=B, Hin=2 * h, Win=2 * w, Hout=2 * h, Wout=2 * w, Cout=Cout, ksize=3,
                     bias=dev(bias), residual=nhwc(res))
        rows.append(("conv3x3(up2 on load)" + tag, rel(nchw(y), y_ref), TOL))
        upd = torch.full((B, 2 * h, 2 * w, Cin), float("nan"), device=DEV)
        hip.up2_fwd(xd, upd)
        y2 = torch.full((B, 2 * h, 2 * w, Cout), float("nan"), device=DEV)
        hip.conv_fwd([upd], wp, y2, B=B, Hin=2 * h, Win=2 * w, Hout=2 * h, Wout=2 * w, Cout=Cout, ksize=3, bias=dev(bias), residual=nhwc(res))
        rows.append(("conv3x3(up2 on load) vs up2_fwd + conv" + tag, rel(y, y2), 2e-6))
    # the weight gradient with the same sampling on load (wgrad3_kernel<..., UP>: upsampled maps >= 32 wide; 2x2-tile and one-tile
    # blocks, partial tiles, K-split with and without the deferred reduction) vs fp64 autograd through F.interpolate + F.conv2d
    for (B, h, w, Cin, Cout) in ((2, 16, 16, 24, 12), (1, 22, 37, 48, 24), (1, 17, 16, 12, 12), (2, 44, 44, 24, 12), (1, 20, 24, 192, 96), (1, 33, 18, 36, 20)):
        x = R(B, Cin, h, w, seed=151)
        wt = R(Cout, Cin, 3, 3, seed=152, scale=0.15).requires_grad_(True)
        bias = R(Cout, seed=153).requires_grad_(True)
        yr = F.conv2d(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True), wt, bias, padding=1)
        dy = R(*yr.shape, seed=154)
        yr.backward(dy)
        dW, db = torch.zeros(Cout, Cin, 3, 3, device=DEV), torch.zeros(Cout, device=DEV)
        hip.conv_wgrad([dict(view=nhwc(x), flags=hip.SRC_UP2)], nhwc(dy), dW, db, B=B, Hin=2 * h, Win=2 * w, Hout=2 * h, Wout=2 * w, Cout=Cout, ksize=3)
        tag = " %dx%d %d->%d" % (h, w, Cin, Cout)
        rows.append(("wgrad3x3(up2 on load) dW" + tag, rel(dW, wt.grad), 2e-4))
        rows.append(("wgrad3x3(up2 on load) db" + tag, rel(db, bias.grad), 2e-4))
    return rows


def check_fused_sources_modes():
    """the on-load / epilogue fusions of round 5 (LMN_SRC_LN, LMN_SRC_UP2, LMN_EP_LN_BWD) in the modes that only the
    end-to-end tests reached -- bf16 STORAGE (PM = 2 instances) and deterministic mode (slot path of the epilogue statistics) --
    against the UNFUSED kernels of the same mode at tight tolerances (the model-level bf16 tolerances cannot see a 1 % single-block
    error), plus LayerNorm rows whose channel 0 is far from the row mean (the M-split pre-pass sums about channel 0)."""
    rows = []
    bf = lambda t: t.to(torch.bfloat16)
    # ---- LayerNorm rows with an outlier in channel 0 (fp32): N-split (12, 24), M-split pre-pass (48, 96, 372)
    for Cn, Co, n in ((12, 36, 77), (24, 48, 130), (48, 96, 257), (96, 192, 64), (372, 744, 50)):
        x = R(n, Cn, seed=161) * 0.7 + 0.2
        x[:, 0] = 250.0 + R(n, seed=162) * 3.0
        x[::3, 0] = -400.0
        g, b = R(Cn, seed=163).abs() + 0.5, R(Cn, seed=164) * 0.3
        w, bias = R(Co, Cn, seed=165, scale=0.2), R(Co, seed=166)
        y_ref = F.linear(F.layer_norm(x, (Cn,), g, b, 1e-5), w, bias)
        stats = torch.full((n, 2), float("nan"), device=DEV)
        src = dict(view=dev(x).view(1, 1, n, Cn), ln=(dev(g), dev(b), 1e-5, stats))
        wp = hip.conv_pack(dev(w).view(Co, Cn, 1, 1), 1, [Cn])
        y = torch.full((1, 1, n, Co), float("nan"), device=DEV)
        hip.conv_fwd([src], wp, y, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co, bias=dev(bias))
        xf = dev(x).double().cpu()   # (statistics of the fp32-rounded input: 250 +- 3 loses 2e-5 absolute to the rounding itself)
        mu, rs = xf.mean(1), 1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)
        tag = " C=%d->%d n=%d" % (Cn, Co, n)
        rows.append(("LN+linear, channel-0 outlier: stats table" + tag, max(rel(stats[:, 0], mu), rel(stats[:, 1], rs)), 2e-5))
        rows.append(("LN+linear, channel-0 outlier: fwd" + tag, rel(y.view(n, Co), y_ref), 2e-4))
    # ---- bf16 storage: fused LN -> linear against ln_fwd + plain conv, its weight gradient against the one over the stored LN output
    mma0 = hip._MMA[0]
    hip._MMA[0] = hip.BF16
    try:
        for Cn, Co, n in ((12, 36, 1043), (24, 72, 300), (48, 96, 257), (96, 192, 64), (372, 744, 50)):
            x = bf(dev(R(n, Cn, seed=171) * 1.7 + 0.6))
            g, b = dev(R(Cn, seed=172).abs() + 0.5), dev(R(Cn, seed=173) * 0.3)
            w, bias = dev(R(Co, Cn, seed=174, scale=0.2)), dev(R(Co, seed=175))
            dy = bf(dev(R(n, Co, seed=176)))
            tag = " C=%d->%d n=%d" % (Cn, Co, n)
            wp = hip.conv_pack(w.view(Co, Cn, 1, 1), 1, [Cn])
            stats = torch.full((n, 2), float("nan"), device=DEV)
            src = dict(view=x.view(1, 1, n, Cn), ln=(g, b, 1e-5, stats))
            yf = torch.full((1, 1, n, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
            hip.conv_fwd([src], wp, yf, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co, bias=bias)
            nrm = torch.full((n, Cn), float("nan"), device=DEV, dtype=torch.bfloat16)
            hip.ln_fwd(x, g, b, nrm)
            yu = torch.full((1, 1, n, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
            hip.conv_fwd([nrm.view(1, 1, n, Cn)], wp, yu, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co, bias=bias)
            # (both round LN(x) to bf16 once; their fp32 values differ in the last bit, so single operands flip by one bf16 ulp)
            rows.append(("bf16 storage LN+linear fwd vs ln_fwd + conv" + tag, rel(yf.float(), yu.float()), 1.2e-2))
            dWf, dbf = torch.zeros(Co, Cn, 1, 1, device=DEV), torch.zeros(Co, device=DEV)
            hip.conv_wgrad([src], dy.view(1, 1, n, Co), dWf, dbf, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co)
            dWu, dbu = torch.zeros(Co, Cn, 1, 1, device=DEV), torch.zeros(Co, device=DEV)
            hip.conv_wgrad([nrm.view(1, 1, n, Cn)], dy.view(1, 1, n, Co), dWu, dbu, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co)
            rows.append(("bf16 storage LN+linear wgrad vs wgrad over ln_fwd" + tag, max(rel(dWf, dWu), rel(dbf, dbu)), 5e-3))
        for (B, h, w_, Cin, Cout) in ((2, 9, 11, 24, 12), (1, 22, 13, 48, 24), (1, 8, 7, 192, 96), (2, 16, 16, 24, 12), (1, 22, 37, 48, 24)):
            x = bf(nhwc(R(B, Cin, h, w_, seed=181)))
            wt, bias = dev(R(Cout, Cin, 3, 3, seed=182, scale=0.15)), dev(R(Cout, seed=183))
            tag = " %dx%d %d->%d" % (h, w_, Cin, Cout)
            wp = hip.conv_pack(wt, 3, [Cin])
            kw = dict(B=B, Hin=2 * h, Win=2 * w_, Hout=2 * h, Wout=2 * w_, Cout=Cout, ksize=3)
            yf = torch.full((B, 2 * h, 2 * w_, Cout), float("nan"), device=DEV, dtype=torch.bfloat16)
            hip.conv_fwd([dict(view=x, flags=hip.SRC_UP2)], wp, yf, bias=bias, **kw)
            up = torch.full((B, 2 * h, 2 * w_, Cin), float("nan"), device=DEV, dtype=torch.bfloat16)
            hip.up2_fwd(x, up)
            yu = torch.full((B, 2 * h, 2 * w_, Cout), float("nan"), device=DEV, dtype=torch.bfloat16)
            hip.conv_fwd([up], wp, yu, bias=bias, **kw)
            rows.append(("bf16 storage conv3x3(up2 on load) vs up2_fwd + conv" + tag, rel(yf.float(), yu.float()), 1.2e-2))
            if 2 * w_ >= 32:
                dy = bf(nhwc(R(B, Cout, 2 * h, 2 * w_, seed=184)))
                dWf, dbf = torch.zeros(Cout, Cin, 3, 3, device=DEV), torch.zeros(Cout, device=DEV)
                hip.conv_wgrad([dict(view=x, flags=hip.SRC_UP2)], dy, dWf, dbf, **kw)
                dWu, dbu = torch.zeros(Cout, Cin, 3, 3, device=DEV), torch.zeros(Cout, device=DEV)
                hip.conv_wgrad([up], dy, dWu, dbu, **kw)
                rows.append(("bf16 storage wgrad3x3(up2 on load) vs wgrad over up2_fwd" + tag, max(rel(dWf, dWu), rel(dbf, dbu)), 5e-3))
    finally:
        hip._MMA[0] = mma0
    # ---- deterministic mode (fp32): LN backward in the conv epilogue (slot path of d gamma / d beta, both row orders) against
    #      linear^T + lmn_ln_bwd, twice (bit-identical); the fused forwards are untouched by the mode but must still agree
    det0 = hip.get_deterministic()
    hip.set_deterministic(True)
    try:
        for Cn, Co, n in ((12, 36, 1043), (24, 72, 777), (48, 144, 333), (20, 8, 35)):
            x = dev(R(n, Cn, seed=191) * 1.7 + 0.6)
            g = dev(R(Cn, seed=192).abs() + 0.5)
            b = dev(R(Cn, seed=193) * 0.3)
            w = dev(R(Co, Cn, seed=194, scale=0.2))
            dy, dres = dev(R(n, Co, seed=195)), dev(R(n, Cn, seed=196))
            tag = " C=%d->%d n=%d" % (Cn, Co, n)
            stats = torch.full((n, 2), float("nan"), device=DEV)
            src = dict(view=x.view(1, 1, n, Cn), ln=(g, b, 1e-5, stats))
            wp = hip.conv_pack(w.view(Co, Cn, 1, 1), 1, [Cn])
            y = torch.full((1, 1, n, Co), float("nan"), device=DEV)
            hip.conv_fwd([src], wp, y, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Co)
            wpt = hip.conv_pack_t(w.view(Co, Cn, 1, 1), 1, 0, Cn, cred=Co)
            dn = torch.full((1, 1, n, Cn), float("nan"), device=DEV)
            hip.conv_fwd([dy.view(1, 1, n, Co)], wpt, dn, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Cn, transposed=1)
            dxu, dgu, dbu = torch.full((n, Cn), float("nan"), device=DEV), torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
            hip.ln_bwd(x, g, dn.view(n, Cn), None, dxu, dgu, dbu)
            for swap in (0, 1):
                got = []
                for rep in range(2):
                    gpair = torch.full((2, Cn), 0.25, device=DEV)
                    dx2 = torch.full((1, 1, n, Cn), float("nan"), device=DEV)
                    hip.conv_fwd([dy.view(1, 1, n, Co)], wpt, dx2, B=1, Hin=1, Win=n, Hout=1, Wout=n, Cout=Cn, transposed=1, epilogue=hip.EP_LN_BWD,
                                 act=swap, aux=x.view(1, 1, n, Cn), p=(g, None, None, None, None, None, stats), residual=dres.view(1, 1, n, Cn),
                                 stats=gpair, stats_mode=hip.STATS_EP)
                    got.append((dx2.clone(), gpair.clone()))
                dgf, dbf = (got[0][1][0], got[0][1][1]) if swap else (got[0][1][1], got[0][1][0])
                rows.append(("deterministic LN_BWD epilogue vs linear^T + ln_bwd: dx" + tag + " swap=%d" % swap, rel(got[0][0].view(n, Cn), dxu + dres), 2e-5))
                rows.append(("deterministic LN_BWD epilogue: dgamma / dbeta" + tag + " swap=%d" % swap, max(rel(dgf - 0.25, dgu), rel(dbf - 0.25, dbu)), 2e-5))
                same = bool(torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1]))
                rows.append(("deterministic LN_BWD epilogue: two launches bit-identical" + tag + " swap=%d" % swap, 0.0 if same else 1.0, 0.5))
    finally:
        hip.set_deterministic(det0)
    return rows


def check_conv_dma3():
    """conv_dma3_kernel (round 6): the LDS-DMA double-buffered 3x3 stride-1 kernel of the small-channel layers (skip-fuser convs, `nat`
    patch embedding and their data gradients at levels 0-1; /root/reference/core/modules.py:22-39,83-143, core/LM_Net.py:58-74) vs fp64
    F.conv2d / its autograd, and against the LDS-tiled kernel on the same call (lmn_conv_dma_config A/B).  Shapes: 12 / 24 source
    channels x one / two cout tiles, maps that are not multiples of the 8 x 16 tile (zero padding and partial tiles come from the
    DMA's bounds check), a one-tile-per-block and a many-tiles-per-block grid (prologue-only and steady-state pipeline), source / output
    slices of wider buffers, residual, SUM_SQ statistics about a shift (4 slices), deterministic slots, the transposed (data-gradient) form."""
    rows = []
    prev = hip.conv_dma_config(1, 1)
    try:
        for (B, H, W, Cin, Cout) in ((2, 40, 50, 12, 12), (1, 33, 37, 24, 12), (2, 24, 32, 12, 24), (1, 17, 64, 24, 24), (3, 96, 112, 24, 12),
                                     (8, 64, 96, 12, 12), (1, 8, 16, 24, 20), (1, 5, 7, 12, 8)):
            tag = " %dx%dx%d %d->%d" % (B, H, W, Cin, Cout)
            x = R(B, Cin + 8, H, W, seed=401)
            w = R(Cout, Cin, 3, 3, seed=402, scale=0.15).requires_grad_(True)
            b = R(Cout, seed=403)
            res = R(B, Cout, H, W, seed=404)
            xs = x[:, 4:4 + Cin].clone().requires_grad_(True)
            z = F.conv2d(xs, w, b, padding=1)
            y_ref = z + res
            xb = nhwc(x)
            wp = hip.conv_pack(dev(w.detach()), 3, [Cin])
            shift = R(Cout, seed=405) * 0.3
            got = {}
            for mode in (1, 0):
                hip.conv_dma_config(mode, 1)
                outb = torch.full((B, H, W, Cout + 12), float("nan"), device=DEV)
                srep = torch.zeros(4, 2, Cout, device=DEV)
                hip.conv_fwd([hip.V(xb, 4, Cin)], wp, hip.V(outb, 8, Cout), B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, ksize=3, bias=dev(b),
                             residual=nhwc(res), stats=srep, stats_mode=hip.STATS_SUM_SQ, stats_rep=4, p=(None, None, None, None, dev(shift)))
                got[mode] = (outb, srep.sum(0))
            hip.conv_dma_config(1, 1)
            rows.append(("conv_dma3 fwd (slices, bias, residual)" + tag, rel(nchw(got[1][0][..., 8:8 + Cout]), y_ref), TOL))
            rows.append(("conv_dma3 fwd vs LDS-tiled kernel" + tag, rel(got[1][0][..., 8:8 + Cout], got[0][0][..., 8:8 + Cout]), 2e-6))
            rows.append(("conv_dma3 untouched slices stay NaN" + tag, 0.0 if bool(torch.isnan(got[1][0][..., :8]).all() and torch.isnan(got[1][0][..., 8 + Cout:]).all()) else 1.0, 0.5))
            zd = (z - shift.view(1, -1, 1, 1)).detach()
            sref = torch.stack([zd.sum((0, 2, 3)), (zd * zd).sum((0, 2, 3))])
            rows.append(("conv_dma3 SUM_SQ statistics about a shift" + tag, rel(got[1][1], sref), 2e-4))
            # plain call (no residual, no statistics, whole tensors), and statistics only (out = None)
            xw = nhwc(xs.detach())
            y1 = torch.full((B, H, W, Cout), float("nan"), device=DEV)
            hip.conv_fwd([xw], wp, y1, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, ksize=3, bias=dev(b))
            rows.append(("conv_dma3 fwd plain" + tag, rel(nchw(y1), z), TOL))
            s1 = torch.zeros(2, Cout, device=DEV)
            hip.conv_fwd([xw], wp, None, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, ksize=3, bias=dev(b), stats=s1, stats_mode=hip.STATS_SUM_SQ)
            zz = z.detach()
            rows.append(("conv_dma3 statistics only" + tag, rel(s1, torch.stack([zz.sum((0, 2, 3)), (zz * zz).sum((0, 2, 3))])), 2e-4))
            # data gradient (transposed form: taps flipped, weights from conv_pack_t): dy has Cout channels -> only when the kernel takes them
            if Cout in (12, 24) and Cin <= 32:
                dy = R(B, Cout, H, W, seed=406)
                z.backward(dy)
                wpt = hip.conv_pack_t(dev(w.detach()), 3)
                acc = R(B, Cin, H, W, seed=407)
                dx = torch.full((B, H, W, Cin), float("nan"), device=DEV)
                hip.conv_fwd([nhwc(dy)], wpt, dx, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cin, ksize=3, transposed=1, residual=nhwc(acc))
                rows.append(("conv_dma3 data gradient (+ accumulate)" + tag, rel(nchw(dx), xs.grad + acc), TOL))
                hip.conv_dma_config(0, 1)
                dx0 = torch.full((B, H, W, Cin), float("nan"), device=DEV)
                hip.conv_fwd([nhwc(dy)], wpt, dx0, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cin, ksize=3, transposed=1, residual=nhwc(acc))
                hip.conv_dma_config(1, 1)
                rows.append(("conv_dma3 data gradient vs LDS-tiled kernel" + tag, rel(dx, dx0), 2e-6))
        # deterministic mode: statistics through the per-block slots, two launches bit-identical
        det0 = hip.get_deterministic()
        hip.set_deterministic(True)
        try:
            B, H, W, Cin, Cout = 2, 40, 50, 24, 12
            x, w, b = nhwc(R(B, Cin, H, W, seed=411)), R(Cout, Cin, 3, 3, seed=412, scale=0.15), dev(R(Cout, seed=413))
            wp = hip.conv_pack(dev(w), 3, [Cin])
            outs = []
            for rep in range(2):
                y = torch.full((B, H, W, Cout), float("nan"), device=DEV)
                st = torch.zeros(2, Cout, device=DEV)
                hip.conv_fwd([x], wp, y, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, ksize=3, bias=b, stats=st, stats_mode=hip.STATS_SUM_SQ)
                outs.append((y, st))
            zr = F.conv2d(nchw(x), w, b.double().cpu(), padding=1)
            rows.append(("conv_dma3 deterministic statistics", rel(outs[0][1], torch.stack([zr.sum((0, 2, 3)), (zr * zr).sum((0, 2, 3))])), 2e-4))
            rows.append(("conv_dma3 deterministic: two launches bit-identical", 0.0 if (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])) else 1.0, 0.5))
        finally:
            hip.set_deterministic(det0)
    finally:
        hip.conv_dma_config(prev if prev >= 0 else 7, 512)
    return rows


def check_conv_dmaM():
    """conv_dmaM_kernel (round 6): the M-split tile of the WIDE 3x3 stride-1 convs with window and weights by LDS-DMA (skip-fuser /
    bottleneck / decoder convs at 88^2 / 44^2 / 22^2 and their data gradients; /root/reference/core/modules.py:83-143, core/LM_Net.py:14-39)
    vs fp64 F.conv2d / its autograd, and against conv_tileM_kernel on the same call (lmn_conv_dma_config bit 2).  Shapes: the model's
    (48 -> 96 / 144, 96 -> 96 / 192, 372 -> 372 on their maps: a last K16 block of 4 channels), one- and two-stage layers (the counted
    waits of the weight ring run out differently there), maps that are not multiples of the tile, a grid whose blocks walk two tiles (the
    drain at a tile's start), source / output slices of wider buffers, residual, SUM_SQ statistics about a shift (4 slices), deterministic
    slots, the transposed (data-gradient) form."""
    rows = []
    prev = hip.conv_dma_config(7, 1)
    try:
        for (B, H, W, Cin, Cout) in ((2, 22, 22, 372, 372), (2, 44, 44, 96, 192), (1, 44, 44, 96, 96), (2, 88, 88, 48, 144), (1, 88, 88, 48, 96),
                                     (1, 19, 37, 52, 100), (9, 30, 41, 20, 112), (1, 5, 7, 36, 96), (2, 20, 20, 16, 96), (1, 12, 12, 8, 96), (40, 64, 64, 16, 100)):
            tag = " %dx%dx%d %d->%d" % (B, H, W, Cin, Cout)
            x = R(B, Cin + 8, H, W, seed=501)
            w = R(Cout, Cin, 3, 3, seed=502, scale=0.05).requires_grad_(True)
            b = R(Cout, seed=503)
            res = R(B, Cout, H, W, seed=504)
            xs = x[:, 4:4 + Cin].clone().requires_grad_(True)
            z = F.conv2d(xs, w, b, padding=1)
            y_ref = z + res
            xb = nhwc(x)
            wp = hip.conv_pack(dev(w.detach()), 3, [Cin])
            shift = R(Cout, seed=505) * 0.3
            got = {}
            for mode in (7, 3):
                hip.conv_dma_config(mode, 1)
                outb = torch.full((B, H, W, Cout + 12), float("nan"), device=DEV)
                srep = torch.zeros(4, 2, Cout, device=DEV)
                hip.conv_fwd([hip.V(xb, 4, Cin)], wp, hip.V(outb, 8, Cout), B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, ksize=3, bias=dev(b),
                             residual=nhwc(res), stats=srep, stats_mode=hip.STATS_SUM_SQ, stats_rep=4, p=(None, None, None, None, dev(shift)))
                got[mode] = (outb, srep.sum(0))
            hip.conv_dma_config(7, 1)
            rows.append(("conv_dmaM fwd (slices, bias, residual)" + tag, rel(nchw(got[7][0][..., 8:8 + Cout]), y_ref), TOL))
            rows.append(("conv_dmaM fwd vs conv_tileM_kernel" + tag, rel(got[7][0][..., 8:8 + Cout], got[3][0][..., 8:8 + Cout]), 4e-6))
            rows.append(("conv_dmaM untouched slices stay NaN" + tag, 0.0 if bool(torch.isnan(got[7][0][..., :8]).all() and torch.isnan(got[7][0][..., 8 + Cout:]).all()) else 1.0, 0.5))
            zd = (z - shift.view(1, -1, 1, 1)).detach()
            sref = torch.stack([zd.sum((0, 2, 3)), (zd * zd).sum((0, 2, 3))])
            rows.append(("conv_dmaM SUM_SQ statistics about a shift" + tag, rel(got[7][1], sref), 2e-4))
            xw = nhwc(xs.detach())
            y1 = torch.full((B, H, W, Cout), float("nan"), device=DEV)
            hip.conv_fwd([xw], wp, y1, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, ksize=3, bias=dev(b))
            rows.append(("conv_dmaM fwd plain" + tag, rel(nchw(y1), z), TOL))
            # data gradient (transposed form: taps flipped, weights from conv_pack_t) when ITS output (Cin channels) is wide enough for the M-split
            dy = R(B, Cout, H, W, seed=506)
            z.backward(dy)
            wpt = hip.conv_pack_t(dev(w.detach()), 3)
            acc = R(B, Cin, H, W, seed=507)
            dx = torch.full((B, H, W, Cin), float("nan"), device=DEV)
            hip.conv_fwd([nhwc(dy)], wpt, dx, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cin, ksize=3, transposed=1, residual=nhwc(acc))
            rows.append(("conv_dmaM data gradient (+ accumulate)" + tag, rel(nchw(dx), xs.grad + acc), TOL))
            hip.conv_dma_config(3, 1)
            dx0 = torch.full((B, H, W, Cin), float("nan"), device=DEV)
            hip.conv_fwd([nhwc(dy)], wpt, dx0, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cin, ksize=3, transposed=1, residual=nhwc(acc))
            hip.conv_dma_config(7, 1)
            rows.append(("conv_dmaM data gradient vs the LDS-tiled kernels" + tag, rel(dx, dx0), 4e-6))
        # deterministic mode: statistics through the per-block slots, two launches bit-identical
        det0 = hip.get_deterministic()
        hip.set_deterministic(True)
        try:
            B, H, W, Cin, Cout = 2, 44, 44, 96, 96
            x, w, b = nhwc(R(B, Cin, H, W, seed=511)), R(Cout, Cin, 3, 3, seed=512, scale=0.05), dev(R(Cout, seed=513))
            wp = hip.conv_pack(dev(w), 3, [Cin])
            outs = []
            for rep in range(2):
                y = torch.full((B, H, W, Cout), float("nan"), device=DEV)
                st = torch.zeros(2, Cout, device=DEV)
                hip.conv_fwd([x], wp, y, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, ksize=3, bias=b, stats=st, stats_mode=hip.STATS_SUM_SQ)
                outs.append((y, st))
            zr = F.conv2d(nchw(x), w, b.double().cpu(), padding=1)
            rows.append(("conv_dmaM deterministic statistics", rel(outs[0][1], torch.stack([zr.sum((0, 2, 3)), (zr * zr).sum((0, 2, 3))])), 2e-4))
            rows.append(("conv_dmaM deterministic: two launches bit-identical", 0.0 if (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])) else 1.0, 0.5))
        finally:
            hip.set_deterministic(det0)
    finally:
        hip.conv_dma_config(prev if prev >= 0 else 7, 512)
    return rows


def check_conv_dma1():
    """conv_dma1_kernel (round 6): the LDS-DMA streaming form of ReparamConv's 1x1 convs at levels 0-1 (/root/reference/core/modules.py:
    537-539, 576-599) -- expand conv (row-planar out + SUM_SQ statistics about a shift, in slices), pointwise + shortcut (row-planar
    source with on-load GELU x gate + NHWC source), SE-gradient conv (row-planar out and aux, per-image sums), three-source folded data
    gradient -- against the LDS-tiled kernel on the SAME call (lmn_conv_dma_config A/B; that kernel is pinned to fp64 by check_conv_fwd /
    check_zpath / check_conv_rp) and against fp64 directly for the expand conv and the pointwise pair.  Channel sets of level 0 (12 / 24),
    level 1 (24 / 48) and the RGB stem (4 -> 24); one tile per block, many tiles per block, several images per block (gate / SE-sum
    hand-over at the image change), deterministic mode."""
    rows = []
    to_rp, to_nhwc = hip.nhwc_to_rp4, hip.rp4_to_nhwc
    prev = hip.conv_dma_config(3, 1)
    mk = lambda *sh: torch.full(sh, float("nan"), device=DEV)
    try:
        for (B, H, W, Cin, E) in [(2, 16, 24, 12, 24), (3, 8, 16, 24, 48), (2, 32, 64, 12, 24), (9, 16, 8, 24, 48), (2, 16, 16, 4, 24), (1, 64, 128, 12, 24), (1, 32, 96, 24, 48)]:
            Cout = Cin
            tag = " Cin=%d E=%d %dx%dx%d" % (Cin, E, B, H, W)
            xr, prer = R(B, Cin, H, W, seed=501), R(B, E, H, W, seed=502)
            x, pre, dy, dh = nhwc(xr), nhwc(prer), nhwc(R(B, Cout, H, W, seed=503)), nhwc(R(B, E, H, W, seed=504))
            wer, ber = R(E, Cin, seed=505, scale=0.3), R(E, seed=506)
            we, be = dev(wer), dev(ber)
            wpwr, wscr, b2r = R(Cout, E, seed=507, scale=0.2), R(Cout, Cin, seed=508, scale=0.3), R(Cout, seed=510)
            wpw, wsc = dev(wpwr), dev(wscr)
            gater = R(B, E, seed=509).abs()
            gate = dev(gater)
            shift = dev(R(E, seed=511) * 0.2)
            pre_r, dh_r = to_rp(pre), to_rp(dh)
            kw = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W)
            res = {}
            for mode in (3, 0):
                hip.conv_dma_config(mode, 1)
                r = {}
                # F1: expand conv
                wpe = hip.conv_pack(we, 1, [Cin])
                z = hip.rp4(mk(B, H, W, E))
                st = torch.zeros(4, 2, E, device=DEV)
                hip.conv_fwd([x], wpe, z, Cout=E, bias=be, stats=st, stats_mode=hip.STATS_SUM_SQ, stats_rep=4, p=(None, None, None, None, shift), **kw)
                r["z"], r["zst"] = to_nhwc(z), st.sum(0)
                if Cin != 4:
                    # F2: pointwise + shortcut
                    n0, n1 = hip.conv_pack_size(1, Cout, [E]), hip.conv_pack_size(1, Cout, [Cin])
                    wp2 = torch.empty(n0 + n1, device=DEV)
                    hip.conv_pack(wpw, 1, [E], out=wp2[:n0])
                    hip.conv_pack(wsc, 1, [Cin], out=wp2[n0:n0 + n1])
                    y = mk(B, H, W, Cout)
                    hip.conv_fwd([dict(view=pre_r, scale=gate, flags=hip.SRC_GELU), x], wp2, y, Cout=Cout, bias=dev(b2r), **kw)
                    r["y"] = y
                    # B1: SE-gradient conv
                    wpt = hip.conv_pack_t(wpw, 1, 0, E, cred=Cout)
                    u = hip.rp4(mk(B, H, W, E))
                    ds = torch.zeros(B, E, device=DEV)
                    hip.conv_fwd([dy], wpt, u, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=pre_r, stats=ds, stats_mode=hip.STATS_EP, **kw)
                    r["u"], r["ds"] = to_nhwc(u), ds
                    # B2: three-source data gradient (any packed operator of that shape: expand^T | square | shortcut^T)
                    wa, wb, wc = dev(R(Cin, E, seed=512, scale=0.2)), dev(R(Cin, Cin, seed=513, scale=0.3)), dev(R(Cin, Cout, seed=514, scale=0.3))
                    m0, m1, m2 = hip.conv_pack_size(1, Cin, [E]), hip.conv_pack_size(1, Cin, [Cin]), hip.conv_pack_size(1, Cin, [Cout])
                    wp3 = torch.empty(m0 + m1 + m2, device=DEV)
                    hip.conv_pack(wa, 1, [E], out=wp3[:m0]); hip.conv_pack(wb, 1, [Cin], out=wp3[m0:m0 + m1]); hip.conv_pack(wc, 1, [Cout], out=wp3[m0 + m1:])
                    dx = mk(B, H, W, Cin)
                    hip.conv_fwd([dh_r, x, dy], wp3, dx, Cout=Cin, bias=dev(R(Cin, seed=515)), **kw)
                    r["dx"] = dx
                res[mode] = r
            hip.conv_dma_config(3, 1)
            for k_ in res[3]:
                scale_ = 1e-2 if k_ in ("zst", "ds") else 1.0       # (sums: float-atomic order)
                rows.append(("conv_dma1 vs LDS-tiled kernel: %s%s" % (k_, tag), rel(res[3][k_], res[0][k_]) * scale_, 2e-6))
            zref = F.conv2d(xr, wer.view(E, Cin, 1, 1), ber)
            rows.append(("conv_dma1 expand conv vs fp64" + tag, rel(nchw(res[3]["z"]), zref), TOL))
            zd = zref - shift.double().cpu().view(1, -1, 1, 1)
            rows.append(("conv_dma1 expand statistics vs fp64" + tag, rel(res[3]["zst"], torch.stack([zd.sum((0, 2, 3)), (zd * zd).sum((0, 2, 3))])), 2e-4))
            if Cin != 4:
                yref = F.conv2d(gelu(prer) * gater.view(B, E, 1, 1), wpwr.view(Cout, E, 1, 1)) + F.conv2d(xr, wscr.view(Cout, Cin, 1, 1), b2r)
                rows.append(("conv_dma1 pointwise + shortcut vs fp64" + tag, rel(nchw(res[3]["y"]), yref), TOL))
        # chained pairs (lmn_conv_chain_t): pointwise + shortcut of block A -> expand conv of block B (statistics about a shift, slices,
        # snapshot), folded data gradient of block B -> SE-gradient conv of block A, against the two launches they replace
        for (B, H, W, Cin, E) in [(2, 16, 24, 12, 24), (3, 8, 16, 24, 48), (2, 32, 64, 12, 24), (9, 16, 8, 12, 24), (1, 64, 128, 24, 48)]:
            tag = " Cin=%d E=%d %dx%dx%d" % (Cin, E, B, H, W)
            x, pre_r, dy = nhwc(R(B, Cin, H, W, seed=531)), to_rp(nhwc(R(B, E, H, W, seed=532))), nhwc(R(B, Cin, H, W, seed=533))
            dh_r, preA_r = to_rp(nhwc(R(B, E, H, W, seed=534))), to_rp(nhwc(R(B, E, H, W, seed=535)))
            gate = dev(R(B, E, seed=536).abs())
            wpw, wsc, weB = dev(R(Cin, E, seed=537, scale=0.2)), dev(R(Cin, Cin, seed=538, scale=0.3)), dev(R(E, Cin, seed=539, scale=0.3))
            bpw, bsc, beB, shift = dev(R(Cin, seed=540)), dev(R(Cin, seed=541)), dev(R(E, seed=542)), dev(R(E, seed=543) * 0.2)
            kw = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W)
            n0, n1 = hip.conv_pack_size(1, Cin, [E]), hip.conv_pack_size(1, Cin, [Cin])
            wp2 = torch.empty(n0 + n1, device=DEV)
            hip.conv_pack(wpw, 1, [E], out=wp2[:n0]); hip.conv_pack(wsc, 1, [Cin], out=wp2[n0:])
            wpeB = hip.conv_pack(weB, 1, [Cin])
            srcs = [dict(view=pre_r, scale=gate, flags=hip.SRC_GELU), x]
            # separate launches
            y0, z0, st0_ = mk(B, H, W, Cin), hip.rp4(mk(B, H, W, E)), torch.zeros(5, 2, E, device=DEV)
            hip.conv_fwd(srcs, wp2, y0, Cout=Cin, bias=bpw, bias2=bsc, **kw)
            hip.conv_fwd([y0], wpeB, z0, Cout=E, bias=beB, stats=st0_, stats_mode=hip.STATS_SUM_SQ, stats_rep=4, stats_snap=True, p=(None, None, None, None, shift), **kw)
            # one launch
            y1, z1, st1_ = mk(B, H, W, Cin), hip.rp4(mk(B, H, W, E)), torch.zeros(5, 2, E, device=DEV)
            ch = dict(wpack=wpeB, Cout=E, out=z1, bias=beB, shift=shift, stats=st1_, stats_mode=hip.STATS_SUM_SQ, stats_rep=4, stats_snap=True)
            if Cin == 24 and E == 48 and H * W % 64 == 0 or Cin == 12:
                ok = hip.conv_fwd(srcs, wp2, y1, Cout=Cin, bias=bpw, bias2=bsc, chain=ch, query_chain=True, **kw)
                rows.append(("conv chain F2 -> F1': accepted" + tag, 0.0 if ok else 1.0, 0.5))
                if ok:
                    hip.conv_fwd(srcs, wp2, y1, Cout=Cin, bias=bpw, bias2=bsc, chain=ch, **kw)
                    rows.append(("conv chain F2 -> F1': first output" + tag, rel(y1, y0), 1e-7))
                    rows.append(("conv chain F2 -> F1': chained output" + tag, rel(to_nhwc(z1), to_nhwc(z0)), 2e-6))
                    rows.append(("conv chain F2 -> F1': statistics slices + snapshot" + tag, max(rel(st1_[:4].sum(0), st0_[:4].sum(0)) * 1e-2, rel(st1_[4, 0], st0_[4, 0])), 2e-6))
            if Cin == 12:
                wa, wb, wc = dev(R(Cin, E, seed=544, scale=0.2)), dev(R(Cin, Cin, seed=545, scale=0.3)), dev(R(Cin, Cin, seed=546, scale=0.3))
                m0, m1 = hip.conv_pack_size(1, Cin, [E]), hip.conv_pack_size(1, Cin, [Cin])
                wp3 = torch.empty(m0 + 2 * m1, device=DEV)
                hip.conv_pack(wa, 1, [E], out=wp3[:m0]); hip.conv_pack(wb, 1, [Cin], out=wp3[m0:m0 + m1]); hip.conv_pack(wc, 1, [Cin], out=wp3[m0 + m1:])
                wptA = hip.conv_pack_t(dev(R(Cin, E, seed=547, scale=0.2)), 1, 0, E, cred=Cin)
                kb = dev(R(Cin, seed=548))
                dx0, u0, ds0 = mk(B, H, W, Cin), hip.rp4(mk(B, H, W, E)), torch.zeros(B, E, device=DEV)
                hip.conv_fwd([dh_r, x, dy], wp3, dx0, Cout=Cin, bias=kb, **kw)
                hip.conv_fwd([dx0], wptA, u0, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=preA_r, stats=ds0, stats_mode=hip.STATS_EP, **kw)
                dx1, u1, ds1 = mk(B, H, W, Cin), hip.rp4(mk(B, H, W, E)), torch.zeros(B, E, device=DEV)
                ch = dict(wpack=wptA, Cout=E, out=u1, aux=preA_r, stats=ds1, epilogue=hip.EP_SE_BWD, stats_mode=hip.STATS_EP)
                ok = hip.conv_fwd([dh_r, x, dy], wp3, dx1, Cout=Cin, bias=kb, chain=ch, query_chain=True, **kw)
                rows.append(("conv chain B2 -> B1': accepted" + tag, 0.0 if ok else 1.0, 0.5))
                if ok:
                    hip.conv_fwd([dh_r, x, dy], wp3, dx1, Cout=Cin, bias=kb, chain=ch, **kw)
                    rows.append(("conv chain B2 -> B1': first output" + tag, rel(dx1, dx0), 1e-7))
                    rows.append(("conv chain B2 -> B1': chained output" + tag, rel(to_nhwc(u1), to_nhwc(u0)), 2e-6))
                    rows.append(("conv chain B2 -> B1': per-image sums" + tag, rel(ds1, ds0) * 1e-2, 2e-6))
        # a call the table does not hold must be refused by the query (and by the launch)
        xb, wq = nhwc(R(1, 96, 8, 16, seed=551)), hip.conv_pack(dev(R(96, 96, seed=552)), 1, [96])
        rows.append(("conv chain: refused outside the instance table", 1.0 if hip.conv_fwd([xb], wq, mk(1, 8, 16, 96), B=1, Hin=8, Win=16, Hout=8, Wout=16, Cout=96,
                     chain=dict(wpack=wq, Cout=96, out=mk(1, 8, 16, 96), stats=torch.zeros(2, 96, device=DEV), stats_mode=hip.STATS_SUM_SQ), query_chain=True) else 0.0, 0.5))
        # deterministic mode: the statistics of F1 and the per-image sums of B1 go through slots; two launches bit-identical
        det0 = hip.get_deterministic()
        hip.set_deterministic(True)
        try:
            B, H, W, Cin, E = 3, 16, 24, 12, 24
            x, pre_r, dy = nhwc(R(B, Cin, H, W, seed=521)), to_rp(nhwc(R(B, E, H, W, seed=522))), nhwc(R(B, Cin, H, W, seed=523))
            we, wpw = dev(R(E, Cin, seed=524, scale=0.3)), dev(R(Cin, E, seed=525, scale=0.2))
            kw = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W)
            outs = {}
            for mode in (3, 3, 0):
                hip.conv_dma_config(mode, 1)
                z, st = hip.rp4(mk(B, H, W, E)), torch.zeros(2, E, device=DEV)
                hip.conv_fwd([x], hip.conv_pack(we, 1, [Cin]), z, Cout=E, stats=st, stats_mode=hip.STATS_SUM_SQ, **kw)
                u, ds = hip.rp4(mk(B, H, W, E)), torch.zeros(B, E, device=DEV)
                hip.conv_fwd([dy], hip.conv_pack_t(wpw, 1, 0, E, cred=Cin), u, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=pre_r, stats=ds, stats_mode=hip.STATS_EP, **kw)
                outs.setdefault(mode, []).append((z.clone(), st.clone(), u.clone(), ds.clone()))
            a, b_, c = outs[3][0], outs[3][1], outs[0][0]
            rows.append(("conv_dma1 deterministic: two launches bit-identical", 0.0 if all(torch.equal(p_, q_) for p_, q_ in zip(a, b_)) else 1.0, 0.5))
            rows.append(("conv_dma1 deterministic vs LDS-tiled kernel", max(rel(a[0], c[0]), rel(a[1], c[1]) * 1e-2, rel(a[2], c[2]), rel(a[3], c[3]) * 1e-2), 2e-6))
        finally:
            hip.set_deterministic(det0)
    finally:
        hip.conv_dma_config(prev if prev >= 0 else 7, 512)
    return rows


def check_bn_tail():
    """BatchNorm(batch stats)+GELU tail: bn_finalize, bnact_fwd, bnact_bwd_stats, bn_bwd_coef, bnact_bwd, colsum."""
    rows = []
    for Cn in (12, 24, 96):
        n = 403
        z = (R(n, Cn, seed=121) * 1.3 + 0.2).requires_grad_(True)
        g, b = (R(Cn, seed=122).abs() + 0.5).requires_grad_(True), R(Cn, seed=123).requires_grad_(True)
        rm, rv = R(Cn, seed=124) * 0.1, R(Cn, seed=125).abs() + 0.5
        rm_ref, rv_ref = rm.clone(), rv.clone()
        h = F.batch_norm(z, rm_ref, rv_ref, g, b, True, 0.1, 1e-5)
        y_ref = gelu(h)
        dy = R(n, Cn, seed=126)
        y_ref.backward(dy)
        zd = dev(z)
        sums = torch.zeros(2, Cn, device=DEV)
        hip.colsum(zd, sums[0])
        hip.colsum((zd * zd).contiguous(), sums[1])
        rows.append(("colsum C=%d" % Cn, rel(sums[0], z.detach().sum(0)), 2e-4))
        mean, rstd, A, shift = (torch.empty(Cn, device=DEV) for _ in range(4))
        rmd, rvd = dev(rm), dev(rv)
        hip.bn_finalize(sums, n, dev(g), dev(b), 1e-5, 0.1, mean, rstd, A, shift, rmd, rvd)
        rows.append(("bn_finalize running_mean C=%d" % Cn, rel(rmd, rm_ref), TOL))
        rows.append(("bn_finalize running_var C=%d" % Cn, rel(rvd, rv_ref), TOL))
        y = torch.full((n, Cn), float("nan"), device=DEV)
        hip.bnact_fwd(zd, A, shift, y, hip.ACT_GELU)
        rows.append(("bnact_fwd C=%d" % Cn, rel(y, y_ref), TOL))
        bst = torch.zeros(2, Cn, device=DEV)
        hip.bnact_bwd_stats(zd, dev(dy), mean, rstd, dev(g), dev(b), bst, hip.ACT_GELU)
        dg, db = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
        c1, c2, c3 = (torch.empty(Cn, device=DEV) for _ in range(3))
        hip.bn_bwd_coef(bst, n, A, dg, db, c1, c2, c3, True)
        rows.append(("bn dgamma C=%d" % Cn, rel(dg, g.grad), 2e-4))
        rows.append(("bn dbeta C=%d" % Cn, rel(db, b.grad), 2e-4))
        dz = torch.full((n, Cn), float("nan"), device=DEV)
        hip.bnact_bwd(zd, dev(dy), mean, rstd, dev(g), dev(b), c1, c2, c3, dz, hip.ACT_GELU)
        rows.append(("bnact_bwd dz C=%d" % Cn, rel(dz, z.grad), 2e-4))
        # the same tails with the finalize / coefficient launches inside (lmn_bnact_fwd_fin / lmn_bnact_bwd_fin): sums taken about a
        # snapshot of the running mean, as the skip fusers' conv leaves them
        about = dev(rm)
        sums2 = torch.zeros(2, 2, Cn, device=DEV)
        hip.colsum((zd - about).contiguous(), sums2[0, 0])
        hip.colsum(((zd - about) * (zd - about)).contiguous(), sums2[0, 1])
        sums2[1, 0] = about
        mean2, rstd2, A2, shift2 = (torch.full((Cn,), float("nan"), device=DEV) for _ in range(4))
        rmd2, rvd2 = dev(rm), dev(rv)
        y2 = torch.full((n, Cn), float("nan"), device=DEV)
        hip.bnact_fwd_fin(zd, dict(mode=hip.FIN_BN, sums=sums2, nrep=1, count=n, gamma=dev(g), beta=dev(b), eps=1e-5, momentum=0.1,
                                   about=sums2[1, 0], mean=mean2, rstd=rstd2, A=A2, shift=shift2, rmean=rmd2, rvar=rvd2), y2, hip.ACT_GELU)
        rows.append(("bnact_fwd_fin y C=%d" % Cn, rel(y2, y_ref), TOL))
        rows.append(("bnact_fwd_fin running_mean C=%d" % Cn, rel(rmd2, rm_ref), TOL))
        rows.append(("bnact_fwd_fin running_var C=%d" % Cn, rel(rvd2, rv_ref), TOL))
        rows.append(("bnact_fwd_fin mean / rstd / A / shift vs bn_finalize C=%d" % Cn,
                     max(rel(mean2, mean), rel(rstd2, rstd), rel(A2, A), rel(shift2, shift)), 2e-5))
        dg2, db2 = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
        dz2 = torch.full((n, Cn), float("nan"), device=DEV)
        hip.bnact_bwd_fin(zd, dev(dy), mean, rstd, dev(g), dev(b), dict(mode=hip.FIN_BN_BWD, sums=bst, nrep=1, count=n, batch_stats=1,
                                                                         Ain=A, dgamma=dg2, dbeta=db2), dz2, hip.ACT_GELU)
        rows.append(("bnact_bwd_fin dz == bn_bwd_coef + bnact_bwd (bitwise) C=%d" % Cn, 0.0 if torch.equal(dz2, dz) else 1.0, 0.5))
        rows.append(("bnact_bwd_fin dgamma / dbeta C=%d" % Cn, max(rel(dg2, g.grad), rel(db2, b.grad)), 2e-4))
    return rows


def check_conv_bn_epilogues(B=2, H=9, W=8, tag=""):
    """expand_conv (1x1 + batch-stat BN + Hardswish) backward through the conv epilogues BN_BWD1/BN_BWD2,
    and the SE_BWD / DGELU epilogues."""
    rows = []
    Cin, E = 12, 24
    x = R(B, Cin, H, W, seed=131)
    w, b = R(E, Cin, 1, 1, seed=132, scale=0.4).requires_grad_(True), R(E, seed=133).requires_grad_(True)
    g, be = (R(E, seed=134).abs() + 0.5).requires_grad_(True), R(E, seed=135).requires_grad_(True)
    z = F.conv2d(x, w, b)
    hh = F.batch_norm(z, None, None, g, be, True, 0.1, 1e-5)
    x1 = hh * torch.clamp(hh + 3, 0, 6) / 6
    dx1 = R(B, E, H, W, seed=136)
    zr = z.detach().clone().requires_grad_(True)
    h2 = F.batch_norm(zr, None, None, g.detach(), be.detach(), True, 0.1, 1e-5)
    (h2 * torch.clamp(h2 + 3, 0, 6) / 6).backward(dx1)
    x1.backward(dx1)
    N = B * H * W
    mean = z.detach().mean((0, 2, 3))
    rstd = 1 / torch.sqrt(z.detach().var((0, 2, 3), unbiased=False) + 1e-5)
    wp = hip.conv_pack(dev(w), 1, [Cin])
    dh = torch.full((B, H, W, E), float("nan"), device=DEV)
    st = torch.zeros(2, E, device=DEV)
    xd = nhwc(x)
    hip.conv_fwd([xd], wp, dh, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=dev(b), epilogue=hip.EP_BN_BWD1,
                 act=hip.ACT_HSWISH, p=(dev(mean), dev(rstd), dev(g), dev(be)), aux=nhwc(dx1), stats=st,
                 stats_mode=hip.STATS_EP)
    A = dev(g.detach() * rstd)
    dg, db_, c1, c2, c3 = (torch.zeros(E, device=DEV) for _ in range(5))
    hip.bn_bwd_coef(st, N, A, dg, db_, c1, c2, c3, True)
    rows.append(("conv BN_BWD1 dh" + tag, rel(nchw(dh), (zr.grad * 0 + 1) * 0 + _dh_ref(z, g, be, dx1)), TOL))
    rows.append(("conv BN_BWD1 dgamma" + tag, rel(dg, g.grad), 2e-4))
    rows.append(("conv BN_BWD1 dbeta" + tag, rel(db_, be.grad), 2e-4))
    dz = torch.full((B, H, W, E), float("nan"), device=DEV)
    hip.conv_fwd([xd], wp, dz, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=dev(b), epilogue=hip.EP_BN_BWD2,
                 p=(dev(mean), dev(rstd), c1, c2, c3), aux=dh)
    rows.append(("conv BN_BWD2 dz" + tag, rel(nchw(dz), zr.grad), 2e-4))
    # the forms the engine uses: pass 1 statistics only (no dh), pass 2 forms dh from dx1 itself (p5 / p6 = gamma / beta)
    st2 = torch.zeros(2, E, device=DEV)
    hip.conv_fwd([xd], wp, None, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=dev(b), epilogue=hip.EP_BN_BWD1,
                 act=hip.ACT_HSWISH, p=(dev(mean), dev(rstd), dev(g), dev(be)), aux=nhwc(dx1), stats=st2,
                 stats_mode=hip.STATS_EP)
    rows.append(("conv BN_BWD1 statistics-only pass" + tag, rel(st2, st), 2e-5))
    dz2 = torch.full((B, H, W, E), float("nan"), device=DEV)
    hip.conv_fwd([xd], wp, dz2, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=dev(b), epilogue=hip.EP_BN_BWD2,
                 act=hip.ACT_HSWISH, p=(dev(mean), dev(rstd), c1, c2, c3, dev(g), dev(be)), aux=nhwc(dx1))
    rows.append(("conv BN_BWD2 with the activation derivative fused (dz from dx1)" + tag, rel(nchw(dz2), zr.grad), 2e-4))

    # the same pass 2 with c1 / c2 / c3 and the gamma / beta gradients formed in-kernel (lmn_bn_fin_t, LMN_FIN_BN_BWD)
    dz3 = torch.full((B, H, W, E), float("nan"), device=DEV)
    dg3, db3 = torch.zeros(E, device=DEV), torch.zeros(E, device=DEV)
    hip.conv_fwd([xd], wp, dz3, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=dev(b), epilogue=hip.EP_BN_BWD2,
                 act=hip.ACT_HSWISH, p=(dev(mean), dev(rstd), None, None, None, dev(g), dev(be)), aux=nhwc(dx1),
                 fin=dict(mode=hip.FIN_BN_BWD, sums=st2, nrep=1, count=N, batch_stats=True, Ain=A, dgamma=dg3, dbeta=db3))
    rows.append(("conv BN_BWD2 with in-kernel coefficients: dz" + tag, rel(nchw(dz3), zr.grad), 2e-4))
    rows.append(("conv BN_BWD2 with in-kernel coefficients: dgamma" + tag, rel(dg3, g.grad), 2e-4))
    rows.append(("conv BN_BWD2 with in-kernel coefficients: dbeta" + tag, rel(db3, be.grad), 2e-4))
    # forward: statistics pass about a running mean (snapshot behind the slices) + applying pass with in-kernel finalize
    # (LMN_FIN_BN) against lmn_bn_finalize and against torch's batch norm; running statistics updated once
    REP = 4
    rm0, rv0 = R(E, seed=141, scale=0.3), R(E, seed=142).abs() + 0.5
    rm_a, rv_a, rm_b, rv_b = dev(rm0).clone(), dev(rv0).clone(), dev(rm0).clone(), dev(rv0).clone()
    sums = torch.zeros(REP + 1, 2, E, device=DEV)
    hip.conv_fwd([xd], wp, None, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=dev(b), stats=sums, stats_mode=hip.STATS_SUM_SQ,
                 stats_rep=REP, stats_snap=True, p=(None, None, None, None, rm_a))
    rows.append(("conv SUM_SQ statistics: shift snapshot" + tag, rel(sums[REP, 0], dev(rm0)), 1e-7))
    m_a, r_a, A_a, s_a = (torch.zeros(E, device=DEV) for _ in range(4))
    x1k = torch.full((B, H, W, E), float("nan"), device=DEV)
    hip.conv_fwd([xd], wp, x1k, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=dev(b), epilogue=hip.EP_AFFINE_ACT,
                 act=hip.ACT_HSWISH, p=(A_a, s_a),
                 fin=dict(mode=hip.FIN_BN, sums=sums, nrep=REP, count=N, gamma=dev(g), beta=dev(be), eps=1e-5, momentum=0.1,
                          about=sums[REP, 0], mean=m_a, rstd=r_a, A=A_a, shift=s_a, rmean=rm_a, rvar=rv_a))
    m_b, r_b, A_b, s_b = (torch.zeros(E, device=DEV) for _ in range(4))
    hip.bn_finalize(sums[:REP], N, dev(g), dev(be), 1e-5, 0.1, m_b, r_b, A_b, s_b, rm_b, rv_b, about=sums[REP, 0])
    rows.append(("conv AFFINE_ACT with in-kernel BN finalize: x1" + tag, rel(nchw(x1k), x1.detach()), 2e-4))
    for nm, ta, tb in (("mean", m_a, m_b), ("rstd", r_a, r_b), ("A", A_a, A_b), ("shift", s_a, s_b), ("running mean", rm_a, rm_b),
                       ("running var", rv_a, rv_b)):
        rows.append(("in-kernel BN finalize == lmn_bn_finalize: %s" % nm + tag, rel(ta, tb), 1e-6))
    rows.append(("in-kernel BN finalize: mean vs torch" + tag, rel(m_a, mean), 2e-5))

    # SE_BWD: o = v, stats[b][c] += v * gelu(aux)
    pre = R(B, E, H, W, seed=137)
    dy = R(B, Cin, H, W, seed=138)
    wpw = R(Cin, E, 1, 1, seed=139, scale=0.3)          # forward pointwise E -> Cin ; data gradient Cin -> E
    u_ref = F.conv_transpose2d(dy, wpw)
    ds_ref = (u_ref * gelu(pre)).sum((2, 3))
    wpt = hip.conv_pack_t(dev(wpw), 1)
    u = torch.full((B, H, W, E), float("nan"), device=DEV)
    ds = torch.zeros(B, E, device=DEV)
    hip.conv_fwd([nhwc(dy)], wpt, u, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD,
                 aux=nhwc(pre), stats=ds, stats_mode=hip.STATS_EP)
    rows.append(("conv SE_BWD u" + tag, rel(nchw(u), u_ref), TOL))
    rows.append(("conv SE_BWD ds" + tag, rel(ds, ds_ref), 2e-4))
    # DGELU
    o = torch.full((B, H, W, E), float("nan"), device=DEV)
    hip.conv_fwd([nhwc(dy)], wpt, o, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, transposed=1, epilogue=hip.EP_DGELU,
                 aux=nhwc(pre))
    pl = pre.clone().requires_grad_(True)
    gelu(pl).backward(u_ref)
    rows.append(("conv DGELU" + tag, rel(nchw(o), pl.grad), TOL))
    return rows


def _dh_ref(z, g, be, dx1):
    """dL/dh for h = BN(z) (batch stats), x1 = hardswish(h): dx1 * hardswish'(h) in fp64."""
    zd = z.detach()
    mean = zd.mean((0, 2, 3), keepdim=True)
    var = zd.var((0, 2, 3), unbiased=False, keepdim=True)
    h = (zd - mean) / torch.sqrt(var + 1e-5) * g.detach().view(1, -1, 1, 1) + be.detach().view(1, -1, 1, 1)
    d = torch.where(h < -3, torch.zeros_like(h), torch.where(h <= 3, h / 3 + 0.5, torch.ones_like(h)))
    return dx1 * d


# ------------------------------------------------------------------------------------------------ resampling / layout
def check_resample():
    rows = []
    for (B, H, W, Cn) in [(2, 5, 7, 12), (1, 11, 11, 24), (1, 22, 22, 48)]:
        x = R(B, Cn, H, W, seed=141).requires_grad_(True)
        y_ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        dy = R(*y_ref.shape, seed=142)
        y_ref.backward(dy)
        y = torch.full((B, 2 * H, 2 * W, Cn), float("nan"), device=DEV)
        hip.up2_fwd(nhwc(x), y)
        rows.append(("up2_fwd %dx%d" % (H, W), rel(nchw(y), y_ref), TOL))
        dx = torch.full((B, H, W, Cn), float("nan"), device=DEV)
        hip.up2_bwd(nhwc(dy), dx)
        rows.append(("up2_bwd %dx%d" % (H, W), rel(nchw(dx), x.grad), TOL))
    for (B, Ho, Wo, f, Cn) in [(2, 3, 4, 16, 12), (1, 5, 3, 8, 24), (2, 4, 4, 2, 96)]:
        x = R(B, Cn, Ho * f, Wo * f, seed=143).requires_grad_(True)
        y_ref = F.adaptive_avg_pool2d(x, (Ho, Wo))
        dy = R(*y_ref.shape, seed=144)
        y_ref.backward(dy)
        buf = torch.zeros(B, Ho, Wo, Cn + 8, device=DEV)
        hip.avgpool_fwd(nhwc(x), hip.V(buf, 4, Cn), f)
        rows.append(("avgpool_fwd f=%d (into slice)" % f, rel(nchw(buf[..., 4:4 + Cn]), y_ref), TOL))
        dybuf = torch.zeros(B, Ho, Wo, Cn + 8, device=DEV)
        dybuf[..., 4:4 + Cn] = nhwc(dy)
        base = R(B, Cn, Ho * f, Wo * f, seed=145)
        dx = nhwc(base).clone()
        hip.avgpool_bwd(hip.V(dybuf, 4, Cn), dx, f, True)
        rows.append(("avgpool_bwd f=%d accumulate" % f, rel(nchw(dx), x.grad + base), TOL))
    return rows


def check_layout_utils():
    rows = []
    x = R(2, 3, 6, 5, seed=151)
    y = torch.full((2, 6, 5, 4), float("nan"), device=DEV)
    hip.nchw_to_nhwc(dev(x), y)
    rows.append(("nchw_to_nhwc (pad to 4)", rel(y[..., :3], x.permute(0, 2, 3, 1)) + float(y[..., 3].abs().max()), 1e-7))
    z = R(2, 6, 5, 4, seed=152)
    o = torch.full((2, 2, 6, 5), float("nan"), device=DEV)
    hip.nhwc_to_nchw(dev(z), o)
    rows.append(("nhwc_to_nchw (first 2 ch)", rel(o, z[..., :2].permute(0, 3, 1, 2)), 1e-7))
    a, b, c = R(1000, seed=153), R(1000, seed=154), R(1000, seed=155)
    out = torch.empty(1000, device=DEV)
    hip.add(dev(a), dev(b), dev(c), None, out)
    rows.append(("add3", rel(out, a + b + c), 1e-6))
    t = torch.empty(777, device=DEV)
    hip.fill(t, 2.5)
    rows.append(("fill", float((t - 2.5).abs().max()), 1e-30))
    src = R(50, 20, seed=156)
    dst = torch.zeros(50, 32, device=DEV)
    hip.copy_slice(hip.V(dev(src), 4, 12), hip.V(dst, 8, 12))
    rows.append(("copy_slice", rel(dst[:, 8:20], src[:, 4:16]) + float(dst[:, :8].abs().max() + dst[:, 20:].abs().max()), 1e-7))
    return rows


ALL_CHECKS = [check_conv_fwd, check_conv_bwd_data, check_conv_wgrad, check_conv_dropout, check_conv_bn_epilogues,
              check_dw, check_se, check_na, check_gattn, check_ln, check_bn_tail, check_resample, check_layout_utils]


def check_conv_large():
    """Full-size feature maps: many tiles per block, several images per block range, K-split reductions."""
    rows = []
    B, H, W = 3, 176, 160
    for (name, cins, cout, k, s) in [("1x1 12->24", [12], 24, 1, 1), ("3x3 12->12", [12], 12, 3, 1),
                                      ("3x3 s2 12->24", [12], 24, 3, 2), ("3x3 cat 24+12->24", [24, 12], 24, 3, 1)]:
        cin = sum(cins)
        x = R(B, cin, H, W, seed=201).requires_grad_(True)
        w = R(cout, cin, k, k, seed=202, scale=1.0 / math.sqrt(cin * k * k)).requires_grad_(True)
        b = R(cout, seed=203).requires_grad_(True)
        y = F.conv2d(x, w, b, stride=s, padding=k // 2)
        dy = R(*y.shape, seed=204)
        y.backward(dy)
        Ho, Wo = y.shape[2:]
        xs, off = [], 0
        for c in cins:
            xs.append(nhwc(x.detach()[:, off:off + c]))
            off += c
        wd = dev(w)
        out = torch.full((B, Ho, Wo, cout), float("nan"), device=DEV)
        stats = torch.zeros(2, cout, device=DEV)
        hip.conv_fwd(xs, hip.conv_pack(wd, k, cins), out, B=B, Hin=H, Win=W, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s,
                     bias=dev(b), stats=stats, stats_mode=hip.STATS_SUM_SQ)
        rows.append(("large conv_fwd " + name, rel(nchw(out), y), TOL))
        yd = y.detach()
        rows.append(("large conv_fwd stats " + name, rel(stats, torch.stack([yd.sum((0, 2, 3)), (yd * yd).sum((0, 2, 3))])), 2e-4))
        dW, db = torch.zeros_like(wd), torch.zeros(cout, device=DEV)
        hip.conv_wgrad(xs, nhwc(dy), dW, db, B=B, Hin=H, Win=W, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s)
        rows.append(("large conv_wgrad dW " + name, rel(dW, w.grad), 2e-4))
        rows.append(("large conv_wgrad db " + name, rel(db, b.grad), 2e-4))
        if len(cins) == 1:
            dx = torch.full((B, H, W, cin), float("nan"), device=DEV)
            hip.conv_fwd([nhwc(dy)], hip.conv_pack_t(wd, k), dx, B=B, Hin=Ho, Win=Wo, Hout=H, Wout=W, Cout=cin, ksize=k,
                         stride=s, transposed=1)
            rows.append(("large conv_bwd_data " + name, rel(nchw(dx), x.grad), TOL))
    # SE_BWD epilogue across several images in one block range
    E, Cin = 24, 12
    pre, dyy = R(B, E, H, W, seed=205), R(B, Cin, H, W, seed=206)
    wpw = R(Cin, E, 1, 1, seed=207, scale=0.3)
    u_ref = F.conv_transpose2d(dyy, wpw)
    ds_ref = (u_ref * gelu(pre)).sum((2, 3))
    u = torch.full((B, H, W, E), float("nan"), device=DEV)
    ds = torch.zeros(B, E, device=DEV)
    hip.conv_fwd([nhwc(dyy)], hip.conv_pack_t(dev(wpw), 1), u, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, transposed=1,
                 epilogue=hip.EP_SE_BWD, aux=nhwc(pre), stats=ds, stats_mode=hip.STATS_EP)
    rows.append(("large conv SE_BWD u", rel(nchw(u), u_ref), TOL))
    rows.append(("large conv SE_BWD ds", rel(ds, ds_ref), 2e-4))
    rows += check_conv_bn_epilogues(B=2, H=176, W=168, tag=" (large)")
    return rows


ALL_CHECKS.append(check_conv_large)


# ------------------------------------------------------------------------------------------------ bf16 matrix-core operands
def check_conv_bf16():
    """Mixed-precision form of the dense contractions (LMN_BF16: operands rounded to bf16 when staged / packed,
    v_mfma_f32_16x16x16_bf16, fp32 accumulation and epilogues).
    (a) EXACTNESS of the wiring: with bf16-representable inputs and weights every product is exact in fp32, so the result
        must match the fp64 reference to fp32 accumulation noise (1e-5) -- a fragment-order or packing bug cannot hide
        behind a rounding tolerance;
    (b) the fp32 checks of the conv family re-run with bf16 operands at the rounding tolerance 1e-2 (two operands rounded
        to 8 significant bits: 2^-8 per product, averaged over the reduction)."""
    rows = []
    hip._MMA[0] = hip.BF16
    try:
        r16 = lambda t: t.to(torch.bfloat16).double()
        cases = [("1x1 12->24", 2, 9, 13, [12], 24, 1, 1), ("1x1 4->24 (padded rgb)", 1, 8, 8, [4], 24, 1, 1),
                 ("1x1 two-src 24+12->12", 2, 6, 7, [24, 12], 12, 1, 1), ("1x1 372->1116", 1, 5, 5, [372], 1116, 1, 1),
                 ("3x3 s1 12->12", 2, 10, 11, [12], 12, 3, 1), ("3x3 s2 12->24", 2, 12, 10, [12], 24, 3, 2),
                 ("3x3 s1 cat 24+24+24->24", 1, 8, 9, [24, 24, 24], 24, 3, 1), ("3x3 s1 100->112 (chunks)", 1, 6, 6, [100], 112, 3, 1),
                 ("3x3 s1 48->144 22x22", 2, 22, 22, [48], 144, 3, 1), ("1x1 96->288 44x44", 1, 44, 44, [96], 288, 1, 1)]
        for name, B, H, W, cins, cout, k, s in cases:
            cin = sum(cins)
            x = r16(R(B, cin, H, W, seed=1))
            w = r16(R(cout, cin, k, k, seed=2, scale=1.0 / math.sqrt(cin * k * k)))
            b = R(cout, seed=3)
            ref = F.conv2d(x, w, b, stride=s, padding=k // 2)
            Ho, Wo = ref.shape[2:]
            xs, off = [], 0
            for c in cins:
                xs.append(nhwc(x[:, off:off + c]))
                off += c
            wp = hip.conv_pack(dev(w), k, cins)
            out = torch.full((B, Ho, Wo, cout), float("nan"), device=DEV)
            hip.conv_fwd(xs, wp, out, B=B, Hin=H, Win=W, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s, bias=dev(b))
            rows.append(("bf16 exact conv_fwd " + name, rel(nchw(out), ref), 1e-5))
            # data gradient (transposed operator) of the same layer, single source only
            if len(cins) == 1:
                dy = r16(R(B, cout, Ho, Wo, seed=4))
                dref = torch.nn.grad.conv2d_input((B, cin, H, W), w, dy, stride=s, padding=k // 2)
                wpt = hip.conv_pack_t(dev(w), k)
                dx = torch.full((B, H, W, cin), float("nan"), device=DEV)
                hip.conv_fwd([nhwc(dy)], wpt, dx, B=B, Hin=Ho, Win=Wo, Hout=H, Wout=W, Cout=cin, ksize=k, stride=s,
                             transposed=1)
                rows.append(("bf16 exact conv_bwd_data " + name, rel(nchw(dx), dref), 1e-5))
        _ROUND16[0] = True       # bf16-representable operands: the weight-gradient kernels must be exact as well
        try:
            for fn in (check_conv_wgrad, check_conv_bwd_data):
                for n, e, t in fn():
                    rows.append(("bf16 exact " + n, e, t))
        finally:
            _ROUND16[0] = False
        for fn in (check_conv_fwd, check_conv_bwd_data, check_conv_bn_epilogues, check_conv_dropout, check_conv_wgrad):
            for n, e, t in fn():
                rows.append(("bf16 " + n, e, max(t, 1e-2) if t > 1e-20 else t))
    finally:
        hip._MMA[0] = hip.F32
    return rows


def check_bf16_storage_rows():
    """bf16 activation STORAGE in the kernels that are not dense contractions (the conv family has check_conv_bf16): the depthwise
    passes and the neighborhood-attention kernels templated on the storage type.  On bf16-representable inputs the bf16-storage
    launch must equal the fp32-storage launch of the same kernel up to the rounding of its outputs to bf16 (2^-8 relative per
    element; statistics, weight and bias-table gradients stay fp32: 2e-4 / float-atomic noise) -- a wrong element size in an address
    computation or a mis-paired half cannot hide behind the model-level bf16 tolerances."""
    rows = []
    bf = lambda t: t.to(torch.bfloat16)
    # ---- depthwise passes (row-planar tensors)
    for (B, H, W, E) in [(2, 20, 19, 24), (1, 33, 61, 8), (1, 9, 130, 16)]:
        x1 = R(B, E, H, W, seed=301).to(torch.bfloat16).double()
        wd = [dev(R(E, 1, 5, 5, seed=302, scale=0.2)), dev(R(E, 1, 3, 3, seed=303, scale=0.3)), dev(R(E, 1, 3, 1, seed=304, scale=0.5)),
              dev(R(E, 1, 1, 3, seed=305, scale=0.5))]
        tag = " E=%d %dx%d" % (E, H, W)
        x32 = nhwcE(x1)
        x16 = hip.rp4(bf(x32))
        st32, st16 = torch.zeros(4, 2, E, device=DEV), torch.zeros(4, 2, E, device=DEV)
        hip.dw_stats(x32, *wd, st32)
        hip.dw_stats(x16, *wd, st16)
        rows.append(("bf16 storage dw_stats" + tag, rel(st16, st32), 2e-4))
        keff, beff = dev(R(E, 25, seed=306, scale=0.2)), dev(R(E, seed=307))
        p32, p16 = torch.full((B, H, W, E), float("nan"), device=DEV), hip.rp4(torch.full((B, H, W, E), float("nan"), device=DEV, dtype=torch.bfloat16))
        g32, g16 = torch.zeros(B, E, device=DEV), torch.zeros(B, E, device=DEV)
        hip.dw_fwd(x32, hip.rp4(p32), g32, keff, beff)
        hip.dw_fwd(x16, p16, g16, keff, beff)
        rows.append(("bf16 storage dw_fwd pre" + tag, rel(p16.float(), p32), 8e-3))
        rows.append(("bf16 storage dw_fwd gsum" + tag, rel(g16, g32), 2e-4))
        # backward pair on the bf16-rounded pre (so that both launches see the same inputs)
        pre_in = p16.float()
        u = nhwcE(R(B, E, H, W, seed=308).to(torch.bfloat16).double())
        sg, dm = dev(torch.rand(B, E, dtype=torch.float64) * 0.8 + 0.1), dev(R(B, E, seed=309) * 0.01)
        d32, d16 = hip.rp4(torch.full((B, H, W, E), float("nan"), device=DEV)), hip.rp4(torch.full((B, H, W, E), float("nan"), device=DEV, dtype=torch.bfloat16))
        b32, b16 = torch.zeros(5, E, device=DEV), torch.zeros(5, E, device=DEV)
        hip.dw_bwd_stats(x32, hip.rp4(pre_in.clone()), u, sg, dm, d32, *wd, b32)
        hip.dw_bwd_stats(x16, p16, hip.rp4(bf(u)), sg, dm, d16, *wd, b16)
        rows.append(("bf16 storage dw_bwd_stats dpre" + tag, rel(d16.float(), d32), 8e-3))
        rows.append(("bf16 storage dw_bwd_stats sums" + tag, rel(b16, b32), 5e-3))       # (sums of the ROUNDED dpre on one side)
        cA, cC, cD = dev(R(4, E, seed=310)), dev(R(4, E, seed=311) * 0.1), dev(R(4, E, seed=312) * 0.01)
        dq = hip.rp4(d16.float().contiguous())          # the same (bf16-representable) dpre for both
        outs = []
        for xin, din, dt_ in ((x32, dq, torch.float32), (x16, d16, torch.bfloat16)):
            dx = hip.rp4(torch.full((B, H, W, E), float("nan"), device=DEV, dtype=dt_))
            gw = [torch.zeros_like(w) for w in wd]
            hip.dw_bwd(xin, din, dx, *wd, cA, cC, cD, *gw)
            outs.append((dx.float(), gw))
        rows.append(("bf16 storage dw_bwd dx1" + tag, rel(outs[1][0], outs[0][0]), 8e-3))
        for i, nm in enumerate(("w5", "w3", "wv", "wh")):
            rows.append(("bf16 storage dw_bwd d%s%s" % (nm, tag), rel(outs[1][1][i], outs[0][1][i]), 2e-4))
    # ---- neighborhood attention (all head dims of the model; the one-pass backward at hd <= 2, the two-pass kernels above)
    for (B, H, W, hd) in [(2, 37, 41, 1), (1, 32, 36, 2), (2, 47, 19, 2), (1, 20, 23, 4), (1, 12, 9, 8)]:
        heads, Cn = 12, 12 * hd
        qkv = dev(R(B, H, W, 3 * Cn, seed=321).to(torch.bfloat16).double())
        rpb = dev(R(heads, 5, 5, seed=322) * 0.5)
        do = dev(R(B, H, W, Cn, seed=323).to(torch.bfloat16).double())
        tag = " hd=%d %dx%d" % (hd, H, W)
        o32, o16 = torch.full((B, H, W, Cn), float("nan"), device=DEV), torch.full((B, H, W, Cn), float("nan"), device=DEV, dtype=torch.bfloat16)
        hip.na_fwd(qkv, rpb, o32, heads)
        hip.na_fwd(bf(qkv), rpb, o16, heads)
        rows.append(("bf16 storage na_fwd" + tag, rel(o16.float(), o32), 8e-3))
        g32, g16 = torch.full_like(qkv, float("nan")), torch.full((B, H, W, 3 * Cn), float("nan"), device=DEV, dtype=torch.bfloat16)
        r32, r16 = torch.zeros_like(rpb), torch.zeros_like(rpb)
        hip.na_bwd(qkv, rpb, do, g32, r32, heads)
        hip.na_bwd(bf(qkv), rpb, bf(do), g16, r16, heads)
        rows.append(("bf16 storage na_bwd dqkv" + tag, rel(g16.float(), g32), 8e-3))
        rows.append(("bf16 storage na_bwd drpb" + tag, rel(r16, r32), 2e-4))
    return rows


def check_bn_shifted_stats():
    """Batch variance when |mean| >> std (a large conv bias: |mean| / std ~ 1e3).  The conv epilogue sums about the
    BatchNorm's running mean (lmn_conv_fwd p4 / lmn_bn_finalize `about`), so E[d^2] - E[d]^2 does not cancel the squared
    mean; the un-shifted single-pass form loses the variance entirely in fp32 at this ratio."""
    rows = []
    B, H, W, cin, cout = 2, 40, 36, 12, 24
    x = R(B, cin, H, W, seed=91)
    w = R(cout, cin, 1, 1, seed=92, scale=0.05)
    b = R(cout, seed=93) * 300.0 + 500.0
    z = F.conv2d(x, w, b)
    mean_ref, var_ref = z.mean((0, 2, 3)), z.var((0, 2, 3), unbiased=False)
    about = dev(b + R(cout, seed=94) * 0.3)                      # a running mean close to (not equal to) the batch mean
    wp = hip.conv_pack(dev(w), 1, [cin])
    for tag, ab in (("about running mean", about), ("about 0 (old form)", None)):
        sums = torch.zeros(16, 2, cout, device=DEV)
        hip.conv_fwd([nhwc(x)], wp, None, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=cout, bias=dev(b), stats=sums,
                     stats_mode=hip.STATS_SUM_SQ, stats_rep=16, p=(None, None, None, None, ab))
        mean, rstd = torch.empty(cout, device=DEV), torch.empty(cout, device=DEV)
        hip.bn_finalize(sums, B * H * W, torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV), 0.0, 0.1, mean, rstd,
                        None, None, None, None, about=ab)
        var = 1.0 / (rstd.double().cpu() ** 2)
        e = float(((var - var_ref).abs() / var_ref).max())
        if ab is not None:
            rows.append(("BN variance at |mean|/std ~ 1e3, sums " + tag, e, 1e-3))
            rows.append(("BN mean at |mean|/std ~ 1e3", rel(mean, mean_ref), 1e-6))
        else:
            rows.append(("(for contrast) un-shifted variance error %.1e is > 10x worse" % e, 0.0 if e > 1e-2 else 1.0, 0.5))
    return rows


ALL_CHECKS.append(check_conv_bf16)
ALL_CHECKS.append(check_ln_linear)
ALL_CHECKS.append(check_conv_up2)
ALL_CHECKS.append(check_bn_shifted_stats)
