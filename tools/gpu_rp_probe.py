"""Row-planar vs NHWC operands of the 1x1 conv family, same values, cold operands (six rotating tensor sets):
    python tools/gpu_rp_probe.py
One line per (call kind, level): time with NHWC tensors, time with the E-wide tensors row-planar."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip

dev = torch.device("cuda", 0)
NS = 6


def timed(fn, n=24):
    for i in range(6):
        fn(i % NS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i % NS)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, H, W, Cin, E, Cout) in [(8, 352, 352, 12, 24, 12), (8, 176, 176, 24, 48, 24), (8, 88, 88, 48, 96, 48), (8, 44, 44, 96, 192, 96)]:
    mk = lambda C: [torch.randn(B, H, W, C, device=dev) for _ in range(NS)]
    x, dy = mk(Cin), mk(Cout)
    pre_n, dh_n = mk(E), mk(E)
    pre_r, dh_r = [hip.rp4(t.clone()) for t in pre_n], [hip.rp4(t.clone()) for t in dh_n]
    out_n = mk(E)
    out_r = [hip.rp4(t.clone()) for t in out_n]
    y = mk(Cout)
    we, be = torch.randn(E, Cin, device=dev) * 0.3, torch.randn(E, device=dev)
    wpw, wsc = torch.randn(Cout, E, device=dev) * 0.2, torch.randn(Cout, Cin, device=dev) * 0.3
    gate = torch.rand(B, E, device=dev)
    kw = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W)
    wpe = hip.conv_pack(we, 1, [Cin])
    n0, n1 = hip.conv_pack_size(1, Cout, [E]), hip.conv_pack_size(1, Cout, [Cin])
    wp2 = torch.empty(n0 + n1, device=dev)
    hip.conv_pack(wpw, 1, [E], out=wp2[:n0]); hip.conv_pack(wsc, 1, [Cin], out=wp2[n0:])
    wpt = hip.conv_pack_t(wpw, 1, 0, E, cred=Cout)
    st = torch.zeros(16, 2, E, device=dev)
    ds = torch.zeros(B, E, device=dev)
    gW, gb = torch.zeros(Cout, E + Cin, device=dev), torch.zeros(Cout, device=dev)
    gE = torch.zeros(E, Cin, device=dev)
    rows = [
        ("expand conv  x -> z (+sums)", lambda i: hip.conv_fwd([x[i]], wpe, out_n[i], Cout=E, bias=be, stats=st, stats_mode=hip.STATS_SUM_SQ, stats_rep=16, **kw),
                                        lambda i: hip.conv_fwd([x[i]], wpe, out_r[i], Cout=E, bias=be, stats=st, stats_mode=hip.STATS_SUM_SQ, stats_rep=16, **kw)),
        ("pointwise    gelu(pre) s, x -> y", lambda i: hip.conv_fwd([dict(view=pre_n[i], scale=gate, flags=hip.SRC_GELU), x[i]], wp2, y[i], Cout=Cout, **kw),
                                              lambda i: hip.conv_fwd([dict(view=pre_r[i], scale=gate, flags=hip.SRC_GELU), x[i]], wp2, y[i], Cout=Cout, **kw)),
        ("SE-gradient  dy -> u, aux pre", lambda i: hip.conv_fwd([dy[i]], wpt, out_n[i], Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=pre_n[i], stats=ds, stats_mode=hip.STATS_EP, **kw),
                                          lambda i: hip.conv_fwd([dy[i]], wpt, out_r[i], Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=pre_r[i], stats=ds, stats_mode=hip.STATS_EP, **kw)),
        ("wgrad        gelu(pre) s, x | dy", lambda i: hip.conv_wgrad([dict(view=pre_n[i], scale=gate, flags=hip.SRC_GELU), x[i]], dy[i], gW, gb, Cout=Cout, **kw),
                                             lambda i: hip.conv_wgrad([dict(view=pre_r[i], scale=gate, flags=hip.SRC_GELU), x[i]], dy[i], gW, gb, Cout=Cout, **kw)),
        ("wgrad        x | dh", lambda i: hip.conv_wgrad([x[i]], dh_n[i], gE, None, Cout=E, **kw), lambda i: hip.conv_wgrad([x[i]], dh_r[i], gE, None, Cout=E, **kw)),
    ]
    print("level %dx%d E=%d" % (H, W, E))
    for name, fa, fb in rows:
        ta, tb = timed(fa), timed(fb)
        ta2, tb2 = timed(fa), timed(fb)
        print("  %-36s NHWC %7.1f / %7.1f us   row-planar %7.1f / %7.1f us" % (name, ta, ta2, tb, tb2))
