"""The four 1x1 conv forms of a ReparamConv block at levels 0 / 1 (batch 8): LDS-DMA streaming kernel vs LDS-tiled kernel, rotating
tensor sets (cold operands).   python tools/gpu_dma1_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip
from tools.gpu_microbench import timeit
B = 8
for (H, Cin, E) in ((352, 12, 24), (176, 24, 48)):
    nset = 4
    mk = lambda c: [torch.randn(B, H, H, c, device="cuda") for _ in range(nset)]
    xs, pres, dys, dhs = mk(Cin), [hip.rp4(t) for t in mk(E)], mk(Cin), [hip.rp4(t) for t in mk(E)]
    zs, us, ys = [hip.rp4(t) for t in mk(E)], [hip.rp4(t) for t in mk(E)], mk(Cin)
    gate = torch.rand(B, E, device="cuda")
    we, wpw, wsc = torch.randn(E, Cin, device="cuda"), torch.randn(Cin, E, device="cuda"), torch.randn(Cin, Cin, device="cuda")
    wpe = hip.conv_pack(we, 1, [Cin])
    n0, n1 = hip.conv_pack_size(1, Cin, [E]), hip.conv_pack_size(1, Cin, [Cin])
    wp2 = torch.empty(n0 + n1, device="cuda"); hip.conv_pack(wpw, 1, [E], out=wp2[:n0]); hip.conv_pack(wsc, 1, [Cin], out=wp2[n0:])
    wpt = hip.conv_pack_t(wpw, 1, 0, E, cred=Cin)
    wp3 = torch.empty(n0 + 2 * n1, device="cuda"); hip.conv_pack(wpw, 1, [E], out=wp3[:n0]); hip.conv_pack(wsc, 1, [Cin], out=wp3[n0:n0 + n1]); hip.conv_pack(wsc, 1, [Cin], out=wp3[n0 + n1:])
    st, ds, bias = torch.zeros(2, E, device="cuda"), torch.zeros(B, E, device="cuda"), torch.randn(E, device="cuda")
    kw = dict(B=B, Hin=H, Win=H, Hout=H, Wout=H)
    c = [0]
    def nxt():
        c[0] = (c[0] + 1) % nset
        return c[0]
    forms = {
        "F1 expand (stats, rp out)": lambda i: hip.conv_fwd([xs[i]], wpe, zs[i], Cout=E, bias=bias, stats=st, stats_mode=hip.STATS_SUM_SQ, **kw),
        "F2 pointwise+shortcut": lambda i: hip.conv_fwd([dict(view=pres[i], scale=gate, flags=hip.SRC_GELU), xs[i]], wp2, ys[i], Cout=Cin, **kw),
        "B1 SE-gradient conv": lambda i: hip.conv_fwd([dys[i]], wpt, us[i], Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=pres[i], stats=ds, stats_mode=hip.STATS_EP, **kw),
        "B2 three-source dx": lambda i: hip.conv_fwd([dhs[i], xs[i], dys[i]], wp3, ys[i], Cout=Cin, **kw),
    }
    for name, fn in forms.items():
        r = []
        for mode in (0, 3):
            hip.conv_dma_config(mode, 1)
            r.append(timeit(lambda: fn(nxt())) * 1e6)
        print("H=%3d %-28s LDS-tiled %7.1f us   LDS-DMA %7.1f us" % (H, name, r[0], r[1]), flush=True)
hip.conv_dma_config(3, 512)
