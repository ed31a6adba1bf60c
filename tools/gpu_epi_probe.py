"""Cost of the conv epilogue variants at level 0 (B=8, 352x352, 1x1 12 -> 24)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import hip
from tools.gpu_microbench import timeit

dev = "cuda"; B, H, Cin, E = 8, 352, 12, 24
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(E, Cin, 1, 1, device=dev); bias = torch.randn(E, device=dev)
wp = hip.conv_pack(w, 1, [Cin])
out = torch.empty(B, H, H, E, device=dev); aux = torch.randn(B, H, H, E, device=dev)
v = [torch.rand(E, device=dev) + 0.5 for _ in range(5)]
st2 = torch.zeros(2, E, device=dev); stb = torch.zeros(B, E, device=dev)
kw = dict(B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=E, bias=bias)
cases = {
    "LINEAR": lambda: hip.conv_fwd([x], wp, out, **kw),
    "AFFINE_ACT hswish": lambda: hip.conv_fwd([x], wp, out, epilogue=hip.EP_AFFINE_ACT, act=hip.ACT_HSWISH, p=(v[0], v[1]), **kw),
    "stats SUM_SQ, no out": lambda: hip.conv_fwd([x], wp, None, stats=st2, stats_mode=hip.STATS_SUM_SQ, **kw),
    "BN_BWD1 (aux, stats, out)": lambda: hip.conv_fwd([x], wp, out, epilogue=hip.EP_BN_BWD1, act=hip.ACT_HSWISH, p=(v[0], v[1], v[2], v[3]), aux=aux, stats=st2, stats_mode=hip.STATS_EP, **kw),
    "BN_BWD1 (aux, stats, no out)": lambda: hip.conv_fwd([x], wp, None, epilogue=hip.EP_BN_BWD1, act=hip.ACT_HSWISH, p=(v[0], v[1], v[2], v[3]), aux=aux, stats=st2, stats_mode=hip.STATS_EP, **kw),
    "BN_BWD2 (aux, out)": lambda: hip.conv_fwd([x], wp, out, epilogue=hip.EP_BN_BWD2, p=tuple(v), aux=aux, **kw),
    "DGELU (aux, out)": lambda: hip.conv_fwd([x], wp, out, epilogue=hip.EP_DGELU, aux=aux, **kw),
}
dy = torch.randn(B, H, H, Cin, device=dev); wpw = torch.randn(Cin, E, 1, 1, device=dev); wpt = hip.conv_pack_t(wpw, 1)
cases["SE_BWD transposed 12->24 (aux, stats[B], out)"] = lambda: hip.conv_fwd([dy], wpt, out, B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=aux, stats=stb, stats_mode=hip.STATS_EP)
for k, f in cases.items():
    print("%-48s %7.1f us" % (k, timeit(f) * 1e6))
