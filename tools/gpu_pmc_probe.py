"""One launch set of the kernels under study (for rocprofv3 --pmc runs).  argv[1]: conv | dw | wgrad"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import hip
dev = "cuda"; B = 8
what = sys.argv[1] if len(sys.argv) > 1 else "conv"


def conv_case(H, cins, cout, k, s, fwd=True, wgrad=False):
    cin = sum(cins)
    xs = [torch.randn(B, H, H, c, device=dev) for c in cins]
    w = torch.randn(cout, cin, k, k, device=dev)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    dy = torch.randn(B, Ho, Ho, cout, device=dev)
    out = torch.empty(B, Ho, Ho, cout, device=dev)
    dW, db = torch.zeros_like(w), torch.zeros(cout, device=dev)
    wp = hip.conv_pack(w, k, cins)
    for _ in range(2):
        if fwd:
            hip.conv_fwd(xs, wp, out, B=B, Hin=H, Win=H, Hout=Ho, Wout=Ho, Cout=cout, ksize=k, stride=s)
        if wgrad:
            hip.conv_wgrad(xs, dy, dW, db, B=B, Hin=H, Win=H, Hout=Ho, Wout=Ho, Cout=cout, ksize=k, stride=s)


if what in ("conv", "wgrad"):
    f, g = what == "conv", what == "wgrad"
    conv_case(352, [12], 24, 1, 1, f, g)
    conv_case(352, [12], 12, 3, 1, f, g)
    conv_case(352, [12, 12, 12], 12, 3, 1, f, g)
    conv_case(176, [48], 24, 3, 1, f, g)
    conv_case(44, [192], 96, 3, 1, f, g)
    conv_case(22, [372], 372, 3, 1, f, g)
if what == "conv1":     # the HBM-side 1x1 layers of level 0 / 1
    conv_case(352, [12], 24, 1, 1, True, False)
    conv_case(352, [24, 12], 12, 1, 1, True, False)
    conv_case(176, [24], 48, 1, 1, True, False)
if what == "wg3":       # 3x3 weight gradients: one-tile and 2x2-tile blocks
    conv_case(352, [12], 12, 3, 1, False, True)
    conv_case(176, [24], 24, 3, 1, False, True)
    conv_case(176, [48], 24, 3, 1, False, True)
    conv_case(88, [96], 48, 3, 1, False, True)
if what == "conv72":
    conv_case(176, [24], 72, 3, 1, True, False)
    conv_case(176, [24, 24, 24], 24, 3, 1, True, False)
    conv_case(88, [24], 80, 1, 1, True, False)
if what == "dw":
    H, E = 352, 24
    x1 = hip.rp4(torch.randn(B, H, H, E, device=dev)); pre = hip.rp4(torch.empty_like(x1)); gsum = torch.zeros(B, E, device=dev)
    keff, beff = torch.randn(E, 25, device=dev), torch.randn(E, device=dev)
    w5, w3, wv, wh = (torch.randn(E, 1, a, b, device=dev) for a, b in ((5, 5), (3, 3), (3, 1), (1, 3)))
    cA = torch.rand(4, E, device=dev); dws = [torch.zeros_like(w) for w in (w5, w3, wv, wh)]; dx1 = hip.rp4(torch.empty_like(x1))
    for _ in range(2):
        hip.dw_fwd(x1, pre, gsum, keff, beff)
        hip.dw_bwd(x1, pre, dx1, w5, w3, wv, wh, cA, cA, cA, *dws)
torch.cuda.synchronize()
if what == "na":
    for (H, C) in [(352, 12), (176, 24)]:
        qkv = torch.randn(B, H, H, 3 * C, device=dev); rpb = torch.randn(12, 5, 5, device=dev)
        out = torch.empty(B, H, H, C, device=dev); dq, drpb, do = torch.empty_like(qkv), torch.zeros_like(rpb), torch.randn_like(out)
        for _ in range(2):
            hip.na_fwd(qkv, rpb, out, 12)
            hip.na_bwd(qkv, rpb, do, dq, drpb, 12)
    torch.cuda.synchronize()
