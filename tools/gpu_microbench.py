"""Micro-benchmarks of the hot kernels at BASELINE shapes (B=8, 352x352): algorithmic GB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from lm_net_amd import hip  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev = "cuda"
    B = 8
    print("device:", torch.cuda.get_device_name(0))
    for (H, E) in [(352, 24), (176, 48), (88, 96), (44, 192)]:
        x1 = hip.rp4(torch.randn(B, H, H, E, device=dev))
        pre = hip.rp4(torch.empty_like(x1))
        gsum = torch.zeros(B, E, device=dev)
        keff, beff = torch.randn(E, 25, device=dev), torch.randn(E, device=dev)
        t = timeit(lambda: hip.dw_fwd(x1, pre, gsum, keff, beff))
        by = 2 * x1.numel() * 4
        print("dw_fwd      H=%3d E=%3d  %8.1f us  %7.1f GB/s alg (%.1f%% of 8 TB/s)" % (H, E, t * 1e6, by / t / 1e9, by / t / 8e12 * 100))
        w5, w3, wv, wh = (torch.randn(E, 1, a, b, device=dev) for a, b in ((5, 5), (3, 3), (3, 1), (1, 3)))
        st = torch.zeros(4, 2, E, device=dev)
        t = timeit(lambda: hip.dw_stats(x1, w5, w3, wv, wh, st))
        print("dw_stats    H=%3d E=%3d  %8.1f us  %7.1f GB/s" % (H, E, t * 1e6, x1.numel() * 4 / t / 1e9))
        u, s, dm, dpre, bst = hip.rp4(torch.randn_like(x1)), torch.rand(B, E, device=dev), torch.zeros(B, E, device=dev), hip.rp4(torch.empty_like(x1)), torch.zeros(5, E, device=dev)
        t = timeit(lambda: hip.dw_bwd_stats(x1, pre, u, s, dm, dpre, w5, w3, wv, wh, bst))
        print("dw_bwd_stat H=%3d E=%3d  %8.1f us  %7.1f GB/s" % (H, E, t * 1e6, 4 * x1.numel() * 4 / t / 1e9))
        cA = torch.rand(4, E, device=dev)
        dws = [torch.zeros_like(w) for w in (w5, w3, wv, wh)]
        dx1 = hip.rp4(torch.empty_like(x1))
        t = timeit(lambda: hip.dw_bwd(x1, dpre, dx1, w5, w3, wv, wh, cA, cA, cA, *dws))
        print("dw_bwd      H=%3d E=%3d  %8.1f us  %7.1f GB/s" % (H, E, t * 1e6, 3 * x1.numel() * 4 / t / 1e9))
    for (H, C) in [(352, 12), (176, 24), (88, 48), (44, 96)]:
        qkv = torch.randn(B, H, H, 3 * C, device=dev)
        rpb = torch.randn(12, 5, 5, device=dev)
        out = torch.empty(B, H, H, C, device=dev)
        t = timeit(lambda: hip.na_fwd(qkv, rpb, out, 12))
        by = 4 * B * H * H * C * 4
        print("na_fwd      H=%3d C=%3d  %8.1f us  %7.1f GB/s alg (%.1f%% of 8 TB/s)" % (H, C, t * 1e6, by / t / 1e9, by / t / 8e12 * 100))
        dq, drpb, do = torch.empty_like(qkv), torch.zeros_like(rpb), torch.randn_like(out)
        t = timeit(lambda: hip.na_bwd(qkv, rpb, do, dq, drpb, 12))
        print("na_bwd      H=%3d C=%3d  %8.1f us  %7.1f GB/s alg" % (H, C, t * 1e6, 7 * B * H * H * C * 4 / t / 1e9))
    convs = [("1x1 12->24 L0", 352, [12], 24, 1, 1), ("1x1 24+12->12 L0", 352, [24, 12], 12, 1, 1), ("3x3 12->12 L0", 352, [12], 12, 3, 1),
             ("3x3 s2 12->24", 352, [12], 24, 3, 2), ("3x3 36->12 L0 fuse", 352, [12, 12, 12], 12, 3, 1), ("3x3 48->24 L1", 176, [48], 24, 3, 1),
             ("1x1 96->192 L3", 44, [96], 192, 1, 1), ("3x3 192->96 L3", 44, [192], 96, 3, 1), ("3x3 372->372 gft", 22, [372], 372, 3, 1),
             ("1x1 372->1116 gft", 22, [372], 1116, 1, 1)]
    for name, H, cins, cout, k, s in convs:
        cin = sum(cins)
        xs = [torch.randn(B, H, H, c, device=dev) for c in cins]
        w = torch.randn(cout, cin, k, k, device=dev)
        wp = hip.conv_pack(w, k, cins)
        Ho = (H + 2 * (k // 2) - k) // s + 1
        out = torch.empty(B, Ho, Ho, cout, device=dev)
        t = timeit(lambda: hip.conv_fwd(xs, wp, out, B=B, Hin=H, Win=H, Hout=Ho, Wout=Ho, Cout=cout, ksize=k, stride=s))
        fl = 2.0 * B * Ho * Ho * cout * cin * k * k
        by = (sum(x.numel() for x in xs) + out.numel()) * 4
        print("conv_fwd %-20s %8.1f us  %6.2f TFLOP/s  %7.1f GB/s alg" % (name, t * 1e6, fl / t / 1e12, by / t / 1e9))
        dW, db = torch.zeros_like(w), torch.zeros(cout, device=dev)
        dy = torch.randn_like(out)
        t = timeit(lambda: hip.conv_wgrad(xs, dy, dW, db, B=B, Hin=H, Win=H, Hout=Ho, Wout=Ho, Cout=cout, ksize=k, stride=s))
        print("conv_wgrad %-18s %8.1f us  %6.2f TFLOP/s" % (name, t * 1e6, fl / t / 1e12))
        if len(cins) == 1:
            wpt = hip.conv_pack_t(w, k)
            dx = torch.empty_like(xs[0])
            t = timeit(lambda: hip.conv_fwd([dy], wpt, dx, B=B, Hin=Ho, Win=Ho, Hout=H, Wout=H, Cout=cin, ksize=k, stride=s, transposed=1))
            print("conv_bwd_data %-15s %8.1f us  %6.2f TFLOP/s" % (name, t * 1e6, fl / t / 1e12))


if __name__ == "__main__":
    main()
