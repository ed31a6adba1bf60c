"""Neighborhood-attention kernels alone at the four level shapes of the 352x352 / batch 8 step (forward, backward pair).
   python tools/gpu_na_probe.py       (LMN_NA_QGRID=<blocks>: cap of the query pass grid, A/B)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip
from tools.gpu_microbench import timeit
dev = "cuda"; B = 8; heads = 12
tot = 0.0
for (H, C) in [(352, 12), (176, 24), (88, 48), (44, 96)]:
    qkv = torch.randn(B, H, H, 3 * C, device=dev) * 0.5
    rpb = torch.randn(heads, 5, 5, device=dev) * 0.1
    out = torch.empty(B, H, H, C, device=dev); do = torch.randn_like(out)
    dqkv = torch.empty_like(qkv); drpb = torch.zeros_like(rpb)
    stat = torch.empty(B * H * H * 2 * heads, device=dev)
    tf = timeit(lambda: hip.na_fwd(qkv, rpb, out, heads)) * 1e6
    tb = timeit(lambda: hip.na_bwd(qkv, rpb, do, dqkv, drpb, heads, stat=stat)) * 1e6
    by = 4.0 * B * H * H * C
    print("H=%3d C=%2d  fwd %7.1f us (%.2f TB/s)   bwd pair %7.1f us (%.2f TB/s)" % (H, C, tf, 4 * by / tf / 1e6, tb, 7 * by / tb / 1e6), flush=True)
    tot += tf + tb
print("sum %.1f us" % tot)
