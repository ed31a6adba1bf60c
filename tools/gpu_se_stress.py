"""Stress of the in-kernel squeeze-excite hand-off of lmn_dw_fwd (arrival counter per image): the fused gate must equal
lmn_se_fwd on the finished sums, whichever block arrives last, under load (a second stream keeps the GPU busy)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip
hip.load()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
bad = tot = 0
side = torch.cuda.Stream()
junk = torch.randn(64 << 20, device=dev)
for (B, H, W, E) in [(8, 352, 352, 24), (8, 176, 176, 48), (8, 88, 88, 96), (8, 44, 44, 192), (2, 64, 96, 24), (3, 40, 130, 8)]:
    R = max(E // 4, 1)
    x1 = hip.rp4(torch.randn(B, H, W, E, device=dev))
    keff, beff = torch.randn(E, 25, device=dev) * 0.2, torch.randn(E, device=dev) * 0.1
    w1, b1 = torch.randn(R, E, device=dev) * 0.4, torch.randn(R, device=dev)
    w2, b2 = torch.randn(E, R, device=dev) * 0.8, torch.randn(E, device=dev)
    for it in range(60):
        pre = hip.rp4(torch.empty(B, H, W, E, device=dev))
        gs, gs2 = torch.zeros(B, E, device=dev), torch.zeros(B, E, device=dev)
        s, h = torch.full((B, E), float("nan"), device=dev), torch.full((B, R), float("nan"), device=dev)
        with torch.cuda.stream(side):
            junk.mul_(1.0001)
        hip.dw_fwd(x1, pre, gs, keff, beff, se=dict(ticket=torch.zeros(B, device=dev), fc1w=w1, fc1b=b1, fc2w=w2, fc2b=b2, s=s,
                                                    hidden=h, inv_hw=1.0 / (H * W)))
        s2, h2 = torch.empty(B, E, device=dev), torch.empty(B, R, device=dev)
        hip.se_fwd(gs, 1.0 / (H * W), w1, b1, w2, b2, s2, h2)
        torch.cuda.synchronize()
        tot += 1
        if not (torch.equal(s, s2) and torch.equal(h, h2)):
            bad += 1
            print("MISMATCH", (B, H, W, E), it, float((s - s2).abs().max()), float((h - h2).abs().max()))
print("runs %d, mismatches %d" % (tot, bad))
