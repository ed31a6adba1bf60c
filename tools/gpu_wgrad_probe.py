import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import hip
dev = "cuda"; B = 8
for name, H, cins, cout, k, s in [("1x1 12->24 L0", 352, [12], 24, 1, 1), ("3x3 12->12 L0", 352, [12], 12, 3, 1), ("3x3 48->24 L1", 176, [48], 24, 3, 1)]:
    cin = sum(cins)
    xs = [torch.randn(B, H, H, c, device=dev) for c in cins]
    w = torch.randn(cout, cin, k, k, device=dev)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    dy = torch.randn(B, Ho, Ho, cout, device=dev)
    dW, db = torch.zeros_like(w), torch.zeros(cout, device=dev)
    for _ in range(3):
        hip.conv_wgrad(xs, dy, dW, db, B=B, Hin=H, Win=H, Hout=Ho, Wout=Ho, Cout=cout, ksize=k, stride=s)
torch.cuda.synchronize()
