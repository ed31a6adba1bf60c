"""The GEMM-shaped 1x1 convs of the bottleneck (GFT linears, 22x22x8 = 3872 tokens) and of levels 3 / 4, forward and data gradient:
   python tools/gpu_gemm_probe.py      (LMN_CONVM_TP=<pixels> forces the tile of the M-split kernel, LMN_MSPLIT_MIN its threshold)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip
from tools.gpu_microbench import timeit
B = 8
tot = 0.0
for name, H, cin, cout in [("GFT qkv 372->1116", 22, 372, 1116), ("GFT proj 372->372", 22, 372, 372), ("GFT fc1 372->744", 22, 372, 744),
                           ("GFT fc2 744->372", 22, 744, 372), ("L3 expand 96->192", 44, 96, 192), ("L3 point 192->96", 44, 192, 96),
                           ("L4 qkv^T 1116->372", 22, 1116, 372)]:
    x = torch.randn(B, H, H, cin, device="cuda"); out = torch.empty(B, H, H, cout, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda"); wp = hip.conv_pack(w, 1, [cin])
    t = timeit(lambda: hip.conv_fwd([x], wp, out, B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=cout, ksize=1)) * 1e6
    fl = 2.0 * B * H * H * cin * cout
    print("%-22s %7.1f us  %6.1f TF/s" % (name, t, fl / t / 1e6), flush=True)
    tot += t
print("sum %.1f us" % tot)
