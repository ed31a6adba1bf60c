"""Fixed cost of a conv launch on the small maps: 1x1 convs of the level-3 / level-4 maps with the input channels swept from 16
up (the time at K = 16 is launch + block prologue + epilogue; the slope is staging + MFMA).
    python tools/gpu_convK_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip

def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

B = 8
for H, cout in ((44, 192), (44, 96), (22, 372), (88, 96)):
    line = "H=%d Cout=%d:" % (H, cout)
    for cin in (16, 32, 64, 96, 192, 384):
        x = torch.randn(B, H, H, cin, device="cuda"); out = torch.empty(B, H, H, cout, device="cuda")
        w = torch.randn(cout, cin, 1, 1, device="cuda"); wp = hip.conv_pack(w, 1, [cin])
        t = timeit(lambda: hip.conv_fwd([x], wp, out, B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=cout, ksize=1))
        line += "  K=%d %.1f us" % (cin, t * 1e6)
    print(line)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
z = torch.zeros(64, device="cuda")
t = timeit(lambda: hip.fill(z, 0.0))
print("fill of 64 floats (launch floor): %.1f us" % (t * 1e6))
