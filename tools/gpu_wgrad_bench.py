"""1x1 weight-gradient micro-benchmark at the LM-Net layer shapes (B=8, 352x352).
    python tools/gpu_wgrad_bench.py [bf16|bf16s]      (LMN_WGRAD_DIRECT=1 selects the old direct-from-global kernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

B = 8
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
if mode != "f32":
    hip._MMA[0] = hip.BF16
dt = torch.bfloat16 if mode == "bf16s" else torch.float32
L = [("L0 expand 12->24", 352, [12], 24, False), ("L0 pw+sc 24+12->12", 352, [24, 12], 12, True), ("L0 qkv 12->36", 352, [12], 36, False),
     ("L0 fc1 12->24", 352, [12], 24, False), ("L0 fc2 24->12", 352, [24], 12, True),
     ("L1 expand 24->48", 176, [24], 48, False), ("L1 pw+sc 48+24->24", 176, [48, 24], 24, True), ("L1 qkv 24->72", 176, [24], 72, False),
     ("L2 expand 48->96", 88, [48], 96, False), ("L2 pw+sc 96+48->48", 88, [96, 48], 48, True),
     ("L3 expand 96->192", 44, [96], 192, False), ("L3 pw+sc 192+96->96", 44, [192, 96], 96, True), ("L4 372->1116", 22, [372], 1116, False)]
tot = 0.0
for name, H, cins, cout, tf in L:
    xs = [torch.randn(B, H, H, c, device="cuda").to(dt) for c in cins]
    dy = torch.randn(B, H, H, cout, device="cuda").to(dt)
    sc = torch.rand(B, cins[0], device="cuda")
    srcs = [dict(view=xs[0], scale=sc, flags=hip.SRC_GELU)] + xs[1:] if tf else xs
    dW = torch.zeros(cout, sum(cins), 1, 1, device="cuda"); db = torch.zeros(cout, device="cuda")
    t = timeit(lambda: hip.conv_wgrad(srcs, dy, dW, db, B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=cout))
    by = (sum(x.numel() for x in xs) + dy.numel()) * xs[0].element_size()
    print("%-24s %7.1f us  %6.0f GB/s  %5.1f TF" % (name, t * 1e6, by / t / 1e9, 2.0 * B * H * H * cout * sum(cins) / t / 1e12))
    tot += t
print("sum %.1f us (%s, %s)" % (tot * 1e6, mode, "direct" if os.environ.get("LMN_WGRAD_DIRECT") == "1" else "wave-staged"))
