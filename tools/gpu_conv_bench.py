"""Conv family micro-benchmark at the LM-Net layer shapes (B=8, 352x352): forward and data-gradient launches, on COLD
operands (every call works on the next of NSET tensor sets, together larger than the 256 MB memory-side cache -- as the
layers of a training step do; NSET=1 python tools/gpu_conv_bench.py gives the hot-cache numbers).
    python tools/gpu_conv_bench.py [bf16]"""
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
NSET = int(os.environ.get("NSET", "6"))
if len(sys.argv) > 1 and sys.argv[1] == "bf16":
    hip._MMA[0] = hip.BF16
L = [("L0 1x1 4->24", 352, [4], 24, 1, 1), ("L0 1x1 12->24", 352, [12], 24, 1, 1), ("L0 1x1 24+12->12", 352, [24, 12], 12, 1, 1),
     ("L0 1x1 12->36 qkv", 352, [12], 36, 1, 1), ("L0 3x3 12->12", 352, [12], 12, 3, 1), ("L0 3x3 24->12 cat2", 352, [12, 12], 12, 3, 1),
     ("L0 3x3 s2 12->24", 352, [12], 24, 3, 2), ("L0 3x3 24->12 up", 352, [24], 12, 3, 1),
     ("L1 1x1 24->48", 176, [24], 48, 1, 1), ("L1 1x1 48+24->24", 176, [48, 24], 24, 1, 1), ("L1 3x3 24->24", 176, [24], 24, 3, 1),
     ("L1 3x3 72->24 cat3", 176, [24, 24, 24], 24, 3, 1), ("L1 3x3 48->24 up", 176, [48], 24, 3, 1),
     ("L2 1x1 48->96", 88, [48], 96, 1, 1), ("L2 1x1 96+48->48", 88, [96, 48], 48, 1, 1), ("L2 3x3 48->48", 88, [48], 48, 3, 1),
     ("L2 3x3 144->48 cat3", 88, [48, 48, 48], 48, 3, 1),
     ("L3 1x1 96->192", 44, [96], 192, 1, 1), ("L3 1x1 192+96->96", 44, [192, 96], 96, 1, 1), ("L3 3x3 96->96", 44, [96], 96, 3, 1),
     ("L3 3x3 192->96 cat2", 44, [96, 96], 96, 3, 1), ("L4 3x3 372->372", 22, [372], 372, 3, 1), ("L4 1x1 372->1116", 22, [372], 1116, 1, 1)]
tot_f = tot_t = 0.0
for name, H, cins, cout, k, s in L:
    cin = sum(cins)
    Ho = (H + 2 * (k // 2) - k) // s + 1
    one = (B * H * H * cin + B * Ho * Ho * cout) * 4
    nset = max(1, min(NSET, int(1.2e9 // one)))
    xss = [[torch.randn(B, H, H, c, device="cuda") for c in cins] for _ in range(nset)]
    w = torch.randn(cout, cin, k, k, device="cuda")
    wp = hip.conv_pack(w, k, cins)
    outs = [torch.empty(B, Ho, Ho, cout, device="cuda") for _ in range(nset)]
    ctr = [0]
    def fwd():
        ctr[0] = (ctr[0] + 1) % nset
        hip.conv_fwd(xss[ctr[0]], wp, outs[ctr[0]], B=B, Hin=H, Win=H, Hout=Ho, Wout=Ho, Cout=cout, ksize=k, stride=s)
    xs, out = xss[0], outs[0]
    t = timeit(fwd)
    fl = 2.0 * B * Ho * Ho * cout * cin * k * k
    by = (sum(x.numel() for x in xs) + out.numel()) * 4
    line = "%-22s fwd %7.1f us %6.1f TF %6.0f GB/s" % (name, t * 1e6, fl / t / 1e12, by / t / 1e9)
    tot_f += t
    if len(cins) == 1 and s == 1:
        wpt = hip.conv_pack_t(w, k)
        def bwd():  # dy = an output buffer, dx = an input buffer of the next set
            ctr[0] = (ctr[0] + 1) % nset
            hip.conv_fwd([outs[ctr[0]]], wpt, xss[ctr[0]][0], B=B, Hin=Ho, Win=Ho, Hout=H, Wout=H, Cout=cin, ksize=k, stride=s, transposed=1)
        t2 = timeit(bwd)
        line += "   dgrad %7.1f us %6.1f TF" % (t2 * 1e6, fl / t2 / 1e12)
        tot_t += t2
    print(line)
print("sum fwd %.1f us, sum dgrad %.1f us  (pipe=%s, mma=%s)" % (tot_f * 1e6, tot_t * 1e6, os.environ.get("LMN_CONV_PIPE", "0"), "bf16" if hip._MMA[0] else "f32"))
