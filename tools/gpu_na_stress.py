"""Repeats lmn_na_bwd on fixed inputs (optionally beside a second stream that keeps the GPU busy) and counts runs whose dqkv / drpb
differ bitwise from the first (deterministic mode: no float atomics).   python tools/gpu_na_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import hip
hip.load()
hip.set_deterministic(True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda"
side = torch.cuda.Stream()
junk = torch.randn(32 << 20, device=dev)
for dt in (torch.float32, torch.bfloat16):
    for (B, H, C) in [(8, 352, 12), (8, 176, 24), (2, 64, 24), (3, 47, 12)]:
        heads = 12
        qkv = (torch.randn(B, H, H, 3 * C, device=dev) * 0.5).to(dt)
        rpb = torch.randn(heads, 5, 5, device=dev) * 0.1
        do = torch.randn(B, H, H, C, device=dev).to(dt)
        ref = None
        bad = 0
        # something that leaves different LDS / register contents behind between the runs: a 3x3 conv on random data
        cx = torch.randn(2, 64, 64, 48, device=dev)
        cw = hip.conv_pack(torch.randn(48, 48, 3, 3, device=dev), 3, [48])
        cy = torch.empty(2, 64, 64, 48, device=dev)
        for r in range(reps):
            if r % 3 == 1:
                cx.normal_()
                hip.conv_fwd([cx], cw, cy, B=2, Hin=64, Win=64, Hout=64, Wout=64, Cout=48, ksize=3)
            if r % 3 == 2:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        hip.conv_fwd([cx], cw, cy, B=2, Hin=64, Win=64, Hout=64, Wout=64, Cout=48, ksize=3)
            dqkv = torch.full_like(qkv, float("nan"))
            drpb = torch.zeros_like(rpb)
            if r % 2:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        junk.mul_(1.0001)
            hip.na_bwd(qkv, rpb, do, dqkv, drpb, heads)
            torch.cuda.synchronize()
            cur = (dqkv.clone(), drpb.clone())
            if ref is None:
                ref = cur
            elif not (torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])):
                bad += 1
                if bad <= 3:
                    d = (ref[0].float() - cur[0].float()).abs()
                    idx = torch.nonzero(d.reshape(B, H, H, 3, C).amax(dim=(3, 4)) > 0)
                    print("   run %d differs: %d pixels, e.g. %s, max %.3e; drpb equal %s" % (r, idx.shape[0], idx[:4].tolist(), float(d.max()), torch.equal(ref[1], cur[1])))
        print("%s B=%d %dx%d C=%d: %d of %d runs differ from the first%s" % (str(dt).split(".")[1], B, H, H, C, bad, reps - 1, "  (nan in output!)" if torch.isnan(ref[0].float()).any() else ""), flush=True)
hip.set_deterministic(False)
