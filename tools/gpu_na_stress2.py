"""Producer -> neighborhood-attention backward on one stream, beside a busy second stream: `do` is written by a 1x1 conv right before
lmn_na_bwd reads it (as in engine.nat_bwd); the result is compared with the same call repeated after a device synchronisation.
    python tools/gpu_na_stress2.py [reps] [bf16|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import hip
hip.load()
hip.set_deterministic(True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
if dt == torch.bfloat16:
    hip._MMA[0] = hip.BF16
dev = "cuda"
side = torch.cuda.Stream()
main = torch.cuda.Stream()
B, H, C, heads = 8, 176, 24, 12
qkv = (torch.randn(B, H, H, 3 * C, device=dev) * 0.5).to(dt)
rpb = torch.randn(heads, 5, 5, device=dev) * 0.1
w = torch.randn(C, C, 1, 1, device=dev) * 0.3
wp = hip.conv_pack(w, 1, [C])
# side work: 3x3 weight gradients (LDS- and MFMA-heavy), as beside the backward chain
SIDE = os.environ.get("SIDE", "wgrad")         # wgrad | wgrad_f32 | conv | mul | none
sdt = torch.float32 if SIDE == "wgrad_f32" else dt
if os.environ.get("SIDE_DT") == "bf16":
    sdt = torch.bfloat16
elif os.environ.get("SIDE_DT") == "f32":
    sdt = torch.float32
SMMA = hip.BF16 if sdt == torch.bfloat16 else hip.F32
SK = int(os.environ.get("SIDE_K", "3"))
SC = int(os.environ.get("SIDE_C", "24"))
sx = torch.randn(8, 176, 176, SC, device=dev).to(sdt)
sdy = torch.randn(8, 176, 176, SC, device=dev).to(sdt)
hip._MMA[0] = SMMA
scw = hip.conv_pack(torch.randn(SC, SC, SK, SK, device=dev), SK, [SC])
hip._MMA[0] = hip.BF16 if dt == torch.bfloat16 else hip.F32
scy = torch.empty(8, 176, 176, SC, device=dev, dtype=sdt)
sdW, sdb = torch.zeros(24, 24, 3, 3, device=dev), torch.zeros(24, device=dev)
bad = 0
keepalive = []
qkv0 = qkv.clone()
canary = [torch.zeros(1 << 20, device=dev) for _ in range(8)]      # 4 MB canaries allocated around the working tensors
for r in range(reps):
    da = torch.randn(B, H, H, C, device=dev).to(dt)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
        for _ in range(6):
            if SIDE in ("wgrad", "wgrad_f32"):
                mm = hip._MMA[0]
                if SIDE == "wgrad_f32":
                    hip._MMA[0] = hip.F32
                hip.conv_wgrad([sx], sdy, sdW, sdb, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=24, ksize=3)
                hip._MMA[0] = mm
            elif SIDE == "conv":
                mm = hip._MMA[0]
                hip._MMA[0] = SMMA
                hip.conv_fwd([sx], scw, scy, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=SK)
                hip._MMA[0] = mm
            elif SIDE == "mul":
                sx.mul_(1.0001)
    with torch.cuda.stream(main):
        hip._STREAM[0] = hip.C.c_void_p(main.cuda_stream)
        do = torch.empty(B, H, H, C, device=dev, dtype=dt)
        if os.environ.get("FRESH") == "1":
            keepalive.append(do)        # never free: every repetition gets an address that was never read before
        mode = os.environ.get("PRODUCER", "conv")
        if mode == "conv":
            hip.conv_fwd([da.view(1, 1, -1, C)], wp, do.view(1, 1, -1, C), B=1, Hin=1, Win=B * H * H, Hout=1, Wout=B * H * H, Cout=C, ksize=1)
        elif mode == "copy":
            do.copy_(da)
        elif mode == "mul":
            torch.mul(da, 0.5, out=do)
        if os.environ.get("SYNC") == "1":
            torch.cuda.current_stream().synchronize()
        dq1 = torch.empty_like(qkv)
        hip.na_bwd(qkv, rpb, do, dq1, torch.zeros_like(rpb), heads)
    hip._STREAM[0] = None
    torch.cuda.synchronize()
    dq2 = torch.empty_like(qkv)
    hip.na_bwd(qkv, rpb, do, dq2, torch.zeros_like(rpb), heads)
    torch.cuda.synchronize()
    if not torch.equal(qkv, qkv0):
        d = (qkv.float() - qkv0.float()).abs()
        idx = torch.nonzero(d > 0)
        print("   rep %d: INPUT qkv CHANGED in %d elements (parts %s)" % (r, idx.shape[0], torch.bincount(idx[:, 3] // C, minlength=3).tolist()), flush=True)
        qkv.copy_(qkv0)
    for ci, cn in enumerate(canary):
        if float(cn.abs().max()) != 0.0:
            nz = torch.nonzero(cn).flatten()
            print("   rep %d: canary %d has %d nonzero floats, first at %d" % (r, ci, nz.numel(), int(nz[0])), flush=True)
            cn.zero_()
    if not torch.equal(dq1, dq2):
        bad += 1
        if os.environ.get("TRUTH") == "1":
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
            from oracle import natten_ref
            hd = C // heads
            qf = qkv.float().clone().requires_grad_(True)
            q_, k_, v_ = qf.reshape(B, H, H, 3, heads, hd).permute(3, 0, 4, 1, 2, 5).unbind(0)
            attn = torch.softmax(natten_ref.na2d_qkrpb(q_ * hd ** -0.5, k_, rpb, 3), -1)
            o_ = natten_ref.na2d_av(attn, v_, 3).permute(0, 2, 3, 1, 4).reshape(B, H, H, C)
            o_.backward(do.float())
            m = (dq1 != dq2)
            e1 = (dq1.float() - qf.grad)[m].abs().max(); e2 = (dq2.float() - qf.grad)[m].abs().max()
            print("      at the differing elements: |concurrent - reference| max %.3e, |quiet - reference| max %.3e" % (float(e1), float(e2)), flush=True)
        d = (dq1.float() - dq2.float()).abs()
        idx = torch.nonzero(d > 0)
        print("   rep %d: %d elements differ (dq/dk/dv %s), max %.3e" % (r, idx.shape[0], torch.bincount(idx[:, 3] // C, minlength=3).tolist(), float(d.max())), flush=True)
print("%s: %d of %d producer->consumer runs differ from the quiet re-run" % (str(dt).split(".")[1], bad, reps))
hip.set_deterministic(False)
