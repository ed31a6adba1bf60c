"""What exactly is wrong in a failing run of the attention backward beside the x2 convs?  

    python tools/gpu_x2_forensics.py        # loads liblmnet_hip_x2.so

Runs the stress of tests/test_na_stress_gpu.py (fp32 storage, 176 x 176, C = 24, head_dim 2) until a repetition differs from its quiet
re-run, then explains the difference WITHOUT touching the kernel: with S = sum_n p_n k_n,
    dq = scale * sum_n p_n (dp_n - dsum) k_n    =>    a wrong dsum' shifts dq by  -scale (dsum' - dsum) S   (both channels of the head, one factor),
so the fit of the observed dq error to S says whether ONLY dsum moved and by how much; the same for dk through its nine queries.
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LMNET_HIP_LIB", os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_x2.so"))
import torch  # noqa: E402
from lm_net_amd import hip  # noqa: E402
hip.load()
print("library:", hip.LIB_PATH, flush=True)
dev = "cuda"
B, H, C, heads = 8, 176, 24, 12
hd = C // heads
dt = torch.float32
hip.set_deterministic(True)
side, main = torch.cuda.Stream(), torch.cuda.Stream()
rnd = lambda *s: torch.randn(*s, device=dev)
qkv = (rnd(B, H, H, 3 * C) * 0.5).to(dt)
rpb = rnd(heads, 5, 5) * 0.1
SC = 24
sx = rnd(8, 176, 176, SC).to(torch.bfloat16)
hip._MMA[0] = hip.BF16
scw = hip.conv_pack(rnd(SC, SC, 3, 3), 3, [SC])
scy = torch.empty(8, 176, 176, SC, device=dev, dtype=torch.bfloat16)
found = None
for r in range(40):
    do = rnd(B, H, H, C).to(dt)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
        hip._MMA[0] = hip.BF16
        for _ in range(12):
            hip.conv_fwd([sx], scw, scy, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=3)
    with torch.cuda.stream(main):
        hip._STREAM[0] = hip.C.c_void_p(main.cuda_stream)
        hip._MMA[0] = hip.F32
        dq1 = torch.empty_like(qkv)
        hip.na_bwd(qkv, rpb, do, dq1, torch.zeros_like(rpb), heads)
    hip._STREAM[0] = None
    torch.cuda.synchronize()
    dq2 = torch.empty_like(qkv)
    hip.na_bwd(qkv, rpb, do, dq2, torch.zeros_like(rpb), heads)
    torch.cuda.synchronize()
    if not torch.equal(dq1, dq2):
        found = (r, do, dq1, dq2)
        break
hip.set_deterministic(False)
if found is None:
    print("no failing repetition in 40")
    sys.exit(0)
r, do, dq1, dq2 = found
print("repetition %d differs" % r)
# ---- float64 reference pieces
f = torch.float64
q = qkv[..., :C].to(f).view(B, H, H, heads, hd)
k = qkv[..., C:2 * C].to(f).view(B, H, H, heads, hd)
v = qkv[..., 2 * C:].to(f).view(B, H, H, heads, hd)
dO = do.to(f).view(B, H, H, heads, hd)
scale = hd ** -0.5
idx = torch.arange(H, device=dev)
ws = (idx - 1).clamp(0, H - 3)
sc, dp, kn = [], [], []
for ki in range(3):
    for kj in range(3):
        ny, nx = ws + ki, ws + kj
        kk = k[:, ny][:, :, nx]
        vv = v[:, ny][:, :, nx]
        bias = rpb.to(f)[:, (ny - idx + 2)][:, :, (nx - idx + 2)]            # [heads, H, W]
        sc.append(scale * (q * kk).sum(-1) + bias.permute(1, 2, 0)[None])
        dp.append((dO * vv).sum(-1))
        kn.append(kk)
sc = torch.stack(sc, 0); dp = torch.stack(dp, 0); kn = torch.stack(kn, 0)       # [9, B, H, W, heads(, hd)]
p = torch.softmax(sc, 0)
dsum = (p * dp).sum(0)                                                             # [B, H, W, heads]
S = (p[..., None] * kn).sum(0)                                                     # [B, H, W, heads, hd]
dq_ref = scale * ((p * (dp - dsum[None]))[..., None] * kn).sum(0)
e_ok = (dq2[..., :C].to(f).view(B, H, H, heads, hd) - dq_ref).abs().max().item()
print("quiet run vs float64 reference: max |dq error| %.2e" % e_ok)
d = (dq1[..., :C].to(f) - dq2[..., :C].to(f)).view(B, H, H, heads, hd)
badph = (d.abs().amax(-1) > 0)
nb = int(badph.sum())
print("(pixel, head) pairs with a different dq: %d of %d;  dk elements different: %d, dv elements different: %d" % (
    nb, badph.numel(), int((dq1[..., C:2 * C] != dq2[..., C:2 * C]).sum()), int((dq1[..., 2 * C:] != dq2[..., 2 * C:]).sum())))
ix = badph.nonzero()
dS = d[badph]; SS = S[badph]
delta = -(dS * SS).sum(-1) / (SS * SS).sum(-1) / scale          # dsum' - dsum if only dsum moved
resid = (dS + scale * delta[:, None] * SS).abs().amax(-1) / dS.abs().amax(-1)
print("fit of the dq error to -scale * delta * S:  median relative residual %.2e, 90th percentile %.2e  (small = ONLY dsum moved)" % (
    resid.median().item(), resid.quantile(0.9).item()))
ds_true = dsum[badph]
ratio = delta / ds_true
print("delta / dsum: median %.4f  10%% %.4f  90%% %.4f   (-1 = dsum' is 0)" % (ratio.median().item(), ratio.quantile(0.1).item(), ratio.quantile(0.9).item()))
# is dsum' the dsum of ANOTHER (pixel, head)?  compare with neighbours in the thread layout: same pixel other heads, next pixels same head
dsp = ds_true + delta
cands = {}
for name, sh in (("same pixel, head+1", (0, 0, 1)), ("same pixel, head-1", (0, 0, -1)), ("pixel x+1", (0, 1, 0)), ("pixel x-1", (0, -1, 0)), ("pixel y+1", (1, 0, 0)), ("pixel y-1", (-1, 0, 0))):
    rolled = torch.roll(dsum, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))
    cands[name] = ((rolled[badph] - dsp).abs() < 1e-4 * (1 + dsp.abs())).float().mean().item()
print("dsum' equals the true dsum of:", {k_: round(v_, 3) for k_, v_ in cands.items()})
# one missing / doubled term?
terms = (p * dp)[:, badph]                                        # [9, nb]
for nm, sgn in (("one term missing", -1.0), ("one term doubled", 1.0)):
    hit = ((terms * sgn - delta[None]).abs() < 1e-4 * (1 + delta.abs()[None])).any(0).float().mean().item()
    print("%s: %.3f of the cases" % (nm, hit))
# p from a wrong normaliser?  dsum' = dsum * c
print("positions (b, y, x, head) of the first 24 and their tile-local coordinates (tile 15 x 17):")
for t in range(min(24, ix.shape[0])):
    b_, y_, x_, h_ = ix[t].tolist()
    print("   b %d y %3d x %3d head %2d | y%%15 %2d x%%17 %2d | dsum % .5f delta % .5f ratio % .4f resid %.1e" % (
        b_, y_, x_, h_, y_ % 15, x_ % 17, ds_true[t].item(), delta[t].item(), ratio[t].item(), resid[t].item()))
# histogram over tile-local coordinates and heads
ty = (ix[:, 1] % 15); tx = (ix[:, 2] % 17)
print("by head:", torch.bincount(ix[:, 3], minlength=heads).tolist())
print("by y%15:", torch.bincount(ty, minlength=15).tolist())
print("by x%17:", torch.bincount(tx, minlength=17).tolist())
print("by image:", torch.bincount(ix[:, 0], minlength=B).tolist())
