"""Cost of the conv epilogue variants at level 0 (B=8, 352x352, 1x1 12 -> 24), on COLD operands: every call works on the
next of NSET tensor sets (together larger than the 256 MB memory-side cache), as the layers of a training step do.
    python tools/gpu_epi_probe.py [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import hip
from tools.gpu_microbench import timeit

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
dt = torch.float32
if mode == "bf16":
    hip._MMA[0] = hip.BF16; dt = torch.bfloat16
dev = "cuda"; B, H, Cin, E = 8, int(os.environ.get("EPI_H", "352")), int(os.environ.get("EPI_CIN", "12")), int(os.environ.get("EPI_E", "24"))   # EPI_H=176 EPI_CIN=24 EPI_E=48: level 1
NSET = 6
xs = [torch.randn(B, H, H, Cin, device=dev).to(dt) for _ in range(NSET)]
outs = [torch.empty(B, H, H, E, device=dev, dtype=dt) for _ in range(NSET)]
auxs = [torch.randn(B, H, H, E, device=dev).to(dt) for _ in range(NSET)]
dys = [torch.randn(B, H, H, Cin, device=dev).to(dt) for _ in range(NSET)]
w = torch.randn(E, Cin, 1, 1, device=dev); bias = torch.randn(E, device=dev)
wp = hip.conv_pack(w, 1, [Cin])
v = [torch.rand(E, device=dev) + 0.5 for _ in range(7)]
REP = int(os.environ.get("STATS_REP", "16"))
st2 = torch.zeros(REP, 2, E, device=dev); stb = torch.zeros(B, E, device=dev)
kw = dict(B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=E, bias=bias)
wpw = torch.randn(Cin, E, 1, 1, device=dev); wpt = hip.conv_pack_t(wpw, 1)
ctr = [0]
def rot():
    ctr[0] = (ctr[0] + 1) % NSET
    i = ctr[0]
    return xs[i], outs[i], auxs[i], dys[i]
def case(fn):
    def run():
        x, out, aux, dy = rot()
        fn(x, out, aux, dy)
    return run
cases = {
    "LINEAR": case(lambda x, out, aux, dy: hip.conv_fwd([x], wp, out, **kw)),
    "AFFINE_ACT hswish": case(lambda x, out, aux, dy: hip.conv_fwd([x], wp, out, epilogue=hip.EP_AFFINE_ACT, act=hip.ACT_HSWISH, p=(v[0], v[1]), **kw)),
    "stats SUM_SQ, no out": case(lambda x, out, aux, dy: hip.conv_fwd([x], wp, None, stats=st2, stats_mode=hip.STATS_SUM_SQ, stats_rep=REP, **kw)),
    "BN_BWD1 (aux, stats, no out)": case(lambda x, out, aux, dy: hip.conv_fwd([x], wp, None, epilogue=hip.EP_BN_BWD1, act=hip.ACT_HSWISH, p=(v[0], v[1], v[2], v[3]), aux=aux, stats=st2, stats_mode=hip.STATS_EP, stats_rep=REP, **kw)),
    "BN_BWD2 (aux, out)": case(lambda x, out, aux, dy: hip.conv_fwd([x], wp, out, epilogue=hip.EP_BN_BWD2, act=hip.ACT_HSWISH, p=tuple(v), aux=aux, **kw)),
    "DGELU (aux, out)": case(lambda x, out, aux, dy: hip.conv_fwd([x], wp, out, epilogue=hip.EP_DGELU, aux=aux, **kw)),
    "SE_BWD transposed 12->24 (aux, stats[B], out)": case(lambda x, out, aux, dy: hip.conv_fwd([dy], wpt, out, B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=E, transposed=1, epilogue=hip.EP_SE_BWD, aux=aux, stats=stb, stats_mode=hip.STATS_EP)),
}
esz = 2 if mode == "bf16" else 4
npx = B * H * H
byts = {"LINEAR": Cin + E, "AFFINE_ACT hswish": Cin + E, "stats SUM_SQ, no out": Cin, "BN_BWD1 (aux, stats, no out)": Cin + E,
        "BN_BWD2 (aux, out)": Cin + 2 * E, "DGELU (aux, out)": Cin + 2 * E, "SE_BWD transposed 12->24 (aux, stats[B], out)": Cin + 2 * E}
for k, f in cases.items():
    t = timeit(f)
    print("%-48s %7.1f us  %6.0f GB/s" % (k, t * 1e6, byts[k] * npx * esz / t / 1e9))
