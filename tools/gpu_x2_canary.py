"""Decisive experiment for the bf16 "cross-kernel corruption" of round 4.

    python tools/gpu_x2_canary.py x2      # library built with -DLMN_MFMA_X2 (make -C lm_net_amd/csrc x2): v_mfma_f32_16x16x32_bf16 restored
    python tools/gpu_x2_canary.py product # the product library (two 16x16x16 MFMAs): control

Part A: the attention backward (current kernels: one-pass at head_dim 2, two-pass at head_dim 4) beside bf16 3x3 convs / weight
        gradients of the loaded library, bit-compared with a quiet re-run (as tests/test_na_stress_gpu.py) -- does the round-4
        symptom still reproduce with the instruction back?
Part B: tools/micro/canary.hip (no attention code: VGPR / LDS patterns, FMA chains, exp / rcp softmax sums, DPP and bpermute
        reductions, global reloads, all re-checked bit for bit) beside the same convs.  A canary hit means the instruction damages
        co-resident waves; no hit while part A fails means the defect is in na.hip.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
which = sys.argv[1] if len(sys.argv) > 1 else "x2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if which == "x2":
    os.environ["LMNET_HIP_LIB"] = os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_x2.so")
import torch  # noqa: E402
from lm_net_amd import hip  # noqa: E402
import test_na_stress_gpu as S  # noqa: E402

hip.load()
print("library:", hip.LIB_PATH, flush=True)
hip.set_deterministic(True)
# ---- part A
for dt in (torch.bfloat16, torch.float32):
    for (B, H, Cc) in ((8, 176, 24), (8, 88, 48)):
        for kind in ("conv", "wgrad"):
            bad = S._stress(dt, kind, torch.bfloat16, B, H, Cc, reps, nside=12)
            print("A  na_bwd %-8s %dx%d C=%d beside bf16 %-5s: %d of %d runs differ from the quiet re-run%s" % (
                str(dt).split(".")[1], H, H, Cc, kind, len(bad), reps, ("  first (rep, elements, max) %s" % (bad[:3],)) if bad else ""), flush=True)
hip.set_deterministic(False)

# ---- part B
can = C.CDLL(os.path.join(ROOT, "tools", "micro", "libcanary.so"))
dev = "cuda"
ncb = 1 << 20
cbuf = ((torch.arange(ncb, device=dev) & 1023).float() * 0.5).contiguous()
side, main = torch.cuda.Stream(), torch.cuda.Stream()
SC = 24
for kind in ("conv", "wgrad", "none"):
    for sdt in ((torch.bfloat16, torch.float32) if kind != "none" else (torch.float32,)):
        hip._MMA[0] = hip.BF16 if sdt == torch.bfloat16 else hip.F32
        sx, sdy = torch.randn(8, 176, 176, SC, device=dev).to(sdt), torch.randn(8, 176, 176, SC, device=dev).to(sdt)
        scw = hip.conv_pack(torch.randn(SC, SC, 3, 3, device=dev), 3, [SC])
        scy = torch.empty(8, 176, 176, SC, device=dev, dtype=sdt)
        sdW, sdb = torch.zeros(SC, SC, 3, 3, device=dev), torch.zeros(SC, device=dev)
        err = torch.zeros(8, device=dev, dtype=torch.int32)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        t_side = t_can = 0.0
        for r in range(reps):
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
                ev[0].record(side)
                for _ in range(12):
                    if kind == "wgrad":
                        hip.conv_wgrad([sx], sdy, sdW, sdb, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=3)
                    elif kind == "conv":
                        hip.conv_fwd([sx], scw, scy, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=3)
                ev[1].record(side)
            # the canary: 2 blocks per CU (the convs keep their slots beside it), running while the side launches run
            ev[2].record(main)
            rc = can.launch_canary(C.c_void_p(err.data_ptr()), C.c_void_p(cbuf.data_ptr()), ncb, 512, 1500, r + 1, C.c_void_p(main.cuda_stream))
            ev[3].record(main)
            assert rc == 0, rc
            hip._STREAM[0] = None
            torch.cuda.synchronize()
            t_side += ev[0].elapsed_time(ev[1]); t_can += ev[2].elapsed_time(ev[3])
            if r == reps - 1:
                print("   (last repetition: canary started %.3f ms after the side work, side %.3f ms, canary %.3f ms)" % (
                    ev[0].elapsed_time(ev[2]), ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3])))
        torch.cuda.synchronize()
        e = err.tolist()
        print("B  canary beside %-5s %-8s (side %.2f ms, canary %.2f ms per repetition): launches %d; hits  vgpr %d  lds %d  fma %d  exp/rcp %d  dpp %d  bpermute %d  global %d" % (
            kind, str(sdt).split(".")[1] if kind != "none" else "-", t_side / reps, t_can / reps, e[7], e[0], e[1], e[2], e[3], e[4], e[5], e[6]), flush=True)
hip._MMA[0] = hip.F32
