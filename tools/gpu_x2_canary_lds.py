"""Second canary run: the LDS staging -> barrier -> window read cycle (tools/micro/canary.hip canary_lds_kernel) and the
packed-fp32 chain of the first canary beside the x2 convs on the same compute units.   python tools/gpu_x2_canary_lds.py [reps]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ.setdefault("LMNET_HIP_LIB", os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_x2.so"))
import torch  # noqa: E402
from lm_net_amd import hip  # noqa: E402
hip.load()
print("library:", hip.LIB_PATH, flush=True)
can = C.CDLL(os.path.join(ROOT, "tools", "micro", "libcanary.so"))
dev = "cuda"
ncb = 1 << 20
cbuf = ((torch.arange(ncb, device=dev) & 1023).float() * 0.5).contiguous()
side, main = torch.cuda.Stream(), torch.cuda.Stream()
SC = 24
SIDE_DT = torch.float32 if os.environ.get("CANARY_SIDE") == "f32" else torch.bfloat16
for which in ("regs",) if os.environ.get("CANARY_REGS_ONLY") else ("lds", "regs"):
    for kind in ("conv", "wgrad", "none"):
        hip._MMA[0] = hip.BF16 if SIDE_DT == torch.bfloat16 else hip.F32
        sdt = SIDE_DT
        sx, sdy = torch.randn(8, 176, 176, SC, device=dev).to(sdt), torch.randn(8, 176, 176, SC, device=dev).to(sdt)
        scw = hip.conv_pack(torch.randn(SC, SC, 3, 3, device=dev), 3, [SC])
        scy = torch.empty(8, 176, 176, SC, device=dev, dtype=sdt)
        sdW, sdb = torch.zeros(SC, SC, 3, 3, device=dev), torch.zeros(SC, device=dev)
        err = torch.zeros(32, device=dev, dtype=torch.int32)
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
            ev[2].record(main)
            if which == "lds":   # 512 blocks of 512 threads, 56 KB of LDS each (2 per CU beside the convs), ~1-3 ms
                rc = can.launch_canary_lds(C.c_void_p(err.data_ptr()), C.c_void_p(cbuf.data_ptr()), ncb, 512, 400, r + 1, 3584, C.c_void_p(main.cuda_stream))
            else:
                rc = can.launch_canary(C.c_void_p(err.data_ptr()), C.c_void_p(cbuf.data_ptr()), ncb, 512, 1500, r + 1, C.c_void_p(main.cuda_stream))
            ev[3].record(main)
            assert rc == 0, rc
            hip._STREAM[0] = None
            torch.cuda.synchronize()
            t_side += ev[0].elapsed_time(ev[1]); t_can += ev[2].elapsed_time(ev[3])
        e = err.tolist()
        if which == "lds":
            print("canary_lds beside %s %-5s (side %.2f ms, canary %.2f ms per repetition): launches %d; hits  pattern slots %d  global-sourced slots %d" % (
                str(SIDE_DT).split(".")[1], kind, t_side / reps, t_can / reps, e[10], e[8], e[9]), flush=True)
        else:
            print("canary     beside %s %-5s (side %.2f ms, canary %.2f ms per repetition): launches %d; hits  vgpr %d lds %d fma %d PACKED-fma %d exp/rcp %d dpp %d bpermute %d global %d | op_sel forms: pk_fma[0,1,0] %d  pk_mov[1,0]+pk_add %d  pk_mul[1,0] %d  pk_fma plain-sel %d" % (
                str(SIDE_DT).split(".")[1], kind, t_side / reps, t_can / reps, e[7], e[0], e[1], e[2], e[11], e[3], e[4], e[5], e[6], e[12], e[13], e[14], e[15]), flush=True)
            print("           form matrix (tools/micro/canary.hip 2d): " + "  ".join("%d:%d" % (c, e[16 + c]) for c in range(8)), flush=True)
hip._MMA[0] = hip.F32
