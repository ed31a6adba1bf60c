"""Where does the attention backward have to run, relative to the `v_mfma_f32_16x16x32_bf16` convs, to go wrong?  

    python tools/gpu_x2_cumask.py [reps]        # loads liblmnet_hip_x2.so (make -C lm_net_amd/csrc x2)

Two HIP streams created with hipExtStreamCreateWithCUMask; tools/micro/cuid.hip verifies on which (XCC, SE, CU) their blocks run.
  shared    both streams may use every compute unit (the round-4 / test_na_stress_gpu scenario)
  disjoint  the co-runner on one half of the compute units of every XCD, the attention backward on the other half: no wave of the
            two kernels shares a CU (SIMD, LDS, L1), they still share each XCD's L2, the fabric, HBM and the power rail
  xcd       the co-runner on XCDs 0-3, the attention backward on XCDs 4-7: nothing shared below the fabric
A failure that survives `disjoint` is not a register-file / LDS / issue effect of co-resident waves.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
os.environ.setdefault("LMNET_HIP_LIB", os.path.join(ROOT, "lm_net_amd", "csrc", "liblmnet_hip_x2.so"))
import torch  # noqa: E402
from lm_net_amd import hip  # noqa: E402
import test_na_stress_gpu as S  # noqa: E402

hip.load()
print("library:", hip.LIB_PATH, flush=True)
torch.cuda.init()
torch.zeros(1, device="cuda")
rt = C.CDLL("libamdhip64.so")
cuid = C.CDLL(os.path.join(ROOT, "tools", "micro", "libcuid.so"))
NW = 8      # 256 mask bits


def masked_stream(bits):
    words = (C.c_uint32 * NW)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(NW)])
    st = C.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(NW), words)
    assert rc == 0, "hipExtStreamCreateWithCUMask rc %d" % rc
    return torch.cuda.ExternalStream(st.value)


def where(stream, n=4096):
    out = torch.full((n,), -1, device="cuda", dtype=torch.int32)
    torch.cuda.synchronize()
    rc = cuid.launch_cuid(C.c_void_p(out.data_ptr()), n, 200000, C.c_void_p(stream.cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    return set(out.tolist())


full = [1] * 256
all_units = sorted(where(masked_stream(full)))
print("units seen with a full mask: %d  (xcc values %s)" % (len(all_units), sorted({u >> 16 for u in all_units})), flush=True)
# which mask bit is which unit?  probe a few single bits
for b in (0, 1, 2, 8, 9, 31, 32, 33, 128, 255):
    bits = [0] * 256; bits[b] = 1
    u = where(masked_stream(bits), 64)
    print("  mask bit %3d -> %s" % (b, sorted("xcc%d se%d cu%02x" % (x >> 16, (x >> 8) & 7, x & 0x1F) for x in u)), flush=True)

layouts = {}
layouts["shared"] = (full, full)
# candidates for "half of every XCD" / "half of the XCDs" under the two plausible bit orders (bit % 8 = XCD, or bit // 32 = XCD):
layouts["even/odd bits"] = ([1 if b % 2 == 0 else 0 for b in range(256)], [1 if b % 2 == 1 else 0 for b in range(256)])
layouts["bit%8 < 4 / >= 4"] = ([1 if b % 8 < 4 else 0 for b in range(256)], [1 if b % 8 >= 4 else 0 for b in range(256)])
layouts["bit//16 even / odd"] = ([1 if (b // 16) % 2 == 0 else 0 for b in range(256)], [1 if (b // 16) % 2 == 1 else 0 for b in range(256)])
layouts["low 128 / high 128"] = ([1 if b < 128 else 0 for b in range(256)], [1 if b >= 128 else 0 for b in range(256)])

hip.set_deterministic(True)
for name, (ms, mm) in layouts.items():
    side, main = masked_stream(ms), masked_stream(mm)
    us, um = where(side), where(main)
    common = us & um
    xs, xm = sorted({u >> 16 for u in us}), sorted({u >> 16 for u in um})
    print("\n== %s: co-runner on %d units (XCDs %s), attention backward on %d units (XCDs %s), %d units in common" % (
        name, len(us), xs, len(um), xm, len(common)), flush=True)
    for kind in ("conv", "wgrad"):
        bad = S._stress(torch.bfloat16, kind, torch.bfloat16, 8, 176, 24, reps, nside=24 if name != "shared" else 12, streams=(side, main))
        print("   na_bwd bf16 176x176 C=24 beside x2 bf16 %-5s: %d of %d runs differ from the quiet re-run%s" % (
            kind, len(bad), reps, ("  first (rep, elements, max) %s" % (bad[:3],)) if bad else ""), flush=True)
hip.set_deterministic(False)
