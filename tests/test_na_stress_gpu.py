"""GPU: the neighborhood-attention backward beside MFMA-heavy work on a second stream (the stand-alone
reproducer of round 4 was tools/gpu_na_stress2.py).

Round 4 found ~1 % errors in dq / dk of `lmn_na_bwd` (wrong sum_n p_n dp_n for ~0.4 % of the queries, dv always right) whenever a
bf16 3x3 conv or weight gradient that issued `v_mfma_f32_16x16x32_bf16` ran on another stream.  Round 5 root-caused it: packed
fp32 VALU instructions whose src1 takes the HIGH half for both lanes (what the SLP vectoriser makes of the odd head of a channel quad at
head_dim 2) return wrong results while another wave on the same compute unit executes that MFMA -- reproduced by tools/micro/canary.hip
without attention code, absent beside the other MFMA instructions and on disjoint compute units.  The product library does not issue the
instruction (csrc/conv_common.h mfma_bf16x2).  This file keeps both scenarios under `-m gpu`: (1) producer (1x1 conv writing dO) ->
lmn_na_bwd on one stream, 3x3 convs / 3x3 weight gradients in bf16 AND fp32 on a second stream, every result BIT-EQUAL to the same call
repeated on a quiet device -- for the one-pass kernel (head_dim <= 2) and the two-pass kernels (head_dim 4), in fp32 and bf16 storage;
(2) the canary, affected instruction forms included, beside the same co-runners: every category zero.  The co-runner must still be running
when the victim runs: ten side launches per repetition (with four, issued before a host-side copy, it had finished first and the library
built with -DLMN_MFMA_X2 passed 240 of 240; with ten it fails 18 of 20 -- gpurun_out/r05c: the test has teeth).  Semantics of the op itself: oracle/natten_ref.py via tests/kernel_checks.py
check_na; call site /root/reference/core/modules.py:509,517.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stress(dt, side_kind, side_dt, B, H, C, reps, nside=10, gpu_rand=True, streams=None):
    from lm_net_amd import hip
    hip.load()
    dev = "cuda"
    heads = 12
    mma_of = lambda d: hip.BF16 if d == torch.bfloat16 else hip.F32
    side, main = streams if streams is not None else (torch.cuda.Stream(), torch.cuda.Stream())   # (tools/gpu_x2_cumask.py passes CU-masked streams)
    g = torch.Generator(device="cpu").manual_seed(1234)
    rnd = (lambda *s: torch.randn(*s, device=dev)) if gpu_rand else (lambda *s: torch.randn(*s, generator=g).to(dev))
    qkv = (rnd(B, H, H, 3 * C) * 0.5).to(dt)
    rpb = rnd(heads, 5, 5) * 0.1
    hip._MMA[0] = mma_of(dt)
    wp = hip.conv_pack(rnd(C, C, 1, 1) * 0.3, 1, [C])
    SC = 24
    sx, sdy = rnd(8, 176, 176, SC).to(side_dt), rnd(8, 176, 176, SC).to(side_dt)
    hip._MMA[0] = mma_of(side_dt)
    scw = hip.conv_pack(rnd(SC, SC, 3, 3), 3, [SC])
    scy = torch.empty(8, 176, 176, SC, device=dev, dtype=side_dt)
    sdW, sdb = torch.zeros(SC, SC, 3, 3, device=dev), torch.zeros(SC, device=dev)
    qkv0 = qkv.clone()
    bad = []
    try:
        for r in range(reps):
            da = rnd(B, H, H, C).to(dt)
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
                hip._MMA[0] = mma_of(side_dt)
                for _ in range(nside):
                    if side_kind == "wgrad":
                        hip.conv_wgrad([sx], sdy, sdW, sdb, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=3)
                    else:
                        hip.conv_fwd([sx], scw, scy, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=3)
            with torch.cuda.stream(main):
                hip._STREAM[0] = hip.C.c_void_p(main.cuda_stream)
                hip._MMA[0] = mma_of(dt)
                do = torch.empty(B, H, H, C, device=dev, dtype=dt)
                hip.conv_fwd([da.view(1, 1, -1, C)], wp, do.view(1, 1, -1, C), B=1, Hin=1, Win=B * H * H, Hout=1, Wout=B * H * H, Cout=C, ksize=1)
                dq1 = torch.empty_like(qkv)
                hip.na_bwd(qkv, rpb, do, dq1, torch.zeros_like(rpb), heads)
            hip._STREAM[0] = None
            torch.cuda.synchronize()
            dq2 = torch.empty_like(qkv)
            hip.na_bwd(qkv, rpb, do, dq2, torch.zeros_like(rpb), heads)
            torch.cuda.synchronize()
            assert torch.equal(qkv, qkv0), "the attention backward's INPUT changed"
            if not torch.equal(dq1, dq2):
                d = (dq1.float() - dq2.float()).abs()
                bad.append((r, int((d > 0).sum()), float(d.max())))
    finally:
        hip._STREAM[0] = None
        hip._MMA[0] = hip.F32
    return bad


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(8, 176, 24), (8, 88, 48)], ids=["one-pass-hd2", "two-pass-hd4"])
def test_na_backward_is_bit_stable_beside_convs_on_a_second_stream(dt, shape):
    from lm_net_amd import hip
    B, H, C = shape
    hip.load()
    hip.set_deterministic(True)     # (fixed-order bias-table gradient: the comparison is bitwise)
    try:
        for side_kind, side_dt in (("conv", torch.bfloat16), ("wgrad", torch.bfloat16), ("conv", torch.float32), ("wgrad", torch.float32)):
            bad = _stress(dt, side_kind, side_dt, B, H, C, reps=8 if H == 176 else 7)
            assert not bad, "na_bwd differs from its quiet re-run beside %s %s: (rep, elements, max) %s" % (side_dt, side_kind, bad[:4])
    finally:
        hip.set_deterministic(False)


CANARY_CATS = ["vgpr", "lds", "fma", "exp/rcp", "dpp", "bpermute", "global", None, "lds-tile pattern", "lds-tile global", None, "packed fma",
               "pk_fma src1-high", "pk_mov+pk_add", "pk_mul src0 swap", "pk_fma src1-low"] + ["form %d" % c for c in range(8)]


def _canary_lib():
    import ctypes
    import os
    import subprocess
    micro = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "micro")
    so = os.path.join(micro, "libcanary.so")
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(micro, "canary.hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(micro, "canary.hip")])
    return ctypes.CDLL(so)


def test_register_canary_is_clean_beside_the_product_convs():
    """Root cause of the round-4 corruption (gpurun_out/r05v, r05w): packed fp32 VALU instructions whose src1 selects the HIGH half
    for both lanes (`v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[.,1,.] op_sel_hi:[.,1,.]`, what the SLP vectoriser makes of the
    attention backward's odd head at head_dim 2) return wrong results while another wave on the SAME compute unit executes
    `v_mfma_f32_16x16x32_bf16` -- 0.4-12 M wrong results per 30 launches of tools/micro/canary.hip beside the -DLMN_MFMA_X2 convs, none
    beside `v_mfma_f32_16x16x16_bf16` / `v_mfma_f32_16x16x4_f32`, none when the two kernels run on disjoint compute units.  The product
    library does not issue the instruction; this test keeps every category of the canary (incl. the affected forms) at zero beside the
    product's bf16 and fp32 3x3 convs and weight gradients."""
    import ctypes as C
    from lm_net_amd import hip
    hip.load()
    can = _canary_lib()
    dev = "cuda"
    ncb = 1 << 20
    cbuf = ((torch.arange(ncb, device=dev) & 1023).float() * 0.5).contiguous()
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    SC, reps = 24, 6
    try:
        for sdt in (torch.bfloat16, torch.float32):
            hip._MMA[0] = hip.BF16 if sdt == torch.bfloat16 else hip.F32
            sx, sdy = torch.randn(8, 176, 176, SC, device=dev).to(sdt), torch.randn(8, 176, 176, SC, device=dev).to(sdt)
            scw = hip.conv_pack(torch.randn(SC, SC, 3, 3, device=dev), 3, [SC])
            scy = torch.empty(8, 176, 176, SC, device=dev, dtype=sdt)
            sdW, sdb = torch.zeros(SC, SC, 3, 3, device=dev), torch.zeros(SC, device=dev)
            for kind in ("conv", "wgrad"):
                err = torch.zeros(32, device=dev, dtype=torch.int32)
                for r in range(reps):
                    torch.cuda.synchronize()
                    with torch.cuda.stream(side):
                        hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
                        for _ in range(12):
                            if kind == "wgrad":
                                hip.conv_wgrad([sx], sdy, sdW, sdb, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=3)
                            else:
                                hip.conv_fwd([sx], scw, scy, B=8, Hin=176, Win=176, Hout=176, Wout=176, Cout=SC, ksize=3)
                    rc = can.launch_canary(C.c_void_p(err.data_ptr()), C.c_void_p(cbuf.data_ptr()), ncb, 512, 300, r + 1, C.c_void_p(main.cuda_stream))
                    assert rc == 0
                    hip._STREAM[0] = None
                    torch.cuda.synchronize()
                e = err.tolist()
                assert e[7] == reps, "canary launches that ran to the end: %d of %d" % (e[7], reps)
                hits = {CANARY_CATS[i]: e[i] for i in range(len(CANARY_CATS)) if CANARY_CATS[i] is not None and e[i]}
                assert not hits, "canary hits beside the product's %s %s: %s" % (sdt, kind, hits)
    finally:
        hip._STREAM[0] = None
        hip._MMA[0] = hip.F32
