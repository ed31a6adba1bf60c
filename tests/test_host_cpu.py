"""CPU: host-side logic of the product that needs no GPU -- the C-ABI library loads and exports every
symbol include/lmnet_hip.h declares, struct mirrors match, state_dict is checkpoint-compatible with the
reference, and the product refuses to run without the device (no silent fallback)."""
import ctypes
import json
import os
import re

import pytest
import torch

from helpers import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    from lm_net_amd import hip
    lib = hip.load()
    hdr = open(os.path.join(ROOT, "include", "lmnet_hip.h")).read()
    declared = set(re.findall(r"\b(lmn_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"lmn_src_t", "lmn_stream_t"}
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lmn_abi_version() == hip.ABI_VERSION
    assert lib.lmn_sizeof_conv_args() == ctypes.sizeof(hip.ConvArgs)
    assert lib.lmn_sizeof_src() == ctypes.sizeof(hip.SrcT)
    assert lib.lmn_sizeof_wgrad_args() == ctypes.sizeof(hip.WgradArgs)


def test_pack_size_arithmetic():
    from lm_net_amd import hip
    # taps * K16-blocks * cout tiles * 256
    assert hip.conv_pack_size(1, 12, [12]) == 1 * 1 * 1 * 256
    assert hip.conv_pack_size(3, 96, [96, 96]) == 9 * 12 * 6 * 256
    assert hip.conv_pack_size(1, 1116, [372]) == 24 * 70 * 256


def test_state_dict_matches_reference_keys_and_loads_oracle_checkpoint():
    from lm_net_amd import LM_Net
    from oracle.lmnet_ref import LM_Net as Oracle
    keys = json.load(open(os.path.join(GOLDEN, "keys.json")))
    m = LM_Net(3, 2)
    sd = m.state_dict()
    assert list(sd.keys()) == list(keys["train"].keys()) and len(sd) == 766
    assert all(list(v.shape) == keys["train"][k] for k, v in sd.items())
    assert sum(p.numel() for p in m.parameters()) == 3966566
    m.load_state_dict(Oracle(3, 2).state_dict())          # a reference-format checkpoint loads as is
    m.structural_reparam()
    sd = m.state_dict()
    assert list(sd.keys()) == list(keys["deploy"].keys()) and len(sd) == 510


def test_reparam_folding_matches_oracle():
    from lm_net_amd import LM_Net
    from oracle.lmnet_ref import LM_Net as Oracle
    from tools.detweights import fill_module
    a, b = LM_Net(3, 2, filters=[12] * 5), Oracle(3, 2, filters=[12] * 5)
    fill_module(a); fill_module(b)
    a.structural_reparam(); b.structural_reparam()
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == k2 and torch.allclose(v, v2, atol=1e-6), k


def test_no_cpu_fallback_and_input_validation():
    from lm_net_amd import LM_Net
    m = LM_Net(3, 2, filters=[12] * 5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError):                      # parameter containers have no compute path
        m.conv1[0](torch.zeros(1, 3, 32, 32))
    with pytest.raises(AssertionError):
        LM_Net(3, 2, filters=[10, 20, 40, 80, 160])       # 12 heads need multiples of 12


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lm_net_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", "").replace("CPU oracle", "") or fn == "__none__", fn


def test_rows_around_the_path_have_no_cpu_fallback_either():
    """SegLoss / FusedAdamW / ConfusionMeter refuse CPU tensors instead of falling back."""
    from lm_net_amd import LM_Net
    from lm_net_amd.loss import SegLoss
    from lm_net_amd.metrics import ConfusionMeter
    from lm_net_amd.optim import FusedAdamW
    lg, y = torch.zeros(1, 2, 32, 32, requires_grad=True), torch.zeros(1, 32, 32, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SegLoss()(lg, y)
    with pytest.raises(ValueError):
        SegLoss(ce_weight=(1.0, 2.0, 3.0), dice_weight=(1.0, 2.0, 3.0))(lg, y)      # class-count mismatch
    with pytest.raises(RuntimeError, match="GPU first"):
        FusedAdamW(LM_Net(3, 2, filters=[12] * 5))
    with pytest.raises(TypeError):
        FusedAdamW(torch.nn.Linear(2, 2))
    m = ConfusionMeter(2, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.update(lg.detach(), y)
    m.total += torch.tensor([[50.0, 10.0], [5.0, 35.0]], dtype=torch.float64)        # rows = label, cols = prediction
    r = m.compute()
    assert abs(r["dice"][1] - 2 * 35 / (2 * 35 + 10 + 5)) < 1e-12 and abs(r["iou"][1] - 35 / 50) < 1e-12
    assert abs(r["accuracy"] - 0.85) < 1e-12


def test_pack_plan_records_once_and_drops_on_storage_change():
    """hip.PackPlan bookkeeping (no launches): jobs are keyed by weight storage and dropped when it moves."""
    from lm_net_amd import hip
    plan = hip.PackPlan()
    w = torch.nn.Parameter(torch.zeros(4, 4))
    out = torch.zeros(16)
    plan.record(("k",), w, out, dict(w=w.data_ptr(), ksize=1, Cout=4, Cin=4, c=[4], transposed=0, row_off=0, rows=0))
    assert plan.lookup(("k",)) is None            # not fresh until refresh() has re-packed
    plan.fresh = True
    assert plan.lookup(("k",)) is out
    w.data = torch.ones(4, 4)                     # storage replaced (model.to(), load via .data=)
    plan.refresh()
    assert not plan.jobs and not plan.fresh


def test_wgrad_reduce_job_geometry_is_host_arithmetic():
    """lmn_conv_wgrad_job (no GPU call): the deferred reduction it describes fits the workspace lmn_conv_wgrad_workspace asks
    for, covers every cout x cin tile set, and small problems keep reducing by themselves (nblk == 0)."""
    from lm_net_amd import hip
    lib = hip.load()
    assert lib.lmn_sizeof_reduce_job() == ctypes.sizeof(hip.ReduceJob)

    def job(B, H, W, cin, cout, k):
        a = hip.WgradArgs()
        a.B, a.Hout, a.Wout, a.Hin, a.Win, a.ksize, a.stride, a.nsrc, a.Cout = B, H, W, H, W, k, 1, 1, cout
        a.src[0].ptr, a.src[0].C, a.src[0].cstride = 0x10000, cin, cin
        a.dy, a.dy_cstride, a.dW = 0x20000, cout, 0x30000
        need = int(lib.lmn_conv_wgrad_workspace(ctypes.byref(a)))
        a.workspace, a.workspace_floats, a.defer_reduce = 0x40000 if need else None, need, 1
        j = hip.ReduceJob()
        assert lib.lmn_conv_wgrad_job(ctypes.byref(a), ctypes.byref(j)) == 0
        return need, j

    for (B, H, W, cin, cout, k) in [(8, 352, 352, 12, 24, 1), (8, 176, 176, 72, 24, 3), (8, 44, 44, 192, 96, 1), (8, 22, 22, 372, 372, 3)]:
        need, j = job(B, H, W, cin, cout, k)
        assert need > 0 and j.nblk > 0
        assert j.gy * j.nblk * j.per <= need
        assert j.per == j.taps * j.NMT * j.NNT * 256 + j.NMT * 16 and j.taps == k * k
        assert j.gy == -(-j.NMTT // j.NMT) * j.nsets_n and j.NMTT == -(-cout // 16) and j.NNTT == -(-cin // 16)
        assert j.ksl in (1, 2, 4, 8, 16, 32, 64) and j.blocks_per_set * (4096 // j.ksl) >= j.per
    need, j = job(1, 4, 4, 12, 12, 1)        # 16 pixels: one block, atomics
    assert j.nblk == 0


def test_stream_key_of_the_default_stream_is_zero_not_none():
    """hip.stream_key(): the launch stream as a dictionary key.  `ctypes.c_void_p(0).value` is None -- the deferred weight-gradient
    reductions of a serial run on the default stream were once filed under 0 and looked up under None (never flushed)."""
    import ctypes
    from lm_net_amd import hip
    saved = hip._STREAM[0]
    try:
        hip._STREAM[0] = ctypes.c_void_p(0)
        assert hip.stream_key() == 0 and isinstance(hip.stream_key(), int)
        hip._STREAM[0] = ctypes.c_void_p(0x7f00deadbeef)
        assert hip.stream_key() == 0x7f00deadbeef
        assert hip.wgrad_reduce_pending(0) == 0 and hip.wgrad_reduce_pending() == 0
    finally:
        hip._STREAM[0] = saved


def test_row_planar_layout_helpers_follow_the_documented_index():
    """include/lmnet_hip.h, Conventions: element (b, y, x, c) of a row-planar (RP4) tensor sits at
    ((b*H + y)*(C/4) + (c >> 2))*4W + 4x + (c & 3).  hip.nhwc_to_rp4 / rp4_to_nhwc are what tests and callers holding NHWC data use
    to talk to the depthwise entries: check them against the formula element by element, that they invert each other, that the
    marker attribute travels with rp4() and that V() picks the image width up."""
    from lm_net_amd import hip
    B, H, W, Cn = 2, 3, 5, 12
    t = torch.arange(B * H * W * Cn, dtype=torch.float32).reshape(B, H, W, Cn)
    r = hip.nhwc_to_rp4(t)
    assert r.shape == t.shape and hip.is_rp4(r) and not hip.is_rp4(t)
    flat = r.reshape(-1)
    for b in range(B):
        for y in range(H):
            for x in range(W):
                for c in range(Cn):
                    off = ((b * H + y) * (Cn // 4) + (c >> 2)) * 4 * W + 4 * x + (c & 3)
                    assert float(flat[off]) == float(t[b, y, x, c]), (b, y, x, c)
    assert torch.equal(hip.rp4_to_nhwc(r), t)
    assert hip.V(r).rp == W and hip.V(t).rp == 0
    with pytest.raises(AssertionError):
        hip.V(r, 4, 4)          # a channel slice of a row-planar tensor is not a strided view: refused


def test_priority_stream_table_never_refuses():
    """lmn_set_priority_stream (ABI 14) is a scheduling hint: pure host state, levels 0..3, and a caller that cycles through more than
    eight streams replaces the oldest entry instead of getting an error."""
    import ctypes as C
    from lm_net_amd import hip
    lib = hip.load()
    for i in range(20):
        assert lib.lmn_set_priority_stream(C.c_void_p(0x1000 + 64 * i), 3) == 0
    assert lib.lmn_set_priority_stream(C.c_void_p(0x1000), 4) != 0        # level out of range
    for i in range(20):
        assert lib.lmn_set_priority_stream(C.c_void_p(0x1000 + 64 * i), 0) == 0
