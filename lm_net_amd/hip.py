"""ctypes binding of ``liblmnet_hip.so`` (the C-ABI declared in ``include/lmnet_hip.h``).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic op of the
LM-Net path is a hand-written gfx950 kernel behind this boundary.  There is NO fallback: if the
shared library is missing the import raises, and calling any op with a non-device tensor raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.abspath(os.environ["LMNET_HIP_LIB"]) if os.environ.get("LMNET_HIP_LIB") else os.path.join(_HERE, "liblmnet_hip.so")   # (override: A/B runs against another build)

# ---- constants mirrored from include/lmnet_hip.h
SRC_GELU, SRC_DROP, SRC_LN, SRC_UP2 = 1, 2, 4, 8
EP_LINEAR, EP_AFFINE_ACT, EP_DGELU, EP_BN_BWD1, EP_BN_BWD2, EP_SE_BWD, EP_LN_BWD = 0, 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_HSWISH, ACT_GELU = 0, 1, 2
STATS_NONE, STATS_SUM_SQ, STATS_EP = 0, 1, 2
F32, BF16 = 0, 1            # matrix-core operand type of the dense contractions (LMN_F32 / LMN_BF16)
_MMA = [F32]                # ... of the pass in flight (engine.begin_pass)
ABI_VERSION = 14


class SrcT(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scale", C.c_void_p), ("C", C.c_int32), ("cstride", C.c_int32),
                ("flags", C.c_int32), ("drop_seed", C.c_uint32), ("drop_p", C.c_float), ("rp_w", C.c_int32),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_eps", C.c_float), ("_pad0", C.c_int32)]


class BnFin(C.Structure):
    _fields_ = [("mode", C.c_int32), ("nrep", C.c_int32), ("count", C.c_float), ("eps", C.c_float), ("momentum", C.c_float),
                ("batch_stats", C.c_int32), ("sums", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("about", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("A", C.c_void_p), ("shift", C.c_void_p),
                ("rmean", C.c_void_p), ("rvar", C.c_void_p), ("Ain", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


FIN_BN, FIN_BN_BWD = 1, 2


class ConvChain(C.Structure):
    """Mirror of lmn_conv_chain_t: a second 1x1 conv applied to the output tile of a 1x1 conv_fwd call."""
    _fields_ = [("wpack", C.c_void_p), ("bias", C.c_void_p), ("shift", C.c_void_p), ("aux", C.c_void_p), ("out", C.c_void_p),
                ("stats", C.c_void_p), ("Cout", C.c_int32), ("out_cstride", C.c_int32), ("out_rp_w", C.c_int32),
                ("aux_cstride", C.c_int32), ("aux_rp_w", C.c_int32), ("epilogue", C.c_int32), ("stats_mode", C.c_int32),
                ("stats_rep", C.c_int32), ("stats_snap", C.c_int32)]


class ConvArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("transposed", C.c_int32), ("nsrc", C.c_int32),
                ("Cout", C.c_int32), ("src", SrcT * 3), ("wpack", C.c_void_p), ("bias", C.c_void_p),
                ("p0", C.c_void_p), ("p1", C.c_void_p), ("p2", C.c_void_p), ("p3", C.c_void_p), ("p4", C.c_void_p),
                ("aux", C.c_void_p), ("residual", C.c_void_p), ("out", C.c_void_p), ("stats", C.c_void_p),
                ("aux_cstride", C.c_int32), ("res_cstride", C.c_int32), ("out_cstride", C.c_int32),
                ("epilogue", C.c_int32), ("act", C.c_int32), ("stats_mode", C.c_int32),
                ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("seed_ctr", C.c_void_p), ("bias2", C.c_void_p),
                ("stats_rep", C.c_int32), ("mma_dtype", C.c_int32), ("act_dtype", C.c_int32), ("out_rp_w", C.c_int32),
                ("p5", C.c_void_p), ("p6", C.c_void_p), ("fin", BnFin), ("stats_snap", C.c_int32), ("aux_rp_w", C.c_int32),
                ("chain", ConvChain)]


class WgradArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("Hout", C.c_int32), ("Wout", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("nsrc", C.c_int32), ("Cout", C.c_int32),
                ("src", SrcT * 3), ("dy", C.c_void_p), ("dy_cstride", C.c_int32), ("dy_flags", C.c_int32),
                ("dy_seed", C.c_uint32), ("dy_p", C.c_float), ("dW", C.c_void_p), ("db", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_floats", C.c_int64), ("seed_ctr", C.c_void_p),
                ("dW_src", C.c_void_p * 3), ("db2", C.c_void_p), ("mma_dtype", C.c_int32), ("act_dtype", C.c_int32),
                ("defer_reduce", C.c_int32), ("dy_rp_w", C.c_int32)]


class DwPre(C.Structure):
    """Mirror of lmn_dw_pre_t (z-path of ReparamConv: the depthwise kernels form x1 = Hardswish(A z + shift) themselves)."""
    _fields_ = [("A", C.c_void_p), ("shift", C.c_void_p), ("fin", BnFin)]


def _fill_fin(f, fin):
    f.mode, f.nrep, f.count = fin["mode"], fin["nrep"], float(fin["count"])
    f.eps, f.momentum, f.batch_stats = float(fin.get("eps", 0.0)), float(fin.get("momentum", 0.0)), int(fin.get("batch_stats", 1))
    for k in ("sums", "gamma", "beta", "about", "mean", "rstd", "A", "shift", "rmean", "rvar", "Ain", "dgamma", "dbeta"):
        t = fin.get(k)
        setattr(f, k, t.data_ptr() if t is not None else None)


def _dw_pre(zpre):
    """zpre: None | dict(A=, shift=) | dict(fin=dict(...)) (lmn_dw_stats only: A / shift are formed from the batch sums)"""
    if zpre is None:
        return None
    p = DwPre()
    if zpre.get("fin") is not None:
        _fill_fin(p.fin, zpre["fin"])
    else:
        p.A, p.shift = _p(zpre["A"]).value, _p(zpre["shift"]).value
    return C.byref(p)


class SeFuse(C.Structure):
    """Mirror of lmn_se_fuse_t."""
    _fields_ = [("ticket", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("s", C.c_void_p), ("hidden", C.c_void_p), ("inv_hw", C.c_float), ("R", C.c_int32)]


class SeBwd(C.Structure):
    """Mirror of lmn_se_bwd_t."""
    _fields_ = [("ds", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p), ("hidden", C.c_void_p), ("dvec", C.c_void_p),
                ("inv_hw", C.c_float), ("R", C.c_int32)]


def _se_fuse(se):
    """se: None | dict(ticket=[B] zeroed fp32/int32 tensor, fc1w, fc1b, fc2w, fc2b, s, hidden, inv_hw)"""
    if se is None:
        return None
    f = SeFuse()
    f.ticket = se["ticket"].data_ptr()
    f.w1, f.b1, f.w2, f.b2 = (_p(se[k]).value for k in ("fc1w", "fc1b", "fc2w", "fc2b"))
    f.s, f.hidden = _p(se["s"]).value, _p(se["hidden"]).value
    f.inv_hw, f.R = float(se["inv_hw"]), se["fc1w"].shape[0]
    return C.byref(f)


class ReduceJob(C.Structure):
    """Mirror of lmn_reduce_job_t (the deferred second stage of a weight gradient's K-split reduction)."""
    _fields_ = [("partial", C.c_void_p), ("first_block", C.c_int64),
                ("nblk", C.c_int32), ("per", C.c_int32), ("gy", C.c_int32), ("nsets_n", C.c_int32),
                ("taps", C.c_int32), ("NMT", C.c_int32), ("NNT", C.c_int32), ("nsrc", C.c_int32),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("NMTT", C.c_int32), ("NNTT", C.c_int32),
                ("srcC", C.c_int32 * 3), ("ntile_off", C.c_int32 * 3), ("cbase", C.c_int32 * 3),
                ("ksl", C.c_int32), ("blocks_per_set", C.c_int32), ("_pad", C.c_int32),
                ("dW", C.c_void_p), ("dW_src", C.c_void_p * 3), ("db", C.c_void_p), ("db2", C.c_void_p)]


# every symbol include/lmnet_hip.h declares (the CPU test suite checks the library exports all of them)
SYMBOLS = [
    "lmn_abi_version", "lmn_sizeof_conv_args", "lmn_sizeof_src", "lmn_sizeof_wgrad_args", "lmn_last_error",
    "lmn_conv_pack_size", "lmn_conv_pack", "lmn_conv_pack_batch", "lmn_sizeof_pack_job", "lmn_conv_fwd", "lmn_conv_dma_config", "lmn_conv_chain_ok", "lmn_conv_wgrad", "lmn_conv_wgrad_workspace",
    "lmn_conv_wgrad_job", "lmn_conv_wgrad_up2_ok", "lmn_wgrad_reduce_batch", "lmn_sizeof_reduce_job", "lmn_reparam_fold", "lmn_affine2", "lmn_reparam_wfin", "lmn_bnact_fwd_fin", "lmn_bnact_bwd_fin",
    "lmn_dw_stats", "lmn_dw_fwd", "lmn_dw_merge", "lmn_dw_finalize_merge", "lmn_dw_bwd_stats", "lmn_dw_bwd_coef", "lmn_dw_bwd", "lmn_dw_fwd_bn", "lmn_dw_bwd_bn",
    "lmn_se_fwd", "lmn_se_bwd", "lmn_se_bwd_dm", "lmn_se_bwd_params", "lmn_na_fwd", "lmn_na_bwd", "lmn_plan_host_profile", "lmn_set_deterministic", "lmn_get_deterministic", "lmn_gattn_fwd", "lmn_gattn_bwd",
    "lmn_ln_fwd", "lmn_ln_bwd", "lmn_bnact_fwd", "lmn_bnact_bwd_stats", "lmn_bnact_bwd",
    "lmn_bn_finalize", "lmn_bn_fold", "lmn_bn_bwd_coef", "lmn_up2_fwd", "lmn_up2_bwd", "lmn_avgpool_fwd", "lmn_avgpool_bwd",
    "lmn_nchw_to_nhwc", "lmn_nhwc_to_nchw", "lmn_adamw_step", "lmn_segloss_fwd", "lmn_segloss_bwd", "lmn_confusion", "lmn_preprocess_u8", "lmn_fill", "lmn_add", "lmn_colsum", "lmn_copy_slice", "lmn_copy2d",
    "lmn_stream_wait", "lmn_event_record", "lmn_event_wait", "lmn_set_priority_stream", "lmn_plan_create", "lmn_plan_destroy", "lmn_plan_record_begin", "lmn_plan_record_end", "lmn_plan_size",
    "lmn_plan_run", "lmn_prof_begin", "lmn_prof_end",
]

_lib = None


def load():
    """Load the HIP library (once).  Raises if it has not been built -- there is no CPU/torch fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            "lm_net_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C lm_net_amd/csrc`). The LM-Net hot path has no non-HIP fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise RuntimeError("lm_net_amd: %s does not export %s" % (LIB_PATH, name))
    lib.lmn_last_error.restype = C.c_char_p
    lib.lmn_conv_pack_size.restype = C.c_int64
    lib.lmn_conv_wgrad_workspace.restype = C.c_int64
    lib.lmn_plan_create.restype = C.c_void_p
    lib.lmn_plan_record_end.restype = C.c_int64
    lib.lmn_plan_size.restype = C.c_int64
    lib.lmn_prof_end.restype = C.c_int64
    if lib.lmn_abi_version() != ABI_VERSION:
        raise RuntimeError("lm_net_amd: ABI version mismatch")
    if (lib.lmn_sizeof_conv_args() != C.sizeof(ConvArgs) or lib.lmn_sizeof_src() != C.sizeof(SrcT)
            or lib.lmn_sizeof_wgrad_args() != C.sizeof(WgradArgs) or lib.lmn_sizeof_reduce_job() != C.sizeof(ReduceJob)):
        raise RuntimeError("lm_net_amd: argument struct layout differs between hip.py and lmnet_hip.h")
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("lm_net_amd.%s failed (code %d): %s" % (what, rc, load().lmn_last_error().decode()))


def _p(t):
    """fp32 device pointer (parameters, statistics, workspaces, fp32 boundary tensors)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("lm_net_amd: device tensor required (got a CPU tensor); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError("lm_net_amd: fp32 tensor required, got %s" % t.dtype)
    return C.c_void_p(t.data_ptr())


def _pa(t):
    """activation pointer: fp32 or bf16 storage (the call's act_dtype says which, see _dt)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("lm_net_amd: device tensor required (got a CPU tensor); the HIP path has no CPU fallback")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("lm_net_amd: fp32 or bf16 activation tensor required, got %s" % t.dtype)
    return C.c_void_p(t.data_ptr())


def _dt(*ts):
    """act_dtype of a call: the storage type shared by all of its activation tensors."""
    d = None
    for t in ts:
        if t is None:
            continue
        t = t.t if isinstance(t, V) else t
        if d is None:
            d = t.dtype
        elif t.dtype != d:
            raise RuntimeError("lm_net_amd: activation tensors of one call must share a storage type (%s vs %s)" % (d, t.dtype))
    return BF16 if d == torch.bfloat16 else F32


_ALLOC = [None]    # allocator of the pass in flight (engine.begin_pass): fn(device, shape) -> fp32 tensor


def _default_alloc(device, shape):
    return torch.empty(shape, device=device, dtype=torch.float32)


_SEED_CTR = [None]  # device int32[1] added to every dropout seed (graph mode: bumped once per step), or None
_STREAM = [None]   # stream handle of the pass in flight (engine.begin_pass): torch.cuda.current_stream() costs ~3 us


def _stream():
    return _STREAM[0] if _STREAM[0] is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)


def stream_key():
    """The launch stream as a dictionary key: its handle as an int (0 = the legacy default stream; `c_void_p(0).value` is None,
    which once made the deferred reductions of a serial run on the default stream unreachable)."""
    v = _STREAM[0].value if _STREAM[0] is not None else torch.cuda.current_stream().cuda_stream
    return int(v or 0)


def _i64(v):
    return C.c_int64(int(v))


def _f(v):
    return C.c_float(float(v))


def rp4(t):
    """Mark ``t`` (shape [B, H, W, C], contiguous) as ROW-PLANAR: its memory holds element (b, y, x, c) at
    ((b*H + y) * (C/4) + (c >> 2)) * 4*W + 4*x + (c & 3) (include/lmnet_hip.h, Conventions) -- the layout of the E-wide tensors
    inside a ReparamConv block.  The shape stays [B, H, W, C] (it names the sizes); the conv family picks the layout up from the
    mark (lmn_src_t.rp_w / out_rp_w / aux_rp_w / dy_rp_w), the depthwise family takes nothing else.
    The mark is a Python attribute: clone() / view() / detach() / slicing return UNMARKED tensors -- the depthwise wrappers
    refuse those (_rp_req) instead of reading NHWC memory as row-planar."""
    if t.dim() != 4 or not t.is_contiguous() or t.shape[-1] % 4 != 0:
        raise RuntimeError("lm_net_amd: rp4() takes a contiguous [B, H, W, C] tensor with C %% 4 == 0 (got shape %s, contiguous=%s)"
                           % (tuple(t.shape), t.is_contiguous()))
    t._lmn_rp = True
    return t


def is_rp4(t):
    return getattr(t, "_lmn_rp", False)


def _rp_req(name, **ts):
    """The depthwise family reads and writes row-planar tensors only: every activation operand must carry the rp4 mark."""
    for k, t in ts.items():
        if t is not None and not getattr(t, "_lmn_rp", False):
            raise RuntimeError("lm_net_amd: %s: operand `%s` is not marked row-planar (hip.rp4 / hip.nhwc_to_rp4); "
                               "an NHWC tensor here would be read with permuted channels" % (name, k))


def nhwc_to_rp4(t):
    """torch-side layout conversion (tests, debugging): NHWC values -> a row-planar tensor of the same shape."""
    B, H, W, Cn = t.shape
    return rp4(t.reshape(B, H, W, Cn // 4, 4).permute(0, 1, 3, 2, 4).contiguous().reshape(B, H, W, Cn))


def rp4_to_nhwc(t):
    """torch-side layout conversion (tests, debugging): the NHWC values of a row-planar tensor."""
    B, H, W, Cn = t.shape
    return t.reshape(B, H, Cn // 4, W, 4).permute(0, 1, 3, 2, 4).contiguous().reshape(B, H, W, Cn)


class V:
    """A channel slice [off, off+C) of an NHWC tensor ``t`` of shape [..., Ctot] (contiguous); rp: image width W when the
    tensor is row-planar (whole tensors only), else 0."""
    __slots__ = ("t", "off", "C", "rp")

    def __init__(self, t, off=0, Cn=None):
        assert t.is_contiguous()
        self.t, self.off = t, off
        self.C = t.shape[-1] - off if Cn is None else Cn
        self.rp = 0
        if getattr(t, "_lmn_rp", False):
            assert off == 0 and self.C == t.shape[-1] and t.dim() == 4, "row-planar tensors are used whole"
            self.rp = t.shape[2]

    @property
    def ptr(self):
        if not self.t.is_cuda or self.t.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("lm_net_amd: fp32 or bf16 device tensor required")
        return self.t.data_ptr() + self.t.element_size() * self.off

    @property
    def cstride(self):
        return self.t.shape[-1]


def _as_view(x):
    return x if isinstance(x, V) else V(x)


def _fill_src(dst, s):
    """s: V | tensor | dict(view=, scale=, flags=, drop_seed=, drop_p=)"""
    if isinstance(s, dict):
        v = _as_view(s["view"])
        scale = s.get("scale")
        dst.ptr, dst.C, dst.cstride, dst.rp_w = v.ptr, v.C, v.cstride, v.rp
        dst.scale = scale.data_ptr() if scale is not None else None
        dst.flags = s.get("flags", 0)
        dst.drop_seed = s.get("drop_seed", 0)
        dst.drop_p = s.get("drop_p", 0.0)
        ln = s.get("ln")          # (gamma, beta, eps, stats [pixels, 2] or None): LayerNorm on load (SRC_LN is set here)
        if ln is not None:
            g_, b_, eps_, st_ = ln
            dst.flags |= SRC_LN
            dst.ln_gamma, dst.ln_beta, dst.ln_eps = _p(g_).value, _p(b_).value, float(eps_)
            dst.ln_stats = _p(st_).value if st_ is not None else None
        else:
            dst.ln_gamma = dst.ln_beta = dst.ln_stats = None
            dst.ln_eps = 0.0
    else:
        v = _as_view(s)
        dst.ptr, dst.C, dst.cstride, dst.rp_w = v.ptr, v.C, v.cstride, v.rp
        dst.scale, dst.flags, dst.drop_seed, dst.drop_p = None, 0, 0, 0.0
        dst.ln_gamma = dst.ln_beta = dst.ln_stats = None
        dst.ln_eps = 0.0
    return v.C


# ------------------------------------------------------------------------------------------ conv family
def conv_pack_size(ksize, rows, src_channels):
    arr = (C.c_int32 * len(src_channels))(*src_channels)
    return int(load().lmn_conv_pack_size(ksize, rows, len(src_channels), arr))


class PackJob(C.Structure):
    """Mirror of lmn_pack_job_t."""
    _fields_ = [("w", C.c_void_p), ("wpack", C.c_void_p), ("total", C.c_int64), ("first_block", C.c_int64),
                ("ksize", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32), ("nsrc", C.c_int32),
                ("c", C.c_int32 * 3), ("transposed", C.c_int32), ("row_off", C.c_int32), ("rows", C.c_int32),
                ("dtype", C.c_int32), ("_pad", C.c_int32)]


class PackPlan:
    """Packed forms of the PERSISTENT weights (nn.Parameter storage) used by one pass (forward or backward).

    The first pass runs every pack as its own launch and records it; later passes re-pack all recorded jobs
    with ONE ``lmn_conv_pack_batch`` launch in ``refresh()`` and ``conv_pack``/``conv_pack_t`` then only look
    their buffer up.  Jobs hold a reference to the weight tensor; if a parameter's storage moved
    (``model.to(...)``) the plan is dropped and re-recorded."""

    def __init__(self):
        self.jobs = {}        # key -> [weight tensor, packed tensor, PackJob fields]
        self.buffers = {}     # key -> plan-owned buffer that several jobs pack slices of
        self.table = None     # device copy of the job array
        self.blocks = 0
        self.fresh = False

    def refresh(self):
        self.fresh = False
        if not self.jobs:
            return
        if any(j[0].data_ptr() != j[2]["w"] for j in self.jobs.values()):
            self.jobs, self.buffers, self.table = {}, {}, None
            return
        if self.table is None:
            arr = (PackJob * len(self.jobs))()
            blk = 0
            for i, (wt, out, f) in enumerate(self.jobs.values()):
                a = arr[i]
                a.w, a.wpack, a.total, a.first_block = f["w"], out.data_ptr(), out.numel(), blk
                a.ksize, a.Cout, a.Cin, a.nsrc = f["ksize"], f["Cout"], f["Cin"], len(f["c"])
                for k, cc in enumerate(f["c"]):
                    a.c[k] = cc
                a.transposed, a.row_off, a.rows = f["transposed"], f["row_off"], f["rows"]
                a.dtype = f.get("dtype", F32)
                blk += (out.numel() + 1023) // 1024
            if C.sizeof(PackJob) != load().lmn_sizeof_pack_job():
                raise RuntimeError("lm_net_amd: lmn_pack_job_t layout differs between hip.py and lmnet_hip.h")
            dev = next(iter(self.jobs.values()))[1].device
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.table, self.blocks = host.to(dev), blk
        _check(load().lmn_conv_pack_batch(C.c_void_p(self.table.data_ptr()), len(self.jobs), _i64(self.blocks), _stream()),
               "conv_pack_batch")
        self.fresh = True

    def buffer(self, key, n, device):
        b = self.buffers.get(key)
        if b is None or b.numel() != n or b.device != device:
            b = self.buffers[key] = torch.empty(n, device=device, dtype=torch.float32)
        return b

    def lookup(self, key):
        j = self.jobs.get(key) if self.fresh else None
        return j[1] if j is not None else None

    def record(self, key, w, out, fields):
        self.jobs[key] = [w, out, fields]
        self.table = None     # rebuilt by the next refresh()


_PLAN = [None]                # the plan of the pass in flight (set by engine.begin_pass)


def _planned(w, out, persistent):
    return _PLAN[0] is not None and (out is None or persistent) and isinstance(w, torch.nn.Parameter)


def conv_pack(w, ksize, src_channels, out=None, persistent=False):
    """Pack a torch-layout weight [Cout, Cin(,k,k)] for the forward operator.  ``persistent``: ``out`` is a
    slice of a ``PackPlan.buffer`` (so the job may be recorded and replayed by the plan)."""
    cout, cin = w.shape[0], w.shape[1]
    planned = _planned(w, out, persistent)
    dt = _MMA[0]
    if planned:
        key = (w.data_ptr(), ksize, tuple(src_channels), 0, 0, 0, out.data_ptr() if out is not None else 0, dt)
        hit = _PLAN[0].lookup(key)
        if hit is not None:
            return hit
    n = conv_pack_size(ksize, cout, src_channels)      # floats of the fp32 form (the bf16 form fills the first half)
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    arr = (C.c_int32 * len(src_channels))(*src_channels)
    _check(load().lmn_conv_pack(_p(w), _p(out), ksize, cout, cin, len(src_channels), arr, 0, 0, 0, dt, _stream()), "conv_pack")
    if planned:
        _PLAN[0].record(key, w, out, dict(w=w.data_ptr(), ksize=ksize, Cout=cout, Cin=cin, c=list(src_channels),
                                          transposed=0, row_off=0, rows=0, dtype=dt))
    return out


def conv_pack_t(w, ksize, row_off=0, rows=None, out=None, cred=None, persistent=False):
    """Pack the data-gradient operator of a forward weight [Cout, Cin(,k,k)]: rows = input channels
    [row_off, row_off+rows), reduction over Cout.  `rows` may exceed the weight's Cin and `cred` (channels of the dy
    operand) its Cout by the zero padding to a multiple of 4 (RGB input as NHWC4, 2-class head on 4 rows)."""
    cout, cin = w.shape[0], w.shape[1]
    rows = cin - row_off if rows is None else rows
    cred = cout if cred is None else cred
    planned = _planned(w, out, persistent)
    dt = _MMA[0]
    if planned:
        key = (w.data_ptr(), ksize, (cred,), 1, row_off, rows, out.data_ptr() if out is not None else 0, dt)
        hit = _PLAN[0].lookup(key)
        if hit is not None:
            return hit
    n = conv_pack_size(ksize, rows, [cred])
    if out is None:
        out = torch.empty(n, device=w.device, dtype=torch.float32)
    arr = (C.c_int32 * 1)(cred)
    _check(load().lmn_conv_pack(_p(w), _p(out), ksize, cout, cin, 1, arr, 1, row_off, rows, dt, _stream()), "conv_pack_t")
    if planned:
        _PLAN[0].record(key, w, out, dict(w=w.data_ptr(), ksize=ksize, Cout=cout, Cin=cin, c=[cred],
                                          transposed=1, row_off=row_off, rows=rows, dtype=dt))
    return out


def conv_fwd(srcs, wpack, out, *, B, Hin, Win, Hout, Wout, Cout, ksize=1, stride=1, transposed=0, bias=None, bias2=None,
             epilogue=EP_LINEAR, act=ACT_NONE, p=(), aux=None, residual=None, stats=None, stats_mode=STATS_NONE,
             drop_p=0.0, drop_seed=0, stats_rep=1, stats_snap=False, fin=None, chain=None, query_chain=False):
    """chain: dict(wpack=, Cout=, out=, bias=None, shift=None, aux=None, stats=None, epilogue=EP_LINEAR, stats_mode=STATS_NONE, stats_rep=1,
    stats_snap=False) -- a second 1x1 conv on the output tile (lmn_conv_chain_t); query_chain=True: nothing is launched, returns whether
    the library takes the call with its chain (lmn_conv_chain_ok).
    fin: dict(mode=FIN_BN | FIN_BN_BWD, sums=[nrep(+1), 2, C] tensor, nrep, count, ...) -- in-kernel BatchNorm bookkeeping
    (lmn_bn_fin_t): tensors for gamma / beta / about / mean / rstd / A / shift / rmean / rvar / Ain / dgamma / dbeta."""
    a = ConvArgs()
    a.B, a.Hout, a.Wout, a.Hin, a.Win = B, Hout, Wout, Hin, Win
    a.ksize, a.stride, a.transposed, a.nsrc, a.Cout = ksize, stride, transposed, len(srcs), Cout
    for i, s in enumerate(srcs):
        _fill_src(a.src[i], s)
    a.wpack = wpack.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.bias2 = bias2.data_ptr() if bias2 is not None else None
    ps = [t.data_ptr() if t is not None else None for t in p] + [None] * (7 - len(p))
    a.p0, a.p1, a.p2, a.p3, a.p4, a.p5, a.p6 = ps
    if aux is not None:
        v = _as_view(aux)
        a.aux, a.aux_cstride, a.aux_rp_w = v.ptr, v.cstride, v.rp
    if residual is not None:
        v = _as_view(residual)
        assert not v.rp, "conv_fwd: a row-planar residual is not supported"
        a.residual, a.res_cstride = v.ptr, v.cstride
    if out is not None:
        v = _as_view(out)
        a.out, a.out_cstride, a.out_rp_w = v.ptr, v.cstride, v.rp
    a.stats = stats.data_ptr() if stats is not None else None
    a.stats_rep = stats_rep
    a.stats_snap = int(bool(stats_snap))
    if fin is not None:
        _fill_fin(a.fin, fin)
    a.epilogue, a.act, a.stats_mode = epilogue, act, stats_mode
    a.drop_p, a.drop_seed = drop_p, drop_seed
    a.mma_dtype = _MMA[0]
    a.act_dtype = _dt(*[(s["view"] if isinstance(s, dict) else s) for s in srcs], aux, residual, out)
    a.seed_ctr = _SEED_CTR[0].data_ptr() if _SEED_CTR[0] is not None else None
    if chain is not None:
        c = a.chain
        c.wpack = chain["wpack"].data_ptr()
        c.bias = chain["bias"].data_ptr() if chain.get("bias") is not None else None
        c.shift = chain["shift"].data_ptr() if chain.get("shift") is not None else None
        c.stats = chain["stats"].data_ptr() if chain.get("stats") is not None else None
        vo = _as_view(chain["out"])
        c.out, c.out_cstride, c.out_rp_w, c.Cout = vo.ptr, vo.cstride, vo.rp, chain["Cout"]
        if chain.get("aux") is not None:
            va = _as_view(chain["aux"])
            c.aux, c.aux_cstride, c.aux_rp_w = va.ptr, va.cstride, va.rp
        c.epilogue, c.stats_mode = chain.get("epilogue", EP_LINEAR), chain.get("stats_mode", STATS_NONE)
        c.stats_rep, c.stats_snap = chain.get("stats_rep", 1), int(bool(chain.get("stats_snap", False)))
    if query_chain:
        return bool(load().lmn_conv_chain_ok(C.byref(a)))
    _check(load().lmn_conv_fwd(C.byref(a), _stream()), "conv_fwd")


def conv_dma_config(mode=-1, min_tiles=-1):
    """lmn_conv_dma_config: which LDS-DMA kernel forms lmn_conv_fwd may pick (bit 0 conv_dma3, bit 1 conv_dma1, bit 2 conv_dmaM; 0 = the
    LDS-tiled kernels everywhere), smallest call (in 8x16-pixel tiles) that takes the small-channel forms; -1 keeps a value.  Returns the
    previous mode (-1: not set yet, the LMN_CONV_DMA default applies)."""
    return int(load().lmn_conv_dma_config(int(mode), int(min_tiles)))


def conv_wgrad_up2_ok(src, dy, *, B, Hin, Win, Cout):
    """True when conv_wgrad takes `src` (the half-resolution map) with SRC_UP2 for a 3x3 stride-1 weight gradient over the Hin x Win
    upsampled image (lmn_conv_wgrad_up2_ok: the library's own launch predicate, host arithmetic only)."""
    a = WgradArgs()
    a.B, a.Hout, a.Wout, a.Hin, a.Win = B, Hin, Win, Hin, Win
    a.ksize, a.stride, a.nsrc, a.Cout = 3, 1, 1, Cout
    _fill_src(a.src[0], dict(view=src, flags=SRC_UP2))
    v = _as_view(dy)
    a.dy, a.dy_cstride, a.dy_rp_w = v.ptr, v.cstride, v.rp
    a.dW = 0x1000   # (any non-null pointer: the query touches no memory)
    a.mma_dtype = _MMA[0]
    a.act_dtype = _dt(src, dy)
    return bool(load().lmn_conv_wgrad_up2_ok(C.byref(a)))


def conv_wgrad(srcs, dy, dW, db, *, B, Hin, Win, Hout, Wout, Cout, ksize=1, stride=1, dy_flags=0, dy_seed=0, dy_p=0.0,
               dW_src=None, db2=None, defer=False):
    """dW_src: optional list (one entry per source, None = use the columns of dW) of per-source gradient tensors.
    defer: leave the second stage of the K-split reduction to `wgrad_reduce_flush()` (one launch for all deferred calls of the
    launch stream); the call then writes its block partials to a workspace of its own.  Returns that workspace (or None)."""
    a = WgradArgs()
    a.B, a.Hout, a.Wout, a.Hin, a.Win = B, Hout, Wout, Hin, Win
    a.ksize, a.stride, a.nsrc, a.Cout = ksize, stride, len(srcs), Cout
    for i, s in enumerate(srcs):
        _fill_src(a.src[i], s)
    v = _as_view(dy)
    a.dy, a.dy_cstride, a.dy_rp_w = v.ptr, v.cstride, v.rp
    a.dy_flags, a.dy_seed, a.dy_p = dy_flags, dy_seed, dy_p
    a.dW = dW.data_ptr() if dW is not None else None
    a.db = db.data_ptr() if db is not None else None
    a.db2 = db2.data_ptr() if db2 is not None else None
    a.mma_dtype = _MMA[0]
    a.act_dtype = _dt(*[(s["view"] if isinstance(s, dict) else s) for s in srcs], dy)
    if dW_src is not None:
        for i, t in enumerate(dW_src):
            a.dW_src[i] = t.data_ptr() if t is not None else None
    lib = load()
    need = int(lib.lmn_conv_wgrad_workspace(C.byref(a)))
    ws = None
    if need > 0 and defer:
        # the call's own workspace, sized to the partials it will really write (lmn_conv_wgrad_workspace is the upper bound)
        a.workspace, a.workspace_floats = 0x1000, need      # (any non-null pointer: the geometry query does not touch it)
        job = ReduceJob()
        _check(lib.lmn_conv_wgrad_job(C.byref(a), C.byref(job)), "conv_wgrad_job")
        if job.nblk > 0:
            nfl = job.gy * job.nblk * job.per
            ws = (_ALLOC[0] or _default_alloc)(v.t.device, (nfl,))
            a.workspace, a.workspace_floats = ws.data_ptr(), nfl
            a.defer_reduce = 1
            job.partial = ws.data_ptr()
            _RJOBS.setdefault(stream_key(), []).append((job, ws))
        else:
            a.workspace, a.workspace_floats = None, 0
            need = 0
    elif need > 0:
        wsh = _workspace(v.t.device, need)
        a.workspace, a.workspace_floats = wsh.data_ptr(), wsh.numel()
    a.seed_ctr = _SEED_CTR[0].data_ptr() if _SEED_CTR[0] is not None else None
    _check(lib.lmn_conv_wgrad(C.byref(a), _stream()), "conv_wgrad")
    return ws


_RJOBS = {}          # launch stream handle -> [(ReduceJob, workspace tensor)] deferred since the last flush on that stream
_RTABLES = {}        # bytes of a job table -> its device copy (pointers repeat from step to step: no H2D copy per flush)


def wgrad_reduce_pending(key=None):
    key = int(key or 0) if key is not None else stream_key()
    return len(_RJOBS.get(key, ()))


def _reduce_targets(j):
    return {int(v) for v in (j.dW, j.dW_src[0], j.dW_src[1], j.dW_src[2], j.db, j.db2) if v}


def wgrad_reduce_flush():
    """Sum the block partials of every deferred weight gradient of the CURRENT launch stream in one launch.  Returns the
    device job tables (the caller keeps them alive while a recorded plan references them); [] when nothing was pending.
    The batched kernel adds into its destinations with plain read-modify-writes, so two jobs of one launch must not share a
    destination (a weight used twice, a module called twice, accumulation into one explicit dW): such jobs are split into
    consecutive launches, which the stream serialises as the per-layer reduce launches used to."""
    key = stream_key()
    jobs = _RJOBS.pop(key, None)
    if not jobs:
        return []
    groups, seen = [[]], set()
    for job in jobs:
        t = _reduce_targets(job[0])
        if groups[-1] and (t & seen):
            groups.append([])
            seen = set()
        groups[-1].append(job)
        seen |= t
    tabs = []
    for grp in groups:
        arr = (ReduceJob * len(grp))()
        blk = 0
        for i, (j, _) in enumerate(grp):
            j.first_block = blk
            C.memmove(C.byref(arr[i]), C.byref(j), C.sizeof(ReduceJob))
            blk += j.gy * j.blocks_per_set
        raw = bytes(arr)
        dev = grp[0][1].device
        tab = _RTABLES.get((dev, raw))
        if tab is None:
            if len(_RTABLES) > 4096:
                _RTABLES.clear()
            tab = _RTABLES[(dev, raw)] = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        _check(load().lmn_wgrad_reduce_batch(C.c_void_p(tab.data_ptr()), len(grp), _i64(blk), _stream()), "wgrad_reduce_batch")
        tabs.append(tab)
    return tabs


def wgrad_reduce_drop():
    """Forget deferred jobs (a pass that raised)."""
    _RJOBS.clear()


_ws_cache = {}


def _workspace(device, nfloats):
    """Scratch buffer of the weight-gradient K-split reduction, one per (device, launch stream): users of one
    stream reuse it in stream order.  Allocated once at the library's cap (lmn_conv_wgrad_workspace never asks for
    more than 16 M floats), so it is never re-allocated under kernels still in flight on another stream."""
    key = (device, stream_key())
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nfloats:
        ws = torch.empty(max(nfloats, 16 << 20), device=device, dtype=torch.float32)
        _ws_cache[key] = ws
    return ws


# ------------------------------------------------------------------------------------------ depthwise block
def dw_stats(x1, w5, w3, wv, wh, stats, zpre=None):
    """zpre: the tensor is z (z-path, see _dw_pre); with zpre['fin'] the expand conv's BatchNorm is finalised here."""
    B, H, W, E = x1.shape
    _rp_req("dw_stats", x1=x1)
    _check(load().lmn_dw_stats(_pa(x1), B, H, W, E, _p(w5), _p(w3), _p(wv), _p(wh), _p(stats), _dw_pre(zpre), _dt(x1), _stream()), "dw_stats")


def dw_merge(w5, w3, wv, wh, A, shift, keff, beff):
    _check(load().lmn_dw_merge(_p(w5), _p(w3), _p(wv), _p(wh), _p(A), _p(shift), _p(keff), _p(beff), w5.shape[0],
                               _stream()), "dw_merge")


def dw_fwd(x1, pre, gsum, keff, beff, se=None, zpre=None):
    """se: squeeze-excite gate formed inside the pass (see _se_fuse), or None; zpre: the tensor is z (see _dw_pre)."""
    B, H, W, E = x1.shape
    _rp_req("dw_fwd", x1=x1, pre=pre)
    _check(load().lmn_dw_fwd(_pa(x1), _pa(pre), _p(gsum), B, H, W, E, _p(keff), _p(beff), _se_fuse(se), _dw_pre(zpre), _dt(x1, pre), _stream()), "dw_fwd")


def dw_finalize_merge(stats, count, bns, ws, mean, rstd, A, keff, beff):
    """bns: the four BatchNorm2d modules (branch order 5x5, 3x3, 3x1, 1x3); ws: the four depthwise weights."""
    P4 = C.c_void_p * 4
    F4 = C.c_float * 4
    E = mean.shape[-1]
    _check(load().lmn_dw_finalize_merge(
        _p(stats), _f(count), P4(*[b.weight.data_ptr() for b in bns]), P4(*[b.bias.data_ptr() for b in bns]),
        P4(*[b.running_mean.data_ptr() for b in bns]), P4(*[b.running_var.data_ptr() for b in bns]),
        F4(*[b.eps for b in bns]), F4(*[(b.momentum if b.momentum is not None else 0.1) for b in bns]),
        _p(ws[0]), _p(ws[1]), _p(ws[2]), _p(ws[3]), _p(mean), _p(rstd), _p(A), _p(keff), _p(beff), E, _stream()),
        "dw_finalize_merge")


def dw_fwd_bn(x1, pre, gsum, stats, count, bns, ws, mean, rstd, A, se=None, zpre=None):
    """dw_finalize_merge + dw_fwd in one launch (training): bns / ws as in dw_finalize_merge."""
    P4 = C.c_void_p * 4
    F4 = C.c_float * 4
    B, H, W, E = x1.shape
    _rp_req("dw_fwd_bn", x1=x1, pre=pre)
    _check(load().lmn_dw_fwd_bn(
        _pa(x1), _pa(pre), _p(gsum), B, H, W, E, _p(stats), _f(count), P4(*[b.weight.data_ptr() for b in bns]),
        P4(*[b.bias.data_ptr() for b in bns]), P4(*[b.running_mean.data_ptr() for b in bns]),
        P4(*[b.running_var.data_ptr() for b in bns]), F4(*[b.eps for b in bns]),
        F4(*[(b.momentum if b.momentum is not None else 0.1) for b in bns]), _p(ws[0]), _p(ws[1]), _p(ws[2]), _p(ws[3]),
        _p(mean), _p(rstd), _p(A), _se_fuse(se), _dw_pre(zpre), _dt(x1, pre), _stream()), "dw_fwd_bn")


def dw_bwd_bn(x1, dpre, dx1, w5, w3, wv, wh, bstats, mean, rstd, A, count, batch_stats, dgs, dbs, dw5, dw3, dwv, dwh, part=0,
              zpre=None, hstats=None):
    """dw_bwd_coef + dw_bwd in one launch (part 1: dx1 + gamma / beta gradients only, part 2: the weight gradients only).
    zpre / hstats: z-path -- x1 is z, `dx1` receives dh = dx1 * Hardswish'(A z + shift), hstats [2][E] += (sum dh, sum dh z)."""
    P4 = C.c_void_p * 4
    B, H, W, E = x1.shape
    _rp_req("dw_bwd_bn", x1=x1, dpre=dpre, dx1=dx1)
    _check(load().lmn_dw_bwd_bn(_pa(x1), _pa(dpre), _pa(dx1), B, H, W, E, _p(w5), _p(w3), _p(wv), _p(wh), _p(bstats), _p(mean),
                                _p(rstd), _p(A), _f(count), int(batch_stats), P4(*[t.data_ptr() for t in dgs]),
                                P4(*[t.data_ptr() for t in dbs]), _p(dw5), _p(dw3), _p(dwv), _p(dwh), part, _dw_pre(zpre),
                                _p(hstats), _dt(x1, dpre, dx1), _stream()), "dw_bwd_bn")


def dw_bwd_stats(x1, pre, u, s, dm, dpre, w5, w3, wv, wh, bstats, seb=None, zpre=None):
    """seb: dict(ds, fc1w, fc2w, hidden, dvec, inv_hw) -- the squeeze-excite backward formed inside the pass (dm may be None)."""
    B, H, W, E = x1.shape
    sb = None
    if seb is not None:
        f = SeBwd()
        f.ds, f.w1, f.w2, f.hidden, f.dvec = (_p(seb[k]).value for k in ("ds", "fc1w", "fc2w", "hidden", "dvec"))
        f.inv_hw, f.R = float(seb["inv_hw"]), seb["fc1w"].shape[0]
        sb = C.byref(f)
    _rp_req("dw_bwd_stats", x1=x1, pre=pre, u=u, dpre=dpre)
    _check(load().lmn_dw_bwd_stats(_pa(x1), _pa(pre), _pa(u), _p(s), _p(dm), _pa(dpre), B, H, W, E, _p(w5), _p(w3), _p(wv),
                                   _p(wh), _p(bstats), sb, _dw_pre(zpre), _dt(x1, pre, u, dpre), _stream()), "dw_bwd_stats")


def dw_bwd_coef(bstats, mean, rstd, A, count, batch_stats, cA, cC, cD, dgs, dbs):
    E = A.shape[-1]
    _check(load().lmn_dw_bwd_coef(_p(bstats), _p(mean), _p(rstd), _p(A), _f(count), int(batch_stats), _p(cA), _p(cC),
                                  _p(cD), *[_p(t) for t in dgs], *[_p(t) for t in dbs], E, _stream()), "dw_bwd_coef")


def dw_bwd(x1, dpre, dx1, w5, w3, wv, wh, cA, cC, cD, dw5, dw3, dwv, dwh):
    B, H, W, E = x1.shape
    _rp_req("dw_bwd", x1=x1, dpre=dpre, dx1=dx1)
    _check(load().lmn_dw_bwd(_pa(x1), _pa(dpre), _pa(dx1), B, H, W, E, _p(w5), _p(w3), _p(wv), _p(wh), _p(cA), _p(cC),
                             _p(cD), _p(dw5), _p(dw3), _p(dwv), _p(dwh), _dt(x1, dpre, dx1), _stream()), "dw_bwd")


def se_fwd(gsum, inv_hw, w1, b1, w2, b2, s, hidden):
    B, E = gsum.shape
    _check(load().lmn_se_fwd(_p(gsum), _f(inv_hw), _p(w1), _p(b1), _p(w2), _p(b2), _p(s), _p(hidden), B, E, w1.shape[0],
                             _stream()), "se_fwd")


def se_bwd(ds, gsum, inv_hw, w1, b1, w2, b2, hidden, dm, dw1, db1, dw2, db2):
    B, E = gsum.shape
    _check(load().lmn_se_bwd(_p(ds), _p(gsum), _f(inv_hw), _p(w1), _p(b1), _p(w2), _p(b2), _p(hidden), _p(dm), _p(dw1),
                             _p(db1), _p(dw2), _p(db2), B, E, w1.shape[0], _stream()), "se_bwd")


def se_bwd_dm(ds, s, inv_hw, w1, w2, hidden, dm, dvec):
    B, E = s.shape
    _check(load().lmn_se_bwd_dm(_p(ds), _p(s), _f(inv_hw), _p(w1), _p(w2), _p(hidden), _p(dm), _p(dvec), B, E, w1.shape[0],
                                _stream()), "se_bwd_dm")


def se_bwd_params(dvec, gsum, inv_hw, hidden, dw1, db1, dw2, db2):
    B, E = gsum.shape
    _check(load().lmn_se_bwd_params(_p(dvec), _p(gsum), _f(inv_hw), _p(hidden), _p(dw1), _p(db1), _p(dw2), _p(db2), B, E,
                                    dw1.shape[0], _stream()), "se_bwd_params")


# ------------------------------------------------------------------------------------------ attention
def _na_k(rpb, heads, direct):
    K = (rpb.shape[-1] + 1) // 2          # rpb: [heads][2K-1][2K-1]
    assert tuple(rpb.shape) == (heads, 2 * K - 1, 2 * K - 1), tuple(rpb.shape)
    return -K if direct else K            # negative: the direct (run-time K) kernels also at K = 3


def na_fwd(qkv, rpb, out, heads, direct=False):
    B, H, W, C3 = qkv.shape
    hd = C3 // 3 // heads
    _check(load().lmn_na_fwd(_pa(qkv), _p(rpb), _pa(out), B, H, W, heads, hd, _na_k(rpb, heads, direct), _f(hd ** -0.5),
                             _dt(qkv, out), _stream()), "na_fwd")


def na_bwd(qkv, rpb, dout, dqkv, drpb, heads, stat=None, direct=False):
    B, H, W, C3 = qkv.shape
    hd = C3 // 3 // heads
    if stat is None:
        stat = (_ALLOC[0] or _default_alloc)(qkv.device, (B * H * W * 2 * heads,))
    _check(load().lmn_na_bwd(_pa(qkv), _p(rpb), _pa(dout), _pa(dqkv), _p(drpb), _p(stat), B, H, W, heads, hd,
                             _na_k(rpb, heads, direct), _f(hd ** -0.5), _dt(qkv, dout, dqkv), _stream()), "na_bwd")


def gattn_fwd(qkv, out, lse, heads):
    B, N, C3 = qkv.shape
    hd = C3 // 3 // heads
    _check(load().lmn_gattn_fwd(_pa(qkv), _pa(out), _p(lse), B, N, heads, hd, _f(hd ** -0.5), _dt(qkv, out), _stream()), "gattn_fwd")


def gattn_bwd(qkv, out, dout, lse, dqkv, delta, heads):
    B, N, C3 = qkv.shape
    hd = C3 // 3 // heads
    _check(load().lmn_gattn_bwd(_pa(qkv), _pa(out), _pa(dout), _p(lse), _pa(dqkv), _p(delta), B, N, heads, hd,
                                _f(hd ** -0.5), _dt(qkv, out, dout, dqkv), _stream()), "gattn_bwd")


# ------------------------------------------------------------------------------------------ norms
def ln_fwd(x, gamma, beta, y):
    Cn = x.shape[-1]
    _check(load().lmn_ln_fwd(_pa(x), _p(gamma), _p(beta), _pa(y), _i64(x.numel() // Cn), Cn, _dt(x, y), _stream()), "ln_fwd")


def ln_bwd(x, gamma, dy, dres, dx, dgamma, dbeta):
    Cn = x.shape[-1]
    _check(load().lmn_ln_bwd(_pa(x), _p(gamma), _pa(dy), _pa(dres), _pa(dx), _p(dgamma), _p(dbeta), _i64(x.numel() // Cn),
                             Cn, _dt(x, dy, dres, dx), _stream()), "ln_bwd")


def bnact_fwd(z, a, b, y, act):
    Cn = z.shape[-1]
    _check(load().lmn_bnact_fwd(_pa(z), _p(a), _p(b), _pa(y), _i64(z.numel() // Cn), Cn, act, _dt(z, y), _stream()), "bnact_fwd")


def bnact_fwd_fin(z, fin, y, act):
    """lmn_bnact_fwd_fin: y = act(A z + shift) with A / shift formed in the kernel from the batch sums (fin: dict as for conv_fwd)."""
    Cn = z.shape[-1]
    f = BnFin()
    _fill_fin(f, fin)
    _check(load().lmn_bnact_fwd_fin(_pa(z), C.byref(f), _pa(y), _i64(z.numel() // Cn), Cn, act, _dt(z, y), _stream()), "bnact_fwd_fin")


def bnact_bwd_fin(z, dy, mean, rstd, gamma, beta, fin, dz, act):
    """lmn_bnact_bwd_fin: dz with c1 / c2 / c3 formed in the kernel from the sums of bnact_bwd_stats (fin: mode FIN_BN_BWD)."""
    Cn = z.shape[-1]
    f = BnFin()
    _fill_fin(f, fin)
    _check(load().lmn_bnact_bwd_fin(_pa(z), _pa(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), C.byref(f), _pa(dz),
                                    _i64(z.numel() // Cn), Cn, act, _dt(z, dy, dz), _stream()), "bnact_bwd_fin")


def bnact_bwd_stats(z, dy, mean, rstd, gamma, beta, stats, act):
    Cn = z.shape[-1]
    _check(load().lmn_bnact_bwd_stats(_pa(z), _pa(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(stats),
                                      _i64(z.numel() // Cn), Cn, act, _dt(z, dy), _stream()), "bnact_bwd_stats")


def bnact_bwd(z, dy, mean, rstd, gamma, beta, c1, c2, c3, dz, act):
    Cn = z.shape[-1]
    _check(load().lmn_bnact_bwd(_pa(z), _pa(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(c1), _p(c2), _p(c3), _pa(dz),
                                _i64(z.numel() // Cn), Cn, act, _dt(z, dy, dz), _stream()), "bnact_bwd")


def bn_finalize(sums, count, gamma, beta, eps, momentum, mean, rstd, A, shift, running_mean, running_var, about=None):
    """about: the per-channel value the sums were taken about (conv_fwd(..., p=(None,)*4 + (about,))), or None."""
    nrep = sums.shape[0] if sums.dim() == 3 else 1      # [nrep][2][C] slices (conv_fwd(stats_rep=nrep)) or [2][C]
    _check(load().lmn_bn_finalize(_p(sums), nrep, _f(count), _p(gamma), _p(beta), _f(eps), _f(momentum), _p(mean), _p(rstd),
                                  _p(A), _p(shift), _p(running_mean), _p(running_var), _p(about), gamma.numel(), _stream()),
           "bn_finalize")


def bn_fold(running_mean, running_var, gamma, beta, eps, mean, rstd, A, shift):
    _check(load().lmn_bn_fold(_p(running_mean), _p(running_var), _p(gamma), _p(beta), _f(eps), _p(mean), _p(rstd), _p(A),
                              _p(shift), gamma.numel(), _stream()), "bn_fold")


def bn_bwd_coef(bstats, count, A, dgamma, dbeta, c1, c2, c3, batch_stats=True):
    nrep = bstats.shape[0] if bstats.dim() == 3 else 1
    _check(load().lmn_bn_bwd_coef(_p(bstats), nrep, _f(count), int(batch_stats), _p(A), _p(dgamma), _p(dbeta), _p(c1), _p(c2),
                                  _p(c3), A.numel(), _stream()), "bn_bwd_coef")


# ------------------------------------------------------------------------------------------ resampling / layout / utils
def up2_fwd(x, y):
    x, y = _as_view(x), _as_view(y)
    B, Hin, Win = x.t.shape[:3]
    _check(load().lmn_up2_fwd(C.c_void_p(x.ptr), C.c_void_p(y.ptr), B, Hin, Win, x.C, x.cstride, y.cstride, _dt(x, y),
                              _stream()), "up2_fwd")


def up2_bwd(dy, dx):
    dy, dx = _as_view(dy), _as_view(dx)
    B, Hin, Win = dx.t.shape[:3]
    _check(load().lmn_up2_bwd(C.c_void_p(dy.ptr), C.c_void_p(dx.ptr), B, Hin, Win, dx.C, dy.cstride, dx.cstride,
                              _dt(dy, dx), _stream()), "up2_bwd")


def avgpool_fwd(x, y, f):
    x, y = _as_view(x), _as_view(y)
    B, Hout, Wout = y.t.shape[:3]
    _check(load().lmn_avgpool_fwd(C.c_void_p(x.ptr), C.c_void_p(y.ptr), B, Hout, Wout, f, x.C, x.cstride, y.cstride,
                                  _dt(x, y), _stream()), "avgpool_fwd")


def avgpool_bwd(dy, dx, f, accumulate):
    dy, dx = _as_view(dy), _as_view(dx)
    B, Hout, Wout = dy.t.shape[:3]
    _check(load().lmn_avgpool_bwd(C.c_void_p(dy.ptr), C.c_void_p(dx.ptr), B, Hout, Wout, f, dx.C, dy.cstride,
                                  dx.cstride, int(accumulate), _dt(dy, dx), _stream()), "avgpool_bwd")


def nchw_to_nhwc(x, y):
    B, Cn, H, W = x.shape
    _check(load().lmn_nchw_to_nhwc(_p(x), _pa(y), B, Cn, H, W, y.shape[-1], _dt(y), _stream()), "nchw_to_nhwc")


def nhwc_to_nchw(x, y):
    B, Cn, H, W = y.shape
    _check(load().lmn_nhwc_to_nchw(_pa(x), _p(y), B, Cn, H, W, x.shape[-1], _dt(x), _stream()), "nhwc_to_nchw")


def _pl(t):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.int64 or not t.is_contiguous():
        raise RuntimeError("lm_net_amd: contiguous int64 device tensor required for labels")
    return C.c_void_p(t.data_ptr())


def segloss_fwd(logits, target, w_ce, w_dice, label_smoothing, smooth, sums, coef, loss):
    B, Cn = logits.shape[0], logits.shape[1]
    hw = logits.numel() // (B * Cn)
    _check(load().lmn_segloss_fwd(_p(logits), _pl(target), _p(w_ce), _p(w_dice), B, Cn, _i64(hw), _f(label_smoothing),
                                  _f(smooth), _p(sums), _p(coef), _p(loss), _stream()), "segloss_fwd")


def segloss_bwd(logits, target, w_ce, coef, gscale, dlogits):
    B, Cn = logits.shape[0], logits.shape[1]
    hw = logits.numel() // (B * Cn)
    _check(load().lmn_segloss_bwd(_p(logits), _pl(target), _p(w_ce), _p(coef), _p(gscale), B, Cn, _i64(hw), _p(dlogits),
                                  _stream()), "segloss_bwd")


def confusion(logits, target, counts):
    B, Cn = logits.shape[0], logits.shape[1]
    hw = logits.numel() // (B * Cn)
    _check(load().lmn_confusion(_p(logits), _pl(target), B, Cn, _i64(hw), _p(counts), _stream()), "confusion")


def preprocess_u8(images, masks, flips, out, labels, mean, std):
    """uint8 HWC images [B,Hs,Ws,3] / masks [B,Hs,Ws] -> fp32 NCHW `out` [B,3,H,W] / int64 `labels` [B,H,W]."""
    def raw(t, dt):
        if t is None:
            return None
        if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
            raise RuntimeError("lm_net_amd.preprocess_u8: contiguous %s device tensor required" % dt)
        return C.c_void_p(t.data_ptr())
    ref = images if images is not None else masks
    B, Hs, Ws = ref.shape[0], ref.shape[1], ref.shape[2]
    dst = out if out is not None else labels
    H, W = dst.shape[-2], dst.shape[-1]
    m3, s3 = (C.c_double * 3)(*[float(v) for v in mean]), (C.c_double * 3)(*[float(v) for v in std])
    _check(load().lmn_preprocess_u8(raw(images, torch.uint8), raw(masks, torch.uint8), raw(flips, torch.uint8), B, Hs, Ws,
                                    H, W, m3, s3, _p(out), raw(labels, torch.int64), _stream()), "preprocess_u8")


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2):
    _check(load().lmn_adamw_step(_p(p), _p(g), _p(m), _p(v), _i64(p.numel()), _f(lr), _f(beta1), _f(beta2), _f(eps),
                                 _f(weight_decay), _f(bias_corr1), _f(bias_corr2), _stream()), "adamw_step")


def reparam_fold(hstats, mean, rstd, A, count, batch_stats, w_expand, b_expand, w_shortcut, rows, cred, wpack, kbias, coef, dgamma,
                 dbeta):
    """lmn_reparam_fold: BatchNorm backward of the expand conv folded into the packed operators of ONE three-source conv."""
    E, cinw = w_expand.shape[0], w_expand.shape[1]
    _check(load().lmn_reparam_fold(_p(hstats), _p(mean), _p(rstd), _p(A), _f(count), int(batch_stats), _p(w_expand), _p(b_expand),
                                   _p(w_shortcut), E, rows, cinw, cred, w_shortcut.shape[0], _p(wpack), _p(kbias), _p(coef),
                                   _p(dgamma), _p(dbeta), _MMA[0], _stream()), "reparam_fold")


class SeParamsT(C.Structure):
    _fields_ = [("dvec", C.c_void_p), ("gsum", C.c_void_p), ("hidden", C.c_void_p), ("dw1", C.c_void_p), ("db1", C.c_void_p),
                ("dw2", C.c_void_p), ("db2", C.c_void_p), ("inv_hw", C.c_float), ("B", C.c_int32), ("E", C.c_int32), ("R", C.c_int32)]


def reparam_wfin(R, M, m, coef, hstats, w_expand, b_expand, count, dW, db, se=None):
    """lmn_reparam_wfin: expand-conv weight / bias gradient from the raw gradient R = sum dh x^T and the moments of x.
    se = dict(dvec, gsum, inv_hw, hidden, dw1, db1, dw2, db2): the squeeze-excite parameter gradients of the same block in the same launch."""
    E, cinw = w_expand.shape[0], w_expand.shape[1]
    sp = None
    if se is not None:
        Bn, En = se["gsum"].shape
        sp = SeParamsT(*[_p(se[k]).value for k in ("dvec", "gsum", "hidden", "dw1", "db1", "dw2", "db2")],
                       float(se["inv_hw"]), Bn, En, se["dw1"].shape[0])
    _check(load().lmn_reparam_wfin(_p(R), _p(M), _p(m), _p(coef), _p(hstats), _p(w_expand), _p(b_expand), _f(count), E, R.shape[1],
                                   cinw, _p(dW), _p(db), C.byref(sp) if sp is not None else None, _stream()), "reparam_wfin")


def affine2(u, v, coef, y):
    """y = coef[0] * u + coef[1] * v + coef[2] per channel (activation tensors of shape [..., C])."""
    Cn = u.shape[-1]
    rp = is_rp4(u)
    assert rp == is_rp4(v) == is_rp4(y), "affine2: operands of one layout"
    _check(load().lmn_affine2(_pa(u), _pa(v), _p(coef), _pa(y), _i64(u.numel() // Cn), Cn, u.shape[2] if rp else 0, _dt(u, v, y), _stream()), "affine2")


def fill(t, v):
    _check(load().lmn_fill(_p(t), _f(v), _i64(t.numel()), _stream()), "fill")


def add(a, b, c=None, d=None, out=None):
    out = a if out is None else out
    _check(load().lmn_add(_pa(a), _pa(b), _pa(c), _pa(d), _pa(out), _i64(a.numel()), _dt(a, b, c, d, out), _stream()), "add")
    return out


def colsum(x, out):
    x = _as_view(x)
    rows = x.t.numel() // x.cstride
    _check(load().lmn_colsum(C.c_void_p(x.ptr), _p(out), _i64(rows), x.C, x.cstride, _dt(x), _stream()), "colsum")


def copy_slice(x, y):
    x, y = _as_view(x), _as_view(y)
    rows = x.t.numel() // x.cstride
    _check(load().lmn_copy_slice(C.c_void_p(x.ptr), C.c_void_p(y.ptr), _i64(rows), x.C, x.cstride, y.cstride, _dt(x, y),
                                 _stream()), "copy_slice")


def copy2d(x, y, rows, cols, x_stride, y_stride):
    """y[r][0:cols] = x[r][0:cols] (row strides in floats); any cols."""
    _check(load().lmn_copy2d(_p(x), _p(y), _i64(rows), cols, x_stride, y_stride, _stream()), "copy2d")


# ------------------------------------------------------------------------------------------ streams, plans, kernel timer
def stream_wait(waiter, waited):
    """`waiter` (torch stream) waits for everything enqueued on `waited` so far; recorded when a plan is recording."""
    if waiter.cuda_stream == waited.cuda_stream:
        return
    _check(load().lmn_stream_wait(C.c_void_p(waiter.cuda_stream), C.c_void_p(waited.cuda_stream)), "stream_wait")


def event_record(slot, stream):
    _check(load().lmn_event_record(slot, C.c_void_p(stream.cuda_stream)), "event_record")


def event_wait(slot, stream):
    _check(load().lmn_event_wait(slot, C.c_void_p(stream.cuda_stream)), "event_wait")


_PRIO_STREAMS = {}


def set_priority_stream(stream, level=3):
    """Wave priority 0..3 of the kernels launched on a torch stream (include/lmnet_hip.h, lmn_set_priority_stream).  Process-wide."""
    key = stream.cuda_stream
    if _PRIO_STREAMS.get(key, 0) == level:
        return
    _check(load().lmn_set_priority_stream(C.c_void_p(key), level), "set_priority_stream")
    if len(_PRIO_STREAMS) >= 8:      # (the library's table holds 8 streams and replaces the oldest: forget what this side believes is registered)
        _PRIO_STREAMS.clear()
    _PRIO_STREAMS[key] = level


class Plan:
    """A recorded pass (see include/lmnet_hip.h, lmn_plan_*): every C-ABI entry issued between record_begin() and
    record_end() is remembered with its arguments; run() re-issues them in one FFI crossing."""

    def __init__(self):
        self.h = C.c_void_p(load().lmn_plan_create())
        self.marks = []            # op counts at the segment boundaries reported during recording

    def record_begin(self):
        _check(load().lmn_plan_record_begin(self.h), "plan_record_begin")

    def record_end(self, seal=True):
        return int(load().lmn_plan_record_end(self.h, 1 if seal else 0))

    def size(self):
        return int(load().lmn_plan_size(self.h))

    def mark(self):
        self.marks.append(self.size())

    def run(self, lo=0, hi=-1):
        _check(load().lmn_plan_run(self.h, _i64(lo), _i64(hi)), "plan_run")

    def host_profile(self):
        """Runs the plan once; -> {entry point: (ops, host microseconds)}"""
        lib = load()
        lib.lmn_plan_host_profile.restype = C.c_int64
        buf = C.create_string_buffer(1 << 16)
        n = int(lib.lmn_plan_host_profile(self.h, buf, _i64(1 << 16)))
        if n < 0:
            raise RuntimeError("plan_host_profile failed: " + last_error())
        return {a: (int(b), float(c)) for a, b, c in (ln.split("\t") for ln in buf.value.decode().splitlines())}

    def __del__(self):
        try:
            if self.h is not None and _lib is not None:
                _lib.lmn_plan_destroy(self.h)
        except Exception:
            pass
        self.h = None


def set_deterministic(on):
    """Fixed-order reductions in every kernel (include/lmnet_hip.h, lmn_set_deterministic): bit-reproducible steps, slower."""
    _check(load().lmn_set_deterministic(1 if on else 0), "set_deterministic")


def get_deterministic():
    return bool(load().lmn_get_deterministic())


def prof_begin(filter_=None):
    """Time (HIP events on the launch stream) every kernel launch whose name contains one of the '|'-separated substrings."""
    _check(load().lmn_prof_begin(filter_.encode() if filter_ else None), "prof_begin")


def prof_end():
    """-> {kernel name: dict(launches, total_us, flops, bytes)} since prof_begin (synchronises the device)."""
    lib = load()
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        need = int(lib.lmn_prof_end(buf, _i64(cap)))
        if need <= cap:
            break
        cap = need + 16
        # the records are kept until the next prof_begin: a second call re-reads them
    out = {}
    for line in buf.value.decode().splitlines():
        name, n, us, fl, by = line.split("\t")
        out[name] = dict(launches=int(n), total_us=float(us), flops=float(fl), bytes=float(by))
    return out


def prof_timeline():
    """After prof_begin("!..."): -> [(kernel, stream, start_us, end_us)] per launch, on the device clock of the first launch."""
    lib = load()
    cap = 1 << 20
    while True:
        buf = C.create_string_buffer(cap)
        need = int(lib.lmn_prof_end(buf, _i64(cap)))
        if need <= cap:
            break
        cap = need + 16
    out = []
    for line in buf.value.decode().splitlines():
        name, st, t0, t1, _ = line.split("\t")
        out.append((name.strip("()"), int(st, 16), float(t0), float(t1)))
    return out
