"""Forward/backward schedule of the LM-Net hot path on the HIP kernels (one process, one stream).

The reference runs this graph as ~400 eager ATen/natten calls forward and ~800 backward
(core/LM_Net.py:95-123).  Here the whole network is ONE autograd node whose
forward and backward are explicit kernel schedules over NHWC fp32 activations:

  * every row of SURVEY.md section 8a is a fused HIP kernel (or a short chain) reached through
    ``hip.py`` -> ``liblmnet_hip.so``; torch only allocates buffers and provides the stream;
  * torch.cat along channels never materialises: producers write into channel slices of one buffer;
  * train-mode BatchNorm is two passes (statistics, then apply) with the cheap producer recomputed
    rather than stored; running statistics are updated by the same finalize kernel;
  * the backward pass writes every parameter gradient into one flat fp32 buffer laid out in
    backward-completion order, so data-parallel gradient buckets are contiguous slices that can be
    all-reduced on a side stream as soon as their last kernel has been enqueued (``ddp.py``).
"""
import contextlib
import os

import torch

from . import hip
from .hip import V

EPS_BN = 1e-5


class Ctx:
    """Saved tensors of one forward pass (the hand-written autograd tape)."""

    def __init__(self):
        self.t = {}


class Arena:
    """Bump allocator over ONE fp32 buffer, no reuse inside a step: every tensor of a recorded pass keeps its address for
    the life of the plan (288 GB of HBM: a batch-8 352x352 training step needs a few GB this way), and no buffer is ever
    shared between the streams of the schedule, so replays need no allocator bookkeeping."""

    def __init__(self, nfloats, device):
        self.buf = torch.empty(int(nfloats), device=device, dtype=torch.float32)
        self.off = 0

    def alloc(self, shape, dtype=torch.float32):
        n = 1
        for d in shape:
            n *= int(d)
        nf = n if dtype == torch.float32 else (n + 1) // 2      # floats covering n elements
        n_al = (nf + 63) & ~63                 # 256-byte granules
        if self.off + n_al > self.buf.numel():
            raise RuntimeError("lm_net_amd: plan arena exhausted (%d + %d > %d floats)" % (self.off, n_al, self.buf.numel()))
        v = self.buf[self.off:self.off + nf]
        v = v.view(shape) if dtype == torch.float32 else v.view(dtype)[:n].view(shape)
        self.off += n_al
        return v


_ENG = [None]      # engine of the pass in flight


def _E(ref, *shape):
    """fp32 buffer of `shape` on ref's device (ref: tensor or torch.device) from the allocator of the pass in flight:
    per-channel vectors, statistics, packed weights, workspaces and the fp32 tensors at the module boundary."""
    dev = ref if isinstance(ref, torch.device) else ref.device
    eng = _ENG[0]
    if eng is None:
        return torch.empty(shape, device=dev, dtype=torch.float32)
    return eng.alloc(dev, shape)


def _A(ref, *shape):
    """ACTIVATION tensor of `shape`: stored in the pass's activation type (fp32, or bf16 in the mixed-precision mode)."""
    dev = ref if isinstance(ref, torch.device) else ref.device
    eng = _ENG[0]
    if eng is None:
        return torch.empty(shape, device=dev, dtype=torch.float32)
    return eng.alloc(dev, shape, eng.act_dtype)


def _R(ref, *shape):
    """ROW-PLANAR activation tensor [B, H, W, E] (hip.rp4): the E-wide tensors inside a ReparamConv block (z / x1, pre, u, dpre,
    dh / dx1, dz), written and read by the depthwise kernels and by the 1x1 convs either side of them."""
    return hip.rp4(_A(ref, *shape))


class ZeroPool:
    """All small zero-initialised accumulators (BN sums, SE sums, ...) of one forward or backward pass come
    out of ONE buffer zeroed by ONE memset (their total size is learnt on the first pass)."""

    def __init__(self):
        self.need = 0
        self.buf = None
        self.off = 0
        self.count = 0

    def begin(self, device):
        self.buf = None
        if self.need:
            self.buf = _E(device, self.need)
            hip.fill(self.buf, 0.0)
        self.off = self.count = 0

    def get(self, device, *shape):
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) // 4 * 4
        self.count += n4
        if self.buf is not None and self.off + n4 <= self.buf.numel():
            v = self.buf[self.off:self.off + n].view(shape)
            self.off += n4
            return v
        t = _E(device, *shape)
        hip.fill(t, 0.0)
        return t

    def end(self):
        self.need = max(self.need, self.count)
        self.buf = None


_POOL = [None]


def _Z(ref, *shape):
    pool = _POOL[0]
    if pool is None:
        t = _E(ref, *shape)
        hip.fill(t, 0.0)
        return t
    return pool.get(ref if isinstance(ref, torch.device) else ref.device, *shape)


# slices of the conv statistics buffers: a conv's ~1280 blocks add into slice (block % STATS_REP) instead of all into the
# same two cache lines; lmn_bn_finalize / lmn_bn_bwd_coef sum the slices
STATS_REP = 16


def _new_stream(device, priority, wave_prio=0):
    """A stream of the given HIP priority (-1 high, 0 normal, 1 low).  torch clamps positive priorities to 0, so low-priority streams are
    created through the runtime and wrapped (they live as long as the process).  wave_prio > 0: the library's kernels on this stream
    raise their waves' issue priority (hip.set_priority_stream)."""
    s = _new_stream_raw(device, priority)
    if wave_prio > 0:
        hip.set_priority_stream(s, wave_prio)
    return s


def _new_stream_raw(device, priority):
    if priority <= 0:
        return torch.cuda.Stream(device=device, priority=priority)
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = rt.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(priority))      # 1 = hipStreamNonBlocking
    if rc != 0:
        raise RuntimeError("hipStreamCreateWithPriority(%d) failed: %d" % (priority, rc))
    return torch.cuda.ExternalStream(h.value, device=device)


class Engine:
    _instances = 0

    def __init__(self, model):
        self.m = model
        self.G = None            # param -> gradient view (set per backward)
        self.training = True
        self.seed_base = 0x1234567
        self.step = 0
        self.seed_ctr = None      # int32[1] device tensor while a hipGraph of the pass is captured / replayed
        self.overlap_wgrad = True # weight gradients on a side stream (see wgrad)
        self.capturing = False
        self.sides = {}           # issuing stream handle -> [weight-gradient stream, busy]
        self.branch = None        # second compute stream of the backward schedule (LM_Net._backward_body)
        self.branches = []        # further branch streams (branch_stream_n)
        # chain of level k (0 = 352^2 ... 3 = 44^2) -> branch stream index, forward / backward ("0000": one branch stream, rounds 2-5)
        self.branch_map_f = [int(c) for c in os.environ.get("LMN_BRANCH_MAP_F", "0001")]
        self.branch_map_b = [int(c) for c in os.environ.get("LMN_BRANCH_MAP_B", "0101")]
        self.branch_overlap = True
        # BatchNorm bookkeeping inside the consuming conv (lmn_bn_fin_t) instead of separate launches (LMN_FUSE_BN=0: A/B runs)
        self.fuse_bn = os.environ.get("LMN_FUSE_BN", "1") != "0"
        # depthwise backward in two halves (dx1 here, weight gradients on the side stream; LMN_DW_SPLIT=1)
        self.split_dw = os.environ.get("LMN_DW_SPLIT", "0") == "1"
        self.split_se = os.environ.get("LMN_SE_SPLIT", "1") != "0"   # SE backward in two launches (parameter gradients on the side stream)
        # squeeze-excite gate / its backward formed inside the depthwise passes (no se_fwd / se_bwd_dm launches; LMN_FUSE_SE=0: A/B)
        self.fuse_se = int(os.environ.get("LMN_FUSE_SE", "3"))      # bit 0: forward gate, bit 1: backward
        self._fuse_se0 = self.fuse_se
        self.deterministic = False
        if os.environ.get("LMN_DETERMINISTIC", "0") == "1":
            self.set_deterministic(True)
        # z-path of ReparamConv (include/lmnet_hip.h, lmn_dw_pre_t / lmn_reparam_fold): the expand conv's BatchNorm + Hardswish
        # applied inside the depthwise kernels, its backward folded into the weights of one three-source conv (LMN_ZPATH=0: A/B)
        self.zpath = os.environ.get("LMN_ZPATH", "1") != "0"
        self.chain_on = os.environ.get("LMN_CONV_CHAIN", "1") != "0"   # chained 1x1 convs across the two blocks of a stage (hip.conv_fwd(chain=...))
        self.fwd_join = os.environ.get("LMN_FWD_JOIN", "1") != "0"   # the forward ends with a join of its weight-gradient stream (A/B)
        self.zpath_lds = 65536    # bytes of LDS lmn_reparam_fold may use (35 E floats): wider blocks keep the two-pass BatchNorm form (E > 468)
        # weight gradients of the branch chains issued late, beside the encoder's backward (LMN_LAZY_WGRAD=0: A/B runs)
        self.lazy_wgrad = os.environ.get("LMN_LAZY_WGRAD", "1") != "0"
        self.lazy_on = False
        self.lazy_q = []
        self.side_prio = int(os.environ.get("LMN_SIDE_PRIO", "0"))   # HIP priority of the weight-gradient streams (-1: high; A/B runs)
        self.branch_prio = int(os.environ.get("LMN_BRANCH_PRIO", "0"))
        self.prio_main = int(os.environ.get("LMN_PRIO_MAIN", "3"))     # wave priority (s_setprio) of the kernels on the caller's stream ...
        self.prio_branch = int(os.environ.get("LMN_PRIO_BRANCH", "0")) # ... and on the branch streams (0: the default, as the weight-gradient streams)
        self.slot_base = 4 * (Engine._instances % 16)     # this engine's four numbered events (lmn_event_record / wait: 64 per process)
        Engine._instances += 1
        self.zpool_fwd, self.zpool_bwd = ZeroPool(), ZeroPool()
        self.packs_fwd, self.packs_bwd = hip.PackPlan(), hip.PackPlan()
        self.mma = hip.F32        # matrix-core operand type of the dense contractions of the pass (hip.F32 | hip.BF16)
        self.act_dtype = torch.float32   # storage of the activation tensors of the pass (bf16 needs mma = BF16)
        self.arena = None         # plan mode (LM_Net.enable_plans): bump arena all tensors of the pass come from
        self.planning = False     # a plan is being recorded: buffers come from the arena, no allocator stream bookkeeping
        self.alloc_floats = 0     # floats requested since begin_pass (sizes the arena during the eager warm-up steps)
        # second stage of the weight gradients' K-split reductions batched per gradient bucket (LMN_DEFER_REDUCE=0: A/B runs)
        self.defer_reduce = os.environ.get("LMN_DEFER_REDUCE", "1") != "0"
        self.probe = None         # debug hook fn(tag, module, tensors: dict) called in stream order at a few points of the backward
                                  # (tools/gpu_race_locate.py, tools/gpu_glitch_locate.py clone what they need there); None in production
        self.reduce_tabs = []     # device job tables of the batched reductions of the pass in flight (kept by recorded plans)
        self.post_reduce = {}     # launch stream handle -> follow-ups of deferred weight gradients (run after the batched reduction)
        # z-path: expand-conv weight gradient from the raw gradient and the moments of x (lmn_reparam_wfin) instead of a pass that
        # materialises dz (LMN_ZPATH_M=0: lmn_affine2 + plain weight gradient, A/B runs)
        self.zpath_m = os.environ.get("LMN_ZPATH_M", "1") != "0"
        # LayerNorm fused into the consuming Linear (hip.SRC_LN: norm1 -> qkv, norm2 -> fc1 of the transformer blocks; LMN_FUSE_LN=0: the
        # lmn_ln_fwd launches and the n1 / n2 tensors of rounds 1-4, A/B runs)
        self.fuse_ln = os.environ.get("LMN_FUSE_LN", "1") != "0"
        # bilinear x2 sampled where the 3x3 conv stages its window (hip.SRC_UP2: up1..4 and the `convs` branch of the skip fusers; the
        # backward recomputes `up` for the weight gradient on the weight-gradient stream; LMN_FUSE_UP=0: lmn_up2_fwd in the forward)
        self.fuse_up = os.environ.get("LMN_FUSE_UP", "1") != "0"
        # ... and in the weight gradient's staging (LMN_FUSE_UP_WGRAD=0: `up` recomputed on the weight-gradient stream, A/B runs)
        self.fuse_up_wgrad = os.environ.get("LMN_FUSE_UP_WGRAD", "1") != "0"
        # squeeze-excite parameter gradients in the launch of lmn_reparam_wfin (one launch instead of two per block; LMN_FUSE_SE_WFIN=0: A/B)
        self.fuse_se_wfin = os.environ.get("LMN_FUSE_SE_WFIN", "1") != "0"
        self.unpad_side = os.environ.get("LMN_UNPAD_SIDE", "1") != "0"      # un-pad copies of padded weight gradients on the side stream
        # skip fusers: BatchNorm finalize / backward coefficients inside the BN + GELU tails (lmn_bnact_fwd_fin / lmn_bnact_bwd_fin;
        # LMN_FUSE_BN_TAIL=0: the lmn_bn_finalize / lmn_bn_bwd_coef launches, A/B runs)
        self.fuse_bn_tail = os.environ.get("LMN_FUSE_BN_TAIL", "1") != "0"
        # LayerNorm backward in the epilogue of the data-gradient conv of the Linear it feeds (hip.EP_LN_BWD, C <= 48: the neighborhood-
        # attention blocks of levels 0-2; LMN_FUSE_LN_BWD=0: lmn_ln_bwd launches, A/B runs)
        self.fuse_ln_bwd = os.environ.get("LMN_FUSE_LN_BWD", "1") != "0"

    def pm(self):
        """precision mode of the pass: 0 fp32, 1 bf16 MFMA operands on fp32 storage, 2 bf16 storage + bf16 operands."""
        return 0 if self.mma == hip.F32 else (2 if self.act_dtype == torch.bfloat16 else 1)

    def alloc(self, device, shape, dtype=torch.float32):
        n = 1
        for d in shape:
            n *= int(d)
        self.alloc_floats += ((n if dtype == torch.float32 else (n + 1) // 2) + 63) & ~63
        if self.arena is not None:
            return self.arena.alloc(shape, dtype)
        return torch.empty(shape, device=device, dtype=dtype)

    # ------------------------------------------------------------------ small helpers
    def begin_pass(self, backward, device):
        _ENG[0] = self
        self.alloc_floats = 0
        self.reduce_tabs = []
        self.post_reduce = {}
        self.lazy_q, self.lazy_on = [], False
        hip.wgrad_reduce_drop()
        hip._ALLOC[0] = self.alloc
        hip._STREAM[0] = None
        hip._STREAM[0] = hip._stream()          # one stream lookup per pass instead of one per launch
        hip._SEED_CTR[0] = self.seed_ctr
        hip._MMA[0] = self.mma
        pool = self.zpool_bwd if backward else self.zpool_fwd
        pool.begin(device)
        _POOL[0] = pool
        plan = self.packs_bwd if backward else self.packs_fwd
        plan.refresh()            # all persistent weights of this pass re-packed in one launch
        hip._PLAN[0] = plan

    def end_pass(self):
        if _POOL[0] is not None:
            _POOL[0].end()
        _POOL[0] = None
        _ENG[0] = None
        hip._MMA[0] = hip.F32
        hip._ALLOC[0] = None
        hip._PLAN[0] = None
        hip._STREAM[0] = None
        hip._SEED_CTR[0] = None

    def _seed(self, tag):
        if self.seed_ctr is not None:   # graph mode: the step-dependent part lives in device memory (hip._SEED_CTR)
            return (self.seed_base + 0x9E3779B1 * tag) & 0xFFFFFFFF
        return (self.seed_base + 0x9E3779B1 * (self.step * 64 + tag)) & 0xFFFFFFFF

    @staticmethod
    def _t(src):
        v = src["view"] if isinstance(src, dict) else src
        return v.t if isinstance(v, V) else v

    @staticmethod
    def _c(src):
        v = src["view"] if isinstance(src, dict) else src
        return v.C if isinstance(v, V) else v.shape[-1]

    def conv(self, srcs, w, bias, out, *, Hin, Win, k=1, s=1, wp=None, cout=None, **kw):
        """Forward conv of `srcs` (list) with torch-layout weight w [Cout, Cin, k, k] (or [Cout, Cin]).
        cout: rows computed (the weight's Cout rounded up to 4: the extra rows of the packed weight are zeros)."""
        B = self._t(srcs[0]).shape[0]
        if wp is None:
            wp = hip.conv_pack(w, k, [self._c(x) for x in srcs])
        Hout = (Hin + 2 * (k // 2) - k) // s + 1
        Wout = (Win + 2 * (k // 2) - k) // s + 1
        hip.conv_fwd(srcs, wp, out, B=B, Hin=Hin, Win=Win, Hout=Hout, Wout=Wout, Cout=cout or w.shape[0], ksize=k, stride=s,
                     bias=bias, **kw)
        return wp

    def conv_T(self, dy, w, out, *, Hin, Win, k=1, s=1, row_off=0, rows=None, B=None, **kw):
        """Data gradient of a forward conv with weight w whose INPUT was Hin x Win: out = dL/dx (rows slice).
        rows / the channel count of dy may be the weight's Cin / Cout rounded up to 4 (zero-padded operators)."""
        rows = w.shape[1] - row_off if rows is None else rows
        wpt = hip.conv_pack_t(w, k, row_off, rows, cred=self._c(dy))
        Ho = (Hin + 2 * (k // 2) - k) // s + 1
        Wo = (Win + 2 * (k // 2) - k) // s + 1
        B = self._t(dy).shape[0] if B is None else B
        hip.conv_fwd([dy], wpt, out, B=B, Hin=Ho, Win=Wo, Hout=Hin, Wout=Win, Cout=rows, ksize=k, stride=s,
                     transposed=1, **kw)

    def wgrad(self, srcs, dy, w_param, b_param, *, Hin, Win, k=1, s=1, dW=None, db=None, join=True, after=None, deferred=False, keep=(), pre=None, **kw):
        """join=False: an explicit dW is NOT read on the issuing stream right away (no join; the K-split reduction stays deferred
        when after is given).  after: callable run on the gradient's stream right after the batched reduction that completes it.
        pre: callable run on the gradient's stream right before the launch (recomputes an operand the forward did not keep: the
        bilinear x2 `up` tensor of a LMN_SRC_UP2 conv); the tensors it touches go into `keep`."""
        d = dy.t if isinstance(dy, V) else dy
        if self.lazy_on and not (dW is not None and join):
            # late scheduling (LM_Net._backward_body): the weight gradients of the skip / neighborhood-attention chains have no
            # consumer before the optimizer; they are issued once the compute chain has reached the encoder's small feature maps,
            # which cannot fill the GPU, instead of next to the level-0 chains that already saturate it
            G_ = self.G
            if dW is None and w_param is not None:
                dW = G_[w_param]
            if db is None and b_param is not None:
                db = G_[b_param]
            self.lazy_q.append(lambda: self.wgrad(srcs, dy, None, None, Hin=Hin, Win=Win, k=k, s=s, dW=dW, db=db, join=False,
                                                  after=after, deferred=True, keep=keep, pre=pre, **kw))
            return
        B = d.shape[0]
        Ho = (Hin + 2 * (k // 2) - k) // s + 1
        Wo = (Win + 2 * (k // 2) - k) // s + 1
        explicit = dW is not None and join             # caller reads the result on the main stream right away
        if dW is None and w_param is not None:
            dW = self.G[w_param]
        db = (self.G[b_param] if b_param is not None else None) if db is None else db
        cout = dW.shape[0] if dW is not None else kw["dW_src"][0].shape[0]
        # second stage of the K-split reduction: deferred and batched per gradient bucket (flush_reduce) unless the caller
        # reads the result right away
        defer = self.defer_reduce and not explicit and not self.capturing and (join or deferred or after is not None)   # (a capture cannot upload the job table)
        if not self.overlap_wgrad or self.capturing:
            if pre is not None:
                pre()
            hip.conv_wgrad(srcs, dy, dW, db, B=B, Hin=Hin, Win=Win, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s, defer=defer, **kw)
            if after is not None:
                if defer:
                    self.post_reduce.setdefault(hip.stream_key(), []).append(after)
                else:
                    after()
            return
        # Weight gradients feed nothing downstream in the backward chain: they run on a side stream and overlap the
        # data-gradient chain on the main stream (at batch 8 most kernels of levels 2-4 cannot fill 256 CUs alone).
        # Order: side waits for everything enqueued on main so far; inputs are marked as in use on the side stream so
        # the caching allocator does not recycle them early; main joins the side stream at the end of backward
        # (and before every data-parallel bucket hand-over).
        main = torch.cuda.current_stream(d.device)
        side = self._side_stream(main)
        hip.stream_wait(side, main)
        saved = hip._STREAM[0]
        hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
        try:
            if pre is not None:
                pre()
            ws = hip.conv_wgrad(srcs, dy, dW, db, B=B, Hin=Hin, Win=Win, Hout=Ho, Wout=Wo, Cout=cout, ksize=k, stride=s, defer=defer, **kw)
        finally:
            hip._STREAM[0] = saved
        if after is not None:
            if defer:
                self.post_reduce.setdefault(side.cuda_stream, []).append(after)
            else:
                hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
                try:
                    after()
                finally:
                    hip._STREAM[0] = saved
        if self.arena is None:          # caching-allocator tensors: keep them alive for the side stream
            d.record_stream(side)
            if ws is not None:
                ws.record_stream(side)
            for t in keep:              # (tensors only the follow-up `after` touches: without this mark the allocator hands their memory
                t.record_stream(side)   #  to the next main-stream allocation while the side stream has not run the follow-up yet)
            if dW is not None:
                dW.record_stream(side)
            if db is not None:
                db.record_stream(side)
            for src in srcs:
                self._t(src).record_stream(side)
                if isinstance(src, dict) and src.get("scale") is not None:
                    src["scale"].record_stream(side)
                if isinstance(src, dict) and src.get("ln") is not None and src["ln"][3] is not None:
                    src["ln"][3].record_stream(side)      # the (mean, rstd) table of a LayerNorm-on-load source
        if explicit:
            self.join_side(d.device)

    def wgrad_unpad(self, srcs, dy, dWp, db, H, W, unpad, keep=()):
        """Weight gradient into a padded scratch `dWp`, then `unpad` (layout copies into the parameter's gradient).  The copies run
        on the weight-gradient stream right after the reduction that completes the scratch: the main stream does not stop for a
        gradient that nothing in the backward chain reads (LMN_UNPAD_SIDE=0: the round-5 order, main joins and copies)."""
        if self.unpad_side:
            self.wgrad(srcs, dy, None, None, Hin=H, Win=W, dW=dWp, db=db, join=False, keep=(dWp,) + tuple(keep), after=unpad)
        else:
            self.wgrad(srcs, dy, None, None, Hin=H, Win=W, dW=dWp, db=db)
            unpad()

    def set_deterministic(self, on):
        """Fixed-order reductions in every kernel (hip.set_deterministic: process-wide) and no ticket-based squeeze-excite gate in the
        depthwise forward (its sums arrive by float atomics in that form)."""
        self.deterministic = bool(on)
        self.fuse_se = (self._fuse_se0 & ~1) if on else self._fuse_se0
        try:
            hip.set_deterministic(on)
        except Exception:
            if on:
                raise

    def side_call(self, ref, fn, keep=()):
        """Run fn() (library launches) on the weight-gradient side stream of the current stream, ordered after everything
        enqueued so far; `keep`: tensors the side work reads (caching-allocator bookkeeping outside plan mode)."""
        if not self.overlap_wgrad or self.capturing:
            fn()
            return
        main = torch.cuda.current_stream(ref.device)
        side = self._side_stream(main)
        hip.stream_wait(side, main)
        saved = hip._STREAM[0]
        hip._STREAM[0] = hip.C.c_void_p(side.cuda_stream)
        try:
            fn()
        finally:
            hip._STREAM[0] = saved
        if self.arena is None:
            for t in keep:
                t.record_stream(side)

    def _side_stream(self, issuing):
        """The weight-gradient stream paired with the issuing stream (main, or the branch stream of LM_Net's
        backward): each chain joins only its own weight gradients."""
        key = issuing.cuda_stream
        ent = self.sides.get(key)
        if ent is None:
            ent = self.sides[key] = [_new_stream(issuing.device, self.side_prio), False]
        ent[1] = True
        return ent[0]

    def flush_reduce(self, device):
        """ONE launch per stream for the deferred K-split reductions (hip.conv_wgrad(defer=True)) of the weight gradients issued
        from the current stream: on its weight-gradient stream, and on the stream itself where they ran there.  Then the
        follow-ups registered for those gradients (wgrad(after=...)) on the same stream."""
        cur = torch.cuda.current_stream(device)
        keys = [(cur.cuda_stream, None)]
        ent = self.sides.get(cur.cuda_stream)
        if ent is not None:
            keys.append((ent[0].cuda_stream, ent[0]))
        for key, stream in keys:
            fns = self.post_reduce.pop(key, None)
            if not hip.wgrad_reduce_pending(key) and not fns:
                continue
            saved = hip._STREAM[0]
            hip._STREAM[0] = hip.C.c_void_p(key)
            try:
                tabs = hip.wgrad_reduce_flush()
                for fn in fns or ():
                    fn()
            finally:
                hip._STREAM[0] = saved
            for tab in tabs:
                self.reduce_tabs.append(tab)
                if self.arena is None and stream is not None:
                    tab.record_stream(stream)

    def join_side(self, device):
        """Make the current stream wait for its weight-gradient stream (whose deferred reductions are launched first)."""
        self.flush_reduce(device)
        cur = torch.cuda.current_stream(device)
        ent = self.sides.get(cur.cuda_stream)
        if ent is not None and ent[1]:
            hip.stream_wait(cur, ent[0])
            ent[1] = False

    @contextlib.contextmanager
    def on_stream(self, s):
        """Launch everything inside the block on stream `s` (torch allocations and the C-ABI calls alike)."""
        prev = hip._STREAM[0]
        with torch.cuda.stream(s):
            hip._STREAM[0] = hip.C.c_void_p(s.cuda_stream)
            try:
                yield
            finally:
                hip._STREAM[0] = prev

    def branch_stream(self, device):
        if self.branch is None or self.branch.device != device:
            self.branch = _new_stream(device, self.branch_prio, self.prio_branch)
        return self.branch

    def branch_stream_n(self, device, i):
        """Branch stream i (0 = branch_stream): the skip / neighborhood-attention chains of the four levels are spread over them by
        LMN_BRANCH_MAP_F / LMN_BRANCH_MAP_B (one digit per level 0..3; round 6: small-map chains beside the level-0 / 1 ones)."""
        if i == 0:
            return self.branch_stream(device)
        while len(self.branches) < i:
            self.branches.append(None)
        b = self.branches[i - 1]
        if b is None or b.device != device:
            b = self.branches[i - 1] = _new_stream(device, self.branch_prio, self.prio_branch)
        return b

    def bn_stats(self, bn, sums, count, ref):
        """(mean, rstd, A, shift) of a BatchNorm from batch sums [2,C] (training; taken about the running mean, see
        lmn_bn_finalize) or running stats (eval)."""
        C = bn.weight.numel()
        mean, rstd, A, shift = (_E(ref, C) for _ in range(4))
        if self.training:
            hip.bn_finalize(sums, count, bn.weight, bn.bias, bn.eps, bn.momentum if bn.momentum is not None else 0.1,
                            mean, rstd, A, shift, bn.running_mean, bn.running_var, about=bn.running_mean)
        else:
            hip.bn_fold(bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, mean, rstd, A, shift)
        return mean, rstd, A, shift

    # ------------------------------------------------------------------ ReparamConv  (rows A1, A2, A3)
    def reparam_fwd(self, m, x, cx, out=None, chain=None, pre_z=None):
        """x: NHWC tensor [B,H,W,Cin] (Cin possibly zero-padded to 4).  Returns y [B,H,W,Cout].
        chain: the NEXT ReparamConv of the stage -- its expand conv (z + batch sums) is computed from this block's output tile inside
        the pointwise + shortcut launch (hip.conv_fwd(chain=...), lmn_conv_chain_t) where the library takes the pair; returns
        (y, pre_z) then, pre_z = None when the pair was not taken.  pre_z: what such a launch left for THIS block (its z-path skips
        the expand conv)."""
        B, H, W, _ = x.shape
        E, Cout, N = m.cexp, m.cout, B * H * W
        ec, ebn = m.expand_conv[0], m.expand_conv[1]
        wpe = hip.conv_pack(ec.weight, 1, [x.shape[-1]])   # (a 3-channel weight on the NHWC4 input: zero column packed)
        x1 = pre_z["x1"] if pre_z is not None else _R(x, B, H, W, E)
        # (lmn_reparam_fold keeps two [E]-wide operator panels of the expand conv in LDS: wider blocks take the two-pass BatchNorm form)
        zpath = self.training and self.fuse_bn and self.zpath and not m.deploy and 35 * E * 4 <= self.zpath_lds
        zp = None
        if zpath:
            # z-path: ONE pass writes z = conv(x) + bias and its batch sums; the depthwise kernels form x1 = Hardswish(A1 z + sh1)
            # themselves when they stage their rows (lmn_dw_pre_t), the first of them (lmn_dw_stats) finalises the BatchNorm --
            # no statistics-only conv, no second read of x (x1 below HOLDS z)
            if pre_z is not None:
                sums1 = pre_z["sums1"]                       # x1 (z) and the batch sums were written by the previous block's pointwise launch
            else:
                sums1 = _Z(x, STATS_REP + 1, 2, E)
                hip.conv_fwd([x], wpe, x1, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, stats=sums1,
                             stats_mode=hip.STATS_SUM_SQ, stats_rep=STATS_REP, stats_snap=True,
                             p=(None, None, None, None, ebn.running_mean))
            mean1, rstd1, A1, sh1 = (_E(x, E) for _ in range(4))
            zfin = dict(mode=hip.FIN_BN, sums=sums1, nrep=STATS_REP, count=N, gamma=ebn.weight, beta=ebn.bias, eps=ebn.eps,
                        momentum=ebn.momentum if ebn.momentum is not None else 0.1, about=sums1[STATS_REP, 0],
                        mean=mean1, rstd=rstd1, A=A1, shift=sh1, rmean=ebn.running_mean, rvar=ebn.running_var)
            zp = dict(A=A1, shift=sh1)
            if self.zpath_m and cx is not None:
                # moments of x for the backward's weight gradient: M = sum x x^T, m = sum x -- one weight-gradient launch over
                # (x, x) on the weight-gradient stream, which idles in the forward
                zp["M"], zp["m"] = _Z(x, x.shape[-1], x.shape[-1]), _Z(x, x.shape[-1])
                self.wgrad([x], x, None, None, Hin=H, Win=W, dW=zp["M"], db=zp["m"], join=False, deferred=True)   # (reduced at the end of the pass)
        elif self.training and self.fuse_bn:
            # statistics pass (sums about the running mean, whose snapshot lands behind the slices), then the applying pass
            # forms mean / rstd / A / shift itself (lmn_bn_fin_t): no lmn_bn_finalize launch in between
            sums1 = _Z(x, STATS_REP + 1, 2, E)
            hip.conv_fwd([x], wpe, None, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, stats=sums1,
                         stats_mode=hip.STATS_SUM_SQ, stats_rep=STATS_REP, stats_snap=True,
                         p=(None, None, None, None, ebn.running_mean))
            mean1, rstd1, A1, sh1 = (_E(x, E) for _ in range(4))
            fin = dict(mode=hip.FIN_BN, sums=sums1, nrep=STATS_REP, count=N, gamma=ebn.weight, beta=ebn.bias, eps=ebn.eps,
                       momentum=ebn.momentum if ebn.momentum is not None else 0.1, about=sums1[STATS_REP, 0],
                       mean=mean1, rstd=rstd1, A=A1, shift=sh1, rmean=ebn.running_mean, rvar=ebn.running_var)
            hip.conv_fwd([x], wpe, x1, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, epilogue=hip.EP_AFFINE_ACT,
                         act=hip.ACT_HSWISH, p=(A1, sh1), fin=fin)
        else:
            sums1 = None
            if self.training:
                sums1 = _Z(x, STATS_REP, 2, E)
                hip.conv_fwd([x], wpe, None, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, stats=sums1,
                             stats_mode=hip.STATS_SUM_SQ, stats_rep=STATS_REP, p=(None, None, None, None, ebn.running_mean))
            mean1, rstd1, A1, sh1 = self.bn_stats(ebn, sums1, N, x)
            hip.conv_fwd([x], wpe, x1, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, epilogue=hip.EP_AFFINE_ACT,
                         act=hip.ACT_HSWISH, p=(A1, sh1))
        # depthwise branches
        if m.deploy:
            keff, beff = m.fuse_conv.weight, m.fuse_conv.bias
            bmean = brstd = bA = None
        else:
            brs = m.branches()
            ws = [b.conv.weight for b in brs]
            bmean, brstd, bA, bshift = (_E(x, 4, E) for _ in range(4))
            st2 = None
            if self.training:
                st2 = _Z(x, 4, 2, E)
                hip.dw_stats(x1, *ws, st2, zpre=dict(fin=zfin) if zpath else None)
            keff = beff = None
            if self.training and self.fuse_bn:
                pass            # finalize + merge happen inside the depthwise pass itself (hip.dw_fwd_bn below)
            elif self.training:   # the four finalizes and the merge in one launch
                keff, beff = _E(x, E, 25), _E(x, E)
                hip.dw_finalize_merge(st2, N, [b.bn for b in brs], ws, bmean, brstd, bA, keff, beff)
            else:
                keff, beff = _E(x, E, 25), _E(x, E)
                for i, b in enumerate(brs):
                    bn = b.bn
                    hip.bn_fold(bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, bmean[i], brstd[i], bA[i], bshift[i])
                hip.dw_merge(*ws, bA, bshift, keff, beff)
        pre = _R(x, B, H, W, E)
        gsum = _Z(x, B, E)
        se = m.se
        R = se.fc1.weight.shape[0]
        sgate, hid = _E(x, B, E), _E(x, B, R)
        # the squeeze-excite gate is formed by the block of the depthwise pass that completes an image's sums (no launch)
        sef = dict(ticket=_Z(x, B), fc1w=se.fc1.weight, fc1b=se.fc1.bias, fc2w=se.fc2.weight, fc2b=se.fc2.bias, s=sgate,
                   hidden=hid, inv_hw=1.0 / (H * W)) if (self.fuse_se & 1) else None
        if keff is None:
            hip.dw_fwd_bn(x1, pre, gsum, st2, N, [b.bn for b in brs], ws, bmean, brstd, bA, se=sef, zpre=zp)
        else:
            hip.dw_fwd(x1, pre, gsum, keff, beff, se=sef)
        if sef is None:
            hip.se_fwd(gsum, 1.0 / (H * W), se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias, sgate, hid)
        # pointwise(g*s) + shortcut(x): one conv over two sources
        wpw, wsc = m.pointwise_conv[0].weight, m.shortcut[0].weight
        wp3 = self._pack2(wpw, wsc, E, x.shape[-1], Cout, x)
        y = _A(x, B, H, W, Cout) if out is None else out
        kwc = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cout, bias=m.pointwise_conv[0].bias, bias2=m.shortcut[0].bias)
        srcs = [dict(view=pre, scale=sgate, flags=hip.SRC_GELU), x]
        pz = None
        if chain is not None and self.chain_on and out is None:
            # the next block's expand conv on this launch's output tile (its z-path conditions, its arguments: see the z-path branch above)
            m2 = chain
            E2 = m2.cexp
            ec2, ebn2 = m2.expand_conv[0], m2.expand_conv[1]
            if (self.training and self.fuse_bn and self.zpath and not m2.deploy and 35 * E2 * 4 <= self.zpath_lds
                    and ec2.weight.shape[1] == Cout):
                x1n, sums1n = _R(x, B, H, W, E2), _Z(x, STATS_REP + 1, 2, E2)
                ch = dict(wpack=hip.conv_pack(ec2.weight, 1, [Cout]), Cout=E2, out=x1n, bias=ec2.bias, shift=ebn2.running_mean,
                          stats=sums1n, stats_mode=hip.STATS_SUM_SQ, stats_rep=STATS_REP, stats_snap=True)
                if hip.conv_fwd(srcs, wp3, y, chain=ch, query_chain=True, **kwc):
                    hip.conv_fwd(srcs, wp3, y, chain=ch, **kwc)
                    pz = dict(x1=x1n, sums1=sums1n)
        if pz is None:
            hip.conv_fwd(srcs, wp3, y, **kwc)
        if cx is not None:
            cx.t[m] = dict(x=x, x1=x1, pre=pre, gsum=gsum, s=sgate, hid=hid, wpe=wpe, mean1=mean1, rstd1=rstd1, A1=A1,
                           bmean=bmean, brstd=brstd, bA=bA, zp=zp)        # (zp: x1 holds z, the z-path)
        return (y, pz) if chain is not None else y

    @staticmethod
    def _pack2(w0, w1, c0, c1, cout, ref):
        """1x1 conv over two sources with separate weights: pack each into its K-block range."""
        n0 = hip.conv_pack_size(1, cout, [c0])
        n1 = hip.conv_pack_size(1, cout, [c1])
        plan = hip._PLAN[0]
        own = plan is not None and isinstance(w0, torch.nn.Parameter) and isinstance(w1, torch.nn.Parameter)
        if own:
            wp = plan.buffer(("pack2", w0.data_ptr(), w1.data_ptr(), hip._MMA[0]), n0 + n1, ref.device)
        else:
            wp = _E(ref, n0 + n1)
        h = 2 if hip._MMA[0] == hip.BF16 else 1      # bf16 fragments take half the space: source 1 starts at n0 / 2
        hip.conv_pack(w0, 1, [c0], out=wp[:n0 // h], persistent=own)
        hip.conv_pack(w1, 1, [c1], out=wp[n0 // h:(n0 + n1) // h], persistent=own)
        return wp

    @staticmethod
    def _pack2_t(w0, w1, cred0, cred1, rows, ref):
        """Data-gradient operators of two 1x1 convs that share their input (rows = its channels), as ONE two-source conv:
        dx = W0^T . dy0 + W1^T . dy1.  Each transposed operator is packed into its K-block range."""
        n0 = hip.conv_pack_size(1, rows, [cred0])
        n1 = hip.conv_pack_size(1, rows, [cred1])
        plan = hip._PLAN[0]
        own = plan is not None and isinstance(w0, torch.nn.Parameter) and isinstance(w1, torch.nn.Parameter)
        if own:
            wp = plan.buffer(("pack2t", w0.data_ptr(), w1.data_ptr(), rows, hip._MMA[0]), n0 + n1, ref.device)
        else:
            wp = _E(ref, n0 + n1)
        h = 2 if hip._MMA[0] == hip.BF16 else 1
        hip.conv_pack_t(w0, 1, 0, rows, out=wp[:n0 // h], cred=cred0, persistent=own)
        hip.conv_pack_t(w1, 1, 0, rows, out=wp[n0 // h:(n0 + n1) // h], cred=cred1, persistent=own)
        return wp

    def reparam_bwd(self, m, dy, cx, need_dx=True, chain=None, pre_u=None):
        """chain: the PREVIOUS ReparamConv of the stage (it consumes this block's dx as its dy): its SE-gradient conv is computed from
        this block's dx tile inside the folded data-gradient launch where the library takes the pair; returns (dx, pre_u) then.
        pre_u = (u, ds) left by such a launch for THIS block, which then skips its SE-gradient conv."""
        if m.deploy:
            raise NotImplementedError("backward through a deployed (re-parameterised) ReparamConv is not supported; "
                                      "deploy form is inference-only, as in the reference")
        S = cx.t[m]
        x, x1, pre, sgate = S["x"], S["x1"], S["pre"], S["s"]
        B, H, W, Cin = x.shape
        E, Cout, N = m.cexp, m.cout, B * H * W
        G = self.G
        pw, sc, ec, ebn, se = m.pointwise_conv[0], m.shortcut[0], m.expand_conv[0], m.expand_conv[1], m.se
        # ---- A3 backward
        cw = sc.weight.shape[1]          # real input channels (3 for the RGB stem, carried as NHWC4)
        if cw == Cin:  # one pass over dy for both convs (they were one conv over two sources in the forward)
            self.wgrad([dict(view=pre, scale=sgate, flags=hip.SRC_GELU), x], dy, None, pw.bias, Hin=H, Win=W,
                       dW_src=[G[pw.weight], G[sc.weight]], db2=G[sc.bias])
        else:  # padded RGB input: gradient of the padded weight, keep the real columns
            self.wgrad([dict(view=pre, scale=sgate, flags=hip.SRC_GELU)], dy, pw.weight, pw.bias, Hin=H, Win=W)
            dWp = _Z(x, Cout, Cin)
            gsc = G[sc.weight]
            self.wgrad_unpad([x], dy, dWp, G[sc.bias], H, W, lambda d=dWp, g=gsc: hip.copy2d(d, g, Cout, cw, Cin, cw))   # un-pad (layout copy)
        if pre_u is not None:
            u, ds = pre_u                    # written by the next block's data-gradient launch (chained SE-gradient conv)
        else:
            u = _R(x, B, H, W, E)
            ds = _Z(x, B, E)
            self.conv_T(dy, pw.weight, u, Hin=H, Win=W, epilogue=hip.EP_SE_BWD, aux=pre, stats=ds, stats_mode=hip.STATS_EP)
        # (the shortcut's data gradient W_sc^T . dy is the second source of the LAST conv of this function: no dx_sc tensor)
        # ---- SE backward
        dm = None
        seb = None
        if self.fuse_se & 2:
            # dm is formed inside the depthwise statistics pass (every block for its own channels); parameter gradients on the
            # weight-gradient stream once that pass has written dvec
            dvec = _E(x, B, E + se.fc1.weight.shape[0])
            seb = dict(ds=ds, fc1w=se.fc1.weight, fc2w=se.fc2.weight, hidden=S["hid"], dvec=dvec, inv_hw=1.0 / (H * W))
        elif self.split_se:
            dm = _E(x, B, E)
            # dm (critical path: it feeds the depthwise backward) here, the parameter gradients on the weight-gradient stream
            dvec = _E(x, B, E + se.fc1.weight.shape[0])
            hip.se_bwd_dm(ds, sgate, 1.0 / (H * W), se.fc1.weight, se.fc2.weight, S["hid"], dm, dvec)
            gs_, hid_ = S["gsum"], S["hid"]
            self.side_call(x, lambda: hip.se_bwd_params(dvec, gs_, 1.0 / (H * W), hid_, G[se.fc1.weight], G[se.fc1.bias],
                                                        G[se.fc2.weight], G[se.fc2.bias]), keep=(dvec, gs_, hid_))
        else:
            dm = _E(x, B, E)
            hip.se_bwd(ds, S["gsum"], 1.0 / (H * W), se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias, S["hid"], dm,
                       G[se.fc1.weight], G[se.fc1.bias], G[se.fc2.weight], G[se.fc2.bias])
        # ---- A2 backward
        brs = m.branches()
        ws = [b.conv.weight for b in brs]
        dpre = _R(x, B, H, W, E)
        bst = _Z(x, 5, E)
        zp = S.get("zp")
        hip.dw_bwd_stats(x1, pre, u, sgate, dm, dpre, *ws, bst, seb=seb, zpre=zp)
        sep = None   # squeeze-excite parameter gradients inside lmn_reparam_wfin's launch (same stream, later in this block's backward)
        if seb is not None:
            gs_, hid_ = S["gsum"], S["hid"]
            if self.fuse_se_wfin and zp is not None and zp.get("M") is not None:
                sep = dict(dvec=dvec, gsum=gs_, inv_hw=1.0 / (H * W), hidden=hid_, dw1=G[se.fc1.weight], db1=G[se.fc1.bias],
                           dw2=G[se.fc2.weight], db2=G[se.fc2.bias])
            else:
                self.side_call(x, lambda: hip.se_bwd_params(dvec, gs_, 1.0 / (H * W), hid_, G[se.fc1.weight], G[se.fc1.bias],
                                                            G[se.fc2.weight], G[se.fc2.bias]), keep=(dvec, gs_, hid_))
        dx1 = u  # reuse
        if zp is not None:
            # ---- z-path: the depthwise backward writes dh = dx1 * Hardswish'(A1 z + sh1) and its BatchNorm-backward sums; ONE tiny
            # launch folds the BatchNorm backward into the operators of ONE conv over (dh, x, dy) -> dx.  The weight gradient's
            # dz = a dh + b z + c is materialised beside the critical path.
            hst = _Z(x, 2, E)
            hip.dw_bwd_bn(x1, dpre, dx1, *ws, bst, S["bmean"], S["brstd"], S["bA"], N, self.training,
                          [G[b.bn.weight] for b in brs], [G[b.bn.bias] for b in brs], *[G[w] for w in ws], zpre=zp, hstats=hst)
            dh = dx1
            cdy = self._c(dy)
            wp3 = _E(x, hip.conv_pack_size(1, Cin, [E, Cin, cdy]))
            kb, coef = _E(x, Cin), _E(x, 3, E)
            hip.reparam_fold(hst, S["mean1"], S["rstd1"], S["A1"], N, self.training, ec.weight, ec.bias, sc.weight, Cin, cdy,
                             wp3, kb, coef, G[ebn.weight], G[ebn.bias])
            if zp.get("M") is not None:
                # dW_e = diag(a) R + diag(b) (W_e M + b_e m^T) + c m^T with R = sum dh x^T: the raw gradient into a scratch, the
                # closed form right after the batched reduction that completes it (same stream)
                Rw = _Z(x, E, Cin)
                Mx, mx, gw, gb = zp["M"], zp["m"], G[ec.weight], G[ec.bias]
                self.wgrad([x], dh, None, None, Hin=H, Win=W, dW=Rw, db=None, join=False,
                           keep=(Rw, Mx, mx, coef, hst) + ((sep["dvec"], sep["gsum"], sep["hidden"]) if sep else ()),
                           after=lambda: hip.reparam_wfin(Rw, Mx, mx, coef, hst, ec.weight, ec.bias, N, gw, gb, se=sep))
            else:
                dz = dpre  # (reuse: its last reader on this stream was the depthwise backward)
                z_ = x1
                self.side_call(x, lambda: hip.affine2(dh, z_, coef, dz), keep=(dh, z_, coef, dz))
                if cw == Cin:
                    self.wgrad([x], dz, ec.weight, ec.bias, Hin=H, Win=W)
                else:
                    dWp = _Z(x, E, Cin)
                    gec = G[ec.weight]
                    self.wgrad_unpad([x], dz, dWp, G[ec.bias], H, W, lambda d=dWp, g=gec: hip.copy2d(d, g, E, cw, Cin, cw))    # un-pad (layout copy)
            if not need_dx:
                return (None, None) if chain is not None else None
            dx = _A(x, B, H, W, Cin)
            kwc = dict(B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cin, bias=kb)
            pu = None
            if chain is not None and self.chain_on and not chain.deploy:
                SA = cx.t[chain]
                EA = chain.cexp
                pwa = chain.pointwise_conv[0]
                if pwa.weight.shape[0] == Cin:
                    ua, dsa = _R(x, B, H, W, EA), _Z(x, B, EA)
                    ch = dict(wpack=hip.conv_pack_t(pwa.weight, 1, 0, EA, cred=Cin), Cout=EA, out=ua, aux=SA["pre"], stats=dsa,
                              epilogue=hip.EP_SE_BWD, stats_mode=hip.STATS_EP)
                    if hip.conv_fwd([dh, x, dy], wp3, dx, chain=ch, query_chain=True, **kwc):
                        hip.conv_fwd([dh, x, dy], wp3, dx, chain=ch, **kwc)
                        pu = (ua, dsa)
            if pu is None:
                hip.conv_fwd([dh, x, dy], wp3, dx, **kwc)
            return (dx, pu) if chain is not None else dx
        split = self.fuse_bn and self.split_dw and self.overlap_wgrad and not self.capturing
        if split:
            # dx1 (critical path) on this stream, the four weight gradients of the same pass on the side stream: the halves share
            # the branch-output recomputation (+37 % VALU work in total), but dx1 alone is 27 % shorter and the weight-gradient half
            # overlaps the HBM-bound BN-backward convs that follow
            args = (x1, dpre, dx1, *ws, bst, S["bmean"], S["brstd"], S["bA"], N, self.training,
                    [G[b.bn.weight] for b in brs], [G[b.bn.bias] for b in brs], *[G[w] for w in ws])
            hip.dw_bwd_bn(*args, part=1)
            self.side_call(x1, lambda: hip.dw_bwd_bn(*args, part=2), keep=(x1, dpre))
        elif self.fuse_bn:   # the coefficients of f_b and the gamma / beta gradients are formed inside the pass
            hip.dw_bwd_bn(x1, dpre, dx1, *ws, bst, S["bmean"], S["brstd"], S["bA"], N, self.training,
                          [G[b.bn.weight] for b in brs], [G[b.bn.bias] for b in brs], *[G[w] for w in ws])
        else:
            cA, cC, cD = (_E(x, 4, E) for _ in range(3))
            hip.dw_bwd_coef(bst, S["bmean"], S["brstd"], S["bA"], N, self.training, cA, cC, cD,
                            [G[b.bn.weight] for b in brs], [G[b.bn.bias] for b in brs])
            hip.dw_bwd(x1, dpre, dx1, *ws, cA, cC, cD, *[G[w] for w in ws])
        # ---- A1 backward: Hardswish' and BatchNorm backward fused into the recomputed 1x1 conv
        wpe = S["wpe"]
        # pass 1 is statistics only (dh = dx1 * hswish'(h) is not written); pass 2 forms dh again from dx1 and turns it into
        # dz in the same epilogue: one E-wide write and one E-wide read fewer than writing dh in between
        st = _Z(x, STATS_REP, 2, E)
        if self.probe is not None:
            self.probe("reparam_bwd:stats0", m, dict(st=st))
        hip.conv_fwd([x], wpe, None, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, epilogue=hip.EP_BN_BWD1,
                     act=hip.ACT_HSWISH, p=(S["mean1"], S["rstd1"], ebn.weight, ebn.bias), aux=dx1, stats=st,
                     stats_mode=hip.STATS_EP, stats_rep=STATS_REP)
        dz = _R(x, B, H, W, E) if split else dpre  # (reuse, unless the side stream still reads dpre)
        if self.fuse_bn:   # c1, c2, c3 and the gamma / beta gradients are formed inside pass 2 (lmn_bn_fin_t)
            fin = dict(mode=hip.FIN_BN_BWD, sums=st, nrep=STATS_REP, count=N, batch_stats=self.training, Ain=S["A1"],
                       dgamma=G[ebn.weight], dbeta=G[ebn.bias])
            hip.conv_fwd([x], wpe, dz, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, epilogue=hip.EP_BN_BWD2,
                         act=hip.ACT_HSWISH, p=(S["mean1"], S["rstd1"], None, None, None, ebn.weight, ebn.bias), aux=dx1, fin=fin)
        else:
            c1, c2, c3 = (_E(x, E) for _ in range(3))
            hip.bn_bwd_coef(st, N, S["A1"], G[ebn.weight], G[ebn.bias], c1, c2, c3, self.training)
            hip.conv_fwd([x], wpe, dz, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=E, bias=ec.bias, epilogue=hip.EP_BN_BWD2,
                         act=hip.ACT_HSWISH, p=(S["mean1"], S["rstd1"], c1, c2, c3, ebn.weight, ebn.bias), aux=dx1)
        if self.probe is not None:
            self.probe("reparam_bwd", m, dict(wpe=wpe, dy=dy, u_dx1=dx1, st=st, dz=dz, bst=bst, ds=ds, x=x, mean1=S["mean1"], rstd1=S["rstd1"], A1=S["A1"]))
        if cw == Cin:
            self.wgrad([x], dz, ec.weight, ec.bias, Hin=H, Win=W)
        else:
            dWp = _Z(x, E, Cin)
            gec = G[ec.weight]
            self.wgrad_unpad([x], dz, dWp, G[ec.bias], H, W, lambda d=dWp, g=gec: hip.copy2d(d, g, E, cw, Cin, cw))    # un-pad (layout copy)
        if not need_dx:
            return (None, None) if chain is not None else None
        # dx = W_e^T . dz + W_sc^T . dy: one conv over two sources (the forward's expand + shortcut share x)
        dx = _A(x, B, H, W, Cin)
        wpt = self._pack2_t(ec.weight, sc.weight, E, self._c(dy), Cin, x)
        hip.conv_fwd([dz, dy], wpt, dx, B=B, Hin=H, Win=W, Hout=H, Wout=W, Cout=Cin)
        return (dx, None) if chain is not None else dx

    def stage_fwd(self, seq, x, cx):
        """Sequential(ReparamConv, ReparamConv) (core/LM_Net.py:11-57).  Where the library takes the pair (levels 0-1, fp32, training),
        block A's pointwise + shortcut launch also computes block B's expand conv from its output tile (one launch and one read of the
        intermediate tensor less)."""
        y, pz = self.reparam_fwd(seq[0], x, cx, chain=seq[1])
        return self.reparam_fwd(seq[1], y, cx, pre_z=pz)

    def stage_bwd(self, seq, dy, cx, need_dx=True):
        """... and block B's folded data gradient also computes block A's SE-gradient conv (u = W_p^T dy and the per-image sums)."""
        dxb, pu = self.reparam_bwd(seq[1], dy, cx, chain=seq[0])
        return self.reparam_bwd(seq[0], dxb, cx, need_dx, pre_u=pu)

    # ------------------------------------------------------------------ plain 3x3 conv rows (A4, parts of A9/A10)
    def conv3_fwd(self, conv, x, out, s=1, **kw):
        x_t = x.t if isinstance(x, V) else x
        H, W = x_t.shape[1:3]
        self.conv([x], conv.weight, conv.bias, out, Hin=H, Win=W, k=3, s=s, **kw)

    def conv3_up_fwd(self, conv, x, out, **kw):
        """out = conv3x3(bilinear_x2(x)) with the upsampling sampled where the conv stages its window (hip.SRC_UP2): `up` is never
        written (/root/reference/core/LM_Net.py:58-74, core/modules.py:94,129)."""
        h, w = x.shape[1:3]
        self.conv([dict(view=x, flags=hip.SRC_UP2)], conv.weight, conv.bias, out, Hin=2 * h, Win=2 * w, k=3, **kw)

    def conv3_up_bwd(self, conv, x, dy, dx_small):
        """Backward of conv3_up_fwd.  Weight gradient: the 3x3 weight gradient with the same on-load sampling where the library's own
        predicate (hip.conv_wgrad_up2_ok: the wgrad3 form, maps >= 32 wide, 32-bit offsets) accepts it; otherwise `up` is recomputed
        on the weight-gradient stream right before the launch (off the compute chain; the forward kept nothing).  Data gradient:
        dup = conv^T(dy), dx = bilinear_x2^T(dup) into dx_small (written, not accumulated)."""
        B, h, w, C = x.shape
        if self.fuse_up_wgrad and hip.conv_wgrad_up2_ok(x, dy, B=B, Hin=2 * h, Win=2 * w, Cout=conv.weight.shape[0]):
            self.wgrad([dict(view=x, flags=hip.SRC_UP2)], dy, conv.weight, conv.bias, Hin=2 * h, Win=2 * w, k=3)
        else:
            up = _A(x, B, 2 * h, 2 * w, C)
            self.wgrad([up], dy, conv.weight, conv.bias, Hin=2 * h, Win=2 * w, k=3, pre=lambda: hip.up2_fwd(x, up), keep=(x, up))
        dup = _A(x, B, 2 * h, 2 * w, C)
        self.conv_T(dy, conv.weight, dup, Hin=2 * h, Win=2 * w, k=3)
        hip.up2_bwd(dup, dx_small)

    def conv3_bwd(self, conv, x, dy, s=1, dx=None, accumulate=False):
        """weight/bias grads; if dx is given: dx (+)= data gradient."""
        x_t = x.t if isinstance(x, V) else x
        H, W = x_t.shape[1:3]
        self.wgrad([x], dy, conv.weight, conv.bias, Hin=H, Win=W, k=3, s=s)
        if dx is not None:
            self.conv_T(dy, conv.weight, dx, Hin=H, Win=W, k=3, s=s, residual=dx if accumulate else None)

    # ------------------------------------------------------------------ skip fusers (A9)
    def skip_fwd(self, m, xs_in, cx):
        """M3Skip(xl, xm, xs) / M2Skip(xl, xs).  All convs write straight into slices of the cat buffer."""
        three = len(xs_in) == 3
        bottom = (not three) and m.model_type == "bottom"
        fconv, fbn = m.fuse_conv[0], m.fuse_conv[1]
        C = fbn.weight.numel()
        if three:
            xl, xm, xsm = xs_in
            B, H, W, _ = xm.shape
        elif bottom:
            xl, xsm = xs_in
            B, H, W, _ = xsm.shape
        else:
            xl, xsm = xs_in
            B, H, W, _ = xl.shape
        nb = 3 if three else 2
        cat = _A(xl, B, H, W, nb * C)
        up = None
        self.conv3_fwd(m.convl[0], xl, V(cat, 0, C), s=2 if (three or bottom) else 1)
        if three:
            self.conv3_fwd(m.convm[0], xm, V(cat, C, C))
        if bottom:
            self.conv3_fwd(m.convs[0], xsm, V(cat, C, C))
        elif self.fuse_up:
            self.conv3_up_fwd(m.convs[1], xsm, V(cat, (nb - 1) * C, C))
        else:
            up = _A(xl, B, H, W, xsm.shape[-1])
            hip.up2_fwd(xsm, up)
            self.conv3_fwd(m.convs[1], up, V(cat, (nb - 1) * C, C))
        z = _A(xl, B, H, W, C)
        y = _A(xl, B, H, W, C)
        if self.training and self.fuse_bn_tail:
            # the conv leaves its batch sums (about the running mean, whose snapshot lands behind the slice) and the BN + GELU tail
            # finalises the BatchNorm itself (lmn_bnact_fwd_fin): no lmn_bn_finalize launch in between
            sums = _Z(xl, 2, 2, C)
            self.conv([cat], fconv.weight, fconv.bias, z, Hin=H, Win=W, k=3, stats=sums, stats_mode=hip.STATS_SUM_SQ, stats_rep=1,
                      stats_snap=True, p=(None, None, None, None, fbn.running_mean))
            mean, rstd, A, shift = (_E(xl, C) for _ in range(4))
            fin = dict(mode=hip.FIN_BN, sums=sums, nrep=1, count=B * H * W, gamma=fbn.weight, beta=fbn.bias, eps=fbn.eps,
                       momentum=fbn.momentum if fbn.momentum is not None else 0.1, about=sums[1, 0], mean=mean, rstd=rstd, A=A,
                       shift=shift, rmean=fbn.running_mean, rvar=fbn.running_var)
            hip.bnact_fwd_fin(z, fin, y, hip.ACT_GELU)
        else:
            sums = _Z(xl, 2, C) if self.training else None
            self.conv([cat], fconv.weight, fconv.bias, z, Hin=H, Win=W, k=3, stats=sums,
                      stats_mode=hip.STATS_SUM_SQ if self.training else hip.STATS_NONE,
                      p=(None, None, None, None, fbn.running_mean) if self.training else ())
            mean, rstd, A, shift = self.bn_stats(fbn, sums, B * H * W, xl)
            hip.bnact_fwd(z, A, shift, y, hip.ACT_GELU)
        if cx is not None:
            cx.t[m] = dict(xs=xs_in, cat=cat, up=up, z=z, mean=mean, rstd=rstd, A=A)
        return y

    def skip_bwd(self, m, dy, cx, gacc, order=None):
        """gacc: dict tensor-id -> GradSlot for the encoder activations (accumulated in place).
        order = (event slot to wait for or None, event slot to record, stream): the chains of several levels accumulate into the SAME
        encoder gradients; when they run on two branch streams the accumulating tail of each waits for the previous chain's (in fork
        order: the sums keep the single-stream order, bit for bit)."""
        S = cx.t[m]
        xs_in, cat, up, z = S["xs"], S["cat"], S["up"], S["z"]
        three = len(xs_in) == 3
        bottom = (not three) and m.model_type == "bottom"
        fconv, fbn = m.fuse_conv[0], m.fuse_conv[1]
        C = fbn.weight.numel()
        B, H, W, _ = z.shape
        G = self.G
        st = _Z(z, 2, C)
        hip.bnact_bwd_stats(z, dy, S["mean"], S["rstd"], fbn.weight, fbn.bias, st, hip.ACT_GELU)
        dz = _A(z, B, H, W, C)
        if self.fuse_bn_tail:   # c1 / c2 / c3 and the gamma / beta gradients inside the applying pass (lmn_bnact_bwd_fin)
            fin = dict(mode=hip.FIN_BN_BWD, sums=st, nrep=1, count=B * H * W, batch_stats=int(self.training), Ain=S["A"],
                       dgamma=G[fbn.weight], dbeta=G[fbn.bias])
            hip.bnact_bwd_fin(z, dy, S["mean"], S["rstd"], fbn.weight, fbn.bias, fin, dz, hip.ACT_GELU)
        else:
            c1, c2, c3 = (_E(z, C) for _ in range(3))
            hip.bn_bwd_coef(st, B * H * W, S["A"], G[fbn.weight], G[fbn.bias], c1, c2, c3, self.training)
            hip.bnact_bwd(z, dy, S["mean"], S["rstd"], fbn.weight, fbn.bias, c1, c2, c3, dz, hip.ACT_GELU)
        self.wgrad([cat], dz, fconv.weight, fconv.bias, Hin=H, Win=W, k=3)
        nb = 3 if three else 2
        dcat = _A(z, B, H, W, nb * C)
        self.conv_T(dz, fconv.weight, dcat, Hin=H, Win=W, k=3)
        xl = xs_in[0]
        sl = 2 if (three or bottom) else 1
        if order is not None and order[0] is not None:
            hip.event_wait(order[0], order[2])
        self._acc_conv(m.convl[0], xl, V(dcat, 0, C), sl, gacc)
        if three:
            self._acc_conv(m.convm[0], xs_in[1], V(dcat, C, C), 1, gacc)
        xsm = xs_in[-1]
        if bottom:
            self._acc_conv(m.convs[0], xsm, V(dcat, C, C), 1, gacc)
        elif up is None:      # (fuse_up: the forward sampled the upsampling on load and kept no `up`)
            slot = gacc[id(xsm)]
            first = slot.g is None
            tgt = _A(z, *xsm.shape)
            self.conv3_up_bwd(m.convs[1], xsm, V(dcat, (nb - 1) * C, C), tgt)
            if first:
                slot.g = tgt
            else:
                hip.add(slot.g, tgt)
        else:
            dup = _A(z, *up.shape)
            self.conv3_bwd(m.convs[1], up, V(dcat, (nb - 1) * C, C), dx=dup)
            slot = gacc[id(xsm)]
            if slot.g is None:
                slot.g = _A(z, *xsm.shape)
                hip.up2_bwd(dup, slot.g)
            else:
                tmp = _A(z, *xsm.shape)
                hip.up2_bwd(dup, tmp)
                hip.add(slot.g, tmp)
        if order is not None:
            hip.event_record(order[1], order[2])

    def _acc_conv(self, conv, x, dy, s, gacc):
        slot = gacc[id(x)]
        first = slot.g is None
        if first:
            slot.g = _A(x, *x.shape)
        self.conv3_bwd(conv, x, dy, s=s, dx=slot.g, accumulate=not first)

    # ------------------------------------------------------------------ transformer pieces (A6, A7, A8)
    @staticmethod
    def _ln_src(x, norm, stats):
        """Source descriptor `LayerNorm(x)` (hip.SRC_LN): the conv / weight gradient normalises x where it stages it -- the
        normalised tensor never exists in HBM.  stats: [pixels, 2] table of (mean, rstd), written by the forward conv, read by the
        weight gradient."""
        Cn = x.shape[-1]
        return dict(view=x.view(1, 1, -1, Cn), ln=(norm.weight, norm.bias, norm.eps, stats))

    def _mlp_fwd(self, mlp, n2, a_res, y, tagbase):
        """y = drop(fc2(drop(gelu(fc1(n2))))) + a_res ; returns a1 (pre-GELU), seeds.  n2: tensor, or an _ln_src descriptor (norm2 fused
        into fc1)."""
        Cn, Ch = mlp.fc1.weight.shape[1], mlp.fc1.weight.shape[0]
        n2f = n2 if isinstance(n2, dict) else n2.view(1, 1, -1, Cn)
        yf, af = y.view(1, 1, -1, Cn), a_res.view(1, 1, -1, Cn)
        npx = yf.shape[2]
        a1 = _A(y, 1, 1, npx, Ch)
        p = mlp.dropout.p if self.training else 0.0
        s1, s2 = self._seed(tagbase), self._seed(tagbase + 1)
        self.conv([n2f], mlp.fc1.weight, mlp.fc1.bias, a1, Hin=1, Win=npx)
        src = dict(view=a1, flags=hip.SRC_GELU | (hip.SRC_DROP if p > 0 else 0), drop_seed=s1, drop_p=p)
        self.conv([src], mlp.fc2.weight, mlp.fc2.bias, yf, Hin=1, Win=npx, residual=af, drop_p=p, drop_seed=s2)
        return a1, (p, s1, s2)

    def _ln_bwd_ep(self, nsrc, norm, residual, out):
        """Keyword arguments that make a `conv_T` call finish the LayerNorm backward itself (hip.EP_LN_BWD: the data gradient of the Linear
        behind `norm`, then dx = rstd (g - mean g - zh mean(g zh)) + residual in the epilogue, d gamma / d beta as its channel statistics),
        or None when the call cannot take it: no (mean, rstd) table (LMN_FUSE_LN=0), more than 48 channels, or the two gradient slices of
        the LayerNorm not adjacent in the flat buffer (the epilogue's [2][C] statistics land in them directly)."""
        if not (self.fuse_ln_bwd and isinstance(nsrc, dict) and nsrc.get("ln") is not None):
            return None
        x = nsrc["view"]
        Cn = x.shape[-1]
        gw, gb = self.G[norm.weight], self.G[norm.bias]
        if Cn > 48 or gw.dtype != torch.float32:
            return None
        if gw.data_ptr() + 4 * Cn == gb.data_ptr():
            first, swap = gw, 1          # memory order (d gamma, d beta)
        elif gb.data_ptr() + 4 * Cn == gw.data_ptr():
            first, swap = gb, 0
        else:
            return None
        return dict(epilogue=hip.EP_LN_BWD, act=swap, aux=x, p=(norm.weight, None, None, None, None, None, nsrc["ln"][3]),
                    residual=residual.view(1, 1, -1, Cn), stats=first, stats_mode=hip.STATS_EP), out.view(1, 1, -1, Cn)

    def _mlp_bwd(self, mlp, n2, a1, dy, drop, ln_bwd=None):
        """returns dn2; accumulates fc1/fc2 grads.  dy is the gradient of the block output.  ln_bwd = (norm, residual, out): the LayerNorm
        in front of fc1 is differentiated in the last conv's epilogue when _ln_bwd_ep allows (returns None then: `out` holds the result)."""
        p, s1, s2 = drop
        Cn, Ch = mlp.fc1.weight.shape[1], mlp.fc1.weight.shape[0]
        n2f, dyf = (n2 if isinstance(n2, dict) else n2.view(1, 1, -1, Cn)), dy.view(1, 1, -1, Cn)     # (n2: tensor or _ln_src descriptor)
        npx = dyf.shape[2]
        dflag = hip.SRC_DROP if p > 0 else 0
        hsrc = dict(view=a1, flags=hip.SRC_GELU | dflag, drop_seed=s1, drop_p=p)
        self.wgrad([hsrc], dyf, mlp.fc2.weight, mlp.fc2.bias, Hin=1, Win=npx, dy_flags=dflag, dy_seed=s2, dy_p=p)
        da1 = _A(dy, 1, 1, npx, Ch)
        self.conv_T(dict(view=dyf, flags=dflag, drop_seed=s2, drop_p=p), mlp.fc2.weight, da1, Hin=1, Win=npx,
                    epilogue=hip.EP_DGELU, aux=a1, drop_p=p, drop_seed=s1)
        self.wgrad([n2f], da1, mlp.fc1.weight, mlp.fc1.bias, Hin=1, Win=npx)
        ep = self._ln_bwd_ep(n2, *ln_bwd) if ln_bwd is not None else None
        if ep is not None:
            self.conv_T(da1, mlp.fc1.weight, ep[1], Hin=1, Win=npx, **ep[0])
            return None
        dn2 = _A(dy, 1, 1, npx, Cn)
        self.conv_T(da1, mlp.fc1.weight, dn2, Hin=1, Win=npx)
        return dn2.view(dy.shape)

    def _lin(self, lin, x, out, **kw):
        """x: tensor, or an _ln_src descriptor (the LayerNorm in front of the Linear is applied where the conv stages its input)."""
        Cn = lin.weight.shape[1]
        xf = x if isinstance(x, dict) else x.view(1, 1, -1, Cn)
        npx = out.numel() // lin.weight.shape[0]
        self.conv([xf], lin.weight, lin.bias, out.view(1, 1, npx, -1), Hin=1, Win=npx, **kw)

    def _lin_bwd(self, lin, x, dy, dx, ln_bwd=None):
        """ln_bwd = (norm, residual, out) as in _mlp_bwd: returns True when the data-gradient conv finished the LayerNorm backward (`dx`
        is then untouched, `out` holds d(input of the LayerNorm))."""
        Cn, Co = lin.weight.shape[1], dy.shape[-1]
        xf, dyf = (x if isinstance(x, dict) else x.view(1, 1, -1, Cn)), dy.view(1, 1, -1, Co)
        npx = dyf.shape[2]
        self.wgrad([xf], dyf, lin.weight, lin.bias, Hin=1, Win=npx)
        ep = self._ln_bwd_ep(x, *ln_bwd) if ln_bwd is not None else None
        if ep is not None:
            self.conv_T(dyf, lin.weight, ep[1], Hin=1, Win=npx, **ep[0])
            return True
        self.conv_T(dyf, lin.weight, dx.view(1, 1, npx, Cn), Hin=1, Win=npx)
        return False

    def nat_fwd(self, m, x, cx, tag):
        B, H, W, C = x.shape
        heads = m.att1.num_heads
        e = _A(x, B, H, W, C)
        self.conv3_fwd(m.patchembedding.patch_embeddings, x, e)
        # norm1 -> qkv and norm2 -> fc1 are ONE row each (SURVEY 8a "LN1+qkv"; core/modules.py:516-518): the LayerNorm is applied
        # where the Linear stages its input (hip.SRC_LN), n1 / n2 never cross HBM; the (mean, rstd) tables serve the weight gradients
        if self.fuse_ln:
            st1, st2 = _E(x, B * H * W, 2), _E(x, B * H * W, 2)
            n1 = self._ln_src(e, m.norm1, st1)
        else:
            n1 = _A(x, B, H, W, C)
            hip.ln_fwd(e, m.norm1.weight, m.norm1.bias, n1)
        qkv = _A(x, B, H, W, 3 * C)
        self._lin(m.att1.qkv, n1, qkv)
        o = _A(x, B, H, W, C)
        hip.na_fwd(qkv, m.att1.rpb, o, heads)
        a = _A(x, B, H, W, C)
        self._lin(m.att1.proj, o, a, residual=e.view(1, 1, -1, C))
        if self.fuse_ln:
            n2 = self._ln_src(a, m.norm2, st2)
        else:
            n2 = _A(x, B, H, W, C)
            hip.ln_fwd(a, m.norm2.weight, m.norm2.bias, n2)
        y = _A(x, B, H, W, C)
        a1, drop = self._mlp_fwd(m.mlp, n2, a, y, tag)
        if cx is not None:
            cx.t[m] = dict(x=x, e=e, n1=n1, qkv=qkv, o=o, a=a, n2=n2, a1=a1, drop=drop)
        return y

    def nat_bwd(self, m, dy, cx):
        S = cx.t[m]
        x, e, n1, qkv, o, a, n2 = S["x"], S["e"], S["n1"], S["qkv"], S["o"], S["a"], S["n2"]
        B, H, W, C = x.shape
        G = self.G
        da = _A(x, B, H, W, C)
        dn2 = self._mlp_bwd(m.mlp, n2, S["a1"], dy, S["drop"], ln_bwd=(m.norm2, dy, da))
        if dn2 is not None:     # (else: the last conv of the Mlp backward wrote da -- LayerNorm backward in its epilogue)
            hip.ln_bwd(a, m.norm2.weight, dn2, dy, da, G[m.norm2.weight], G[m.norm2.bias])
        do = _A(x, B, H, W, C)
        self._lin_bwd(m.att1.proj, o, da, do)
        dqkv = _A(x, B, H, W, 3 * C)
        hip.na_bwd(qkv, m.att1.rpb, do, dqkv, G[m.att1.rpb], m.att1.num_heads)
        if self.probe is not None:
            self.probe("nat_bwd:na", m, dict(qkv=qkv, do=do, dqkv=dqkv, rpb=m.att1.rpb, heads=m.att1.num_heads))
        dn1 = do
        de = _A(x, B, H, W, C)
        if not self._lin_bwd(m.att1.qkv, n1, dqkv, dn1, ln_bwd=(m.norm1, da, de)):
            hip.ln_bwd(e, m.norm1.weight, dn1, da, de, G[m.norm1.weight], G[m.norm1.bias])
        dx = _A(x, B, H, W, C)
        self.conv3_bwd(m.patchembedding.patch_embeddings, x, de, dx=dx)
        if self.probe is not None:
            self.probe("nat_bwd", m, dict(dy=dy, dn2=dn2, da=da, dqkv=dqkv, dn1=dn1, de=de, dx=dx, qkv=qkv, o=o, e=e))
        return dx

    def gft_fwd(self, m, catp, cx, tag):
        B, h, w, C = catp.shape
        N = h * w
        heads = m.attention.num_heads
        e = _A(catp, B, h, w, C)
        self.conv3_fwd(m.patchembedding.patch_embeddings, catp, e)
        if self.fuse_ln:      # (as in nat_fwd: norm1 -> qkv, norm2 -> fc1 fused; core/modules.py:343-344)
            st1, st2 = _E(catp, B * N, 2), _E(catp, B * N, 2)
            n1 = self._ln_src(e, m.norm1, st1)
        else:
            n1 = _A(catp, B, N, C)
            hip.ln_fwd(e, m.norm1.weight, m.norm1.bias, n1)
        qkv = _A(catp, B, N, 3 * C)
        self._lin(m.attention.qkv, n1, qkv)
        o = _A(catp, B, N, C)
        lse = _E(catp, B, heads, N)
        hip.gattn_fwd(qkv, o, lse, heads)
        a = _A(catp, B, N, C)
        self._lin(m.attention.proj, o, a, residual=e.view(1, 1, -1, C))
        if self.fuse_ln:
            n2 = self._ln_src(a, m.norm2, st2)
        else:
            n2 = _A(catp, B, N, C)
            hip.ln_fwd(a, m.norm2.weight, m.norm2.bias, n2)
        y = _A(catp, B, N, C)
        a1, drop = self._mlp_fwd(m.mlp, n2, a, y, tag)
        cv = m.conv[0]
        x5 = _A(catp, B, h, w, cv.weight.shape[0])
        self._lin(cv, y, x5)
        if cx is not None:
            cx.t[m] = dict(catp=catp, e=e, n1=n1, qkv=qkv, o=o, lse=lse, a=a, n2=n2, a1=a1, drop=drop, y=y)
        return x5

    def gft_bwd(self, m, dx5, cx):
        S = cx.t[m]
        catp, e, n1, qkv, o, a, n2, y = S["catp"], S["e"], S["n1"], S["qkv"], S["o"], S["a"], S["n2"], S["y"]
        B, h, w, C = catp.shape
        N = h * w
        G = self.G
        heads = m.attention.num_heads
        dy = _A(catp, B, N, C)
        self._lin_bwd(m.conv[0], y, dx5, dy)
        dn2 = self._mlp_bwd(m.mlp, n2, S["a1"], dy, S["drop"])
        da = _A(catp, B, N, C)
        hip.ln_bwd(a, m.norm2.weight, dn2, dy, da, G[m.norm2.weight], G[m.norm2.bias])
        do = _A(catp, B, N, C)
        self._lin_bwd(m.attention.proj, o, da, do)
        dqkv = _A(catp, B, N, 3 * C)
        delta = _E(catp, B, heads, N)
        hip.gattn_bwd(qkv, o, do, S["lse"], dqkv, delta, heads)
        dn1 = do
        self._lin_bwd(m.attention.qkv, n1, dqkv, dn1)
        de = _A(catp, B, h, w, C)
        hip.ln_bwd(e, m.norm1.weight, dn1, da, de, G[m.norm1.weight], G[m.norm1.bias])
        dcat = _A(catp, B, h, w, C)
        self.conv3_bwd(m.patchembedding.patch_embeddings, catp, de, dx=dcat)
        return dcat

    # ------------------------------------------------------------------ decoder up rows (A10)
    def up_fwd(self, seq, x, skip, cx):
        """conv3x3(bilinear_x2(x)) + skip"""
        B, h, w, C = x.shape
        conv = seq[1]
        out = _A(x, B, 2 * h, 2 * w, conv.weight.shape[0])
        if self.fuse_up:
            self.conv3_up_fwd(conv, x, out, residual=skip)
            if cx is not None:
                cx.t[seq] = dict(up=None, x=x)
            return out
        up = _A(x, B, 2 * h, 2 * w, C)
        hip.up2_fwd(x, up)
        self.conv3_fwd(conv, up, out, residual=skip)
        if cx is not None:
            cx.t[seq] = dict(up=up)
        return out

    def up_bwd(self, seq, dt, cx, xshape):
        up = cx.t[seq]["up"]
        if up is None:
            dx = _A(dt, *xshape)
            self.conv3_up_bwd(seq[1], cx.t[seq]["x"], dt, dx)
            return dx
        dup = _A(dt, *up.shape)
        self.conv3_bwd(seq[1], up, dt, dx=dup)
        dx = _A(dt, *xshape)
        hip.up2_bwd(dup, dx)
        return dx


class GradSlot:
    __slots__ = ("g",)

    def __init__(self):
        self.g = None
