"""``LM_Net`` -- drop-in replacement of the reference's ``core.LM_Net.LM_Net`` (core/LM_Net.py:5-123).

Same constructor, ``forward(x) -> logits`` and ``structural_reparam()``; same ``state_dict`` keys and
shapes (766 keys in train form, 510 after deploy), so ``train.py`` runs unchanged with
``from lm_net_amd import LM_Net`` in place of ``from core.LM_Net import LM_Net``.  The arithmetic runs
exclusively on the hand-written HIP kernels of ``liblmnet_hip.so``; a CPU tensor or a missing library
raises -- there is no eager/PyTorch fallback path.
"""
import torch
import torch.nn as nn

from . import hip
from .engine import Arena, Ctx, Engine, GradSlot, _A, _E, _Z
from .hip import V
from .modules import (GFT, NUM_HEADS, M2Skip, M3Skip, NeighborhoodTransformer, PyramidPool, ReparamConv, _conv, stage)

# backward-completion order of the top-level blocks: the flat gradient buffer is laid out in this
# order so that a data-parallel bucket (a contiguous slice) is complete as early as possible.
BACKWARD_ORDER = ["output_layer", "dconv4", "up4", "dconv3", "up3", "dconv2", "up2", "dconv1", "up1",
                  "natt4", "natt3", "natt2", "natt1", "skip4", "skip3", "skip2", "skip1", "gft",
                  "down4", "conv4", "down3", "conv3", "down2", "conv2", "down1", "conv1"]


class _LMNetFunction(torch.autograd.Function):
    """The whole network as one autograd node (explicit forward/backward kernel schedules)."""

    @staticmethod
    def forward(ctx, x, model, *params):
        cx = Ctx() if model._save_tape else None
        out = model._forward_impl(x, cx)
        ctx.model, ctx.cx, ctx.x_req = model, cx, x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        dx, grads = model._backward_impl(ctx.cx, dlogits.contiguous(), ctx.x_req)
        ctx.cx = None
        return (dx, None) + tuple(grads)


class _GraphedStep:
    """One captured training step for a fixed input shape: forward and backward hipGraphs sharing a memory pool."""

    def __init__(self):
        self.fwd = self.bwd = None
        self.x = self.out = self.dlogits = None
        self.cx = None
        self.flat = None          # static flat gradient buffer (views = the returned gradients)
        self.grads = None
        self.warm = 0


class _LMNetGraphFunction(torch.autograd.Function):
    """The captured step as an autograd node: forward = replay of the forward graph on a static input copy,
    backward = replay of the backward graph on a static copy of dlogits."""

    @staticmethod
    def forward(ctx, x, model, gs, *params):
        gs.x.copy_(x)
        gs.fwd.replay()
        ctx.model, ctx.gs = model, gs
        return gs.out.clone()

    @staticmethod
    def backward(ctx, dlogits):
        model, gs = ctx.model, ctx.gs
        gs.dlogits.copy_(dlogits)
        gs.bwd.replay()
        model._grad_flat = gs.flat
        if model.grad_begin_hook is not None:       # data parallel: one bucketed all-reduce pass after the replay
            model.grad_begin_hook(gs.flat)
            model.grad_ready_hook(0, gs.flat.numel())
            model.grad_finish_hook()
        grads = gs.grads
        p0 = model._param_list()[0]
        if p0.grad is not None and p0.grad.data_ptr() == grads[0].data_ptr():
            grads = [g.clone() for g in grads]      # accumulation into .grad that aliases the static buffer
        return (None, None, None) + tuple(grads)


class _gc_paused:
    """No cyclic garbage collection inside a hipGraph capture: the capture runs a whole pass of Python (thousands of allocations),
    and a collection in the middle of it may finalise objects of EARLIER models -- graphs, plans, streams, device tensors -- whose
    destructors call into HIP, which aborts the process while a stream is capturing."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class _PlannedStep:
    """One recorded pass pair (forward, backward) for a fixed input shape and mode: two lmn plans over one arena."""

    def __init__(self):
        self.warm = 0
        self.need_fwd = self.need_bwd = 0   # floats of arena a pass needs (measured on the eager warm-up steps)
        self.arena = None
        self.fwd = self.bwd = None
        self.x = self.out = self.dlogits = None
        self.cx = None
        self.flat = None           # static flat gradient buffer
        self.views = None          # per-parameter views of it (direct_grads)
        self.G = None
        self.sizes = None
        self.shapes = None
        self.marks = []            # backward: (op index, block name, producer streams) at every finished gradient block
        self.pending = False       # a forward whose backward has not run yet
        self.rec = None            # hip.Plan being recorded by the backward (its size marks the gradient-block boundaries)
        self.key = None


class _LMNetPlanFunction(torch.autograd.Function):
    """The recorded step as an autograd node: forward / backward = one lmn_plan_run each (per gradient bucket when a
    data-parallel reducer is attached)."""

    @staticmethod
    def forward(ctx, x, model, ps, *params):
        ctx.model, ctx.ps = model, ps
        return model._plan_forward(ps, x)

    @staticmethod
    def backward(ctx, dlogits):
        return (None, None, None) + tuple(ctx.model._plan_backward(ctx.ps, dlogits))


class _LMNetPlanDirectFunction(torch.autograd.Function):
    """enable_plans(direct_grads=True): the node has TWO tensor inputs (the batch and a one-element anchor) instead of the
    batch and ~510 parameters, and its backward assigns the gradient views to `.grad` itself.  The autograd engine then runs
    no AccumulateGrad node per parameter (2.5 ms of host time per step: at batch 1 the step is host-bound)."""

    @staticmethod
    def forward(ctx, x, anchor, model, ps):
        ctx.model, ctx.ps = model, ps
        return model._plan_forward(ps, x)

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        grads = model._plan_backward(ctx.ps, dlogits, direct=True)
        if grads is not None:                      # (None: accumulated in place into the .grad tensors that alias the buffer)
            for p, g in zip(model._param_list(), grads):
                if not p.requires_grad:
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad += g
        return None, None, None, None


class LM_Net(nn.Module):
    def __init__(self, channel, n_classes=2, filters=[12, 24, 48, 96, 192], deep_supervision=False, na_kernel_size=3):
        # (na_kernel_size is not in the reference signature, core/LM_Net.py:6: its four NAT blocks are built with kernel_size 3,
        #  core/modules.py:509 -- other odd sizes run the run-time-K neighborhood-attention kernels, csrc/na.hip)
        super().__init__()
        f = list(filters)
        assert all(c % NUM_HEADS == 0 for c in f[:4]) and sum(f) % NUM_HEADS == 0, \
            "filters[0..3] and sum(filters) must be multiples of 12 (12 attention heads)"
        self.deep_supervision = deep_supervision      # stored, never read (as in the reference)
        self.filters = f
        self.channel, self.n_classes = channel, n_classes
        self.conv1 = stage(channel, f[1], f[0]); self.down1 = nn.Sequential(_conv(f[0], f[1], 3, 2))
        self.conv2 = stage(f[1], f[2], f[1]);    self.down2 = nn.Sequential(_conv(f[1], f[2], 3, 2))
        self.conv3 = stage(f[2], f[3], f[2]);    self.down3 = nn.Sequential(_conv(f[2], f[3], 3, 2))
        self.conv4 = stage(f[3], f[4], f[3]);    self.down4 = nn.Sequential(_conv(f[3], f[4], 3, 2))
        self.dconv1 = stage(f[3], f[4], f[3])
        self.dconv2 = stage(f[2], f[3], f[2])
        self.dconv3 = stage(f[1], f[2], f[1])
        self.dconv4 = stage(f[0], f[1], f[0])
        self.pyramidpool = PyramidPool()
        self.gft = GFT(sum(f), 2, f[4], NUM_HEADS)
        up = lambda ci, co: nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True), _conv(ci, co, 3))
        self.up1, self.up2, self.up3, self.up4 = up(f[4], f[3]), up(f[3], f[2]), up(f[2], f[1]), up(f[1], f[0])
        self.skip1 = M2Skip([f[2], f[3]], "bottom")
        self.skip2 = M3Skip([f[1], f[2], f[3]])
        self.skip3 = M3Skip([f[0], f[1], f[2]])
        self.skip4 = M2Skip([f[0], f[1]], "top")
        self.natt1 = NeighborhoodTransformer(f[3], NUM_HEADS, na_kernel_size)
        self.natt2 = NeighborhoodTransformer(f[2], NUM_HEADS, na_kernel_size)
        self.natt3 = NeighborhoodTransformer(f[1], NUM_HEADS, na_kernel_size)
        self.natt4 = NeighborhoodTransformer(f[0], NUM_HEADS, na_kernel_size)
        self.output_layer = nn.Conv2d(f[0], n_classes, 1)
        self._engine = Engine(self)
        self._grad_flat = None
        self._grad_layout = None
        self._recording = None
        self._pend = None          # gradient blocks enqueued but not handed on yet: [lo, hi) of the flat buffer (see _done)
        self._emitted = False
        self._save_tape = False
        self._keep_taps = False
        self._taps = None
        self.use_graphs = False   # capture the training step into hipGraphs (see enable_graphs)
        self._graphs = {}
        # precision of the pass: None = follow torch.autocast (bf16 / fp16 autocast -> "bf16", else "fp32"), or pinned:
        #   "fp32"      fp32 activations, exact fp32 MFMA                     (BASELINE configs[1], the default)
        #   "bf16"      bf16 ACTIVATION STORAGE + bf16 MFMA operands; fp32 accumulators, BatchNorm / LayerNorm / softmax
        #               statistics, master weights and weight gradients     (BASELINE configs[2])
        #   "bf16-mma"  fp32 activation storage, bf16 MFMA operands only
        # (custom autograd nodes are opaque to autocast, so the module reads the autocast state itself: the reference's
        #  AMP branch, utils/train_eval_utils.py:130-138, keeps working)
        self.compute_dtype = None
        self.use_plans = False    # replay recorded C-side schedules (see enable_plans)
        self._direct_grads = False
        self._plans = {}
        self._head_bias4 = None
        # data-parallel hooks (ddp.py): begin(flat) at the start of backward, ready(lo, hi) when the
        # flat-gradient slice [lo,hi) has been enqueued, finish() at the end of backward
        self.grad_begin_hook = None
        self.grad_ready_hook = None
        self.grad_finish_hook = None

    # ------------------------------------------------------------------ public API of the reference
    def structural_reparam(self):
        for m in list(self.modules()):
            if hasattr(m, "switch_to_deploy"):
                m.switch_to_deploy()
        self._grad_layout = None
        self._graphs = {}
        self._plans = {}
        self.__dict__["_param_cache"] = None
        self.__dict__["_bn_cache"] = None

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("lm_net_amd.LM_Net runs on the HIP device only (input is on %s); there is no CPU path. "
                               "Move the model and the batch to 'cuda'." % x.device)
        if x.dim() != 4 or x.shape[1] != self.channel:
            raise ValueError("expected input [B,%d,H,W], got %s" % (self.channel, tuple(x.shape)))
        if x.shape[2] % 16 or x.shape[3] % 16 or x.shape[2] < 32 or x.shape[3] < 32:
            raise ValueError("H and W must be multiples of 16 and >= 32 (got %dx%d)" % (x.shape[2], x.shape[3]))
        hip.load()
        # the step is as long as the kernel chain on the caller's stream: its conv / depthwise kernels run at a raised wave priority
        hip.set_priority_stream(torch.cuda.current_stream(x.device), self._engine.prio_main)
        if hip.get_deterministic() != self._engine.deterministic:     # the switch is process-wide: every model follows it
            self._engine.set_deterministic(hip.get_deterministic())
        params = self._param_list()
        if params and not params[0].is_cuda:
            raise RuntimeError("lm_net_amd.LM_Net: parameters are on %s; call model.to('cuda')" % params[0].device)
        x = x.float().contiguous() if x.dtype != torch.float32 or not x.is_contiguous() else x
        self._engine.mma, self._engine.act_dtype = self._precision()
        self._save_tape = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        if self.use_plans and not x.requires_grad and not self._keep_taps and (self._save_tape or not torch.is_grad_enabled()):
            ps = self._plan_for(x)
            if ps is not None:
                if self._save_tape:
                    if self._direct_grads:
                        return _LMNetPlanDirectFunction.apply(x, self._grad_anchor(x.device), self, ps)
                    return _LMNetPlanFunction.apply(x, self, ps, *params)
                return self._plan_forward(ps, x)
        graphs = self.use_graphs and not self._engine.deterministic       # (no capture in deterministic mode: see enable_graphs)
        if graphs and self.training and self._save_tape and not x.requires_grad and not self._keep_taps:
            gs = self._graph_for(x)
            if gs is not None:
                return _LMNetGraphFunction.apply(x, self, gs, *params)
        if graphs and not self.training and not self._save_tape and not self._keep_taps:
            out = self._infer_replay(x)
            if out is not None:
                return out
        return _LMNetFunction.apply(x, self, *params)

    @property
    def deterministic(self):
        """Bit-reproducible passes: every cross-block float reduction of the library is summed in a fixed order
        (include/lmnet_hip.h, lmn_set_deterministic -- a PROCESS-WIDE switch) and the squeeze-excite gate is computed by its own
        launch instead of by the last block of the depthwise forward.  Slower (extra launches); default off, or LMN_DETERMINISTIC=1."""
        return self._engine.deterministic

    @deterministic.setter
    def deterministic(self, on):
        self._engine.set_deterministic(bool(on))

    def _precision(self):
        """(matrix-core operand type, activation storage type) of the next pass."""
        cd = self.compute_dtype
        if cd is None:
            amp = torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') in (torch.bfloat16, torch.float16)
            cd = "bf16" if amp else "fp32"
        if cd in ("bf16", "bfloat16", torch.bfloat16):       # bf16 activation storage + bf16 MFMA operands
            return hip.BF16, torch.bfloat16
        if cd in ("bf16-mma", "bf16_mma"):                   # fp32 activation storage, bf16 MFMA operands only
            return hip.BF16, torch.float32
        if cd in ("fp32", "f32", "float32", torch.float32):
            return hip.F32, torch.float32
        raise ValueError("LM_Net.compute_dtype must be None, 'fp32', 'bf16' or 'bf16-mma' (got %r)" % (cd,))

    # ------------------------------------------------------------------ recorded C-side schedules (lmn_plan_*)
    def enable_plans(self, on=True, direct_grads=False):
        """Run every pass of a repeated input shape as ONE call into the library (include/lmnet_hip.h, lmn_plan_*): the
        first two passes of a (shape, mode) run launch by launch and size the arena, the third is recorded while it runs,
        later ones are `lmn_plan_run` replays -- same kernels, same four HIP streams and cross-stream events as the
        host-launched schedule (unlike a hipGraph capture, which cannot carry the stream forks), ~2 ms of host time per
        step instead of 16-20.  Contract while enabled: fixed parameter storage; the returned gradients are views of one
        static buffer (as `zero_grad(set_to_none=True)` expects; a `.grad` that still aliases it is accumulated into
        correctly); data-parallel buckets are reported between plan segments; dropout draws from a device-side counter.
        direct_grads=True: the backward node writes `.grad` itself instead of returning ~510 gradients to the autograd engine
        (one AccumulateGrad node each, 2.5 ms of host time per step).  `loss.backward()` then behaves as before (including
        accumulation into existing `.grad`); `torch.autograd.grad(loss, parameters)` and tensor hooks on parameters do not see
        these gradients -- hence opt-in."""
        self.use_plans = bool(on)
        self._direct_grads = bool(on and direct_grads)
        if not on:
            self._plans = {}
            self._engine.seed_ctr = None if not self.use_graphs else self._engine.seed_ctr
        return self

    def _grad_anchor(self, dev):
        a = self.__dict__.get("_anchor")
        if a is None or a.device != dev:
            a = torch.zeros(1, device=dev, requires_grad=True)      # what makes the direct node part of the autograd graph
            self.__dict__["_anchor"] = a
        return a

    def _plan_for(self, x):
        taped = bool(self._save_tape)          # a backward may follow (training, or eval-mode BatchNorm with gradients enabled)
        key = self._plan_key(x)
        ps = self._plans.get(key)
        if ps is None:
            ps = self._plans[key] = _PlannedStep()
            ps.key = key
        if ps.fwd is not None:
            if ps.pending and taped:
                return None            # a second forward before the backward of the first: that one runs launch by launch
            return ps
        ps.warm += 1
        if ps.warm <= 2:
            return None
        if taped and ps.need_bwd == 0:
            return None                # no backward has been seen for this shape yet: keep warming up
        eng = self._engine
        if eng.seed_ctr is None:
            eng.seed_ctr = torch.zeros(1, device=x.device, dtype=torch.int32)
        return ps

    def _plan_key(self, x):
        """(shape, device, mode, precision, launch stream): recorded launches carry the stream that was current while recording,
        so a call under another current stream gets its own plan instead of racing with this one."""
        return (tuple(x.shape), x.device, self.training, self._save_tape, self._engine.pm(),
                torch.cuda.current_stream(x.device).cuda_stream, self._engine.deterministic)

    def _drop_plan(self, ps):
        """A recording failed: forget the (possibly truncated) plans of this shape; it runs launch by launch from now on."""
        for k, v in list(self._plans.items()):
            if v is ps:
                del self._plans[k]
        ps.fwd = ps.bwd = None

    def _plan_forward(self, ps, x):
        eng = self._engine
        with torch.cuda.device(x.device):
            if ps.fwd is None:                          # record while running
                L = self._ensure_grad_layout()
                # the arena holds the backward's tensors whenever a tape is kept (eval-mode BatchNorm with gradients too)
                need = ps.need_fwd + (ps.need_bwd + L["total"] if self._save_tape else 0)
                ps.arena = Arena(int(need * 1.02) + (1 << 16), x.device)
                ps.x = ps.arena.alloc(tuple(x.shape))
                ps.x.copy_(x)
                ps.cx = Ctx() if self._save_tape else None
                plan = hip.Plan()
                eng.arena, eng.planning = ps.arena, True
                ok = False
                try:
                    plan.record_begin()
                    out = self._forward_impl(ps.x, ps.cx)
                    ok = True
                finally:
                    plan.record_end()
                    eng.arena, eng.planning = None, False
                    if not ok:
                        self._drop_plan(ps)
                ps.out, ps.fwd = out, plan              # (assigned only once the recording is complete)
                ps.keep = [eng.packs_fwd.table, eng.packs_bwd.table] + list(eng.reduce_tabs)    # device job tables the recorded launches read
            else:
                ps.x.copy_(x)
                self._step_bookkeeping()
                if ps.cx is not None:
                    ps.cx.training = self.training
                    ps.cx.mma, ps.cx.act_dtype = eng.mma, eng.act_dtype
                ps.fwd.run()
            ps.pending = ps.cx is not None
            return ps.out.clone()

    def _plan_backward(self, ps, dlogits, direct=False):
        eng = self._engine
        L = self._ensure_grad_layout()
        params = self._param_list()
        with torch.cuda.device(dlogits.device):
            lo_b = ps.flat.data_ptr() if ps.flat is not None else 0
            hi_b = lo_b + 4 * ps.flat.numel() if ps.flat is not None else 0
            alias = [p.grad is not None and lo_b <= p.grad.data_ptr() < hi_b for p in params] if ps.flat is not None else []
            acc = any(alias)
            old = ps.flat.clone() if acc else None      # some .grad still aliases the static buffer: keep its values
            if ps.bwd is None:                          # record while running
                ps.dlogits = ps.arena.alloc(tuple(dlogits.shape))
                ps.dlogits.copy_(dlogits)
                ps.flat = ps.arena.alloc((L["total"],))
                ps.G = {}
                for p in L["order"]:
                    a, b = L["offs"][id(p)]
                    ps.G[p] = ps.flat[a:b].view(p.shape)
                ps.sizes = [(L["offs"][id(p)][0], p.numel(), tuple(p.shape)) for p in params]
                plan = hip.Plan()
                ps.rec = plan
                ps.marks = []
                eng.arena, eng.planning = ps.arena, True
                ok = False
                try:
                    plan.record_begin()
                    self._backward_impl(ps.cx, ps.dlogits, False, plan=ps)
                    ok = True
                finally:
                    plan.record_end()
                    eng.arena, eng.planning = None, False
                    ps.rec = None
                    if not ok:
                        self._drop_plan(ps)
                ps.bwd = plan
                ps.keep += [eng.packs_fwd.table, eng.packs_bwd.table] + list(eng.reduce_tabs)
            else:
                ps.dlogits.copy_(dlogits)
                self._grad_flat = ps.flat
                if self.grad_begin_hook is not None:
                    self.grad_begin_hook(ps.flat)
                if self.grad_ready_hook is None:
                    ps.bwd.run()
                else:
                    lo = 0
                    for idx, rng, streams in ps.marks:
                        ps.bwd.run(lo, idx)
                        lo = idx
                        self.grad_ready_hook(rng[0], rng[1], streams)
                    ps.bwd.run(lo, -1)
                if self.grad_finish_hook is not None:
                    self.grad_finish_hook()
            ps.pending = False
            # fresh views every time: AccumulateGrad then takes them as .grad without a copy.  A parameter whose .grad still
            # aliases the static buffer gets its gradient from a copy (autograd then adds it to the kept values, see `old`)
            if old is None:
                if direct:                              # the same view objects every step: assigned to .grad by the node
                    if ps.views is None:
                        ps.views = [ps.flat[a:a + n].view(shp) for a, n, shp in ps.sizes]
                    return ps.views
                return [ps.flat[a:a + n].view(shp) for a, n, shp in ps.sizes]
            if direct and all(al or not p.requires_grad for al, p in zip(alias, params)):
                ps.flat.add_(old)                       # every .grad aliases the buffer: accumulate in place, one launch
                return None
            new = ps.flat.clone()
            ps.flat.copy_(old)                          # the aliased .grad tensors keep their accumulated values
            return [new[a:a + n].view(shp) for a, n, shp in ps.sizes]

    # ------------------------------------------------------------------ hipGraph capture of the training step
    def enable_graphs(self, on=True):
        """Run training steps as two hipGraph replays (forward, backward) per input shape instead of ~1700 launches
        (and eval / no-grad forwards as one replay, see _infer_replay):
        the step at batch 8 / 352x352 carries ~8 ms of per-launch cost.  The first two steps of a shape run eagerly
        (warm-up: workspaces, pack plans, allocator), the third is captured.  Contract while enabled: fixed parameter
        storage, `zero_grad(set_to_none=True)` semantics (returned gradients are views of one static buffer; they
        are cloned if a `.grad` still aliases it), data-parallel gradients are all-reduced after the backward
        replay instead of bucket by bucket inside it.  Dropout draws a new mask per replay from a device-side
        counter."""
        if on and self._engine.deterministic:
            raise RuntimeError("enable_graphs(): hipGraph capture is not available in deterministic mode (its slot scratch cannot be "
                               "allocated inside a capture); use enable_plans()")
        self.use_graphs = bool(on)
        if not on:
            self._graphs = {}
            self._engine.seed_ctr = None
        return self

    def _infer_replay(self, x):
        """Eval / no-grad forward of a fixed input shape as ONE hipGraph replay (the
        launch-minimal inference schedule).  Two eager calls of a shape warm the workspaces up, the third is captured;
        weight packing and BatchNorm folding are kernels inside the graph, so parameter / running-stat updates between
        calls are honoured; `structural_reparam()` drops the graphs.  Returns None while warming up."""
        key = ("infer", tuple(x.shape), x.device, self._engine.pm())
        gs = self._graphs.get(key)
        if gs is None:
            gs = self._graphs[key] = _GraphedStep()
        if gs.fwd is None:
            gs.warm += 1
            if gs.warm <= 2:
                return None
            eng = self._engine
            gs.x = x.clone()
            torch.cuda.synchronize(x.device)
            eng.capturing = True
            try:
                gs.fwd = torch.cuda.CUDAGraph()
                with _gc_paused(), torch.cuda.graph(gs.fwd):
                    gs.out = self._forward_impl(gs.x, Ctx())
            finally:
                eng.capturing = False
        gs.x.copy_(x)
        gs.fwd.replay()
        return gs.out.clone()

    def _graph_for(self, x):
        key = (tuple(x.shape), x.device, self._engine.pm())
        gs = self._graphs.get(key)
        if gs is None:
            gs = self._graphs[key] = _GraphedStep()
        if gs.fwd is not None:
            return gs
        gs.warm += 1
        if gs.warm <= 2:
            return None                      # eager warm-up steps
        eng = self._engine
        if eng.seed_ctr is None:
            eng.seed_ctr = torch.zeros(1, device=x.device, dtype=torch.int32)
        params = self._param_list()
        gs.x = x.clone()
        # (no extra warm-up pass here: the two eager steps already sized the workspaces / pack plans / zero pools, and
        #  another training-mode forward would update the BatchNorm running statistics once too often)
        torch.cuda.synchronize(x.device)
        hooks = (self.grad_begin_hook, self.grad_ready_hook, self.grad_finish_hook)
        self.grad_begin_hook = self.grad_ready_hook = self.grad_finish_hook = None   # no collectives inside the capture
        eng.capturing = True
        try:
            gs.fwd = torch.cuda.CUDAGraph()
            with _gc_paused(), torch.cuda.graph(gs.fwd):
                gs.cx = Ctx()
                gs.out = self._forward_impl(gs.x, gs.cx)
            gs.dlogits = torch.zeros_like(gs.out)
            gs.bwd = torch.cuda.CUDAGraph()
            with _gc_paused(), torch.cuda.graph(gs.bwd, pool=gs.fwd.pool()):
                _, grads = self._backward_impl(gs.cx, gs.dlogits, False)
            gs.flat, gs.grads = self._grad_flat, grads
        finally:
            eng.capturing = False
            self.grad_begin_hook, self.grad_ready_hook, self.grad_finish_hook = hooks
        return gs

    def _param_list(self):
        """model.parameters() as a cached list (walking 300 sub-modules costs ~1.5 ms per call); the Parameter
        objects are stable -- only structural_reparam() changes the set, and it drops the cache."""
        c = self.__dict__.get("_param_cache")
        if c is None:
            c = self.__dict__["_param_cache"] = list(self.parameters())
        return c

    # ------------------------------------------------------------------ forward schedule (core/LM_Net.py:95-123)
    def _step_bookkeeping(self):
        eng = self._engine
        if self.training:
            eng.step += 1
            if eng.seed_ctr is not None:
                eng.seed_ctr += 0x2545F49          # odd increment: a fresh dropout stream per step, also under replay
            self._bump_num_batches_tracked()

    def _forward_impl(self, x, cx):
        eng = self._engine
        eng.training = self.training
        if cx is not None:
            cx.training = self.training             # the backward of THIS pass uses the BatchNorm mode it ran in
            cx.mma, cx.act_dtype = eng.mma, eng.act_dtype   # ... and the same arithmetic / storage types
        self._step_bookkeeping()
        with torch.cuda.device(x.device):
            eng.begin_pass(False, x.device)
            try:
                out = self._forward_body(x, cx)
                # weight-gradient-style launches of the forward (the moments of the z-path, on the weight-gradient stream): reduced and
                # JOINED here, so that the backward's lmn_reparam_wfin depends on them whatever stream / overlap setting it runs
                # with (the side stream is idle at the end of the forward: the join costs nothing)
                (eng.join_side if eng.fwd_join else eng.flush_reduce)(x.device)
            finally:
                nf = eng.alloc_floats
                eng.end_pass()
        if self.use_plans:
            key = self._plan_key(x)
            ps = self._plans.get(key)
            if ps is not None:
                ps.need_fwd = max(ps.need_fwd, nf)
                if cx is not None:
                    cx.plan_key = key
        return out

    def _forward_body(self, x, cx):
        eng = self._engine
        B, Cin, H, W = x.shape
        c4 = (Cin + 3) // 4 * 4
        dev = x.device
        xin = _A(dev, B, H, W, c4)
        hip.nchw_to_nhwc(x, xin)
        f = self.filters
        main = torch.cuda.current_stream(dev)
        fork = eng.branch_overlap and not eng.capturing
        bst = eng.branch_stream(dev) if fork else None
        sb = eng.slot_base                   # (four lmn event slots per engine: two models of one process do not share them)
        slots = {2: sb, 4: sb + 1, 6: sb + 2, 8: sb + 3}     # lmn event slot of each chain (by dropout tag)

        fstreams = [eng.branch_stream_n(dev, i) for i in eng.branch_map_f] if fork else []   # per level 0..3

        def chain(skip, nat, xs_in, tag):
            """Skip fuser + neighborhood-attention block of one level: needs only encoder outputs and is needed
            only by the decoder stage of its level, so it runs on the branch stream as soon as its inputs exist
            (level 0 -- the heavy one -- right after conv2) while the main stream goes down the encoder and
            through the GFT bottleneck, whose small feature maps cannot fill the GPU."""
            if not fork:
                xs = eng.skip_fwd(skip, xs_in, cx)
                return eng.nat_fwd(nat, xs, cx, tag=tag), None, xs
            bs = fstreams[{8: 0, 6: 1, 4: 2, 2: 3}[tag]]   # (LMN_BRANCH_MAP_F: the small-map chains, which the decoder needs first, beside the others)
            hip.stream_wait(bs, main)
            with eng.on_stream(bs):
                xs = eng.skip_fwd(skip, xs_in, cx)
                out = eng.nat_fwd(nat, xs, cx, tag=tag)
            hip.event_record(slots[tag], bs)           # the point the decoder stage of this level waits for
            return out, slots[tag], xs

        def need(res):
            out, ev, _ = res
            if ev is not None:
                hip.event_wait(ev, main)
                if eng.arena is None:
                    out.record_stream(main)
            return out

        x1 = eng.stage_fwd(self.conv1, xin, cx)
        xd1 = _A(dev, B, H // 2, W // 2, f[1]); eng.conv3_fwd(self.down1[0], x1, xd1, s=2)
        x2 = eng.stage_fwd(self.conv2, xd1, cx)
        r4 = chain(self.skip4, self.natt4, (x1, x2), 8) if fork else None
        xd2 = _A(dev, B, H // 4, W // 4, f[2]); eng.conv3_fwd(self.down2[0], x2, xd2, s=2)
        x3 = eng.stage_fwd(self.conv3, xd2, cx)
        r3 = chain(self.skip3, self.natt3, (x1, x2, x3), 6) if fork else None
        xd3 = _A(dev, B, H // 8, W // 8, f[3]); eng.conv3_fwd(self.down3[0], x3, xd3, s=2)
        x4 = eng.stage_fwd(self.conv4, xd3, cx)
        r1 = chain(self.skip1, self.natt1, (x3, x4), 2) if fork else None
        r2 = chain(self.skip2, self.natt2, (x2, x3, x4), 4) if fork else None
        # PyramidPool: mean-pool x1..x4 onto the 1/16 grid, down4 writes x_down4 straight into its slice
        h, w = H // 16, W // 16
        catp = _A(dev, B, h, w, sum(f))
        off = 0
        for t, fac in ((x1, 16), (x2, 8), (x3, 4), (x4, 2)):
            hip.avgpool_fwd(t, V(catp, off, t.shape[-1]), fac)
            off += t.shape[-1]
        eng.conv3_fwd(self.down4[0], x4, V(catp, off, f[4]), s=2)
        x5 = eng.gft_fwd(self.gft, catp, cx, tag=0)
        if not fork:    # the reference's order (core/LM_Net.py:107-116)
            r1 = chain(self.skip1, self.natt1, (x3, x4), 2)
            r2 = chain(self.skip2, self.natt2, (x2, x3, x4), 4)
            r3 = chain(self.skip3, self.natt3, (x1, x2, x3), 6)
            r4 = chain(self.skip4, self.natt4, (x1, x2), 8)
        x46 = need(r1)
        x6 = eng.stage_fwd(self.dconv1, eng.up_fwd(self.up1, x5, x46, cx), cx)
        x37 = need(r2)
        x7 = eng.stage_fwd(self.dconv2, eng.up_fwd(self.up2, x6, x37, cx), cx)
        x28 = need(r3)
        x8 = eng.stage_fwd(self.dconv3, eng.up_fwd(self.up3, x7, x28, cx), cx)
        x19 = need(r4)
        x9 = eng.stage_fwd(self.dconv4, eng.up_fwd(self.up4, x8, x19, cx), cx)
        if fork:
            for bs_ in {id(t): t for t in [bst] + fstreams}.values():
                hip.stream_wait(main, bs_)
        # segmentation head: computed on rows padded to a multiple of 4 (the packed weight's extra rows are zeros, the
        # bias lives in a persistent 4-vector), then NHWC -> NCHW keeps the first n_classes channels
        ncp = (self.n_classes + 3) // 4 * 4
        bh = self._head_bias(ncp)
        o4 = _A(dev, B, H, W, ncp)
        eng.conv([x9], self.output_layer.weight, bh, o4, Hin=H, Win=W, cout=ncp)
        logits = _E(dev, B, self.n_classes, H, W)
        hip.nhwc_to_nchw(o4, logits)
        if cx is not None:
            cx.t["act"] = dict(xin=xin, x1=x1, x2=x2, x3=x3, x4=x4, xd1=xd1, xd2=xd2, xd3=xd3, catp=catp, x5=x5, x6=x6,
                               x7=x7, x8=x8, x9=x9, ncp=ncp, shape=(B, H, W))
        self._taps = dict(x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, xs1=r1[2], xs2=r2[2], xs3=r3[2], xs4=r4[2], x46=x46, x37=x37,
                          x28=x28, x19=x19, x6=x6, x7=x7, x8=x8, x9=x9) if self._keep_taps else None
        return logits

    def _head_bias(self, ncp):
        """output_layer.bias in a persistent zero-padded [ncp] vector, refreshed by one small copy per pass."""
        b = self.output_layer.bias
        hb = self._head_bias4
        if hb is None or hb.device != b.device or hb.numel() != ncp:
            hb = self._head_bias4 = torch.zeros(ncp, device=b.device, dtype=torch.float32)
        hip.copy2d(b, hb, 1, self.n_classes, self.n_classes, ncp)
        return hb

    def _bump_num_batches_tracked(self):
        """All 84 `num_batches_tracked` buffers are views of ONE int64 tensor: one increment per step."""
        bns = self.__dict__.get("_bn_cache")
        if bns is None:
            bns = self.__dict__["_bn_cache"] = [m for m in self.modules()
                                                if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
        if not bns:
            return
        flat = getattr(self, "_nbt_flat", None)
        dev = bns[0].num_batches_tracked.device
        ok = flat is not None and flat.device == dev and flat.numel() == len(bns) and all(
            m.num_batches_tracked.data_ptr() == flat.data_ptr() + 8 * i for i, m in enumerate(bns))
        if not ok:
            flat = torch.stack([m.num_batches_tracked.detach().to(dev).long() for m in bns]).contiguous()
            for i, m in enumerate(bns):
                m.num_batches_tracked.data = flat[i]
            self._nbt_flat = flat
        flat += 1

    # ------------------------------------------------------------------ flat gradient buffer
    def _ensure_grad_layout(self):
        params = self._param_list()
        key = tuple(id(p) for p in params)
        if self._grad_layout is not None and self._grad_layout["key"] == key and \
                self._grad_layout["device"] == params[0].device:
            return self._grad_layout
        order, seen = [], set()
        for name in BACKWARD_ORDER:
            for p in getattr(self, name).parameters():
                if id(p) not in seen:
                    seen.add(id(p)); order.append(p)
        for p in params:
            if id(p) not in seen:
                seen.add(id(p)); order.append(p)
        offs, pos = {}, 0
        for p in order:
            offs[id(p)] = (pos, pos + p.numel())
            pos += (p.numel() + 3) // 4 * 4            # keep every view 16-byte aligned
        blocks, lo = {}, 0
        for name in BACKWARD_ORDER:
            ps = list(getattr(self, name).parameters())
            hi = max([offs[id(p)][0] + (p.numel() + 3) // 4 * 4 for p in ps], default=lo)
            blocks[name] = (lo, hi)
            lo = hi
        self._grad_layout = dict(key=key, device=params[0].device, order=order, offs=offs, total=pos, blocks=blocks)
        return self._grad_layout

    def _new_grads(self):
        L = self._ensure_grad_layout()
        flat = torch.zeros(L["total"], device=L["device"], dtype=torch.float32)
        G = {}
        for p in L["order"]:
            a, b = L["offs"][id(p)]
            G[p] = flat[a:b].view(p.shape)
        return flat, G

    def _done(self, name, force=False):
        """Block `name` of the flat gradient buffer has been enqueued.  Blocks are handed on in BUCKETS (>= 1 MB for the first,
        >= 4 MB after it, and whenever the issuing stream changes): one batched launch sums the deferred K-split partials of the
        bucket's weight gradients (engine.flush_reduce), then the data-parallel reducer / the plan recorder is told."""
        plan = self._recording
        if self.grad_ready_hook is None and plan is None:
            return
        lo, hi = self._grad_layout["blocks"][name]
        if self._pend is None:
            self._pend = [lo, hi]
        else:
            self._pend[1] = hi
        cap = (1 << 18) if not self._emitted else (1 << 20)        # floats
        # the encoder blocks come last: their weight-gradient reductions / follow-ups are handed on block by block, so that only the
        # last block's are left when the compute chain ends (the step's tail is the weight-gradient stream finishing)
        tail = name in ("down4", "conv4", "down3", "conv3", "down2", "conv2", "down1", "conv1")
        if force or tail or self._pend[1] - self._pend[0] >= cap:
            self._emit_done()

    def _emit_done(self):
        if self._pend is None:
            return
        lo, hi = self._pend
        self._pend = None
        self._emitted = True
        plan = self._recording
        dev = self._grad_flat.device
        if self._grad_flat.is_cuda:
            self._engine.flush_reduce(dev)
        # the block's weight gradients run on the side stream of the current stream: the collective waits for
        # that stream, the compute chain does not (it joins once, at the end of backward)
        cur = torch.cuda.current_stream(dev) if self._grad_flat.is_cuda else None
        ent = self._engine.sides.get(cur.cuda_stream) if cur is not None else None
        streams = ([cur] if cur is not None else []) + ([ent[0]] if ent is not None else [])
        if plan is not None:                     # replays call the hook between plan segments
            plan.marks.append((plan.rec.size(), (lo, hi), streams))
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(lo, hi, streams)

    # ------------------------------------------------------------------ backward schedule
    def _backward_impl(self, cx, dlogits, need_dx, plan=None):
        if cx is None:
            raise RuntimeError("backward called on a forward pass that saved no state")
        eng = self._engine
        eng.training = getattr(cx, "training", eng.training)     # the mode of the forward this tape belongs to
        eng.mma, eng.act_dtype = getattr(cx, "mma", eng.mma), getattr(cx, "act_dtype", eng.act_dtype)
        with torch.cuda.device(dlogits.device):
            if plan is None:
                flat, G = self._new_grads()
            else:
                flat, G = plan.flat, plan.G
            self._grad_flat = flat
            self._recording = plan
            self._pend, self._emitted = None, False
            if self.grad_begin_hook is not None:
                self.grad_begin_hook(flat)
            eng.G = G
            eng.begin_pass(True, dlogits.device)
            try:
                if plan is not None:
                    hip.fill(flat, 0.0)                          # recorded: every replay starts from zero gradients
                dx = self._backward_body(cx, dlogits, need_dx, G)
                self._emit_done()
                eng.join_side(dlogits.device)
            finally:
                nb = eng.alloc_floats
                eng.end_pass()
                eng.G = None
                self._recording = None
        key = getattr(cx, "plan_key", None)
        if key is not None and key in self._plans:
            self._plans[key].need_bwd = max(self._plans[key].need_bwd, nb + dlogits.numel() + 64)
        if self.grad_finish_hook is not None:
            self.grad_finish_hook()
        return dx, [G[p] for p in self._param_list()]

    def _backward_body(self, cx, dlogits, need_dx, G):
        eng = self._engine
        A = cx.t["act"]
        B, H, W = A["shape"]
        f = self.filters
        dev = dlogits.device
        # head
        ncp, c0 = A["ncp"], f[0]
        dy4 = _A(dev, B, H, W, ncp)
        hip.nchw_to_nhwc(dlogits, dy4)
        dWh, dbh = _Z(dev, ncp, c0), _Z(dev, ncp)
        gwh, gbh, ncl = G[self.output_layer.weight], G[self.output_layer.bias], self.n_classes

        def unpad_head():                                                     # drop the padding rows
            hip.copy2d(dWh, gwh, ncl, c0, c0, c0)
            hip.copy2d(dbh, gbh, 1, ncl, ncp, ncl)
        eng.wgrad_unpad([A["x9"]], dy4, dWh, dbh, H, W, unpad_head, keep=(dbh,))
        dx9 = _A(dev, *A["x9"].shape)
        eng.conv_T(dy4, self.output_layer.weight, dx9, Hin=H, Win=W)
        self._done("output_layer")
        gacc = {id(A[k]): GradSlot() for k in ("x1", "x2", "x3", "x4")}
        main = torch.cuda.current_stream(dev)
        fork = eng.branch_overlap and not eng.capturing
        bst = eng.branch_stream(dev) if fork else None

        # weight gradients of the branch chains: issued late (engine.wgrad, lazy_on).  With a data-parallel reducer attached the
        # natt* / skip* buckets are then handed on from the branch stream after that late issue (their all-reduce, a few MB, runs
        # beside the encoder's backward like the weight gradients themselves)
        lazy = bool(fork and eng.lazy_wgrad)

        bstreams = [eng.branch_stream_n(dev, i) for i in eng.branch_map_b] if fork else []   # per level 0..3
        two = bool(fork and any(t is not bst for t in bstreams))
        sbk = eng.slot_base
        tok = [None]     # event slot recorded by the accumulating tail of the previous chain

        def branch(nat, skip, dt, lvl=0):
            """Backward of one neighborhood-attention block and its skip fuser.  Their only input is dt (the decoder
            stage gradient) and nothing on the decoder chain waits for them, so they run on the branch stream while
            the main stream continues with the coarser decoder stages (few of those kernels fill 256 CUs alone)."""
            if not fork:
                return nat, skip, dt
            bs = bstreams[lvl]                        # (LMN_BRANCH_MAP_B: the small-map chains beside the level-0 / 1 ones)
            hip.stream_wait(bs, main)
            if eng.arena is None:
                dt.record_stream(bs)
            with eng.on_stream(bs):
                eng.lazy_on = lazy
                try:
                    dxs = eng.nat_bwd(nat, dt, cx)
                    eng.skip_bwd(skip, dxs, cx, gacc, order=(tok[0], sbk + lvl, bs) if two else None)
                    tok[0] = sbk + lvl
                finally:
                    eng.lazy_on = False
                eng.join_side(dev)                       # this chain's weight gradients (none when they are issued late)
            return None

        # The flat gradient buffer lists natt4..1 before skip4..1 (BACKWARD_ORDER) and a data-parallel bucket is a
        # contiguous prefix: blocks are reported as soon as every block before them is enqueued too.  The branch stream
        # runs natt_k + skip_k per level, so after level 4..1 are all forked the whole natt/skip range is complete on
        # the BRANCH stream -- it is handed to the reducer from there (the collective waits for the branch stream and its
        # weight-gradient stream only), while the main stream goes on with the GFT backward.
        def branch_blocks_done():
            if not fork or (self.grad_ready_hook is None and self._recording is None):
                return False
            self._emit_done()                            # blocks pending on the main stream leave from there
            with eng.on_stream(bst):
                for name in ("natt4", "natt3", "natt2", "natt1", "skip4", "skip3", "skip2", "skip1"):
                    self._done(name, force=name == "skip1")
            return True

        # decoder (the branch work of level k is forked as soon as dt_k exists)
        dt4 = eng.stage_bwd(self.dconv4, dx9, cx); self._done("dconv4")
        p4 = branch(self.natt4, self.skip4, dt4, 0)
        dx8 = eng.up_bwd(self.up4, dt4, cx, A["x8"].shape); self._done("up4")
        dt3 = eng.stage_bwd(self.dconv3, dx8, cx); self._done("dconv3")
        p3 = branch(self.natt3, self.skip3, dt3, 1)
        dx7 = eng.up_bwd(self.up3, dt3, cx, A["x7"].shape); self._done("up3")
        dt2 = eng.stage_bwd(self.dconv2, dx7, cx); self._done("dconv2")
        p2 = branch(self.natt2, self.skip2, dt2, 2)
        dx6 = eng.up_bwd(self.up2, dt2, cx, A["x6"].shape); self._done("up2")
        dt1 = eng.stage_bwd(self.dconv1, dx6, cx); self._done("dconv1")
        p1 = branch(self.natt1, self.skip1, dt1, 3)
        dx5 = eng.up_bwd(self.up1, dt1, cx, A["x5"].shape); self._done("up1")
        if fork:
            for bs_ in {id(t): t for t in bstreams}.values():
                if bs_ is not bst:
                    hip.stream_wait(bst, bs_)            # (the first branch stream carries on for all: late weight gradients, bucket hand-over, join)
            reported = False
            if not lazy:
                reported = branch_blocks_done()          # natt* / skip*: complete on the branch stream from here on
            dcat = eng.gft_bwd(self.gft, dx5, cx)        # independent of the branch chains: before the join
            hip.stream_wait(main, bst)                   # join: the encoder gradients in gacc are complete
            if lazy:
                # the chains' weight gradients now: their stream starts once the chains are through (it waits for the branch stream)
                # and runs beside the encoder's backward below
                lq, eng.lazy_q = eng.lazy_q, []
                with eng.on_stream(bst):
                    for fn in lq:
                        fn()
                reported = branch_blocks_done()
            if eng.arena is None:
                for slot in gacc.values():
                    if slot.g is not None:
                        slot.g.record_stream(main)
            if not reported:
                for name in ("natt4", "natt3", "natt2", "natt1", "skip4", "skip3", "skip2", "skip1"):
                    self._done(name)
        else:
            # neighborhood-attention blocks (their outputs were the residual inputs of up1..4: gradient = dt_k)
            dxs = [eng.nat_bwd(p[0], p[2], cx) for p in (p4, p3, p2, p1)]
            for k, name in enumerate(("natt4", "natt3", "natt2", "natt1")):
                self._done(name)
            # skip fusers: accumulate into the encoder activations' gradients
            for k, (p, name) in enumerate(zip((p4, p3, p2, p1), ("skip4", "skip3", "skip2", "skip1"))):
                eng.skip_bwd(p[1], dxs[k], cx, gacc); self._done(name)
        # bottleneck
        if not fork:
            dcat = eng.gft_bwd(self.gft, dx5, cx)
        self._done("gft")
        off = 0
        for k, fac in (("x1", 16), ("x2", 8), ("x3", 4), ("x4", 2)):
            t = A[k]
            hip.avgpool_bwd(V(dcat, off, t.shape[-1]), gacc[id(t)].g, fac, True)
            off += t.shape[-1]
        g1, g2, g3, g4 = (gacc[id(A[k])].g for k in ("x1", "x2", "x3", "x4"))
        # encoder (reverse)
        eng.conv3_bwd(self.down4[0], A["x4"], V(dcat, off, f[4]), s=2, dx=g4, accumulate=True); self._done("down4")
        dxd3 = eng.stage_bwd(self.conv4, g4, cx); self._done("conv4")
        eng.conv3_bwd(self.down3[0], A["x3"], dxd3, s=2, dx=g3, accumulate=True); self._done("down3")
        dxd2 = eng.stage_bwd(self.conv3, g3, cx); self._done("conv3")
        eng.conv3_bwd(self.down2[0], A["x2"], dxd2, s=2, dx=g2, accumulate=True); self._done("down2")
        dxd1 = eng.stage_bwd(self.conv2, g2, cx); self._done("conv2")
        eng.conv3_bwd(self.down1[0], A["x1"], dxd1, s=2, dx=g1, accumulate=True); self._done("down1")
        dxin = eng.stage_bwd(self.conv1, g1, cx, need_dx=need_dx); self._done("conv1")
        if lazy:                                         # the late weight gradients of the branch chains
            with eng.on_stream(bst):
                eng.join_side(dev)
            hip.stream_wait(main, bst)
        dx = None
        if need_dx:
            dx = _E(dev, B, self.channel, H, W)
            hip.nhwc_to_nchw(dxin, dx)
        return dx
