"""Data-parallel training of LM_Net: one process per GPU, gradient all-reduce on RCCL over xGMI,
overlapped with the backward pass.

The reference has no working multi-GPU path (``utils/distributed_utils.py`` is never called from
``train.py``) -- only the scaffolding of plain data parallelism: replicated
model, per-process BatchNorm statistics (``--syncBN`` is parsed but unused), rank-offset data seed.
This module supplies that capability MI355X-first:

  * ``LM_Net``'s backward writes all 3.97 M parameter gradients (15.9 MB fp32) into ONE flat buffer laid
    out in backward-completion order, and reports each finished block ``[lo, hi)``;
  * ``GradReducer`` coalesces finished blocks into a few large buckets (xGMI is point-to-point, 7 links
    per GPU: at 16 MB the collective is latency-bound, so few big messages beat many small ones) and
    launches ``all_reduce`` on a side HIP stream as soon as a bucket is complete, while the compute
    stream keeps running the rest of the backward schedule;
  * the compute stream waits for the side stream once, at the end of backward, so ``optimizer.step()``
    sees averaged gradients.  No activation is exchanged (per-GPU BN statistics, as the reference).

Works with any ``torch.distributed`` backend: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` for the
CPU tests of the bucketing logic.
"""
import torch
import torch.distributed as dist


class GradReducer:
    """Bucketed asynchronous all-reduce (mean) over contiguous slices of a flat gradient buffer."""

    def __init__(self, process_group=None, bucket_bytes=4 << 20, first_bucket_bytes=1 << 20):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes, self.first_bucket_bytes = bucket_bytes, first_bucket_bytes
        self.flat = None
        self.pending_lo = self.pending_hi = 0
        self.works = []
        self.launched = []            # (lo, hi) of every bucket launched in the current backward
        self.launched_before_finish = 0   # how many of them were handed to the backend before finish() (= overlapped)
        self.also_wait = []           # producer streams besides the current one (model's weight-gradient stream)
        self.stream = None

    def begin(self, flat):
        self.flat = flat
        self.pending_lo = self.pending_hi = 0
        self.works, self.launched = [], []
        self.also_wait = []
        if flat.is_cuda and self.stream is None:
            self.stream = torch.cuda.Stream(device=flat.device)

    def ready(self, lo, hi, streams=()):
        """Gradients of flat[lo:hi) have been enqueued on the current stream (and on `streams`: the weight-gradient
        side stream of the model -- the collective waits for them, the compute stream does not).  Blocks arrive in
        order."""
        if self.world == 1 or hi <= lo:
            return
        for s in streams:
            if s not in self.also_wait:
                self.also_wait.append(s)
        assert lo == self.pending_hi, "blocks must be reported contiguously in backward order"
        self.pending_hi = hi
        cap = self.first_bucket_bytes if not self.launched else self.bucket_bytes
        if (self.pending_hi - self.pending_lo) * self.flat.element_size() >= cap:
            self._launch()

    def _launch(self):
        lo, hi = self.pending_lo, self.pending_hi
        if hi <= lo:
            return
        chunk = self.flat[lo:hi]
        avg = dist.ReduceOp.AVG if (self.flat.is_cuda and dist.get_backend(self.pg) == "nccl") else dist.ReduceOp.SUM
        if self.flat.is_cuda:
            self.stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            for s in self.also_wait:
                self.stream.wait_stream(s)
            with torch.cuda.stream(self.stream):
                dist.all_reduce(chunk, op=avg, group=self.pg)
                if avg == dist.ReduceOp.SUM:
                    chunk.mul_(1.0 / self.world)
        else:
            self.works.append((dist.all_reduce(chunk, op=avg, group=self.pg, async_op=True), chunk))
        self.launched.append((lo, hi))
        self.pending_lo = hi

    def finish(self):
        """Flush the tail bucket and make the compute stream wait for every collective."""
        if self.world == 1 or self.flat is None:
            return
        self.launched_before_finish = len(self.launched)
        self._launch()
        if self.flat.is_cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        else:
            for w, chunk in self.works:
                w.wait()
                chunk.mul_(1.0 / self.world)
        self.works = []


def broadcast_state(module, src=0, process_group=None):
    """One-time replication of parameters and buffers from rank `src` (BN buffers are not re-broadcast
    every step: statistics stay per-process, as in the reference)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=process_group)


class DistributedLMNet(torch.nn.Module):
    """``model = DistributedLMNet(LM_Net(...).cuda())`` -- the data-parallel wrapper.  ``forward`` is the
    wrapped model's; gradients are averaged across ranks by the time ``loss.backward()`` returns."""

    def __init__(self, model, process_group=None, bucket_bytes=4 << 20, first_bucket_bytes=1 << 20):
        super().__init__()
        self.module = model
        self.reducer = GradReducer(process_group, bucket_bytes, first_bucket_bytes)
        broadcast_state(model, 0, process_group)
        model.grad_begin_hook = self.reducer.begin
        model.grad_ready_hook = self.reducer.ready
        model.grad_finish_hook = self.reducer.finish

    def forward(self, x):
        return self.module(x)
