"""GPU: the data-parallel path end to end with TWO processes sharing the single test GPU (gloo backend moves the
device buffers through the host; RCCL needs one GPU per rank, which the driver's multi-GPU bench provides).
Checks the stream-ordered bucketed reducer inside a real LM_Net backward: after loss.backward() every rank holds
the MEAN of the per-rank gradients, parameters stay replicated after an optimizer step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, plans=False):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from lm_net_amd import LM_Net
        from lm_net_amd.ddp import DistributedLMNet
        from tools.detweights import det_input, fill_module
        net = LM_Net(3, 2, filters=[12] * 5)
        fill_module(net, seed=rank)                       # different weights per rank before wrapping ...
        net = net.cuda().train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        x = det_input((2, 3, 32, 48), "ddp/x%d" % rank).cuda()   # rank-specific shard of the batch
        model = DistributedLMNet(net, bucket_bytes=64 << 10, first_bucket_bytes=16 << 10)       # ... replicated from rank 0 here
        if plans:       # recorded C-side schedules: the reducer is fed between plan segments (LM_Net._plan_backward)
            net.enable_plans()
            for _ in range(4):      # 2 warm-up passes, the recording pass, one replay: the checks below run on replays
                model(x).square().mean().backward()
                net.zero_grad(set_to_none=True)
            assert any(p.bwd is not None for p in net._plans.values()), "no plan was recorded"
        # local (un-reduced) gradients: run once with the hooks detached
        hooks = (net.grad_begin_hook, net.grad_ready_hook, net.grad_finish_hook)
        net.grad_begin_hook = net.grad_ready_hook = net.grad_finish_hook = None
        net(x).square().mean().backward()
        local = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        net.zero_grad(set_to_none=True)
        for bn in [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]:
            bn.reset_running_stats()
        net.grad_begin_hook, net.grad_ready_hook, net.grad_finish_hook = hooks
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        model(x).square().mean().backward()
        torch.cuda.synchronize()
        reduced = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        gathered = [torch.zeros_like(local.cpu()) for _ in range(world)]
        dist.all_gather(gathered, local.cpu())
        expect = sum(gathered) / world
        err = float((reduced.cpu() - expect).abs().max() / (expect.abs().max() + 1e-30))
        opt.step()
        w = torch.cat([p.detach().flatten() for p in net.parameters()]).cpu()
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        same = float((ws[0] - ws[1]).abs().max())
        q.put((rank, err, same, len(model.reducer.launched)))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        q.put((rank, repr(e), None, None))


@pytest.mark.parametrize("plans", [False, True])
def test_two_rank_training_step_on_one_gpu(plans):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, plans)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=240) for _ in range(world)]
    [p.join(60) for p in ps]
    for rank, err, same, nb in res:
        assert isinstance(err, float), "rank %d failed: %s" % (rank, err)
        assert err < 1e-5, "rank %d: reduced gradient != mean of local gradients (rel %.3e)" % (rank, err)
        assert same == 0.0, "parameters diverged across ranks after the step"
        assert nb >= 2, "expected several buckets"
