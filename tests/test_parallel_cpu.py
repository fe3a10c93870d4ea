"""CPU, world_size 2, gloo: the N>1 gradient-exchange path (arena slices + bucket plan + async all-reduce).
The HIP kernels are not involved: gradients are delivered into the arena by hand, exactly the way the
backward kernels' `deliver_grad` does it."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(8, 16, 3, bias=False)
        self.bn = nn.BatchNorm2d(16)
        self.fc = nn.Linear(16, 4)
        self.unused = nn.Linear(4, 4)     # never receives a gradient (like the Q4 heads)

    def forward(self, x):
        return x


def _worker(rank, world, port, overlap, q, always=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oaprogressionmmf_amd.arena import deliver_grad, grad_target
        from oaprogressionmmf_amd.parallel import DataParallelRCCL
        torch.manual_seed(100 + rank)          # ranks start from DIFFERENT weights: broadcast must fix that
        m = Tiny()
        ddp = DataParallelRCCL(m, bucket_elems=1024, overlap=overlap, exchange_always=always)
        assert ddp._active == (world > 1 or always)
        a = ddp.arena()
        ref_w = [p.detach().clone() for p in m.parameters()]
        gathered = [torch.zeros_like(a.P) for _ in range(world)]
        dist.all_gather(gathered, a.P)
        assert all(torch.equal(gathered[0], g) for g in gathered), "parameters not broadcast from rank 0"
        assert m.conv.weight.shape == (16, 8, 3, 3) and m.conv.weight.permute(0, 2, 3, 1).is_contiguous()
        trained = [m.fc.bias, m.fc.weight, m.bn.bias, m.bn.weight, m.conv.weight]   # backward order
        for step in range(3):
            ddp(torch.zeros(1))
            for p in m.parameters():
                p.grad = None
            expect = {}
            for i, p in enumerate(trained):
                buf, acc = grad_target(p)
                assert not acc and buf.data_ptr() == p._koaf_grad.data_ptr()
                buf.copy_(torch.full(p.shape, float(rank + 1) * (i + 1) + step))
                deliver_grad(p, buf, acc)
                expect[id(p)] = sum(float(r + 1) * (i + 1) + step for r in range(world))
            ddp.reduce_gradients()
            for p in trained:
                assert torch.allclose(p.grad, torch.full(p.shape, expect[id(p)])), (step, p.shape)
            assert m.unused.weight.grad is None
            assert len(ddp._plan) >= 2, "bucket plan should split the arena"
        # running statistics follow rank 0
        m.bn.running_mean.fill_(float(rank))
        ddp.broadcast_buffers()
        assert float(m.bn.running_mean.sum()) == 0.0
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_gradient_exchange_gloo_world2(overlap):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_world_of_one_exchange_rehearsal():
    """exchange_always=True: a single rank still broadcasts, plans buckets and all-reduces (the 1-GPU RCCL rehearsal mode);
    without it a world of one does no collective at all (no plan is ever built)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(0, 1, _free_port(), True, q, True))
    p.start()
    rank, msg = q.get(timeout=180)
    p.join(timeout=60)
    assert msg == "ok", msg
    from oaprogressionmmf_amd.parallel import DataParallelRCCL
    ddp = DataParallelRCCL(Tiny())                 # no process group: plain single process
    assert ddp.world == 1 and not ddp._active
    ddp.reduce_gradients()
    assert ddp._plan is None


def test_arena_views_and_adam_ranges():
    from oaprogressionmmf_amd.arena import get_arena
    m = Tiny()
    w0 = m.conv.weight.detach().clone()
    a = get_arena(m)
    assert torch.equal(m.conv.weight.detach(), w0)                      # values survive adoption
    assert m.conv.weight.data_ptr() >= a.P.data_ptr()
    sd = m.state_dict()
    assert sd["conv.weight"].shape == (16, 8, 3, 3) and "bn.running_mean" in sd
    m.load_state_dict({k: torch.zeros_like(v) for k, v in sd.items()})   # in-place load keeps the arena
    assert a.valid() and float(a.P.abs().sum()) == 0.0
    runs = a.active_ranges([m.conv.weight, m.bn.weight, m.bn.bias, m.fc.weight, m.fc.bias])
    assert len(runs) == 1                                                 # contiguous: ONE fused Adam launch
    runs = a.active_ranges([m.conv.weight, m.fc.weight])
    assert len(runs) == 2


def test_shard_sampler_slices_one_global_stream():
    """per-rank data shards from ONE global WeightedRandomSampler stream (_data_provider.py:463-483): the union of
    the ranks' shards, batch by batch, is the sequence the reference's single-process sampler draws"""
    from torch.utils.data import WeightedRandomSampler
    from oaprogressionmmf_amd.parallel import shard_sampler
    w = torch.tensor([0.1, 0.9, 0.5, 0.5, 2.0, 0.01, 1.0] * 9)           # 63 samples
    g = torch.Generator().manual_seed(77)
    glob = list(WeightedRandomSampler(w, num_samples=len(w), replacement=True, generator=g))
    world, B = 4, 3
    shards = [shard_sampler(w, r, world, seed=77, batch_size=B) for r in range(world)]
    steps = len(glob) // (B * world)
    assert steps == 5 and all(len(s) == steps * B for s in shards)            # drop_last: equal step counts on every rank
    for b in range(steps):
        gathered = sum((shards[r][b * B:(b + 1) * B] for r in range(world)), [])    # DataParallel's scatter order
        assert gathered == glob[b * B * world:(b + 1) * B * world]
    # without drop_last nothing of the stream is lost
    all_ = [shard_sampler(w, r, world, seed=77, batch_size=B, drop_last=False) for r in range(world)]
    assert sorted(sum(all_, [])) == sorted(glob)
    # ... but the shard lengths then differ (evaluation only); equal=True pads by wrapping around: same length on every rank
    assert len({len(a) for a in all_}) > 1
    eq = [shard_sampler(w, r, world, seed=77, batch_size=B, drop_last=False, equal=True) for r in range(world)]
    assert len({len(a) for a in eq}) == 1 and len(eq[0]) == (steps + 1) * B
    assert sum((eq[r][steps * B:] for r in range(world)), []) == (glob + glob)[steps * B * world:(steps + 1) * B * world]
    eqs = [shard_sampler(w, r, 2, seed=77, drop_last=False, equal=True) for r in range(2)]
    assert len(eqs[0]) == len(eqs[1]) == 32 and eqs[1][-1] == glob[0]
    # strided variant; another epoch = another seed = another stream; bad rank raises
    st = [shard_sampler(w, r, 2, seed=77) for r in range(2)]
    assert st[0] == glob[0:62:2] and st[1] == glob[1:62:2]
    assert shard_sampler(w, 0, world, seed=78, batch_size=B) != shards[0]
    with pytest.raises(ValueError):
        shard_sampler(w, 4, 4, seed=0)
