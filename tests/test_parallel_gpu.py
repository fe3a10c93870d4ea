"""(second test) RCCL itself, world of ONE: the box has one GPU and RCCL refuses two ranks on a device, so the nccl code
path -- process group with device_id, broadcast of the parameter arena, async all-reduce of arena slices on the
staging stream, handle.wait() on the compute stream -- is rehearsed with `exchange_always=True`; summing over one rank
must leave the train step bit-identical to the exchange-free one.

(first test) GPU, world_size 2 on ONE device (gloo moves the buckets; the driver's real runs use RCCL): the data-parallel
train step of a model whose encoders run on separate HIP streams.  The early (overlapped) all-reduce of a gradient
bucket must be ordered behind every stream that wrote into it -- a bucket can hold gradients of two encoders.  With
deterministic kernels and two ranks the reduced gradient must equal, bit for bit, 0.5*g(rank 0 batch) + 0.5*g(rank 1
batch) computed without any exchange."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, backend="gloo"):
    import sys
    from pathlib import Path
    here = Path(__file__).resolve().parent
    sys.path.insert(0, str(here))
    sys.path.insert(0, str(here.parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    # gloo: both ranks share device 0 (the 1-GPU box); nccl (= RCCL): one rank per device, as the driver's multi-GPU runs
    di = rank if backend == "nccl" else 0
    torch.cuda.set_device(di)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", di))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        import procedural as P
        from oaprogressionmmf_amd.config import ConfigDict
        from oaprogressionmmf_amd.models import dict_models
        from oaprogressionmmf_amd.parallel import DataParallelRCCL
        from oaprogressionmmf_amd.various import dict_losses
        dev = torch.device("cuda", di)
        cfg = P.cfg_xr1mr2(xr=(96, 96), mr1=(64, 64, 3), mr2=(64, 64, 2), depth=1)
        B = 2
        loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)

        def make():
            m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None)
            P.fill_state_dict(m.state_dict())
            return m.to(dev).train()

        def batch(r):
            xs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in P.model_inputs(cfg, B, 50 + r)]
            return xs, torch.from_numpy(P.make_target("target", B, 50 + r)).to(dev)
        # reference: both ranks' batches on a model of its own, loss scaled by 1/world, no exchange
        parts = []
        for r in range(world):
            ref = make()       # fresh BatchNorm buffers per batch, like each rank's replica (the batch statistics are
            xs, y = batch(r)   # summed about the running mean, so its value shows in the last bits)
            (loss_fn(ref(*xs)["main"].squeeze(1), y.long().squeeze(1)) / world).backward()
            parts.append({k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None})
        want = {k: parts[0][k] + parts[1][k] for k in parts[0]}
        # data-parallel: a small bucket size makes many buckets straddle encoders / streams
        ddp = DataParallelRCCL(make(), bucket_elems=4 * 1024 * 1024)
        ddp.time_exposed = True
        xs, y = batch(rank)
        bad = {}
        buf0 = {k: b.detach().clone() for k, b in ddp.module.named_buffers()}
        for step in range(3):          # step 0 learns the plan; steps 1, 2 launch from the delivery hooks
            with torch.no_grad():      # same BatchNorm buffers (= statistics shift) as the reference at every step
                for k, b in ddp.module.named_buffers():
                    b.copy_(buf0[k])
            ddp.module.zero_grad()
            loss = loss_fn(ddp(*xs)["main"].squeeze(1), y.long().squeeze(1))
            ddp.scale_loss(loss).backward()
            ddp.reduce_gradients()
            torch.cuda.synchronize()
            got = {k: p.grad for k, p in ddp.module.named_parameters() if p.grad is not None}
            assert sorted(got) == sorted(want)
            bad[step] = [k for k in want if not torch.equal(got[k], want[k])]
        nb = len(ddp._plan)
        ex = ddp.exposed_ms()
        assert len(ex) == 3 and all(v >= 0.0 for v in ex), ex      # one bracket per step around the compute stream's waits
        q.put((rank, None, bad, nb))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None, 0))
    finally:
        dist.destroy_process_group()


def test_two_ranks_overlapped_allreduce_is_exact(dev):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, err, bad, nb in res:
        assert err is None, f"rank {rank}:\n{err}"
        assert nb >= 8, f"only {nb} buckets: the plan does not straddle encoders"
        assert all(len(v) == 0 for v in bad.values()), f"rank {rank}: gradients differ from the exchange-free sum: {bad}"


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_two_ranks_rccl_is_exact():
    """the same assertion over the real backend: backend "nccl" (= RCCL over xGMI), one rank per device -- the first
    multi-GPU box proves the collective path (broadcast of the arena, bucketed async all-reduce on the staging stream,
    waits on the compute stream) bit for bit against the exchange-free sum.  Skipped on 1-GPU boxes."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "nccl")) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, err, bad, nb in res:
        assert err is None, f"rank {rank}:\n{err}"
        assert nb >= 8
        assert all(len(v) == 0 for v in bad.values()), f"rank {rank}: RCCL-reduced gradients differ from the exchange-free sum: {bad}"


def _rccl_worker(port, q):
    import sys
    from pathlib import Path
    here = Path(__file__).resolve().parent
    sys.path.insert(0, str(here))
    sys.path.insert(0, str(here.parent))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import numpy as np
        import procedural as P
        from oaprogressionmmf_amd.config import ConfigDict
        from oaprogressionmmf_amd.models import dict_models
        from oaprogressionmmf_amd.parallel import DataParallelRCCL
        from oaprogressionmmf_amd.run import train_step
        from oaprogressionmmf_amd.various import dict_losses, dict_optimizers
        dev = torch.device("cuda:0")
        cfg = P.cfg_xr1mr2(xr=(96, 96), mr1=(64, 64, 3), mr2=(64, 64, 2), depth=1)
        loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)
        xs = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in P.model_inputs(cfg, 2, 60)]
        y = torch.from_numpy(P.make_target("target", 2, 60)).to(dev)

        def run(wrap):
            m = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None)
            P.fill_state_dict(m.state_dict())
            m = m.to(dev).train()
            model = DataParallelRCCL(m, bucket_elems=4 * 1024 * 1024, exchange_always=True) if wrap else m
            opt = dict_optimizers["Adam"](m.parameters(), lr=1e-4, weight_decay=1e-4)
            losses = [float(train_step(model, loss_fn, opt, xs, y)[1]) for _ in range(3)]
            torch.cuda.synchronize()
            return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}, model
        l0, sd0, _ = run(False)
        l1, sd1, ddp = run(True)
        bad = [k for k in sd0 if not torch.equal(sd0[k], sd1[k])]
        t = torch.ones(1024, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        q.put((None, l0, l1, bad, len(ddp._plan), dist.get_backend(), float(t.sum())))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((traceback.format_exc(), None, None, None, 0, None, 0.0))
    finally:
        dist.destroy_process_group()


def test_rccl_world_of_one_exchange_is_a_noop(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    err, l0, l1, bad, nb, backend, s = q.get(timeout=600)
    p.join(timeout=120)
    assert err is None, err
    assert backend == "nccl" and s == 1024.0
    assert nb >= 8, f"only {nb} buckets went through RCCL"
    assert l0 == l1, (l0, l1)
    assert bad == [], f"state after 3 steps differs with the RCCL exchange in the loop: {bad[:5]}"
