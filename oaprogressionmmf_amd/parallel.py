"""Process-per-GPU data parallelism for the koafusion train step (replaces the reference's single-process
`nn.DataParallel`, koafusion/run/train_prog_fus.py:84).

One exchange step per iteration: an all-reduce (RCCL over xGMI; `backend="nccl"` IS RCCL on ROCm) of the flat
gradient arena.  Buckets are plain [lo, hi) slices of `arena.G` -- no flatten/unflatten copies -- and are
launched `async_op=True` the moment the last gradient of a bucket has been written by the backward kernels,
so the exchange of the transformer/late-stage gradients runs under the remaining encoder backward.
Parameters that never receive a gradient (the mlp_head0 of the cls-less aggregators) are learnt
on the first step and never waited for.

Semantics vs nn.DataParallel: equal shards + mean-reduction loss => mean of per-rank means is
the global mean (call `scale_loss` or divide the loss by world size); BatchNorm statistics stay per replica
exactly as under DataParallel; `broadcast_buffers()` copies rank 0's running stats (DataParallel keeps
replica 0's).
"""
import torch
import torch.distributed as dist
from torch import nn

from .arena import _round_up, get_arena

BUCKET_ELEMS = 32 * 1024 * 1024   # 128 MB of fp32 per all-reduce (xGMI ring is per-link bound; few, large)


class DataParallelRCCL(nn.Module):
    def __init__(self, module: nn.Module, process_group=None, bucket_elems=BUCKET_ELEMS, overlap=True,
                 exchange_always=False):
        super().__init__()
        self.module = module
        self.pg = process_group
        self.bucket_elems = bucket_elems
        self.overlap = overlap
        self.world = dist.get_world_size(self.pg) if dist.is_initialized() else 1
        # exchange_always: run the collectives even in a world of one (RCCL rehearsal on a 1-GPU box; a no-op sum)
        self._active = self.world > 1 or (exchange_always and dist.is_initialized())
        self._arena = None
        self._plan = None          # list of buckets: dict(lo, hi, need=set(param ids))
        self._pending = None
        self._handles = []
        self._seen = []
        self._streams = None       # per bucket: the HIP streams its gradients were delivered on this step
        self._comm = None          # staging stream the early all-reduces are ordered on
        self.time_exposed = False  # bench: bracket the compute stream's waits on the collectives with events
        self._exposed = []

    # -- setup -------------------------------------------------------------------------------------
    def arena(self):
        a = get_arena(self.module)
        if a is not self._arena:
            self._arena = a
            self._plan = None
            a.ready_hook = self._on_ready
            if self._active:
                self.broadcast_parameters()
        return a

    def broadcast_parameters(self):
        a = self._arena
        dist.broadcast(a.P, src=0, group=self.pg)
        self.broadcast_buffers()

    def broadcast_buffers(self):
        a = self._arena
        if a.B.numel() > 1:
            dist.broadcast(a.B, src=0, group=self.pg)

    def scale_loss(self, loss):
        return loss / self.world if self.world > 1 else loss

    # -- forward / gradient exchange -------------------------------------------------------------------
    def forward(self, *inputs, **kw):
        self.arena()
        self._begin_step()
        return self.module(*inputs, **kw)

    def _begin_step(self):
        self._handles = []
        self._seen = []
        if self._plan is not None:
            self._pending = [set(b["need"]) for b in self._plan]
            self._streams = [set() for _ in self._plan]

    def _on_ready(self, p):
        self._seen.append(p)
        if not self._active or self._plan is None or not self.overlap:
            return
        pid = id(p)
        bi = self._where.get(pid)
        if bi is None:
            return
        pend = self._pending[bi]
        pend.discard(pid)
        if p.is_cuda:
            self._streams[bi].add(torch.cuda.current_stream())
        if not pend:
            self._launch(bi)

    def _launch(self, bi):
        """all-reduce of bucket bi.  The encoders run on several HIP streams ("lanes") and a 128 MB bucket can hold
        gradients of two of them, delivered on different streams; the collective is therefore ordered behind EVERY
        stream that delivered into the bucket, on a staging stream of its own so that no compute lane has to wait."""
        b = self._plan[bi]
        g = self._arena.G[b["lo"]:b["hi"]]
        if not g.is_cuda:
            self._handles.append(dist.all_reduce(g, group=self.pg, async_op=True))
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=g.device)
        streams = self._streams[bi] if self._streams is not None else set()
        streams = set(streams) | {torch.cuda.current_stream()}
        for st in streams:
            self._comm.wait_stream(st)
        with torch.cuda.stream(self._comm):
            self._handles.append(dist.all_reduce(g, group=self.pg, async_op=True))

    def _build_plan(self, params):
        a = self._arena
        spans = sorted((a.slot(p)[0], a.slot(p)[1], id(p)) for p in params)
        plan, cur = [], None
        for o, n, pid in spans:
            hi = o + _round_up(n)
            if cur is not None and cur["hi"] == o and (hi - cur["lo"]) <= self.bucket_elems:
                cur["hi"] = hi
                cur["need"].add(pid)
            else:
                cur = dict(lo=o, hi=hi, need={pid})
                plan.append(cur)
        self._plan = plan
        self._where = {pid: i for i, b in enumerate(plan) for pid in b["need"]}

    def reduce_gradients(self):
        """Call after backward(): finishes (or, on the first step, performs) the gradient all-reduce."""
        if not self._active:
            return
        a = self._arena
        if self._plan is None or not self.overlap:
            # first step: learn which parameters receive gradients, reduce everything now
            params = [p for p in a.params if p.grad is not None]
            self._build_plan(params)
            self._streams = None
            for bi in range(len(self._plan)):
                self._launch(bi)
        else:
            # anything not launched from the hook (a parameter skipped this step) goes now
            for bi, pend in enumerate(self._pending):
                if pend:
                    self._launch(bi)
        # exposed communication = how long the caller's (compute) stream sits in these waits: an event when it arrives at
        # the first wait (everything it had to compute is done by then) and one after the last (read by exposed_ms())
        timed = self.time_exposed and a.G.is_cuda and self._handles
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self._handles:
            h.wait()             # the caller's stream waits for the collectives (which ran behind the staging stream)
        if timed:
            e1.record()
            self._exposed.append((e0, e1))
            if len(self._exposed) > 4096:           # (never drained by the caller: keep the newest, the list must not grow with the run)
                del self._exposed[:-1024]
        self._handles = []

    def exposed_ms(self, reset=True):
        """per-step milliseconds the compute stream waited for gradient collectives since the last call (needs
        `time_exposed = True`; synchronises on the recorded events)"""
        out = []
        for e0, e1 in self._exposed:
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        if reset:
            self._exposed = []
        return out

    # module protocol pass-throughs used by the train driver / checkpoint handler
    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        return self.module.load_state_dict(*a, **k)


def shard_sampler(weights, rank, world, seed, num_samples=None, batch_size=None, drop_last=True, equal=False):
    """This rank's share of ONE global `WeightedRandomSampler` stream (SURVEY 8e "one-time": per-rank data shards from one
    global sampler; reference: koafusion/datasets/_data_provider.py:463-483 builds
    `WeightedRandomSampler(weights, num_samples=len(weights), replacement=True)` for the single-process DataParallel run).

    Every rank draws the SAME index sequence (torch.multinomial on a generator seeded with `seed`: call again with
    seed + epoch for the next epoch) and keeps its slice, so that the union over ranks is exactly the sequence a
    single-process run would have consumed:
      * batch_size given (the PER-RANK batch): global batch g = indices [g*B*world, (g+1)*B*world), of which this rank takes the
        contiguous block [rank*B, (rank+1)*B) -- DataParallel's scatter of the global batch along dim 0; with drop_last (the
        reference's train loaders, :483) the ragged tail that cannot fill a global batch is dropped, so all ranks run the
        same number of steps;
      * batch_size None: the strided slice stream[rank::world] (truncated to equal lengths under drop_last).
    drop_last=False is for EVALUATION only: the ragged tail is split over the leading ranks, so shard lengths differ -- under a
    gradient exchange (DataParallelRCCL) ranks with different step counts hang in the last collective.  With `equal=True` the
    tail is instead padded by wrapping around to the start of the stream, every rank gets the same number of indices (a few
    samples are seen twice, as torch's DistributedSampler does); training loops that must keep every sample use that.
    Returns a list of python ints to hand to `DataLoader(sampler=...)` / `Subset`."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    w = torch.as_tensor(weights, dtype=torch.double)
    n = int(num_samples) if num_samples is not None else int(w.numel())
    g = torch.Generator()
    g.manual_seed(int(seed))
    stream = torch.multinomial(w, n, replacement=True, generator=g).tolist()     # what WeightedRandomSampler.__iter__ draws
    if equal and not drop_last:
        unit = world if batch_size is None else int(batch_size) * world
        if n % unit:
            stream = stream + stream[:unit - n % unit]        # wrap around: equal shard lengths on every rank
            n = len(stream)
    if batch_size is None:
        if drop_last:
            stream = stream[:n - n % world]
        return stream[rank::world]
    gb = int(batch_size) * world
    full = n // gb
    out = []
    for b in range(full):
        lo = b * gb + rank * batch_size
        out.extend(stream[lo:lo + batch_size])
    if not drop_last and n % gb:
        tail = stream[full * gb:]
        per = -(-len(tail) // world)
        out.extend(tail[rank * per:(rank + 1) * per])
    return out
