#!/usr/bin/env python3
"""bench.py -- knees/sec of the full XR + 3 x MRI + clinical fusion train step (BASELINE.json metric and shapes).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W] [--batch B] ...
  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / LOCAL_RANK /
  WORLD_SIZE / MASTER_* in the environment), or plainly as `python bench.py --gpus N`: the parent then starts N fresh worker
  processes itself (before anything touches a GPU), one rank per GPU over RCCL, relays rank 0's JSON line and exits non-zero
  if a worker fails.

A "step" is one iteration of the reference's step body (koafusion/run/train_prog_fus.py:132-168) on one synthetic batch
already resident in HBM: zero_grad -> forward (dropout 0.1 as in the recipe, runner.sh:352) -> FocalLoss -> backward ->
gradient all-reduce over RCCL (N > 1) -> fused Adam -> loss.item().  The reference reads loss.item() between forward and
backward (:159-165); here the same value is read after the optimizer step is enqueued, so that the host sync does not drain
the GPU in the middle of the step (`host_sync` in the JSON names this).  Weak scaling: the per-GPU batch is fixed,
`value` = global samples / max-over-ranks time of exactly K steps between barriers.

Workloads:
  syn3 (default)    BASELINE.json's headline configuration: "Full XR + SAG-DESS/COR-IW-TSE/SAG-T2 + clinical transformer
                    fusion, batch 8" on its synthetic tensors -- XR 1x310x310 and three MRI volumes 1x160x384x384 (slice-major,
                    as BASELINE writes them: `fe.mr.volume_layout: ncdhw`, a zero-copy slice fold) + 9 clinical variables --
                    through the registry extension XR1MR3C1CnnTrf (the reference's XR1MR2C1CnnTrf pattern with a third MRI
                    slot): ResNeXt-50 on the radiograph, ResNet-50 slice-wise on 3 x 160 slices of 384^2, 4 x FeaT(depth 4,
                    8 heads, width 2048).  35.2 TFLOP of algorithmic work per knee and step.  A batch of 8 holds 3840 MRI
                    slices: fp32 activations of every conv would need ~750 GB, so layer1-2 of every MRI encoder (and layer3
                    of the encoder whose backward runs last) are rebuilt in backward from their stage inputs + BatchNorm
                    statistics (`activation_recompute` in the JSON).
                    The same run also times two secondaries at the reference's native sizes (10 steps each): `native3` (same
                    model, XR 350^2 + DESS 160x160x64 + TSE 160x160x32 + T2 160x160x25) and `native` (XR1MR2C1CnnTrf exactly
                    as runner.sh:341-363, the largest model the reference registers, parity-pinned by reference fixtures).
  native3 / native  those two as the primary workload.
  syn               BASELINE's synthetic shapes through the pinned 2-MRI class.
  xr1cnn / xr1c1    BASELINE configs 1 / 2 (XR1Cnn B=4; extension XR1C1Cnn = XR + clinical MLP head, B=32) @350^2.
  mr1 / mr1c1       BASELINE config 3: MR1CnnTrf B=4 @160x160x64; extension MR1C1CnnTrf (DESS + clinical) B=4 @384x384x160.
  eval3             inference pass of the native3 model (eval() mode, no autograd); `value` = knees/s scored.
The JSON line carries `roofline` for the dominant kernel (the MFMA GEMM, timed live with events on the launch stream over
one extra instrumented single-stream step) and `cpu_baseline` (the oracle = CPU port of the same step, timed on this box's
host cores on a bounded sample).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this pool (already exported there)

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

# MI355X_MICROARCH.md, chip-level table
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA (v_mfma_f32_32x32x16_{bf16,f16})
MFMA_F32_PEAK_TFLOPS = 157.3     # dense fp32 MFMA (v_mfma_f32_32x32x2_f32): what the 16-bit piece schemes replace
HBM_PEAK_TBPS = 8.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="syn3")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the native3 / native secondaries of the default run")
    ap.add_argument("--recompute", default="auto",
                    help="activation recompute in the encoders: auto (workload default), none, stage (every stage), block "
                         "(one block at a time), or stage indices per MRI encoder, e.g. '012,01,01' (0 = layer1; encoders in "
                         "forward order; one entry = all encoders)")
    ap.add_argument("--graph", action="store_true",
                    help="1 GPU: capture the train step into a HIP graph after 2 eager steps and replay it (run.GraphedTrainStep): "
                         "for the launch-bound small configurations (xr1cnn, xr1c1)")
    ap.add_argument("--breakdown", default="", help="write a per-shape table of the instrumented step to this file")
    ap.add_argument("--inproc", action="store_true",
                    help="1 GPU: run the measurement in THIS process instead of a fresh child of a GPU-free supervisor (profiler runs: "
                         "rocprofv3 must see the measuring process itself, and a process whose GPU the profiler has initialised must "
                         "not start children)")
    ap.add_argument("--serial", action="store_true",
                    help="one HIP stream only (no encoder lanes / wgrad side stream): kernel durations seen by a profiler "
                         "are then not inflated by co-running kernels -- the mode the roofline step always uses")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves (the parent never touches a GPU)
# ------------------------------------------------------------------------------------------------------------------
def worker_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


LAST_WORKER_CODES = []


def launch_workers(n, argv, program=None, timeout_s=None):
    """N fresh child processes of this script, one rank each; rank 0's stdout is relayed.  Returns the exit code.
    All children are polled: on the first non-zero exit (a rank that died after the rendezvous leaves the others inside
    a collective for ever) or after `timeout_s` (KOAF_BENCH_TIMEOUT_S, default 3000) the rest are terminated, then killed,
    and the code is 1.  The parent only ever spawns fresh children -- it never touches a GPU and never re-execs."""
    import tempfile
    import time
    if timeout_s is None:
        timeout_s = float(os.environ.get("KOAF_BENCH_TIMEOUT_S", "3000"))
    port = free_port()
    cmd = [sys.executable, program or str(Path(__file__).resolve())] + list(argv)
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:          # rank 0's stdout goes to a file: no pipe to fill up and block on
        for r in range(n):
            procs.append(subprocess.Popen(cmd, env=worker_env(r, n, port), stdout=out0 if r == 0 else subprocess.DEVNULL))
        t0, bad, timed_out, beat = time.time(), [], False, time.time()
        while True:
            if time.time() - beat > 60:         # (a heartbeat: the supervisor relays rank 0's line only at the end)
                print(f"bench.py: {n} rank(s) running, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
                beat = time.time()
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad or all(rc is not None for rc in rcs):
                break
            if time.time() - t0 > timeout_s:
                timed_out = True
                break
            time.sleep(0.2)
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t1 = time.time()
        for p in live:
            try:
                p.wait(timeout=max(0.1, 10.0 - (time.time() - t1)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        out0.seek(0)
        txt = out0.read()
    if txt:
        sys.stdout.write(txt)
        sys.stdout.flush()
    del LAST_WORKER_CODES[:]
    LAST_WORKER_CODES.extend(p.returncode for p in procs)
    if timed_out:
        print(f"bench.py: workers still running after {timeout_s:.0f} s: terminated", file=sys.stderr)
        return 1
    if bad:
        print(f"bench.py: worker(s) failed: {bad}; the remaining ranks were terminated", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------------
def workload_cfg(name):
    """-> (model config, default per-GPU batch, default recompute policy)"""
    import procedural as P
    if name == "native":
        return P.cfg_full(dropout=0.1), 8, "none"
    if name == "syn":
        # BASELINE.json's synthetic tensor shapes through the oracle-pinned class: the constructor asserts on CONFIG sizes
        # only and forward never checks tensor shapes: legal config sizes, num_slices = 160
        cfg = P.cfg_full(xr=(320, 320), mr1=(320, 320, 160), mr2=(320, 320, 160), dropout=0.1)
        cfg["_tensor_shapes"] = [[310, 310], [384, 384, 160], [384, 384, 160], [16]]
        return cfg, 2, "none"
    if name in ("native3", "eval3"):
        return P.cfg_xr1mr3c1(dropout=0.1), 8, "none"
    if name == "syn3":
        cfg = P.cfg_xr1mr3c1(xr=(320, 320), mr1=(320, 320, 160), mr2=(320, 320, 160), mr3=(320, 320, 160), dropout=0.1)
        cfg["fe"]["mr"]["volume_layout"] = "ncdhw"          # volumes arrive as BASELINE writes them: 1 x 160 x 384 x 384
        cfg["_tensor_shapes"] = [[310, 310], [160, 384, 384], [160, 384, 384], [160, 384, 384], [16]]
        # batch 8 = 3840 slices of 384^2: rebuild layer1-2 everywhere, layer3 too in the two encoders whose backward runs
        # later (their kept activations would sit under the other encoders' layer1 rebuilds: 102 GB each).  Measured:
        # "012,01,01" peaks at 265 GB allocated / 285 reserved of the 288 -- too close; this policy leaves ~50 GB.
        return cfg, 8, "012,012,01"
    if name == "syn3_rcs":
        # the same model and values with the volumes in the REFERENCE's layout (B, 1, R, C, S) (koafusion/models/_xrNmrMcP.py:209-210):
        # the slice fold is then a real strided transpose (koaf_slice_fold) inside the timed step
        cfg, b, pol = workload_cfg("syn3")
        cfg["fe"]["mr"]["volume_layout"] = "rcs"
        cfg["_tensor_shapes"] = [[310, 310], [384, 384, 160], [384, 384, 160], [384, 384, 160], [16]]
        return cfg, b, pol
    if name == "xr1c1":
        return P.cfg_xr1c1(size=350, dropout=0.5), 32, "none"
    if name == "mr1c1":
        cfg = P.cfg_mr1c1(mr=(320, 320, 160), dropout=0.1)
        cfg["_tensor_shapes"] = [[384, 384, 160], [16]]
        return cfg, 4, "none"
    if name == "xr1cnn":
        return P.cfg_xr1cnn(size=350, dropout=0.5), 4, "none"
    if name == "mr1":
        return P.cfg_mr1(shape=(160, 160, 64), dropout=0.1), 4, "none"
    raise SystemExit(f"unknown workload {name}")


WORKLOAD_TEXT = {
    "native": "XR 1x350x350 + DESS 160x160x64 + T2 160x160x25 + 9 clinical (runner.sh:341-363); random-init weights",
    "native3": "XR 1x350x350 + DESS 160x160x64 + TSE 160x160x32 + T2 160x160x25 + 9 clinical (BASELINE config 4 at the "
               "reference's native sizes; 3-MRI registry extension of XR1MR2C1CnnTrf); random-init weights",
    "syn3": "BASELINE config 4 on BASELINE's synthetic tensors: XR 1x310x310 + 3 x MRI 1x160x384x384 (slice-major) + 9 "
            "clinical; random-init weights",
    "syn3_rcs": "BASELINE config 4 on BASELINE's synthetic tensors with the MRI volumes in the reference's layout (B, 1, 384, 384, 160): "
                "the slice fold of koafusion/models/_xrNmrMcP.py:209-210 is a strided transpose inside the step; random-init weights",
    "eval3": "INFERENCE pass (forward + softmax, eval mode) on the native3 shapes; random-init weights",
    "xr1c1": "BASELINE config 2: XR 1x350x350 + 9 clinical, early-fusion MLP head; random-init weights",
    "mr1c1": "BASELINE config 3: SAG-3D-DESS 1x160x384x384 + 9 clinical; random-init weights",
    "syn": "BASELINE synthetic shapes XR 1x310x310 + 2 x MRI 1x160x384x384 + 9 clinical through the pinned 2-MRI class; "
           "random-init weights",
}


def algorithmic_train_gflop_per_sample(name):
    # measured with torch.utils.flop_counter on the imported reference
    # (ResNet-50 4.1705 GFLOP/slice @160^2, 24.02 @384^2; ResNeXt-50 20.877 @350^2, 16.84 @310^2; FeaT 0.2097/token)
    return {"native": 1280.0, "xr1cnn": 62.1, "mr1": 3 * (64 * 4.1705 + 65 * 0.2097),
            "native3": 1280.0 + 3 * (32 * 4.1705 + 64 * 0.2097),
            "eval3": (1280.0 + 11.3) / 3 + (32 * 4.1705 + 64 * 0.2097),
            "syn": 3 * (16.84 + 320 * 24.02 + 641 * 0.2097),
            "syn3": 3 * (16.84 + 480 * 24.02 + 963 * 0.2097),
            "syn3_rcs": 3 * (16.84 + 480 * 24.02 + 963 * 0.2097),
            "xr1c1": 62.1 + 3 * 2 * (9 * 2048 + 2048 * 512) / 1e9,
            "mr1c1": 3 * (160 * 24.02 + 322 * 0.2097)}[name]


def survey_ideal_bytes_per_sample(name):
    """a perfectly fused train step moves 7 x sum(conv elements) x sizeof(fp32) per image -- forward (in + out),
    backward (dY read twice, X read, dX write) and the BatchNorm-recompute read -- with sum(conv elements) = the mean of the
    measured conv-input and conv-output element counts: ResNet-50 5.555 M per 160^2 slice (x 5.76 at 384^2), ResNeXt-50 34.95 M
    at 350^2 (scaled by area).  None where the survey gives no figure."""
    r50_160, rx_350 = 0.5 * (5.44e6 + 5.67e6), 0.5 * (34.4e6 + 35.5e6)
    per = {"syn3": 480 * r50_160 * 5.76 + rx_350 * (310 / 350.0) ** 2, "syn": 320 * r50_160 * 5.76 + rx_350 * (310 / 350.0) ** 2,
           "native": 89 * r50_160 + rx_350, "native3": 121 * r50_160 + rx_350, "xr1cnn": rx_350, "xr1c1": rx_350,
           "mr1": 64 * r50_160, "mr1c1": 160 * r50_160 * 5.76}.get(name)
    return None if per is None else 7.0 * 4.0 * per


def apply_recompute(model, policy):
    """policy: none | stage | block | comma list of stage-index strings per MRI encoder in forward order"""
    from oaprogressionmmf_amd.models import KoafTrunk
    trunks = [m for m in model.modules() if isinstance(m, KoafTrunk)]
    if policy in ("none", "", None):
        return "none"
    if policy in ("stage", "block"):
        for t in trunks:
            t.recompute = True if policy == "stage" else "block"
        return policy
    # radiograph trunks (few images) keep everything; the MRI trunks, in the order they run forward, get their entries
    mri = [getattr(model, f"_fe{i}") for i in range(getattr(model, "n_xr", 1), getattr(model, "n_xr", 1) + getattr(model, "n_mr", 0))]
    if not mri:
        mri = trunks[1:] if len(trunks) > 1 else trunks
    ents = policy.split(",")
    if len(ents) == 1:
        ents = ents * len(mri)
    if len(ents) != len(mri):
        raise SystemExit(f"--recompute {policy}: {len(ents)} entries for {len(mri)} MRI encoders")
    for t, e in zip(mri, ents):
        t.recompute = [int(ch) for ch in e]
    return policy


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle: checker code, used here only as the timed CPU restatement of the same step)
# ------------------------------------------------------------------------------------------------------------------
def _oracle_job(cfg_fn, B, shapes=None):
    import torch
    import procedural as P
    from oracle import koafusion_cpu as O
    cfg = cfg_fn()
    om = O.OracleModel(cfg, fill=None)
    g = torch.Generator().manual_seed(0)
    for k, v in om.sd.items():      # cheap random init (values do not matter for timing)
        if v.dtype.is_floating_point:
            with torch.no_grad():
                if k.endswith("running_var") or (v.dim() == 1 and k.endswith("weight")):
                    v.fill_(1.0)
                elif v.dim() >= 2:
                    v.copy_(torch.randn(v.shape, generator=g) * (1.0 / max(1, v[0].numel())) ** 0.5)
    shapes_cfg = dict(cfg, input_size=shapes) if shapes else cfg
    xs = [torch.from_numpy(a) for a in P.model_inputs(shapes_cfg, B)]
    y = torch.from_numpy(P.make_target("target", B))
    return om, xs, y


def host_threads():
    """threads the CPU baseline runs on: the cores this process may use -- its affinity mask, cut to the container's CPU-time
    quota (cgroup cpu.max) when there is one: with more runnable threads than quota the kernel throttles the whole group and
    a 64-thread run on a 16-core share is slower than a 16-thread one -- and at most 64"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            break
        except (OSError, ValueError, IndexError):
            continue
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    n = min(n, 64)
    return n, f"affinity {aff} cores, cgroup CPU quota {'none' if quota is None else f'{quota:.1f} cores'}"


def cpu_baseline(workload):
    """The oracle (CPU port of the same train step) on this box's host cores; bounded sample."""
    import torch
    import procedural as P
    if workload in ("mr1c1", "eval3", "syn"):
        return None
    n, why_n = host_threads()
    torch.set_num_threads(n)
    if workload == "syn3":
        # One knee of the headline workload is XR + 3 x 160 slices of 384^2 (minutes on a host CPU).  The slice-wise encoders
        # are linear in the slice count, so the sample times the SAME model and tensor sizes with 32 and with 16 of the 160
        # slices per MRI (batch 1, one full train step each, after a 2-slice warm-up step: 96 / 48 images per convolution call
        # keep all cores busy -- with the 4 / 8 slices of the earlier rounds oneDNN could not fill 64 threads and the slope
        # was pessimistic for the CPU) and extrapolates the per-slice cost: t(160) = t(16) + (160 - 16) * (t(32) - t(16)) / 16.
        # Footnote, as BASELINE.md 3 promises: the reference ships OMP_NUM_THREADS=1 (train_prog_fus.py:7-9); the same
        # extrapolation from 2 and 1 slices per MRI on ONE thread.
        def job(S):
            return _oracle_job(lambda: P.cfg_xr1mr3c1(xr=(320, 320), mr1=(320, 320, S), mr2=(320, 320, S), mr3=(320, 320, S),
                                                      dropout=0.1), 1, [[310, 310], [384, 384, S], [384, 384, S], [384, 384, S], [16]])

        def one(S):
            om, xs, y = job(S)
            t0 = time.time()
            om.train_step(xs, y)
            return time.time() - t0
        one(2)                                           # warm-up (thread pool, primitive caches)
        hi, lo = (32, 16) if n >= 16 else (8, 4)         # (a small host would need minutes for 96 images: keep the sample bounded)
        ts = {S: one(S) for S in (hi, lo)}
        per_slice = max((ts[hi] - ts[lo]) / float(hi - lo), 1e-9)
        t160 = ts[lo] + (160 - lo) * per_slice
        torch.set_num_threads(1)
        t1 = {S: one(S) for S in (2, 1)}
        torch.set_num_threads(n)
        t160_1 = t1[1] + 159 * max(t1[2] - t1[1], 1e-9)
        return {"value": round(1.0 / t160, 5), "unit": "knees/s", "cores": n, "kind": "port",
                "sample": f"oracle (CPU port of the same model and step: fwd + focal loss + bwd + Adam) at batch 1 on {n} torch "
                          f"threads with {hi} and with {lo} of the 160 slices per MRI at the full 384x384 / 310x310 sizes (one train "
                          f"step each, after a 2-slice warm-up step): {ts[hi]:.1f} s and {ts[lo]:.1f} s; the slice-wise encoders are "
                          f"linear in the slice count, so one knee at 160 slices = t({lo}) + {160 - lo} x (t({hi}) - t({lo})) / {hi - lo} "
                          f"= {t160:.0f} s",
                "measured_s": {f"slices_{hi}": round(ts[hi], 2), f"slices_{lo}": round(ts[lo], 2)},
                "extrapolated_s_per_knee": round(t160, 1), "host": why_n,
                "extrapolation_validated": "the same line checked once against ONE train step at 144 slices per MRI on a GPU-box host (16 threads, "
                                           "120.8 s measured vs 131.7 s predicted from the 32- and 16-slice steps: the line over-predicts by 9 %, "
                                           "i.e. this baseline is ~9 % pessimistic for the CPU; profiles/r04_cpu_baseline_validation.log, "
                                           "scripts/validate_cpu_baseline.py)",
                "one_thread_footnote": {"value": round(1.0 / t160_1, 6), "unit": "knees/s", "cores": 1,
                                        "why": "the reference ships OMP_NUM_THREADS=1 (train_prog_fus.py:7-9)",
                                        "measured_s": {"slices_2": round(t1[2], 2), "slices_1": round(t1[1], 2)},
                                        "extrapolated_s_per_knee": round(t160_1, 1)}}
    cfg, _, _ = workload_cfg(workload)
    om, xs, y = _oracle_job(lambda: cfg, 1)
    om.train_step(xs, y)                     # warm-up
    steps = 2 if workload in ("native", "native3") else 3
    t0 = time.time()
    for _ in range(steps):
        om.train_step(xs, y)
    dt = time.time() - t0
    return {"value": round(steps / dt, 4), "unit": "knees/s", "cores": n, "kind": "port",
            "sample": f"{steps} full train steps (fwd+focal loss+bwd+Adam) of the same model/shapes at batch 1 "
                      f"on the host CPU ({n} torch threads), after 1 warm-up step"}


# ------------------------------------------------------------------------------------------------------------------
def main(args):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # KOAF_DIST_BACKEND=gloo + KOAF_ONE_DEVICE=1: rehearsal of the N>1 path with all ranks on one GPU (the
    # 1-GPU development box has no second device for RCCL); the driver's real runs use nccl (= RCCL).
    backend = os.environ.get("KOAF_DIST_BACKEND", "nccl")
    if os.environ.get("KOAF_ONE_DEVICE"):
        local = 0
    # KOAF_DIST_REHEARSAL=1: a world of ONE still builds the RCCL process group and runs every collective of the N>1
    # path (parameter broadcast, bucketed gradient all-reduce, barrier, MAX over ranks) -- what a 1-GPU box can rehearse.
    dist_on = world > 1 or bool(os.environ.get("KOAF_DIST_REHEARSAL"))
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import procedural as P
    from oaprogressionmmf_amd import ops
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import dict_models, _common, _encoder
    from oaprogressionmmf_amd.parallel import DataParallelRCCL
    from oaprogressionmmf_amd.various import dict_losses, dict_optimizers, set_ultimate_seed

    if os.environ.get("KOAF_HALO_MODE"):      # A/B runs: force the 3x3 halo kernel shape (koaf.h koaf_set_conv3x3_halo)
        ops.set_conv3x3_halo(int(os.environ["KOAF_HALO_MODE"]))
    set_ultimate_seed(777 + 16 * rank)   # distinct dropout streams per rank; rank 0's parameters are broadcast
    lanes_default = (_common.USE_LANES, _encoder.USE_SIDE_STREAM)
    if args.serial:
        _encoder.USE_SIDE_STREAM = False
        _common.USE_LANES = False

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    comm_ms = []
    jobs_ddp = []
    rank_dt = []          # wall time of the last timed() region on every rank (N > 1)

    job_state = {}

    def make_job(name, batch, recompute="auto", graph=None):
        """model + optimizer + resident synthetic batch of one workload -> (cfg, B, policy, step).  A name ending in `_bf16`
        is the same workload in the bf16 activation-storage mode (config key `activation_storage`), with its own default
        recompute policy where the halved activations allow a lighter one."""
        graph = bool(args.graph) if graph is None else graph
        bf16 = name.endswith("_bf16")
        cfg, bdef, rdef = workload_cfg(name[:-5] if bf16 else name)
        if bf16:
            cfg["activation_storage"] = "bf16"
            rdef = BF16_POLICY.get(name[:-5], rdef)
        B = batch or bdef
        shapes = cfg.pop("_tensor_shapes", None)
        model = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None).to(dev)
        job_state.update(model=model, cfg=cfg, shapes=shapes, B=B)
        policy = apply_recompute(model, rdef if (recompute == "auto" and B >= bdef) else ("none" if recompute == "auto" else recompute))
        ddp = DataParallelRCCL(model, exchange_always=dist_on)
        ddp.time_exposed = dist_on
        jobs_ddp.append(ddp)
        loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)
        opt = dict_optimizers["Adam"](model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=graph)
        if name == "syn3_rcs":
            # the SAME values as syn3's slice-major volumes, laid out as the reference's (B, 1, R, C, S) tensors
            ncd = [[310, 310]] + [[160, 384, 384]] * 3 + [[16]]
            xs = [torch.from_numpy(a).to(dev) for a in P.model_inputs(dict(cfg, input_size=ncd), B, seed=1234 + rank)]
            xs = [x.permute(0, 1, 3, 4, 2).contiguous() if x.dim() == 5 else x for x in xs]
        else:
            xs = [torch.from_numpy(a).to(dev) for a in P.model_inputs(dict(cfg, input_size=shapes) if shapes else cfg, B, seed=1234 + rank)]
        y = torch.from_numpy(P.make_target("target", B, seed=1234 + rank)).to(dev)
        job_state.update(xs=xs, y=y, loss_fn=loss_fn)
        model.train()
        if name == "eval3":
            from oaprogressionmmf_amd.run import predict_batch
            model.eval()

            def step_eval():
                logits, proba = predict_batch(model, xs)
                return float(proba[0, 0].item())       # the driver's per-batch host read (argmax / softmax go to the CPU)
            return cfg, B, policy, step_eval

        if graph:
            if dist_on:
                raise SystemExit("--graph captures single-GPU steps")
            from oaprogressionmmf_amd.run import GraphedTrainStep
            gstep = GraphedTrainStep(model, lambda lg, tg: loss_fn(input=lg, target=tg), opt, xs, y, warmup=2)

            def step_graph():
                return gstep(xs, y)[1].item()
            return cfg, B, policy, step_graph

        def step():
            opt.zero_grad()
            logits = ddp(*xs)["main"]
            loss = loss_fn(input=logits.squeeze(1), target=y.long().squeeze(1))
            ddp.scale_loss(loss).backward()
            if dist_on:
                ddp.reduce_gradients()      # (time_exposed: events around the compute stream's waits on the collectives)
            opt.step()
            return loss.item()          # (see the module docstring: `host_sync`)
        return cfg, B, policy, step

    def timed(step, warmup, steps):
        """W untimed + exactly K timed steps between barriers (max over ranks); per-step event times for the median"""
        for _ in range(warmup):
            step()
        barrier()
        for d in jobs_ddp:
            d.exposed_ms()              # (drop the warm-up steps' brackets)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            lv = step()
            evs[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
        if dist_on:
            mine = torch.tensor([dt], device=dev, dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine)
            rank_dt[:] = [float(t.item()) for t in allr]
            dt = max(rank_dt)               # the job's time is the slowest rank's
        return dt, lv, per

    def free(*objs):
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    rccl_ranks = None
    if dist_on:
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        rccl_ranks = int(one.item())

    cfg, B, policy, step = make_job(args.workload, args.batch, args.recompute)
    torch.cuda.reset_peak_memory_stats()
    dt, lv, per = timed(step, args.warmup, args.steps)
    per_rank_ms = [round(t / args.steps * 1e3, 2) for t in rank_dt] if rank_dt else None
    hbm_gb = (round(torch.cuda.max_memory_allocated() / 2**30, 1), round(torch.cuda.max_memory_reserved() / 2**30, 1))
    comm_exposed, comm_exposed_ranks = None, None
    if dist_on:
        comm_ms = jobs_ddp[0].exposed_ms()
        mine = torch.tensor([statistics.mean(comm_ms) if comm_ms else 0.0], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        comm_exposed_ranks = [round(float(t.item()), 3) for t in allr]
        comm_exposed = max(comm_exposed_ranks)           # (the slowest rank's: what the step pays)

    if args.graph:
        if rank == 0:
            print(json.dumps({"metric": "knees/sec full XR+MRI+clin fusion train step", "value": round(world * B * args.steps / dt, 3),
                              "unit": "knees/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_step_median": round(statistics.median(per), 3),
                              "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": f"{args.workload}: {cfg['name']} train step replayed from a HIP graph "
                                                     f"(run.GraphedTrainStep), per-GPU batch {B}", "last_loss": round(lv, 6)}}))
        return
    # one extra instrumented step: live event timing of every MFMA-GEMM launch on its launch stream
    _encoder.USE_SIDE_STREAM = False      # serialise: per-kernel durations are not inflated by co-running kernels
    _common.USE_LANES = False
    torch.cuda.synchronize()
    torch.cuda.empty_cache()              # the serial step allocates from another stream pool: start it clean
    step()                                # (untimed) settle the allocator on the new pool
    torch.cuda.synchronize()
    ops.PROFILE = []
    step()
    torch.cuda.synchronize()
    prof = [(f, fl, e0.elapsed_time(e1), tag, nb, mpp) for f, fl, e0, e1, tag, nb, mpp in ops.PROFILE if f == "gemm"]
    ops.PROFILE = None
    gemm_ms = sum(p[2] for p in prof)
    gemm_flop = sum(p[1] for p in prof)
    gemm_bytes = sum(p[4] for p in prof)
    n_launch = len(prof)
    achieved_exec = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    # ALGORITHMIC work of the step (per-sample figure x the samples one step processes) over the measured duration of
    # the dominant kernel family: recomputed forward stages are time spent, not work done
    algo_flop = algorithmic_train_gflop_per_sample(args.workload) * 1e9 * B
    achieved = algo_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0

    def call_peak(mpp):
        return BF16_MFMA_PEAK_TFLOPS / mpp

    def call_floor_ms(fl, nb, mpp):
        """roofline time of one call: the larger of its matrix-pipe time and the time to move each operand once"""
        return max(fl / (call_peak(mpp) * 1e12), nb / (HBM_PEAK_TBPS * 1e12)) * 1e3
    floor_ms = sum(call_floor_ms(fl, nb, mpp) for f, fl, ms, tag, nb, mpp in prof)
    hbm_ms = sum(ms for f, fl, ms, tag, nb, mpp in prof if nb / (HBM_PEAK_TBPS * 1e12) > fl / (call_peak(mpp) * 1e12))
    if args.breakdown and rank == 0:
        agg = {}
        for f, fl, ms, tag, nb, mpp in prof:
            a = agg.setdefault((tag, mpp), [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += ms; a[3] += nb; a[4] += call_floor_ms(fl, nb, mpp)
        with open(args.breakdown, "w") as fh:
            fh.write(f"# per-shape GEMM-family calls of one train step ({args.workload}, batch {B}, recompute {policy}); ms from events on "
                     f"the launch stream;\n# mpp = matrix instructions per product (3: fp16 x 2 scheme, 6: bf16 x 3); MB = algorithmic HBM "
                     f"bytes (every operand once); floor = max(FLOP / (2500 TFLOP/s / mpp), bytes / {HBM_PEAK_TBPS:g} TB/s); bound = which "
                     f"term of the floor is larger\n")
            fh.write(f"{'call':48s} {'mpp':>3s} {'n':>4s} {'ms':>9s} {'GFLOP':>10s} {'TFLOP/s':>8s} {'MB':>9s} {'TB/s':>6s} {'floor ms':>9s} {'bound':>5s}\n")
            for (tag, mpp), (n, fl, ms, nb, flo) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                bound = "hbm" if nb / (HBM_PEAK_TBPS * 1e12) > fl / (call_peak(mpp) * 1e12) else "mfma"
                fh.write(f"{tag:48s} {mpp:3d} {n:4d} {ms:9.3f} {fl / 1e9:10.1f} {fl / ms / 1e9 if ms > 0 else 0:8.1f} {nb / 1e6:9.1f} "
                         f"{nb / ms / 1e9 if ms > 0 else 0:6.2f} {flo:9.3f} {bound:>5s}\n")
    bound_s = sum(fl / (call_peak(mpp) * 1e12) for f, fl, ms, tag, nb, mpp in prof)
    peak_mix = gemm_flop / bound_s / 1e12 if bound_s > 0 else call_peak(3)
    f16_share = sum(fl for f, fl, ms, tag, nb, mpp in prof if mpp == 3) / gemm_flop if gemm_flop else 0.0
    # HBM-side bytes per launch of the dominant kernel cannot be counted from inside the process: they come from
    # the committed PMC summary of this same workload (profiles/README.md has the command and the corrections)
    traffic, traffic_src, traffic_step = None, None, None
    for tj in sorted((ROOT / "profiles").glob(f"r*_gemm_traffic_{args.workload}.json"), reverse=True):      # the newest round's
        try:
            tjd = json.loads(tj.read_text())
            if tjd.get("batch") == B:
                traffic, traffic_src = tjd["bytes_per_launch"], f"profiles/{tj.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
                traffic_step = {k: tjd[k] for k in ("bytes_per_step", "act_planes_bytes_per_step", "algorithmic_bytes_per_step",
                                                    "ratio_to_algorithmic", "ratio_to_algorithmic_with_act_planes") if k in tjd}
                break
        except (ValueError, KeyError):
            pass

    traffic_vs_ideal = None
    ideal_b = survey_ideal_bytes_per_sample(args.workload)
    if traffic_step and ideal_b:
        moved = traffic_step.get("bytes_per_step", 0) + traffic_step.get("act_planes_bytes_per_step", 0)
        traffic_vs_ideal = {"ratio": round(moved / (ideal_b * B), 3), "survey_ideal_bytes_per_step": int(ideal_b * B),
                            "pmc_bytes_per_step": int(moved),
                            "ideal_is": "SURVEY 8(d): 7 x sum(conv elements) x 4 B per image (every tensor of a perfectly fused step once: "
                                        "forward in + out, backward dY twice, X, dX, one BatchNorm-recompute read), x the images of the "
                                        "batch; pmc = GEMM-family launches + plane-image pre-passes of the committed PMC summary"}

    def one_step_outputs(perturb=False):
        """eval logits, train logits / loss and the flat gradient arena of ONE forward + backward of the current job's model on
        its batch, dropout seeded alike on every call (the masks are index hashes: the same in either storage mode).
        perturb: MRI / XR inputs rounded to bf16 once -- how far one 8-bit-significand rounding moves the fp32 mode."""
        from oaprogressionmmf_amd.arena import get_arena
        m, xs_, y_, lf = job_state["model"], job_state["xs"], job_state["y"], job_state["loss_fn"]
        if perturb:
            xs_ = [x.bfloat16().float() if x.dim() >= 4 else x for x in xs_]
        with torch.no_grad():
            m.eval()
            ev = m(*xs_)["main"].float().clone()
        m.train()
        bufs = {k: b.detach().clone() for k, b in m.named_buffers()}
        set_ultimate_seed(4242)
        m.zero_grad()
        lg = m(*xs_)["main"]
        ls = lf(input=lg.squeeze(1), target=y_.long().squeeze(1))
        ls.backward()
        torch.cuda.synchronize()
        out = dict(eval=ev, logits=lg.detach().float().clone(), loss=float(ls.detach()), grad=get_arena(m).G.clone())
        with torch.no_grad():
            for k, b in m.named_buffers():
                b.copy_(bufs[k])                    # (the measurement leaves the running statistics as it found them)
        m.zero_grad()
        return out

    def rel_(a, b):
        return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))

    def storage_pair(name, steps2, warm2, timed_fp32):
        """the bf16 activation-storage mode of workload `name` beside its fp32 mode: throughput of both, and the MEASURED
        difference of one train step from the same weights (eval logits, train logits, loss, the whole gradient vector) --
        with, for scale, what a single bf16 rounding of the inputs does to the fp32 mode"""
        res = {}
        if timed_fp32 is None:
            cfgf, Bf, polf, stepf = make_job(name, 0)
            dtf, lvf, perf = timed(stepf, warm2, steps2)
            timed_fp32 = {"value": round(world * Bf * steps2 / dtf, 3), "ms_per_step": round(dtf / steps2 * 1e3, 2), "steps": steps2,
                          "warmup": warm2, "activation_recompute": polf}
            del stepf
        res["fp32"] = timed_fp32
        ref = one_step_outputs()
        sens = one_step_outputs(perturb=True)
        weights = {k: v.detach().clone() for k, v in job_state["model"].state_dict().items()}
        job_state.clear()
        free()
        cfgb, Bb, polb, stepb = make_job(name + "_bf16", 0)
        job_state["model"].load_state_dict(weights)
        del weights
        got = one_step_outputs()
        torch.cuda.reset_peak_memory_stats()
        dtb, lvb, perb = timed(stepb, warm2, steps2)
        val = world * Bb * steps2 / dtb
        gf = algorithmic_train_gflop_per_sample(name)
        res["bf16"] = {"value": round(val, 3), "unit": "knees/s", "ms_per_step": round(dtb / steps2 * 1e3, 2),
                       "ms_per_step_median": round(statistics.median(perb), 2), "steps": steps2, "warmup": warm2,
                       "per_gpu_batch": Bb, "activation_recompute": polb, "last_loss": round(lvb, 6),
                       "hbm_peak_gib": {"allocated": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                                        "reserved": round(torch.cuda.max_memory_reserved() / 2**30, 1)},
                       "speedup_vs_fp32_mode": round(val / res["fp32"]["value"], 3),
                       "algorithmic_tflops": round(gf * val / 1e3 / world, 1),
                       # bf16 activations put the fused conv chain at AI ~ 160 FLOP/B: HBM-bound ceiling 160 x 8 TB/s
                       "frac_of_hbm_ceiling": round(gf * val / 1e3 / world / (160.0 * HBM_PEAK_TBPS), 4),
                       "hbm_ceiling_is": "SURVEY 8(d): 7 x sum(conv in+out elements) x 2 B per slice against its FLOPs = AI 160 FLOP/B; "
                                         "x 8 TB/s = 1280 TFLOP/s algorithmic",
                       "measured_error_vs_fp32_mode": {
                           "eval_logits_rel": float(f"{rel_(got['eval'], ref['eval']):.3e}"),
                           "train_logits_rel": float(f"{rel_(got['logits'], ref['logits']):.3e}"),
                           "loss_abs": float(f"{abs(got['loss'] - ref['loss']):.3e}"),
                           "gradient_rel_l2": float(f"{rel_(got['grad'], ref['grad']):.3e}"),
                           "for_scale_one_bf16_rounding_of_the_inputs_in_fp32_mode": {
                               "eval_logits_rel": float(f"{rel_(sens['eval'], ref['eval']):.3e}"),
                               "train_logits_rel": float(f"{rel_(sens['logits'], ref['logits']):.3e}"),
                               "gradient_rel_l2": float(f"{rel_(sens['grad'], ref['grad']):.3e}")},
                           "what": "one forward + backward from the same weights, batch and dropout masks in both storage modes; "
                                   "rel = ||a - b|| / ||b||, gradient = the whole flat gradient arena.  Storage only: arithmetic, "
                                   "statistics, gradients and parameters stay fp32 (tests/test_bf16_gpu.py: bit-identical to the fp32 mode on "
                                   "widened inputs, kernel by kernel).  ReLU masks are taken from the rounded activations, which alone moves "
                                   "a gradient by ~5e-2 per block; through ~50 train-mode BatchNorms with random weights the map amplifies "
                                   "any perturbation (see the for_scale entry)"}}
        del stepb, ref, sens, got
        job_state.clear()
        free()
        return res

    secondary = {}
    if (args.workload == "syn3" and not args.batch and args.recompute == "auto" and not args.no_secondary
            and args.steps >= 10):
        _common.USE_LANES, _encoder.USE_SIDE_STREAM = (False, False) if args.serial else lanes_default
        try:
            del step
            pr = storage_pair("syn3", 10, 3, {"value": round(world * B * args.steps / dt, 3), "ms_per_step": round(dt / args.steps * 1e3, 2),
                                              "steps": args.steps, "warmup": args.warmup, "activation_recompute": policy})
            secondary["syn3_bf16"] = dict(pr["bf16"], workload=WORKLOAD_TEXT["syn3"] + "; bf16 activation storage")
        except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
            secondary["syn3_bf16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            job_state.clear()
            free()
        try:
            # the reference's tensor layout (B, 1, R, C, S): the slice-fold kernel inside the timed step (5 steps)
            cfgr, Br, polr, stepr = make_job("syn3_rcs", 0)
            dtr, lvr, perr = timed(stepr, 2, 5)
            xv = job_state["xs"][1]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ops.slice_fold(xv, Br, 384, 384, 160)
            e0.record()
            for _ in range(5):
                ops.slice_fold(xv, Br, 384, 384, 160)
            e1.record()
            torch.cuda.synchronize()
            fold_ms = e0.elapsed_time(e1) / 5
            secondary["syn3_rcs"] = {"model": cfgr["name"], "value": round(world * Br * 5 / dtr, 3), "unit": "knees/s",
                                     "ms_per_step": round(dtr / 5 * 1e3, 2), "ms_per_step_median": round(statistics.median(perr), 2),
                                     "steps": 5, "warmup": 2, "per_gpu_batch": Br, "activation_recompute": polr, "last_loss": round(lvr, 6),
                                     "slice_fold": {"ms_per_volume_batch": round(fold_ms, 3), "calls_per_step": 3,
                                                    "tb_per_s": round(2 * xv.numel() * 4 / (fold_ms * 1e-3) / 1e12, 2),
                                                    "what": "koaf_slice_fold of one (8, 1, 384, 384, 160) fp32 volume batch: read + write "
                                                            "755 MB each (koafusion/models/_xrNmrMcP.py:209-210)"},
                                     "workload": WORKLOAD_TEXT["syn3_rcs"]}
            del stepr, xv
            job_state.clear()
            free()
        except Exception as e:  # noqa: BLE001
            secondary["syn3_rcs"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            job_state.clear()
            free()
        try:
            pr = storage_pair("xr1c1", 20, 5, None)
            secondary["xr1c1"] = dict(pr["fp32"], per_gpu_batch=32, workload=WORKLOAD_TEXT["xr1c1"] + " (fp32 mode)")
            secondary["xr1c1_bf16"] = dict(pr["bf16"], workload=WORKLOAD_TEXT["xr1c1"] + "; bf16 activation storage = the configuration "
                                                                                        "as BASELINE.json writes it (batch 32 bf16)")
        except Exception as e:  # noqa: BLE001
            secondary["xr1c1_bf16"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            job_state.clear()
            free()
        # the launch-bound small configurations (BASELINE configs 2 and 1) as ONE HIP graph per train step (run.GraphedTrainStep:
        # same kernels on device-resident step state, bit-identical to the eager step -- tests/test_run_gpu.py)
        if not dist_on:
            for name in ("xr1c1", "xr1cnn"):
                try:
                    cfgg, Bg, polg, stepg = make_job(name, 0, graph=True)
                    dtg, lvg, perg = timed(stepg, 5, 20)
                    secondary[name + "_graph"] = {"model": cfgg["name"], "value": round(world * Bg * 20 / dtg, 3), "unit": "knees/s",
                                                  "ms_per_step": round(dtg / 20 * 1e3, 3), "ms_per_step_median": round(statistics.median(perg), 3),
                                                  "steps": 20, "warmup": 5, "per_gpu_batch": Bg, "last_loss": round(lvg, 6),
                                                  "workload": WORKLOAD_TEXT.get(name, "BASELINE config 1: XR 1x350x350, XR1Cnn; random-init weights")
                                                  + "; the whole train step replayed from one HIP graph"}
                    del stepg
                    job_state.clear()
                    free()
                except Exception as e:  # noqa: BLE001
                    secondary[name + "_graph"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                    job_state.clear()
                    free()
        for name in ("native3", "native"):
            try:
                cfg2, B2, pol2, step2 = make_job(name, 0)
                dt2, lv2, per2 = timed(step2, 3, 10)
                secondary[name] = {"model": cfg2["name"], "value": round(world * B2 * 10 / dt2, 3), "unit": "knees/s",
                                   "ms_per_step": round(dt2 / 10 * 1e3, 2), "ms_per_step_median": round(statistics.median(per2), 2),
                                   "steps": 10, "warmup": 3, "per_gpu_batch": B2, "last_loss": round(lv2, 6),
                                   "workload": WORKLOAD_TEXT[name]}
                del step2
                free()
            except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                secondary[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": ("knees/sec full XR+MRI+clin fusion inference pass" if args.workload == "eval3"
                       else "knees/sec full XR+MRI+clin fusion train step"),
            "value": round(value, 3), "unit": "knees/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "ms_per_step_median": round(statistics.median(per), 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 tensors, fp32 accumulation, every product at fp32 rounding level (forward and gradients alike): "
                          "contractions form each fp32 x fp32 product on the 16-bit matrix pipe from exact pieces of the operands -- "
                          "convolutions: two fp16 pieces of operand x 2^e (e from the tensor's max magnitude) and three products "
                          "(hi*hi, hi*lo, lo*hi; <= 2^-24 relative per operand and for the dropped lo*lo); linear / attention: "
                          "three bf16 pieces, six products",
            "host_sync": "loss.item() is read after optimizer.step() is enqueued; the reference reads the same value between forward "
                         "and backward (train_prog_fus.py:159-165), which would drain the GPU in the middle of the step",
            "config": {"workload": f"{args.workload}: {cfg['name']} "
                                   + ("inference pass (forward + softmax), " if args.workload == "eval3"
                                      else "train step (fwd+FocalLoss+bwd+Adam), ") +
                                   f"per-GPU batch {B}, global batch {world * B}, " + WORKLOAD_TEXT.get(args.workload, "random-init weights"),
                       "parallelism": f"dp{world}", "last_loss": round(lv, 6),
                       "activation_recompute": policy + (" (fallback: the default policy 012,012,01 ran out of memory on this box; "
                                                         "retried once in a fresh process)" if os.environ.get("KOAF_BENCH_OOM_RETRY") else ""),
                       "hbm_peak_gib": {"allocated": hbm_gb[0], "reserved": hbm_gb[1]}},
            "rccl_ranks": rccl_ranks, "comm_exposed_ms": comm_exposed,
            "per_rank": {"ms_per_step": per_rank_ms, "comm_exposed_ms": comm_exposed_ranks} if dist_on else None,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak_mix, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / peak_mix, 4),
                         "achieved_executed": round(achieved_exec, 2), "frac_executed": round(achieved_exec / peak_mix, 4),
                         "traffic": traffic,
                         "traffic_unit": "bytes per GEMM-family kernel launch (average)", "traffic_source": traffic_src,
                         "traffic_per_step": traffic_step,
                         "kernel": "koaf_gemm_kernel + wgrad3x3_ring_kernel (implicit GEMM: conv fwd/dgrad/wgrad, linear, attention; fp32 in/out/accumulate; "
                                   "products on v_mfma_f32_32x32x16_f16 from two scaled fp16 pieces per operand (3 MFMAs per product: "
                                   "convolutions) or on v_mfma_f32_32x32x16_bf16 from three bf16 pieces (6 MFMAs: linear, attention, "
                                   "grouped conv))",
                         "peak_is": "2.5 PFLOP/s dense 16-bit MFMA / MFMAs per product, FLOP-weighted harmonic mean over the step's calls "
                                    f"({100 * f16_share:.0f} % of the executed FLOPs on the 3-MFMA scheme at 833.3, the rest at 416.7 TFLOP/s "
                                    "fp32-equivalent)",
                         "achieved_is": "ALGORITHMIC FLOPs of the step (SURVEY 8(d): step_gflop_per_sample_survey x the per-GPU batch) / the "
                                        "event-timed durations of the step's GEMM-family calls on the launch stream (one instrumented "
                                        "single-stream step); achieved_executed / frac_executed count what the calls execute instead "
                                        "(recomputed forward stages included)",
                         "fp32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                         "achieved_over_fp32_mfma_peak": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                         "traffic_vs_survey_ideal": traffic_vs_ideal,
                         "floor_ms_per_step": round(floor_ms, 2), "frac_of_floor": round(floor_ms / gemm_ms, 4) if gemm_ms > 0 else None,
                         "hbm_bound_share_of_kernel_ms": round(hbm_ms / gemm_ms, 4) if gemm_ms > 0 else None,
                         "floor_is": "sum over the step's calls of max(FLOP / matrix-pipe bound of the call, algorithmic bytes / 8 TB/s): "
                                     "SURVEY 8(d)'s attainable = min(MFMA peak, AI x HBM) applied per call; frac_of_floor = floor / "
                                     "measured kernel time; `frac` above stays the plain achieved / matrix-pipe peak",
                         "launches_per_step": n_launch, "kernel_ms_per_step": round(gemm_ms, 2),
                         "executed_gflop_per_step": round(gemm_flop / 1e9, 1),
                         "algorithmic_gbytes_per_step": round(gemm_bytes / 1e9, 1),
                         "step_gflop_per_sample_survey": algorithmic_train_gflop_per_sample(args.workload),
                         "step_tflops_algorithmic": round(algorithmic_train_gflop_per_sample(args.workload) * value / 1e3 / world, 1)},
        }
        out["numerics"] = ops.numerics_status()       # clamped activations / non-finite operand scales over the whole run: must be 0 / 0
        if secondary:
            out["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload)
            if cb is not None:
                out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


# recompute policies of the bf16 activation-storage mode (half the bytes per saved activation: layer3 is kept everywhere,
# layer2 in the encoder whose backward runs first)
BF16_POLICY = {"syn3": "01,01,0"}

OOM_EXIT = 42
# Every stage of every trunk rebuilt ONE BLOCK AT A TIME: the rebuilt activations alive at once shrink from a whole stage
# (layer1 of one encoder: 102 GB at batch 8) to one block plus the stage's block inputs -- ~20 GB less at the peak, for one
# more forward of most blocks.  (Moving layer3 of the third encoder into the rebuilt set, "012,012,012", does NOT lower the
# peak: it sits at that encoder's layer1 rebuild either way -- measured in tests/test_fullsize_gpu.py.)
FALLBACK_POLICY = "block"


def is_oom(e):
    import torch
    return isinstance(e, torch.cuda.OutOfMemoryError) or "out of memory" in str(e).lower()


def fallback_allowed(args):
    """the headline run with its default policy, not already a retry"""
    return args.workload == "syn3" and args.recompute == "auto" and not args.batch and not os.environ.get("KOAF_BENCH_OOM_RETRY")


def fallback_argv(argv):
    return list(argv) + ["--recompute", FALLBACK_POLICY]


def launch_with_fallback(args, argv, program=None):
    """start the ranks; if one of them ran out of memory on the headline's default policy, ONE more launch with the leaner one
    (FALLBACK_POLICY)"""
    rc = launch_workers(args.gpus, argv, program=program)
    if rc != 0 and OOM_EXIT in LAST_WORKER_CODES and fallback_allowed(args):
        print(f"bench.py: a rank ran out of memory with the default recompute policy: one retry with --recompute {FALLBACK_POLICY}", file=sys.stderr)
        os.environ["KOAF_BENCH_OOM_RETRY"] = "1"
        try:
            rc = launch_workers(args.gpus, fallback_argv(argv), program=program)
        finally:
            del os.environ["KOAF_BENCH_OOM_RETRY"]
    return rc


if __name__ == "__main__":
    _args = parse_args()
    if "WORLD_SIZE" not in os.environ and not _args.inproc:
        # No launcher: this process is a GPU-free SUPERVISOR for every N, 1 included -- it starts the N measuring ranks as fresh
        # children, relays rank 0's line, and on an out-of-memory exit of the headline's default policy starts ONE more set with
        # the leaner policy only after the first set has exited (nothing of the failed run is left on the device, so the
        # retry's headroom and its reported hbm_peak_gib are those of a clean run).
        sys.exit(launch_with_fallback(_args, sys.argv[1:]))
    try:
        main(_args)
        sys.exit(0)
    except Exception as _e:  # noqa: BLE001
        if not is_oom(_e):
            raise
        print(f"bench.py: out of memory: {str(_e)[:300]}", file=sys.stderr)
        sys.exit(OOM_EXIT)              # (a rank: the supervisor / launcher decides; with --inproc there is no retry)
