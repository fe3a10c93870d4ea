#!/usr/bin/env python3
"""bench.py -- knees/sec of the full XR + MRI + clinical fusion train step (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload W] [--batch B] [--recompute]
  (N>1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one iteration of the reference's step body (koafusion/run/train_prog_fus.py:132-168) on one
synthetic batch already resident in HBM: zero_grad -> forward (dropout 0.1 as in the recipe, runner.sh:352) ->
FocalLoss -> loss.item() (the reference's per-step host sync) -> backward -> gradient all-reduce over RCCL
(N>1) -> fused Adam.  Weak scaling: the per-GPU batch is fixed, `value` = global samples / max-over-ranks time.

Workloads:
  native3 (default) BASELINE config 4 literally -- "Full XR + SAG-DESS/COR-IW-TSE/SAG-T2 + clinical transformer fusion,
                    batch 8": the registry extension XR1MR3C1CnnTrf (the reference's XR1MR2C1CnnTrf pattern with a third
                    MRI slot) at the reference's native sizes: XR 1x350x350 (ResNeXt-50), DESS 160x160x64, TSE 160x160x32,
                    T2 160x160x25 (ResNet-50 slice-wise), 9 clinical variables, 4 x FeaT(depth 4, 8 heads, width 2048).
                    The same run also times `native` and reports it under "pinned_reference_model", and (1 GPU) the same model
                    on BASELINE's synthetic tensor shapes at batch 8 under "baseline_synthetic_shapes".
  native            XR1MR2C1CnnTrf exactly as runner.sh:341-363 (the reference's biggest registered model, the
                    reference-pinned 2-MRI mapping of config 4): XR 350^2 + DESS 160x160x64 + T2 160x160x25 + clinical; B=8.
  syn / syn3        BASELINE's synthetic tensor shapes (XR 1x310x310, MRI 1x160x384x384) through the 2-MRI / 3-MRI model
                    (default per-GPU batch 2; batch 8 needs --recompute).
  xr1cnn / xr1c1    BASELINE configs 1 / 2 (XR1Cnn B=4; extension XR1C1Cnn = XR + clinical MLP head, B=32) @350^2.
  mr1 / mr1c1       BASELINE config 3: MR1CnnTrf B=4 @160x160x64; extension MR1C1CnnTrf (DESS + clinical) B=4 @384x384x160.
  eval3             inference pass of the native3 model (eval() mode, no autograd: run.predict_batch = forward + softmax),
                    the evaluation regime of koafusion/run/eval_prog_fus.py; `value` = knees/s scored.
The JSON line carries `roofline` for the dominant kernel (the MFMA GEMM, timed live with events on the
launch stream over one extra instrumented step) and `cpu_baseline` (the oracle = CPU port of the same step,
timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this pool (already exported there)

import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak, v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md, chip-level table)
# The GEMM computes fp32 x fp32 products on the bf16 matrix pipe (koaf_gemm.hip): forward contractions cut each operand
# exactly into three bf16 pieces and issue the six piece products of weight >= 2^-16 as v_mfma_f32_32x32x16_bf16;
# gradient contractions round the operands to 16 significand bits (two pieces) and issue all four products.  The bound of
# a call is the dense bf16 MFMA peak (2.5 PFLOP/s, same table) divided by its MFMAs per product; the bound of the step's
# mix is the FLOP-weighted harmonic mean over its calls.
BF16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_TBPS = 8.0
MFMA_PER_PRODUCT_FWD, MFMA_PER_PRODUCT_BWD = 6, 4


def workload_cfg(name):
    import procedural as P
    if name == "native":
        return P.cfg_full(dropout=0.1), 8
    if name == "syn":
        # BASELINE.json's synthetic tensor shapes (XR 1x310x310, MRI 1x160x384x384 = (384,384,160) in the reference's
        # (R,C,S) layout) through the oracle-pinned class: the constructor asserts on CONFIG sizes only and forward never
        # checks tensor shapes, so the model is built with legal sizes and num_slices=160.
        cfg = P.cfg_full(xr=(320, 320), mr1=(320, 320, 160), mr2=(320, 320, 160), dropout=0.1)
        cfg["_tensor_shapes"] = [[310, 310], [384, 384, 160], [384, 384, 160], [16]]
        return cfg, 2
    if name in ("native3", "eval3"):
        return P.cfg_xr1mr3c1(dropout=0.1), 8
    if name == "syn3":
        cfg = P.cfg_xr1mr3c1(xr=(320, 320), mr1=(320, 320, 160), mr2=(320, 320, 160), mr3=(320, 320, 160), dropout=0.1)
        cfg["_tensor_shapes"] = [[310, 310], [384, 384, 160], [384, 384, 160], [384, 384, 160], [16]]
        return cfg, 2
    if name == "xr1c1":
        return P.cfg_xr1c1(size=350, dropout=0.5), 32
    if name == "mr1c1":
        cfg = P.cfg_mr1c1(mr=(320, 320, 160), dropout=0.1)
        cfg["_tensor_shapes"] = [[384, 384, 160], [16]]
        return cfg, 4
    if name == "xr1cnn":
        return P.cfg_xr1cnn(size=350, dropout=0.5), 4
    if name == "mr1":
        return P.cfg_mr1(shape=(160, 160, 64), dropout=0.1), 4
    raise SystemExit(f"unknown workload {name}")


def algorithmic_train_gflop_per_sample(name):
    # measured with torch.utils.flop_counter on the imported reference
    # (ResNet-50 4.1705 GFLOP/slice @160^2, 24.02 @384^2; ResNeXt-50 20.877 @350^2, 16.84 @310^2; FeaT 0.2097/token)
    return {"native": 1280.0, "xr1cnn": 62.1, "mr1": 3 * (64 * 4.1705 + 65 * 0.2097),
            "native3": 1280.0 + 3 * (32 * 4.1705 + 64 * 0.2097),
            "eval3": (1280.0 + 11.3) / 3 + (32 * 4.1705 + 64 * 0.2097),
            "syn": 3 * (16.84 + 320 * 24.02 + 641 * 0.2097),
            "syn3": 3 * (16.84 + 480 * 24.02 + 963 * 0.2097),
            "xr1c1": 62.1 + 3 * 2 * (9 * 2048 + 2048 * 512) / 1e9,
            "mr1c1": 3 * (160 * 24.02 + 322 * 0.2097)}[name]


def cpu_baseline(cfg, workload):
    """The oracle (CPU port of the same train step) on this box's host cores; bounded sample."""
    import procedural as P
    from oracle import koafusion_cpu as O
    if workload in ("syn", "syn3", "mr1c1", "eval3"):
        return None     # minutes per sample on a host CPU (eval3: no train step to compare): outside the bounded sample
    B = 1
    n = min(len(os.sched_getaffinity(0)), 64)
    torch.set_num_threads(n)
    om = O.OracleModel(cfg, fill=None)
    g = torch.Generator().manual_seed(0)
    for k, v in om.sd.items():      # cheap random init (values do not matter for timing)
        if v.dtype.is_floating_point:
            with torch.no_grad():
                if k.endswith("running_var") or (v.dim() == 1 and k.endswith("weight")):
                    v.fill_(1.0)
                elif v.dim() >= 2:
                    v.copy_(torch.randn(v.shape, generator=g) * (1.0 / max(1, v[0].numel())) ** 0.5)
    xs = [torch.from_numpy(a) for a in P.model_inputs(cfg, B)]
    y = torch.from_numpy(P.make_target("target", B))
    om.train_step(xs, y)                     # warm-up
    steps = 2 if workload in ("native", "native3") else 3
    t0 = time.time()
    for _ in range(steps):
        om.train_step(xs, y)
    dt = time.time() - t0
    return {"value": round(B * steps / dt, 4), "unit": "knees/s", "cores": n, "kind": "port",
            "sample": f"{steps} full train steps (fwd+focal loss+bwd+Adam) of the same model/shapes at batch {B} "
                      f"on the host CPU ({n} torch threads), after 1 warm-up step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="native3")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = workload default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--recompute-mode", default="stage", choices=("block", "stage"),
                    help="granularity of activation recompute: one block at a time (least memory) or one stage at a time")
    ap.add_argument("--no-syn", action="store_true", help="skip the BASELINE-synthetic-shapes addendum of the default run")
    ap.add_argument("--breakdown", default="", help="write a per-shape table of the instrumented step to this file")
    ap.add_argument("--recompute", action="store_true",
                    help="activation recompute in the encoders (keeps stage inputs + BatchNorm statistics only; blocks rebuilt one at a time); "
                         "needed for --workload syn at batch 8")
    ap.add_argument("--serial", action="store_true",
                    help="one HIP stream only (no encoder lanes / wgrad side stream): kernel durations seen by a profiler "
                         "are then not inflated by co-running kernels -- the mode the roofline step always uses")
    args = ap.parse_args()

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
    if args.gpus != world:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import procedural as P
    from oaprogressionmmf_amd import ops
    from oaprogressionmmf_amd.config import ConfigDict
    from oaprogressionmmf_amd.models import dict_models
    from oaprogressionmmf_amd.parallel import DataParallelRCCL
    from oaprogressionmmf_amd.various import dict_losses, dict_optimizers, set_ultimate_seed

    set_ultimate_seed(777 + 16 * rank)   # distinct dropout streams per rank; rank 0's parameters are broadcast
    if args.serial:
        from oaprogressionmmf_amd.models import _common as _c, _encoder as _e
        _e.USE_SIDE_STREAM = False
        _c.USE_LANES = False
    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def make_job(name, batch, recompute=None):
        """model + optimizer + resident synthetic batch of one workload -> (cfg, B, step)"""
        cfg, bdef = workload_cfg(name)
        B = batch or bdef
        model = dict_models[cfg["name"]](config=ConfigDict(cfg), path_weights=None).to(dev)
        if args.recompute if recompute is None else recompute:
            from oaprogressionmmf_amd.models import KoafTrunk
            for m in model.modules():
                if isinstance(m, KoafTrunk):
                    m.recompute = args.recompute_mode if args.recompute_mode == "block" else True
        ddp = DataParallelRCCL(model, exchange_always=dist_on)
        loss_fn = dict_losses["FocalLoss"](reduction="mean", gamma=2.0, num_classes=2)
        opt = dict_optimizers["Adam"](model.parameters(), lr=1e-4, weight_decay=1e-4)
        shapes_cfg = dict(cfg, input_size=cfg.pop("_tensor_shapes")) if "_tensor_shapes" in cfg else cfg
        xs = [torch.from_numpy(a).to(dev) for a in P.model_inputs(shapes_cfg, B, seed=1234 + rank)]
        y = torch.from_numpy(P.make_target("target", B, seed=1234 + rank)).to(dev)
        model.train()
        if name == "eval3":
            from oaprogressionmmf_amd.run import predict_batch
            model.eval()

            def step_eval():
                logits, proba = predict_batch(model, xs)
                return float(proba[0, 0].item())       # the driver's per-batch host read (argmax / softmax go to the CPU)
            return cfg, B, step_eval

        def step():
            opt.zero_grad()
            logits = ddp(*xs)["main"]
            loss = loss_fn(input=logits.squeeze(1), target=y.long().squeeze(1))
            ddp.scale_loss(loss).backward()
            ddp.reduce_gradients()
            opt.step()
            # the reference logs loss.item() every step (:159-163): same value, read after the backward/optimizer
            # kernels are enqueued so the host sync does not drain the GPU between forward and backward
            return loss.item()
        return cfg, B, step

    def timed(step):
        """W untimed + exactly K timed steps between barriers; max over ranks"""
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            lv = step()
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, lv

    pinned = None
    if args.workload == "native3" and not args.batch and not args.recompute:
        # the reference-pinned 2-MRI model of the same configuration, timed first with the same K / W, then freed
        cfg2, B2, step2 = make_job("native", 0)
        dt2, lv2 = timed(step2)
        pinned = {"model": cfg2["name"], "value": round(world * B2 * args.steps / dt2, 3), "unit": "knees/s",
                  "ms_per_step": round(dt2 / args.steps * 1e3, 2), "per_gpu_batch": B2, "last_loss": round(lv2, 6),
                  "note": "runner.sh:341-363 (XR 350^2 + DESS 160x160x64 + T2 160x160x25 + clinical): the largest model "
                          "the reference registers; parity pinned by fixtures of the imported reference"}
        del step2
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    cfg, B, step = make_job(args.workload, args.batch)
    torch.cuda.reset_peak_memory_stats()
    dt, lv = timed(step)
    hbm_gb = (round(torch.cuda.max_memory_allocated() / 2**30, 1), round(torch.cuda.max_memory_reserved() / 2**30, 1))

    # one extra instrumented step: live event timing of every MFMA-GEMM launch on its launch stream
    from oaprogressionmmf_amd.models import _common, _encoder
    _encoder.USE_SIDE_STREAM = False      # serialise: per-kernel durations are not inflated by co-running kernels
    _common.USE_LANES = False
    torch.cuda.synchronize()
    torch.cuda.empty_cache()              # the serial step allocates from another stream pool: start it clean
    step()                                # (untimed) settle the allocator on the new pool
    torch.cuda.synchronize()
    ops.PROFILE = []
    step()
    torch.cuda.synchronize()
    prof_b = [(f, fl, e0.elapsed_time(e1), tag, nb) for f, fl, e0, e1, tag, nb in ops.PROFILE]
    prof = [p[:4] for p in prof_b]
    ops.PROFILE = None
    gemm_ms = sum(ms for f, fl, ms, tag in prof if f == "gemm")
    gemm_flop = sum(fl for f, fl, ms, tag in prof if f == "gemm")
    n_launch = sum(1 for f, *_ in prof if f == "gemm")
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    full_bwd = os.environ.get("KOAF_BWD_PRECISION", "")[:1] == "f"

    def call_peak(tag):
        bwd = any(k in tag for k in ("dgrad", "wgrad", "attn_bwd")) and not full_bwd
        return BF16_MFMA_PEAK_TFLOPS / (MFMA_PER_PRODUCT_BWD if bwd else MFMA_PER_PRODUCT_FWD)

    def call_floor_ms(fl, tag, nb):
        """roofline time of one call: the larger of its matrix-pipe time and the time to move each operand once"""
        return max(fl / (call_peak(tag) * 1e12), nb / (HBM_PEAK_TBPS * 1e12)) * 1e3
    floor_ms = sum(call_floor_ms(fl, tag, nb) for f, fl, ms, tag, nb in prof_b if f == "gemm")
    hbm_ms = sum(ms for f, fl, ms, tag, nb in prof_b
                 if f == "gemm" and nb / (HBM_PEAK_TBPS * 1e12) > fl / (call_peak(tag) * 1e12))
    if args.breakdown and rank == 0:
        agg = {}
        for f, fl, ms, tag, nb in prof_b:
            a = agg.setdefault(tag, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += ms; a[3] += nb; a[4] += call_floor_ms(fl, tag, nb)
        with open(args.breakdown, "w") as fh:
            fh.write(f"# per-shape GEMM-family calls of one train step ({args.workload}, batch {B}); ms from events on the launch stream;\n"
                     f"# MB = algorithmic HBM bytes (every operand once); floor = max(FLOP / matrix-pipe bound of the call, bytes / "
                     f"{HBM_PEAK_TBPS:g} TB/s); bound = which term of the floor is larger\n")
            fh.write(f"{'call':48s} {'n':>4s} {'ms':>9s} {'GFLOP':>10s} {'TFLOP/s':>8s} {'MB':>9s} {'TB/s':>6s} {'floor ms':>9s} {'bound':>5s}\n")
            for tag, (n, fl, ms, nb, flo) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                bound = "hbm" if nb / (HBM_PEAK_TBPS * 1e12) > fl / (call_peak(tag) * 1e12) else "mfma"
                fh.write(f"{tag:48s} {n:4d} {ms:9.3f} {fl / 1e9:10.1f} {fl / ms / 1e9 if ms > 0 else 0:8.1f} {nb / 1e6:9.1f} "
                         f"{nb / ms / 1e9 if ms > 0 else 0:6.2f} {flo:9.3f} {bound:>5s}\n")
    bound_s = sum(fl / (call_peak(tag) * 1e12) for f, fl, ms, tag in prof if f == "gemm")
    peak_mix = gemm_flop / bound_s / 1e12 if bound_s > 0 else BF16_MFMA_PEAK_TFLOPS / MFMA_PER_PRODUCT_FWD
    bwd_share = sum(fl for f, fl, ms, tag in prof if f == "gemm" and any(k in tag for k in ("dgrad", "wgrad", "attn_bwd")))
    bwd_share = bwd_share / gemm_flop if gemm_flop else 0.0
    # HBM-side bytes per launch of the dominant kernel cannot be counted from inside the process: they come from
    # the committed PMC summary of this same workload (profiles/README.md has the command and the corrections)
    traffic, traffic_src = None, None
    tj = ROOT / "profiles" / "r01_gemm_traffic.json"
    if args.workload == "native3" and B == 8 and tj.exists():
        try:
            tjd = json.loads(tj.read_text())
            traffic, traffic_src = tjd["bytes_per_launch"], "profiles/r01_gemm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
        except (ValueError, KeyError):
            pass

    if rank == 0:
        value = world * B * args.steps / dt
        out = {
            "metric": ("knees/sec full XR+MRI+clin fusion inference pass" if args.workload == "eval3"
                       else "knees/sec full XR+MRI+clin fusion train step"),
            "value": round(value, 3), "unit": "knees/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 tensors and accumulation everywhere; contractions form each fp32 x fp32 product from exact bf16 "
                          "pieces of the operands on the bf16 MFMA (forward: all 24 significand bits, error at fp32 rounding "
                          "level; gradient contractions: operands rounded to 16 significand bits, ~7e-6 relative); "
                          "KOAF_BWD_PRECISION=full makes the gradients exact too",
            "config": {"workload": f"{args.workload}: {cfg['name']} "
                                   + ("inference pass (forward + softmax), " if args.workload == "eval3"
                                      else "train step (fwd+FocalLoss+bwd+Adam), ") +
                                   f"per-GPU batch {B}, global batch {world * B}, "
                                   + {"native": "XR 1x350x350 + DESS 160x160x64 + T2 160x160x25 + 9 clinical; random-init weights",
                                      "native3": "XR 1x350x350 + DESS 160x160x64 + TSE 160x160x32 + T2 160x160x25 + 9 clinical "
                                                 "(BASELINE config 4; 3-MRI registry extension of the reference's "
                                                 "XR1MR2C1CnnTrf); random-init weights",
                                      "syn3": "BASELINE synthetic shapes XR 1x310x310 + 3 x MRI 1x160x384x384 + 9 clinical; "
                                              "random-init weights",
                                      "eval3": "INFERENCE pass (forward + softmax, eval mode) on the native3 shapes; random-init weights",
                                      "xr1c1": "BASELINE config 2: XR 1x350x350 + 9 clinical, early-fusion MLP head; random-init weights",
                                      "mr1c1": "BASELINE config 3: SAG-3D-DESS 1x160x384x384 + 9 clinical; random-init weights",
                                      "syn": "BASELINE synthetic shapes XR 1x310x310 + 2 x MRI 1x160x384x384 + 9 clinical "
                                             "(default per-GPU batch 2; batch 8 with --recompute); "
                                             "random-init weights"}.get(args.workload, "random-init weights"),
                       "parallelism": f"dp{world}", "last_loss": round(lv, 6), "activation_recompute": bool(args.recompute),
                       "hbm_peak_gib": {"allocated": hbm_gb[0], "reserved": hbm_gb[1]}},
            **({"pinned_reference_model": pinned} if pinned else {}),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak_mix, 1),
                         "unit": "TFLOP/s", "frac": round(achieved / peak_mix, 4), "traffic": traffic,
                         "traffic_unit": "bytes per koaf_gemm_kernel launch (average)", "traffic_source": traffic_src,
                         "kernel": "koaf_gemm_kernel (implicit GEMM: conv fwd/dgrad/wgrad, linear, attention; fp32 in/out/"
                                   "accumulate; products on v_mfma_f32_32x32x16_bf16 from bf16 pieces of the operands: "
                                   "forward 3 pieces / 6 products (every significand bit), gradients 2 pieces / 4 products "
                                   "(16 significand bits))",
                         "peak_is": "2.5 PFLOP/s dense bf16 MFMA / MFMAs per product, FLOP-weighted harmonic mean over the "
                                    f"step's calls ({100 * bwd_share:.0f} % of the FLOPs are gradient contractions at 625, "
                                    "the rest at 416.7 TFLOP/s fp32-equivalent)",
                         "fp32_mfma_peak": MFMA_F32_PEAK_TFLOPS,
                         "achieved_over_fp32_mfma_peak": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                         "floor_ms_per_step": round(floor_ms, 2), "frac_of_floor": round(floor_ms / gemm_ms, 4) if gemm_ms > 0 else None,
                         "hbm_bound_share_of_kernel_ms": round(hbm_ms / gemm_ms, 4) if gemm_ms > 0 else None,
                         "floor_is": "sum over the step's calls of max(FLOP / matrix-pipe bound of the call, algorithmic bytes / 8 TB/s): "
                                     "SURVEY 8(d)'s attainable = min(MFMA peak, AI x HBM) applied per call; frac_of_floor = floor / "
                                     "measured kernel time; `frac` above stays the plain achieved / matrix-pipe peak",
                         "launches_per_step": n_launch, "kernel_ms_per_step": round(gemm_ms, 2),
                         "algorithmic_gflop_per_step": round(gemm_flop / 1e9, 1),
                         "step_gflop_per_sample_survey": algorithmic_train_gflop_per_sample(args.workload)},
        }
        if world == 1 and args.workload == "native3" and not args.batch and not args.recompute and not args.no_syn:
            # the same model and batch on BASELINE.json's synthetic tensor shapes (XR 1x310x310, MRI 1x160x384x384): the
            # activations of a batch of 8 need activation recompute to fit (142 GB); 1 warm-up + 2 timed steps
            try:
                del step
                import gc
                gc.collect()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
                _common.USE_LANES, _encoder.USE_SIDE_STREAM = not args.serial, not args.serial
                cfg_s, B_s, step_s = make_job("syn3", 8, recompute=True)
                torch.cuda.reset_peak_memory_stats()
                step_s()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    lv_s = step_s()
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t0) / 2
                out["baseline_synthetic_shapes"] = {
                    "workload": "syn3: same model, per-GPU batch 8, XR 1x310x310 + 3 x MRI 1x160x384x384 + 9 clinical, "
                                "activation recompute (one encoder stage at a time)",
                    "value": round(B_s / dts, 3), "unit": "knees/s", "ms_per_step": round(dts * 1e3, 1), "steps": 2, "warmup": 1,
                    "last_loss": round(lv_s, 6),
                    "step_gflop_per_sample_survey": round(algorithmic_train_gflop_per_sample("syn3"), 1),
                    "hbm_peak_gib": {"allocated": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                                     "reserved": round(torch.cuda.max_memory_reserved() / 2**30, 1)}}
                del step_s
                gc.collect()
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001  (the headline line must still be printed)
                out["baseline_synthetic_shapes"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(cfg, args.workload)
            if cb is not None:
                out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
