"""bench.py's CPU baseline for the headline workload is an EXTRAPOLATION: one knee = t(16) + 144 x (t(32) - t(16)) / 16 from two
partial train steps of the oracle (32 and 16 of the 160 slices per MRI).  This script validates the line once: the same oracle,
the same thread count, ONE train step at the largest slice count whose saved activations fit the box's host memory (160 when
~120 GiB are free), beside the two partial steps and the prediction of the line for that point.  (gpurun -- python scripts/validate_cpu_baseline.py > gpurun_out/.../cpu_baseline_validation.log; host CPU only.)"""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import procedural as P
import bench

n, why = bench.host_threads()
torch.set_num_threads(n)


def one(S):
    om, xs, y = bench._oracle_job(lambda: P.cfg_xr1mr3c1(xr=(320, 320), mr1=(320, 320, S), mr2=(320, 320, S), mr3=(320, 320, S), dropout=0.1),
                                  1, [[310, 310], [384, 384, S], [384, 384, S], [384, 384, S], [16]])
    t0 = time.time()
    om.train_step(xs, y)
    return time.time() - t0


def host_memory_gib():
    """what this process may use: MemAvailable, cut to the cgroup's limit when there is one"""
    avail = None
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            avail = int(line.split()[1]) / 2 ** 20
    for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            txt = Path(path).read_text().strip()
            if txt != "max":
                lim = int(txt) / 2 ** 30
                cur = 0.0
                for cp in ("/sys/fs/cgroup/memory.current", "/sys/fs/cgroup/memory/memory.usage_in_bytes"):
                    try:
                        cur = int(Path(cp).read_text()) / 2 ** 30
                        break
                    except (OSError, ValueError):
                        pass
                avail = min(avail, lim - cur) if avail is not None else lim - cur
            break
        except (OSError, ValueError):
            continue
    return avail


# a train step at S slices per MRI saves ~0.75 GiB per slice index (3 MRI x 241 MiB at 384^2): the largest S whose
# activations take at most 40 % of the memory this process may use (a full 160-slice step needs ~120 GiB of host memory)
mem = host_memory_gib()
S3 = 160 if mem is None else max(48, min(160, int(0.4 * mem / 0.75) // 16 * 16))
print(f"host memory available to this process: {mem if mem is None else round(mem, 1)} GiB -> validation point {S3} slices per MRI", flush=True)
one(2)
ts = {}
for S in (32, 16, S3):
    ts[S] = one(S)
    print(f"slices per MRI {S:4d}: one train step {ts[S]:7.1f} s", flush=True)
slope = (ts[32] - ts[16]) / 16
pred = ts[16] + (S3 - 16) * slope
ext = ts[16] + 144 * slope
print(json.dumps({"threads": n, "host": why, "host_memory_gib": None if mem is None else round(mem, 1),
                  "measured_s": {str(k): round(v, 2) for k, v in ts.items()},
                  "validation_point_slices": S3, "predicted_from_32_and_16_s": round(pred, 2), "measured_s_at_validation_point": round(ts[S3], 2),
                  "prediction_over_measured": round(pred / ts[S3], 3), "extrapolated_s_per_knee_160": round(ext, 1),
                  "note": "bench.py extrapolates t(160) = t(16) + 144 (t(32) - t(16)) / 16; this run checks the same line at the largest slice "
                          "count whose saved activations fit the host memory of the box"}))
