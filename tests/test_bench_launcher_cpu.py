"""bench.py --gpus N without a launcher: the parent starts N worker processes with the rendezvous environment of
`torch.distributed.run`, relays rank 0's line and fails if any worker fails (no GPU involved: stub workers)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

STUB = """
import json, os, sys
r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
if "--fail" in sys.argv and r == w - 1:
    sys.exit(3)
print(json.dumps({"rank": r, "world": w, "argv": sys.argv[1:], "port": os.environ["MASTER_PORT"]}))
"""


def test_launcher_starts_ranks_and_relays_rank0(tmp_path, capfd):
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    rc = bench.launch_workers(2, ["--gpus", "2", "--steps", "3"], program=str(stub))
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1                      # only rank 0's line is relayed
    d = json.loads(out[0])
    assert d["rank"] == 0 and d["world"] == 2 and d["argv"] == ["--gpus", "2", "--steps", "3"]
    rc = bench.launch_workers(2, ["--fail"], program=str(stub))
    assert rc == 1
    capfd.readouterr()


HANG = """
import os, sys, time
r = int(os.environ["RANK"])
if r == 1:
    time.sleep(0.5)
    sys.exit(7)          # dies after the "rendezvous"
time.sleep(600)          # the survivors would sit in a collective for ever
"""


def test_launcher_tears_down_survivors_of_a_dead_rank(tmp_path, capfd):
    import time
    import bench
    stub = tmp_path / "hang.py"
    stub.write_text(HANG)
    t0 = time.time()
    rc = bench.launch_workers(3, [], program=str(stub))
    assert rc == 1 and time.time() - t0 < 30
    assert "worker(s) failed" in capfd.readouterr().err
    t0 = time.time()
    rc = bench.launch_workers(1, [], program=str(stub), timeout_s=1.0)       # rank 0 alone only sleeps: the overall limit ends it
    assert rc == 1 and time.time() - t0 < 30
    assert "terminated" in capfd.readouterr().err


OOM = """
import json, os, sys
r = int(os.environ["RANK"])
if "--recompute" not in sys.argv:
    sys.exit(42 if r == 1 else 0)                  # rank 1 runs out of memory on the default policy
assert os.environ.get("KOAF_BENCH_OOM_RETRY") == "1"
if r == 0:
    print(json.dumps({"argv": sys.argv[1:]}))
"""


def test_launcher_retries_once_with_the_lean_policy_after_an_oom(tmp_path, capfd):
    import bench
    stub = tmp_path / "oom.py"
    stub.write_text(OOM)
    args = bench.parse_args(["--gpus", "2"])
    rc = bench.launch_with_fallback(args, ["--gpus", "2"], program=str(stub))
    cap = capfd.readouterr()
    assert rc == 0 and f"one retry with --recompute {bench.FALLBACK_POLICY}" in cap.err
    assert json.loads(cap.out.strip().splitlines()[-1])["argv"] == ["--gpus", "2", "--recompute", bench.FALLBACK_POLICY]
    assert "KOAF_BENCH_OOM_RETRY" not in __import__("os").environ
    args = bench.parse_args(["--gpus", "2", "--recompute", "none"])            # an explicit policy is never overridden
    assert bench.launch_with_fallback(args, ["--gpus", "2"], program=str(stub)) == 1
    capfd.readouterr()


def test_worker_env_and_defaults():
    import bench
    env = bench.worker_env(3, 8, 12345, base={"X": "1"})
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["MASTER_PORT"] == "12345"
    assert env["X"] == "1"
    a = bench.parse_args([])
    assert a.gpus == 1 and a.workload == "syn3" and a.steps >= 20 and a.warmup >= 5
    cfg, B, pol = bench.workload_cfg("syn3")
    assert B == 8 and cfg["_tensor_shapes"][1] == [160, 384, 384] and cfg["fe"]["mr"]["volume_layout"] == "ncdhw"


OOM1 = """
import json, os, sys
assert os.environ["WORLD_SIZE"] == "1" and os.environ["RANK"] == "0"
if "--recompute" not in sys.argv:
    sys.exit(42)                                   # the single rank runs out of memory on the default policy
assert os.environ.get("KOAF_BENCH_OOM_RETRY") == "1"
print(json.dumps({"argv": sys.argv[1:]}))
"""


def test_one_gpu_runs_under_the_gpu_free_supervisor_too(tmp_path, capfd):
    """N = 1 takes the same path as N > 1: the measuring process is a fresh child, and the out-of-memory retry starts only
    after that child has exited (nothing of the failed run stays on the device)"""
    import bench
    stub = tmp_path / "oom1.py"
    stub.write_text(OOM1)
    args = bench.parse_args([])
    assert args.gpus == 1 and not args.inproc
    rc = bench.launch_with_fallback(args, [], program=str(stub))
    cap = capfd.readouterr()
    assert rc == 0 and json.loads(cap.out.strip().splitlines()[-1])["argv"] == ["--recompute", bench.FALLBACK_POLICY]
    src = (ROOT / "bench.py").read_text()
    assert 'if "WORLD_SIZE" not in os.environ and not _args.inproc:' in src
