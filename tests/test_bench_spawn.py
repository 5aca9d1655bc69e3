"""bench.py --gpus N without a launcher: the file starts its own N rank processes before
anything touches a GPU, relays rank 0's JSON line and enforces a deadline.  Covered here on
CPU through bench.py's own main() with --dry-run (ranks rendezvous over gloo, empty step):
the spawn, the rendezvous on 127.0.0.1, barrier + max-over-ranks timing, the relayed line,
the exit code, and the kill of a run that hangs (the renderer the multi-GPU path replaces
needs no launcher either: renderer.rs:63-89)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH] + args, capture_output=True, env=e, timeout=timeout)
    return p, time.time() - t0


def test_self_spawned_ranks_rendezvous_and_rank0_line_is_relayed():
    p, _ = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1                                       # exactly one JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["steps"] == 3
    assert d["config"]["patch_rows_owned_by_all_ranks"] == d["config"]["patch_rows"] == 33   # cyclic rows cover the frame once
    assert d["config"]["rows_per_rank"] == 17


def test_three_ranks():
    p, _ = _run(["--gpus", "3", "--dry-run", "--steps", "2", "--warmup", "0", "--config", "C4"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert d["n_gpus"] == 3 and d["config"]["patch_rows_owned_by_all_ranks"] == 135


def test_a_hung_rank_is_killed_at_the_deadline_and_reported():
    p, took = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0", "--dry-hang-rank", "1", "--deadline", "20"])
    assert p.returncode != 0
    assert took < 90, "the launcher waited %.0f s for a hung rank" % took
    d = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert d["value"] is None and "deadline" in d["error"] and d["n_gpus"] == 2
    assert b"deadline" in p.stderr


def test_gpu_count_comes_from_sysfs_not_from_the_runtime(tmp_path):
    """The launcher counts GPUs from the KFD topology (nodes with SIMDs), honours the
    *_VISIBLE_DEVICES lists, and never imports torch to do so."""
    code = ("import sys; sys.path.insert(0, %r); import bench; n = bench.visible_gpus(); "
            "assert 'torch' not in sys.modules; print(-1 if n is None else n)" % ROOT)
    e = dict(os.environ)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        e.pop(k, None)
    n = int(subprocess.check_output([sys.executable, "-c", code], env=e).decode())
    if os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        assert n >= 0
        import torch
        if torch.cuda.is_available():
            assert n == torch.cuda.device_count()
        m = int(subprocess.check_output([sys.executable, "-c", code], env=dict(e, HIP_VISIBLE_DEVICES="0")).decode())
        assert m == min(n, 1)
    else:
        assert n == -1


def test_without_the_gpus_it_fails_loudly_instead_of_benchmarking_one():
    sys.path.insert(0, ROOT)
    import bench
    have = bench.visible_gpus()
    if have is None:
        import torch
        have = torch.cuda.device_count()
    if have >= 2:
        import pytest
        pytest.skip("two GPUs are present")
    p, _ = _run(["--gpus", "2", "--steps", "2"])
    assert p.returncode == 2 and p.stdout.strip() == b""
    assert b"needs 2 GPUs" in p.stderr


def test_launcher_rank_count_must_match_gpus():
    p, _ = _run(["--gpus", "2", "--steps", "2"], env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and b"--gpus 2 but WORLD_SIZE=3" in p.stderr
