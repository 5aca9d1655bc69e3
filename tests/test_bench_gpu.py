"""bench.py end to end on the GPU box, through its own main(): the one-GPU line, the N > 1
code path started WITHOUT a launcher (two ranks sharing this box's GPU over gloo -- the
transport differs from the 8-GPU run, the sharding, the per-frame collective, the
max-over-ranks timing and the JSON do not), and the library's own exchange through a real
RCCL communicator of one rank (--force-dist), u8 and f64 payloads, checked against the oracle."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH] + args, capture_output=True, env=e, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_one_gpu_line_has_the_contract_fields():
    d = _bench(["--steps", "20", "--warmup", "3", "--check", "--no-sizes"])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["unit"] == "Mpixels/s" and d["dtype"] == "f64"
    assert d["metric"] == "Mpixels/sec at 1920x1080, max-bounce=5" and d["vs_baseline"] is None
    assert d["max_abs_delta_vs_oracle"] < 1e-9 and d["display_bytes_differing_from_oracle"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"].startswith("rmdev_strict::rm_render_static<") and 0.02 < r["kernel_ms"] < 1.0
    assert abs(r["bytes_per_launch"] - 1920 * 1056 * 24) < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["cpu"] and c["value"] > 0 and "frame(s)" in c["sample"]
    assert d["value"] > 100 * c["value"]                      # north star: >= 100x the host-CPU path


def test_two_ranks_self_spawned_on_this_gpu_over_gloo():
    d = _bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--check", "--no-sizes"], env={"RM_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["display_bytes_differing_from_oracle"] == 0
    assert "cyclic patch rows" in d["config"]["sharding"] and "gather at rank 0" in d["config"]["sharding"]
    assert "torch" in d["exchange_paths"]
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--check", "--no-sizes", "--exchange", "allgather"],
               env={"RM_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["display_bytes_differing_from_oracle"] == 0 and "all-gather" in d["config"]["sharding"]
    d = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--check", "--no-sizes", "--payload", "f64"],
               env={"RM_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["max_abs_delta_vs_oracle"] < 1e-9


@pytest.mark.parametrize("payload", ["u8", "f64"])
def test_library_exchange_through_rccl_world_of_one(payload):
    d = _bench(["--force-dist", "--steps", "20", "--warmup", "3", "--check", "--no-sizes", "--no-cpu-baseline", "--payload", payload])
    paths = d["exchange_paths"]
    assert set(paths) == {"torch", "direct"}, d["config"]["collective"]
    assert paths["direct"]["rccl_sees"] == {"rank": 0, "world": 1, "communicators": 1}
    assert paths["direct"]["frames_in_flight"] == 4 and paths["torch"]["frames_in_flight"] == 2
    if payload == "u8":
        assert d["display_bytes_differing_from_oracle"] == 0
    else:
        assert d["max_abs_delta_vs_oracle"] < 1e-9


# ---------------------------------------------------------------- two GPUs and more (self-arming)
def _gpus():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.visible_gpus()
    if n is None:
        import torch
        n = torch.cuda.device_count()
    return n


needs_two = pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs: one rank per GPU over real RCCL (runs by itself on a bigger box)")


@needs_two
@pytest.mark.parametrize("payload", ["u8", "f64"])
@pytest.mark.parametrize("exchange", ["gather", "allgather"])
def test_two_gpus_over_rccl(payload, exchange):
    """The replacement of the Rayon workers (renderer.rs:63-89) with one rank per GPU: cyclic patch
    rows, ONE exchange per frame over RCCL / xGMI -- torch.distributed's and the library's own
    (rm_frame_submit*: four frames in flight), both checked against the oracle and against each
    other on a check frame, every wait bounded, the whole run under bench.py's deadline."""
    d = _bench(["--gpus", "2", "--steps", "10", "--warmup", "2", "--check", "--no-sizes", "--payload", payload,
                "--exchange", exchange, "--deadline", "420"])
    assert d["n_gpus"] == 2 and d["value"] and d["value"] > 0
    if payload == "u8":
        assert d["display_bytes_differing_from_oracle"] == 0
    else:
        assert d["max_abs_delta_vs_oracle"] < 1e-9
    paths = d["exchange_paths"]
    assert "torch" in paths
    assert "direct" in paths, d["config"]["collective"]                # the library's exchange completed and agreed
    assert paths["direct"]["rccl_sees"]["world"] == 2 and paths["direct"]["frames_in_flight"] == 4


@needs_two
def test_rm_walk_two_ranks_match_the_oracle(tmp_path):
    """rm_walk --world 2: two processes, one GPU each, the unique id through a file, rank 0 writes
    every frame's display bytes -- the oracle's to_vec() of the same camera position."""
    import numpy as np

    import __graft_entry__ as G
    import workloads
    O = G.load_oracle()
    exe = os.path.join(G.PKG_DIR, "lib", "rm_walk")
    w, h, depth, frames = 640, 360, 4, 6
    common = ["--world", "2", "--frames", str(frames), "--width", str(w), "--height", str(h), "--depth", str(depth),
              "--step", "0.5,0.25,-1", "--id-file", str(tmp_path / "id.bin"), "--run-id", "test-%d" % os.getpid()]
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", RM_FRAME_TIMEOUT_MS="60000")
    procs = [subprocess.Popen([exe, "--rank", str(r), "--device", str(r)] + common + (["--out", str(tmp_path / "walk")] if r == 0 else []),
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in (0, 1)]
    try:
        outs = [p.communicate(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert [p.returncode for p in procs] == [0, 0], [o[1].decode()[-1500:] for o in outs]
    rows = h // 32 * 32
    so = workloads.oracle_scene(O, "demo")
    for k in range(frames):
        data = (tmp_path / ("walk_%04d.ppm" % k)).read_bytes()
        header = b"P6\n%d %d\n255\n" % (w, rows)
        assert data.startswith(header)
        so.set_camera((0.5 * k, 0.25 * k, -1. * k))
        ref = O.to_vec(O.render(so, w, h, max_depth=depth)[:rows].copy())
        got = np.frombuffer(data[len(header):], np.uint8)
        assert int((got != ref).sum()) <= 2, "frame %d" % k
