#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the render hot path on N MI355X (one process per GPU).

    python bench.py                                   one GPU
    python bench.py --gpus N --steps K --warmup W     N GPUs: starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      (the same under a launcher)

A step is one frame of the workload: BASELINE.json's configs[1] (demo scene, 1920x1080,
depth cap 5) unless --config says otherwise.  Each launch writes the f64 RGB frame (the
reference's FrameBuffer) and its display bytes (`to_vec`, what the UI blits) to HBM; inputs
(the scene) are uploaded before the timed region.

N > 1: the frame is sharded by 32-row patch rows (cyclic ownership: rank r
renders patch rows r, r+N, ...) and ONE exchange per frame, inside the timed region, completes
the display frame (--payload u8, the default) or the f64 frame itself (--payload f64) at the
consumer, rank 0: every peer sends its rows straight to rank 0, all at once, each over its own
xGMI link (--exchange gather, the default; --exchange allgather: one in-place all-gather, the
frame on every rank).  Two implementations are measured, in this order, K steps each:
  1. torch.distributed per frame (batch_isend_irecv / all_gather_into_tensor, two frames in
     flight) -- its numbers are kept whatever happens next;
  2. the library's own exchange (rm_frame_submit / rm_frame_submit_f64: four frames in
     flight, each on a stream of its own, RCCL called from C), first checked byte for
     byte against (1), every wait bounded (RM_ERR_TIMEOUT, never a hang).
`value` is the faster of the two when (2) completed and agreed, else (1)'s; the line says
which and carries both.  Total work is fixed: "strong" scaling.

Started with --gpus N > 1 and no launcher (WORLD_SIZE unset), this file starts the N rank
processes itself -- before anything touches a GPU -- relays rank 0's JSON line and enforces a
deadline: a run that hangs is killed (by process group) and reported, never waited for.

Rank 0 prints ONE JSON line.  Metric definition follows the reference
(renderer.rs:113-120): frame_width x frame_height pixels per frame / wall time.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_PIXEL = 24            # 3 x f64 written per pixel (framebuffer.rs Vec3f), SURVEY.md 8d
FEEDBACK_NOTE = ("frame-to-frame feedback: the tiles that took longest in the previous frame on the stream are dispatched "
                 "first (RM_FEEDBACK=0 switches it off); every tile of every frame is rendered in full")
CLASSIFY_NOTE = ("the first workgroups of the render launch itself (scenes of up to 56 primitives; a launch of its own in front of it for "
                 "larger scenes and for launches that carry the tile-level feedback) test the cone of every tile's primary rays against the "
                 "primitives' bounds, 16 lanes per 32x32 patch (RM_TILE_CLASSIFY=0 switches it off): tiles nothing can be hit in get a wave "
                 "that stores the primary-miss value and leaves; every other pixel is traced in full; kernel_ms and ms_per_step include it")
ORDER_NOTE = ("dispatch order from the launch's own classification: its classifying workgroups put every 32x32 patch behind the first round "
              "into one of sixteen buckets -- by the patch's longest tile in the previous frame while the view stands still (such a launch "
              "dispatches by the order its predecessor laid out and lays out its successor's: nothing waits), by what tiles reaching the same "
              "primitives cost in the previous frame once the camera has moved (the waves behind the first round then wait for the order, "
              "10-20 us into the launch) -- the sky last; the first round renders the first places of the previous order "
              "(RM_PATCH_ORDER=0 switches it off); sky tail: the places of the order nothing can be hit in get one wave instead of sixteen, "
              "their number taken from a hint the earlier launches left in page-locked memory -- exact for a standing view, a guess for a "
              "moved one, whose wrong places are rendered by sixteen waves each at the grid's end (RM_SKY_TAIL=0 switches it off); only the "
              "order and the geometry of a launch depend on any of it: every frame whose view, scene or size differs from its predecessor's "
              "classifies every patch (of eight launches of a standing view one does, the others take their predecessor's words and order: "
              "RM_ORDER_FREEZE=0 makes every launch classify) and every pixel with something to hit is traced in full, exactly once")
MOTION_NOTE = ("the reference renders only after a camera move or a scene change (main.rs:74-78, :119-170): rm_camera_update before "
               "EVERY launch, the camera one button press (+-5 on one axis) from where it was (workloads.camera_walk), the scene "
               "resident; same stream, same outputs, the same kind of HIP-event bracket as the metric's, around whole laps of the closed "
               "walk (at least K steps: `steps` says how many -- K steps from wherever the warm-up stops are K views of a walk whose "
               "views take 40 to 60 us)")
WARMUP_SECONDS = float(os.environ.get("RM_BENCH_WARMUP_S", "0.3"))   # launches before the timed region, on top of --warmup (clocks settle; the profiling passes shorten it)
WARMUP_PROBE = 8                # launches timed to find out how many that is


def kernel_template_args(name):
    """rm_render_static<STACK, POW, WAVES, TPW, STAGED, BVH, CULL, EDGES, ORDER, FEEDBACK, HANDON> -> the arguments as strings"""
    if "<" not in name or ">" not in name:
        return []
    return [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", help="workload id from workloads.CONFIGS")
    ap.add_argument("--payload", choices=["u8", "f64"], default="u8",
                    help="what the per-frame all-gather moves at N > 1: the display bytes (to_vec, 3 B/px) "
                         "or the f64 rows themselves (24 B/px)")
    ap.add_argument("--collective", choices=["direct", "torch"], default="direct",
                    help="N > 1: 'direct' = measure the torch all-gather path, then the library's own frame exchange "
                         "(rm_frame_submit*) and report the latter when it completes and agrees; 'torch' = the torch path only")
    ap.add_argument("--exchange", choices=["gather", "allgather"], default="gather",
                    help="N > 1: 'gather' = every peer sends its chunk straight to rank 0, the consumer (grouped send / recv: "
                         "each over its own xGMI link); 'allgather' = one in-place all-gather, the frame on every rank")
    ap.add_argument("--fast-fp", action="store_true",
                    help="RM_FLAG_FAST_FP flavour of the kernel (FMA, Newton rsqrt): faster, but may decide "
                         "exact-incidence pixels differently from the reference -- not the parity configuration")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the per-frame collective even at world size 1 (smoke test of the N > 1 code path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sizes", action="store_true",
                    help="skip the extras reported beside the metric (other configs, rm_render into host memory)")
    ap.add_argument("--no-motion", action="store_true", help="skip the camera_on_the_move leg (the camera moving before every launch)")
    ap.add_argument("--check", action="store_true", help="also compare the frame with the oracle")
    ap.add_argument("--deadline", type=float, default=900.,
                    help="self-spawned runs (--gpus N > 1 without a launcher): seconds after which the rank processes are killed")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks rendezvous over gloo and go through the step / barrier / max-over-ranks / JSON "
                         "plumbing with an empty step; prints a line marked dry_run with value null (CPU test of the spawn path)")
    ap.add_argument("--dry-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)   # test hook: this rank never finishes
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------
# self-spawn: N rank processes, started before this process has touched a GPU
# ------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _kill_group(proc):
    """Ends exactly the process group this launcher started (never a pattern)."""
    import signal
    for sig in (signal.SIGTERM, signal.SIGKILL):
        if proc.poll() is not None:
            return
        try:
            os.killpg(proc.pid, sig)
        except ProcessLookupError:
            return
        try:
            proc.wait(timeout=5)
        except subprocess.TimeoutExpired:
            pass


def visible_gpus():
    """GPUs this process could open, counted WITHOUT touching the HIP runtime: the KFD topology
    nodes that have SIMDs (CPU nodes have none; a node this container may not use does not read:
    the driver checks the device cgroup), less what HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    leave.  None when the topology is not there to read."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    n = 0
    for path in nodes:
        try:
            for ln in open(path):
                if ln.startswith("simd_count"):
                    n += 1 if int(ln.split()[1]) > 0 else 0
        except (OSError, ValueError, IndexError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start N copies of this file (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment), relay rank 0's JSON line, return the exit
    code.  Nothing here touches the HIP runtime: the GPUs are counted from the KFD topology in
    sysfs (torch.cuda.device_count() only where that is not readable)."""
    n = args.gpus
    if not args.dry_run and os.environ.get("RM_BENCH_BACKEND", "nccl") == "nccl":
        have = visible_gpus()
        if have is None:
            import torch
            have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py --gpus %d needs %d GPUs; this node shows %d\n" % (n, n, have))
            return 2
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    partial = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rm_bench_partial_%d.json" % os.getpid())
    env["RM_BENCH_PARTIAL"] = partial
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().decode().splitlines()), daemon=True)
    reader.start()
    t_end = time.time() + args.deadline
    why = None
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed and why is None:
            why = "rank %d exited with code %d" % (procs.index(failed[0]), failed[0].returncode)
            t_end = min(t_end, time.time() + 30.)          # the others get half a minute to notice
        if time.time() > t_end:
            why = why or "deadline of %.0f s reached" % args.deadline
            for p in procs:
                _kill_group(p)
            break
        time.sleep(0.05)
    reader.join(timeout=5)
    codes = [p.returncode for p in procs]
    line = next((ln for ln in reversed(lines) if ln.startswith("{")), None)
    rc = next((c for c in codes if c), 0)
    if line is None:
        # no number: say why, with whatever rank 0 had measured before it stopped
        part = None
        try:
            part = json.load(open(partial))
        except Exception:
            pass
        line = json.dumps({"metric": "Mpixels/sec at 1920x1080, max-bounce=5", "value": None, "unit": "Mpixels/s",
                           "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "error": why or "no output from rank 0",
                           "rank_exit_codes": codes, "partial": part})
        rc = rc or 5
    try:
        os.remove(partial)
    except OSError:
        pass
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    if why:
        sys.stderr.write("bench.py: %s (rank exit codes %s)\n" % (why, codes))
    return rc


# ------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def csrc_hash():
    """sha256 over the kernel sources (comments and white space apart): ties
    profiles/pmc_latest.json to a build."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rusty-marcher_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc", ".hpp", ".h", ".cpp")):
            text = open(os.path.join(d, name), errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            h.update(name.encode())
            h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def pmc_for(cfg_id, kernel_name, kernel_ms):
    """rocprofv3 PMC figures of a config's render launch (profiles/pmc_<cfg>.json, written by
    profiles/summarise_round.py from passes of their own), quoted only while they belong to the build
    that runs: same hash of the kernel sources, same kernel name.  -> (traffic, fp64_valu, note)"""
    path = os.path.join(ROOT, "profiles", "pmc_%s.json" % cfg_id)
    if not os.path.exists(path):
        return None, None, "no profiles/pmc_%s.json" % cfg_id
    try:
        pmc = json.load(open(path))
        if pmc.get("csrc_sha16") != csrc_hash() or pmc.get("kernel") != kernel_name:
            return None, None, ("profiles/pmc_%s.json was collected on another build of the kernel sources "
                                "(csrc hash / kernel name differ): traffic not reported" % cfg_id)
        valu = {k: pmc.get(k) for k in ("valu_busy_frac", "valu_lanes_active_frac", "valu_wave_instructions",
                                        "salu_wave_instructions", "wait_any_frac", "wait_inst_any_frac")}
        if valu["valu_wave_instructions"] and kernel_ms:
            # the bound that does apply: one 64-lane FP64 VALU instruction per SIMD per 4 cycles
            # (1024 SIMDs, 2.4 GHz peak clock) against this run's kernel time
            valu["valu_issue_frac_of_peak"] = valu["valu_wave_instructions"] * 4.0 / (1024 * 2.4e9 * kernel_ms * 1e-3)
        return pmc.get("hbm_bytes_per_launch"), valu, (
            "traffic and fp64_valu are rocprofv3 PMC figures of this command on this build (profiles/pmc_%s.json: "
            "csrc hash and kernel name match), not measured in this run" % cfg_id)
    except Exception as e:                              # noqa: BLE001
        return None, None, "profiles/pmc_%s.json unreadable: %s" % (cfg_id, e)


def cpu_baseline(O, workloads, cfg, budget_s=25.0):
    """The oracle driven like renderer.rs:63-108 (dynamic 32x32 patches over all host
    threads, per-patch buffers, serial scatter), timed on a bounded sample: 3 warm-up frames,
    then whole frames of the same workload until 20 frames or ~budget_s of wall time."""
    import numpy as np
    scene = workloads.oracle_scene(O, cfg["scene"])
    w, h, depth = cfg["width"], cfg["height"], cfg["max_depth"]
    cores = O.lib().orc_online_cpus()
    # large frames: render a band sample instead of whole frames
    n_rows = h // 32
    rows = n_rows
    est_px_per_s = 1.5e6 * cores if cfg["scene"] == "demo" else 2.0e5 * cores
    while rows > 1 and 23 * (rows * 32 * w) / est_px_per_s > budget_s:
        rows //= 2
    band = ((n_rows - rows) // 2, (n_rows - rows) // 2 + rows)
    frame = np.zeros((h, w, 3), dtype=np.float64)
    t_all = time.perf_counter()
    for _ in range(3):                                                  # warm-up
        O.render(scene, w, h, max_depth=depth, frame=frame, band=band)
    times = []
    while len(times) < 20 and (len(times) < 5 or (time.perf_counter() - t_all) < budget_s):
        t0 = time.perf_counter()
        O.render(scene, w, h, max_depth=depth, frame=frame, band=band)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    # reference metric: full-width rows; scale the band's pixels by the frame convention
    px = w * h if rows == n_rows else rows * 32 * w
    ms = int(med * 1000)
    return {
        "value": px / med / 1e6, "unit": "Mpixels/s", "cores": cores, "cpu": cpu_model(), "kind": "port",
        "sample": "%d frame(s) of patch rows [%d,%d) of %dx%d %s depth %d after 3 warm-up frames, median; "
                  "oracle (-O2, no fast-math) on %d pthreads, dynamic 32x32 patches + serial scatter"
                  % (len(times), band[0], band[1], w, h, cfg["scene"], depth, cores),
        "message": O.status_message(ms, w, h) if rows == n_rows else None,
    }


# ------------------------------------------------------------------------------------------
def dry_main(args):
    """--dry-run: the multi-process plumbing with an empty step (gloo, no GPU)."""
    import datetime

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    import workloads
    cfg = workloads.CONFIGS[args.config]
    n_rows = cfg["height"] // 32
    c_rows, owned = workloads.cyclic_rows(n_rows, world)

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pass
    fence()
    if rank == args.dry_hang_rank:
        while True:                                           # test hook: a rank that never reaches the barrier
            time.sleep(1.)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rows = torch.tensor([len(owned[rank])], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(rows)
    if rank == 0:
        print(json.dumps({"metric": "Mpixels/sec at 1920x1080, max-bounce=5", "value": None, "unit": "Mpixels/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3,
                          "config": {"workload": args.config, "patch_rows_owned_by_all_ranks": int(rows.item()),
                                     "patch_rows": n_rows, "rows_per_rank": c_rows}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------
def rank_main(args):
    import datetime

    import numpy as np

    # stdout carries exactly one JSON line: RCCL prints a version banner to fd 1 when its
    # communicator comes up, so everything else in this process goes to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # Frames in flight run on streams of their own (rm_frame_submit); HIP maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4, shared with every other stream of the
    # process) and two slots on one queue do not overlap.  Must be set before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # every wait of the library's frame exchange is bounded (rm_frame_wait -> RM_ERR_TIMEOUT)
    os.environ.setdefault("RM_FRAME_TIMEOUT_MS", "30000")
    import torch
    import torch.distributed as dist

    import __graft_entry__ as G
    import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start bench.py with --gpus equal to the launcher's rank count "
                         "(or without a launcher: it starts its own ranks)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    # (RM_BENCH_BACKEND=gloo with every rank on GPU 0 is a dry run of the N > 1 code path on
    # a one-GPU box; the real thing is one GPU per rank over RCCL)
    backend = os.environ.get("RM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d needs GPU %d; this node shows %d" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        # a collective that never completes ends the rank (torch's watchdog) instead of hanging it
        tmo = datetime.timedelta(seconds=120)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    pkg = G.load_package()
    L = pkg.lib()
    partial_path = os.environ.get("RM_BENCH_PARTIAL") if rank == 0 else None
    partial = {}

    def note_partial(key, value):
        """What rank 0 has measured so far, for the launcher to report if the run dies later."""
        if partial_path:
            partial[key] = value
            with open(partial_path + ".tmp", "w") as f:
                json.dump(partial, f)
            os.replace(partial_path + ".tmp", partial_path)

    want_direct = use_dist and args.collective == "direct" and backend == "nccl"
    state = {"direct_note": None}

    def make_comm(ctx):
        """RCCL communicator of the library's own frame exchange; the unique id travels
        through torch.distributed.  Every rank learns whether every rank succeeded."""
        ids = [None]
        if rank == 0:
            try:
                ids[0] = pkg.backend.Context.comm_unique_id()
            except Exception as e:                     # noqa: BLE001 -- reported, then the torch path is used
                sys.stderr.write("rm_comm_unique_id failed: %s\n" % e)
        dist.broadcast_object_list(ids, src=0)
        ok = 0
        if ids[0] is not None:
            try:
                ctx.comm_init(rank, world, ids[0])
                ctx.comm_exchange(args.exchange == "allgather")
                ok = 1 if ctx.comm_info()[:2] == (rank, world) else 0
            except Exception as e:                     # noqa: BLE001
                sys.stderr.write("rank %d: rm_comm_init failed: %s\n" % (rank, e))
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            ctx.comm_destroy()
            return False
        return True

    def run_workload(cfg_id, steps, warmup, extras=False, motion=False):
        """Warm-up, then `steps` frames of one workload between fences -- at N > 1 once per
        exchange path; returns the timings and what rank 0 holds afterwards."""
        cfg = workloads.CONFIGS[cfg_id]
        w, h, depth = cfg["width"], cfg["height"], cfg["max_depth"]
        n_rows = h // 32
        unit = 8 if args.payload == "f64" else 1
        # N > 1: cyclic row ownership (rank r renders patch rows r, r+N, ...: every rank gets its
        # share of cheap sky and expensive ground rows); the payload rows are packed per rank.
        cyclic = use_dist
        c_rows, owned = workloads.cyclic_rows(n_rows, world)
        band = (rank, n_rows, world) if cyclic else (0, n_rows)
        n_owned = len(owned[rank])

        ctx = pkg.backend.Context(local_rank)
        scene = workloads.product_scene(pkg, cfg["scene"])
        ctx.upload(scene.flatten())
        fast_flag = 2 if args.fast_fp else 0                                # RM_FLAG_FAST_FP
        params = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth, band)
        params.flags = fast_flag | ((8 if unit == 8 else 4) if cyclic else 0)   # RM_FLAG_F64_COMPACT / RM_FLAG_U8_COMPACT
        plain = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth)   # the library derives the rank's rows
        plain.flags = fast_flag
        kernel_name = ctx.kernel_name(params)
        pad_rows = max(n_rows, world * c_rows) * 32

        n_buf_torch = 2 if use_dist else 1
        n_buf = 4 if use_dist else 1                      # 3 slots are slower than 2 or 4 (profiles/r01_slots_cost.txt)
        # Per frame in flight: the f64 frame (create_frame_buffer zero-fills; u8 payload: this
        # rank's rows of it, the FrameBuffer is distributed over the ranks' HBM), the gather
        # buffer of the payload (world * c patch rows, rank-major chunks) and, at the consumer,
        # the payload in image order.
        frames = [torch.zeros((max(h, pad_rows), w, 3), dtype=torch.float64, device=dev) for _ in range(n_buf)]
        pay_dtype = torch.float64 if unit == 8 else torch.uint8
        gathered = [torch.zeros((world * c_rows * 32, w, 3), dtype=pay_dtype, device=dev) for _ in range(n_buf)] if use_dist else None
        frames8 = None if use_dist else [torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)]
        images = [torch.zeros_like(g) for g in gathered] if use_dist and rank == 0 else None
        my_chunk = [g[rank * c_rows * 32:(rank + 1) * c_rows * 32] for g in gathered] if use_dist else None

        # A dedicated stream: the kernel, the timing events and torch's RCCL op are all ordered
        # on it (torch.cuda.Event only sees the stream it is recorded on).
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        torch.cuda.set_stream(stream)
        stream_ptr = C.c_void_p(stream.cuda_stream)
        p_ref, plain_ref = C.byref(params), C.byref(plain)
        frame_ptrs = [C.c_void_p(f.data_ptr()) for f in frames]
        chunk_ptrs = [C.c_void_p(ch.data_ptr()) for ch in my_chunk] if use_dist else None
        gather_ptrs = [C.c_void_p(g.data_ptr()) for g in gathered] if use_dist else None
        image_ptrs = [C.c_void_p(d.data_ptr()) for d in images] if images else [None] * n_buf
        frame8_ptr = C.c_void_p(frames8[0].data_ptr()) if frames8 else None
        has_rows = n_owned > 0
        pending = [None] * n_buf
        counter = [0]

        def check(st):
            if st != 0:
                raise pkg.BackendError(st, L.rm_last_error(ctx.ptr).decode())

        def step_single():
            check(L.rm_render_device_u8(ctx.ptr, p_ref, frame_ptrs[0], frame8_ptr, stream_ptr))

        # the call pattern the reference has: the camera moves, then render() (main.rs:74-78)
        walk = [pkg._lib.vec3(pkg.Vec3f(*c)) for c in workloads.camera_walk()]

        def step_walk():
            check(L.rm_camera_update(ctx.ptr, walk[counter[0] % len(walk)]))
            counter[0] += 1
            check(L.rm_render_device_u8(ctx.ptr, p_ref, frame_ptrs[0], frame8_ptr, stream_ptr))

        def step_direct():
            """One frame through the library's exchange: render on the slot's own stream, one
            ncclAllGather and the de-interleave at the consumer behind it on the same stream;
            returns at once (a slot is waited for -- bounded -- before it is reused)."""
            b = counter[0] % n_buf
            counter[0] += 1
            if unit == 1:
                check(L.rm_frame_submit(ctx.ptr, plain_ref, frame_ptrs[b], gather_ptrs[b], image_ptrs[b], b))
            else:
                check(L.rm_frame_submit_f64(ctx.ptr, plain_ref, gather_ptrs[b], image_ptrs[b], b))

        def step_torch():
            """One frame: render this rank's rows (f64 rows + their display bytes, the payload
            packed into this rank's chunk), then the single collective of the frame
            (asynchronous: the stream is only made to wait for it when its buffer is about to
            be rendered into again)."""
            b = counter[0] % n_buf_torch
            counter[0] += 1
            if pending[b] is not None:
                for wk in pending[b]:
                    wk.wait()                          # stream-ordered; the host does not block
                pending[b] = None
                if images:
                    workloads.deinterleave_rows(gathered[b], world, images[b])
            if has_rows:
                if unit == 1:
                    check(L.rm_render_device_u8(ctx.ptr, p_ref, frame_ptrs[b], chunk_ptrs[b], stream_ptr))
                else:
                    check(L.rm_render_device(ctx.ptr, p_ref, chunk_ptrs[b], stream_ptr))
            if args.exchange == "allgather":
                pending[b] = [dist.all_gather_into_tensor(gathered[b], my_chunk[b], async_op=True)]
            else:
                pending[b] = workloads.gather_chunks(dist, gathered[b], rank, world)

        def drain(direct):
            if direct:
                for b in range(n_buf):
                    ctx.frame_wait(b)                   # bounded: raises RM_ERR_TIMEOUT
            for i, wks in enumerate(pending):
                if wks is not None:
                    for wk in wks:
                        wk.wait()
                    pending[i] = None
                    if images:
                        workloads.deinterleave_rows(gathered[i], world, images[i])

        def fence(direct=False):
            drain(direct)
            # (a host thread that blocks in the wait wakes up a scheduler tick late on a busy box -- 5 ms were read once
            # on 14 ms of launches: the stream is polled first, the wait then returns at once)
            while not stream.query():
                pass
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize()                 # (the barrier is a collective on the device)

        def timed(step, direct, n_steps=None):
            """--warmup steps + launches until WARMUP_SECONDS have passed (the same count on
            every rank), then EXACTLY `steps` steps between fences; max over ranks.
            (`n_steps`: a leg beside the metric that times another number of steps.)"""
            n = n_steps or steps
            counter[0] = 0
            for _ in range(warmup):
                step()
            fence(direct)
            # how many launches fill WARMUP_SECONDS: from a batch that no longer pays for the first
            # launch of the process (module load, buffers: milliseconds -- counted into the estimate
            # they cut the time-based warm-up of a --warmup 5 run to a few dozen launches)
            t_w = time.perf_counter()
            for _ in range(WARMUP_PROBE):
                step()
            fence(direct)
            per = (time.perf_counter() - t_w) / WARMUP_PROBE
            extra = torch.tensor([int(WARMUP_SECONDS / max(per, 1e-6)) + 1], dtype=torch.int64, device=dev)
            if use_dist:
                dist.broadcast(extra, src=0)
            for _ in range(min(int(extra.item()), 20000)):
                step()
            # HIP events on the launch stream bracket the timed region (one pair: an event per
            # launch would put two extra packets between consecutive kernels)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream); ev1.record(stream)      # (torch creates an event when it is first recorded: not inside the bracket)
            fence(direct)
            t0 = time.perf_counter()
            ev0.record(stream)
            for _ in range(n):
                step()
            t_enq = time.perf_counter()
            ev1.record(stream)
            fence(direct)
            elapsed = time.perf_counter() - t0
            if os.environ.get("RM_BENCH_TRACE"):
                sys.stderr.write("[bench] bracket: %d steps, wall %.1f us, enqueued after %.1f us, events %.1f us\n"
                                 % (n, elapsed * 1e6, (t_enq - t0) * 1e6, ev0.elapsed_time(ev1) * 1e3))
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            if use_dist:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            # average launch duration of this rank's render kernel over the timed region: at
            # N = 1 the events bracket exactly the K launches on their stream; with frames in
            # flight on the slots' own streams there is no per-launch bracket, the per-frame
            # time of the pipeline is reported instead
            kernel_ms = ev0.elapsed_time(ev1) / n if not direct else elapsed / n * 1e3
            return {"elapsed": elapsed, "ms_per_step": elapsed / n * 1e3, "kernel_ms": kernel_ms,
                    "mpx": (w * h) * n / elapsed / 1e6}

        paths = {}
        chosen = "none"
        if not use_dist:
            paths["none"] = timed(step_single, False)
        else:
            paths["torch"] = dict(timed(step_torch, False), frames_in_flight=n_buf_torch)
            chosen = "torch"
            if cfg_id == args.config:
                note_partial("torch_path", dict(paths["torch"], workload=cfg_id, n_gpus=world))
            if want_direct and make_comm(ctx):
                try:
                    # first frame through both paths: the gathered payload must be identical
                    want = gathered[0].clone()
                    want_img = images[0].clone() if images else None
                    gathered[0].zero_()
                    counter[0] = 0
                    step_direct()
                    fence(True)
                    # (gathered at rank 0: a peer holds only its own chunk, in both paths)
                    same = ((rank != 0 and args.exchange == "gather") or torch.equal(gathered[0], want)) and \
                           torch.equal(my_chunk[0], want[rank * c_rows * 32:(rank + 1) * c_rows * 32]) and \
                           (want_img is None or torch.equal(images[0], want_img))
                    flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    if int(flag.item()) != 1:
                        state["direct_note"] = "library exchange disagreed with the torch path on the check frame"
                        ctx.comm_destroy()
                    else:
                        paths["direct"] = dict(timed(step_direct, True), frames_in_flight=n_buf,
                                               rccl_sees=dict(zip(("rank", "world", "communicators"), ctx.comm_info())))
                        # device times of one frame on its stream (stamped frames, outside the timed region)
                        ctx.frame_timing_enable(True)
                        counter[0] = 0
                        for _ in range(n_buf):
                            step_direct()
                        fence(True)
                        t = ctx.frame_timing(0)
                        ctx.frame_timing_enable(False)
                        paths["direct"]["one_frame_on_its_stream_ms"] = {"kernel": t.kernel_ms, "gather": t.gather_ms, "total": t.total_ms}
                        if paths["direct"]["mpx"] >= paths["torch"]["mpx"]:
                            chosen = "direct"                   # `value` is the faster of the two valid exchanges
                except pkg.BackendError as e:
                    if e.status != pkg._lib.RM_ERR_TIMEOUT:
                        raise
                    # A peer never joined (or left) the library's collective.  This rank's streams
                    # are stuck behind it: nothing more can be measured.  Report what the torch
                    # path measured and leave, non-zero, without another collective.
                    sys.stderr.write("rank %d: %s\n" % (rank, e))
                    if rank == 0:
                        emit(result(cfg, paths, "torch", "direct timed out: %s" % e, kernel_name, n_owned, c_rows, cyclic, None, None, None, []))
                    sys.stderr.flush()
                    os._exit(4)
            elif want_direct:
                state["direct_note"] = "library communicator could not be created on every rank"

        best = paths[chosen]
        tiles = None
        moving = None
        if motion and not use_dist:
            # (after the metric's own run: the standing view's history is on the stream, as in a host that has just stopped)
            # (whole laps of the walk: K steps from wherever the warm-up happened to stop are K views of a closed walk whose views
            # take 40 to 60 us -- the driver's 20 steps read 43.7 to 55.3 us from run to run)
            n_walk = -(-max(steps, 1) // len(walk)) * len(walk)
            mv = timed(step_walk, False, n_walk)
            moving = {"value": mv["mpx"], "unit": "Mpixels/s", "steps": n_walk, "ms_per_step": mv["ms_per_step"],
                      "kernel_ms": mv["kernel_ms"], "camera": "every frame one press from the last", "what": MOTION_NOTE}
            # a view held for four frames: its first frame and its fourth, each launch in an event bracket of its own
            firsts, fourths = [], []
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(2)]
            for v in range(12 + min(max(steps, 8), 40)):
                check(L.rm_camera_update(ctx.ptr, walk[(7 * v) % len(walk)]))
                for f in range(4):
                    e = evs[0] if f == 0 else evs[1] if f == 3 else None
                    if e:
                        e[0].record(stream)
                    step_single()
                    if e:
                        e[1].record(stream)
                fence()
                if v >= 12:
                    firsts.append(evs[0][0].elapsed_time(evs[0][1]))
                    fourths.append(evs[1][0].elapsed_time(evs[1][1]))
            moving["first_frame_of_a_view"] = {"kernel_ms_median": float(np.median(firsts)), "fourth_frame_of_the_view_kernel_ms_median": float(np.median(fourths)),
                                               "views": len(firsts), "what": "a view held for four frames, every view a jump to another point of the walk; "
                                                                             "each launch timed by an event pair of its own (which costs a few us)"}
            check(L.rm_camera_update(ctx.ptr, pkg._lib.vec3(scene.camera)))
            counter[0] = 0
            for _ in range(3):
                step_single()
            fence()
        if not use_dist:
            n_t, n_lit = ctx.tile_stats(stream.cuda_stream)
            tiles = {"tiles": n_t, "with_something_to_hit": n_lit,
                     "classified": bool(n_lit != n_t)}
            wgs, tail = ctx.launch_stats()
            tiles.update(workgroups=wgs, sky_tail_patches=tail)
        frame, frame8 = frames[0], (frames8[0] if frames8 else None)
        image = images[0] if images else None
        px_launch = n_owned * 32 * w                                        # pixels one launch writes

        host = piped = None
        if extras and world == 1 and rank == 0 and not use_dist:
            # the drop-in call itself, rm_render into host memory (what
            # Renderer::render hands back), a few frames outside the timed region
            torch.cuda.synchronize()
            host_frame = np.zeros((h, w, 3), dtype=np.float64)
            p_host = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth)
            p_host.flags = fast_flag
            ctx.render(p_host, host_frame)
            ts, tm = [], None
            for _ in range(10):
                t1 = time.perf_counter()
                tm = ctx.render(p_host, host_frame)
                ts.append(time.perf_counter() - t1)
            med = float(np.median(ts))
            host = {"value": w * h / med / 1e6, "unit": "Mpixels/s", "ms_per_frame": med * 1e3,
                    "kernel_ms": tm.kernel_ms, "d2h_ms": tm.d2h_ms,
                    "what": "rm_render(): kernel + device-to-host copy of the f64 frame (the patches that are not black) into the caller's flat pageable memory, median of 10"}
            # the same call into page-locked memory from rm_host_alloc -- what the Rust shim's
            # staging buffer is
            hp = C.c_void_p()
            check(L.rm_host_alloc(ctx.ptr, host_frame.nbytes, C.byref(hp)))
            pinned = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_double)), shape=(h * w * 3,)).reshape(h, w, 3)
            ctx.render(p_host, pinned)
            ts = []
            for _ in range(10):
                t1 = time.perf_counter()
                tm = ctx.render(p_host, pinned)
                ts.append(time.perf_counter() - t1)
            med = float(np.median(ts))
            host["into_page_locked_memory"] = {"value": w * h / med / 1e6, "ms_per_frame": med * 1e3, "kernel_ms": tm.kernel_ms,
                                               "d2h_ms": tm.d2h_ms, "identical": bool(np.array_equal(pinned, host_frame))}
            del pinned
            L.rm_host_free(ctx.ptr, hp)
            st = ctx.hostio_stats()
            host["bytes_over_the_link"] = st["bytes_copied"]
            host["patches_sent"] = "%d of %d (black 32x32 patches are written by the host, not sent)" % (st["patches_sent"], st["patches"])
            # The seam the reference really has: its FrameBuffer is Vec<Vec<Vec3f>>, one
            # heap allocation per scan line (framebuffer.rs:6-22), and its window consumes fb.to_vec()
            # (main.rs:337-346).  rm_seam is compiled host code over the C ABI holding exactly that --
            # a std::vector per row -- run here as a child process while this one idles.
            seam = os.path.join(G.PKG_DIR, "lib", "rm_seam")
            if os.path.exists(seam):
                try:
                    cmd = [seam, "--width", str(w), "--height", str(h), "--depth", str(depth), "--frames", "20"]
                    if cfg["scene"] == "cornell":
                        cmd += ["--scene", workloads.CORNELL]
                    if args.fast_fp:
                        cmd += ["--fast-fp"]
                    if cfg["scene"] in ("demo", "cornell"):
                        # (the host side of these calls -- eight scatter threads -- shares the box's cores with other tenants: one
                        # run of twenty calls read 0.63 ms, the next 1.02; three runs, the quickest kept, all three reported)
                        runs, pr, line = [], None, ""
                        for _ in range(3):
                            pr_k = subprocess.run(cmd, capture_output=True, timeout=300)
                            line_k = pr_k.stdout.decode().strip().splitlines()[-1] if pr_k.stdout.strip() else ""
                            ok_k = pr_k.returncode == 0 and line_k.startswith("{")
                            ms_k = json.loads(line_k)["rows_of_rows"]["ms_per_call"] if ok_k else None
                            if pr is None or (ok_k and (not runs or all(r is None or ms_k <= r for r in runs))):
                                pr, line = pr_k, line_k
                            runs.append(ms_k)
                        if pr.returncode == 0 and line.startswith("{"):
                            sj = json.loads(line)
                            sj["rows_of_rows"]["ms_per_call_of_each_run"] = runs
                            host["rows_of_rows"] = dict(sj["rows_of_rows"], value=sj["rows_of_rows"]["mpx_per_s"], unit="Mpixels/s",
                                                        host_threads=sj["host_threads"])
                            host["display_only"] = sj["display_only"]
                            host["fetch_rows_ms"] = sj["fetch_rows"]["ms_per_call"]
                            host["flat_from_compiled_code"] = sj["flat"]
                        else:
                            host["rows_of_rows"] = {"error": "rm_seam exited with %d: %s" % (pr.returncode, pr.stderr.decode()[-300:])}
                except Exception as e:                   # noqa: BLE001 -- an extra beside the metric: reported, not fatal
                    host["rows_of_rows"] = {"error": "rm_seam: %s" % e}
            # the same frames with four in flight, each on a stream of its own (rm_frame_submit with no
            # communicator): what a renderer gets that need not wait for frame k before starting
            # k+1 -- the ramp and drain of one frame fill with the others.  Not the metric: a step
            # of the metric is one frame at a time, like the reference's render().
            n_slots = 4
            pf = [torch.zeros((h, w, 3), dtype=torch.float64, device=dev) for _ in range(n_slots)]
            pg = [torch.zeros((n_rows * 32, w, 3), dtype=torch.uint8, device=dev) for _ in range(n_slots)]
            n_frames = max(10 * steps, 400)              # (30 frames were 2 ms of wall time: r3's 72.6 us against 62.7 over 600 frames, profiles/slots_cost.py)
            n_warm = n_slots * 20                        # (each slot's stream needs a few frames of its own: order, first round, sky tail)
            # (timed by the host's clock, the submitting thread on a core shared with other tenants: four chunks, each drained,
            # the median chunk reported -- one hiccup of a few milliseconds in a 26-ms leg read 84 us a frame against 63)
            n_chunks = 4
            per_chunk = -(-n_frames // n_chunks)
            k, chunk_s = 0, []
            for _ in range(n_warm):
                ctx.frame_submit(plain, pf[k % n_slots].data_ptr(), pg[k % n_slots].data_ptr(), None, k % n_slots)
                k += 1
            for _ in range(n_chunks):
                for b in range(n_slots):
                    ctx.frame_wait(b)
                t1 = time.perf_counter()
                for _ in range(per_chunk):
                    ctx.frame_submit(plain, pf[k % n_slots].data_ptr(), pg[k % n_slots].data_ptr(), None, k % n_slots)
                    k += 1
                for b in range(n_slots):
                    ctx.frame_wait(b)
                chunk_s.append(time.perf_counter() - t1)
            dt = float(np.median(chunk_s))
            piped = {"value": w * h * per_chunk / dt / 1e6, "unit": "Mpixels/s", "ms_per_step": dt / per_chunk * 1e3,
                     "frames_in_flight": n_slots, "ms_per_step_of_each_chunk": [c / per_chunk * 1e3 for c in chunk_s],
                     "what": "rm_frame_submit, %d chunks of %d frames (each drained; the median chunk), same outputs per frame as the metric" % (n_chunks, per_chunk)}
            if args.check:
                piped["identical_to_single_stream_frame"] = bool(torch.equal(pf[0][:n_rows * 32], frame[:n_rows * 32])
                                                                 and torch.equal(pg[0], frame8[:n_rows * 32]))
            del pf, pg
        checks = None
        if args.check and rank == 0:
            O = G.load_oracle()
            ref = O.render(workloads.oracle_scene(O, cfg["scene"]), w, h, max_depth=depth)[:n_rows * 32]
            checks = {}
            f64 = frame[:n_rows * 32] if not use_dist else (image[:n_rows * 32] if unit == 8 else None)
            if f64 is not None:
                checks["max_abs_delta_vs_oracle"] = float(np.abs(f64.cpu().numpy() - ref).max())
            u8 = frame8[:n_rows * 32] if not use_dist else (image[:n_rows * 32] if unit == 1 else None)
            if u8 is not None:
                checks["display_bytes_differing_from_oracle"] = int((u8.cpu().numpy().reshape(-1) != O.to_vec(ref.copy())).sum())
        res = dict(cfg=cfg, paths=paths, chosen=chosen, kernel_name=kernel_name, n_owned=n_owned, c_rows=c_rows, cyclic=cyclic,
                   host=host, piped=piped, checks=checks, px_launch=px_launch, best=best, tiles=tiles, moving=moving)
        ctx.close()
        torch.cuda.empty_cache()
        return res

    def result(cfg, paths, chosen, note, kernel_name, n_owned, c_rows, cyclic, host, piped, checks, other, tiles=None, moving=None):
        """The JSON line."""
        w, h, depth = cfg["width"], cfg["height"], cfg["max_depth"]
        best = paths[chosen]
        px_launch = n_owned * 32 * w
        kernel_ms = best["kernel_ms"]
        achieved = px_launch * BYTES_PER_PIXEL / (kernel_ms * 1e-3) / 1e9
        traffic, valu, pmc_note = pmc_for(args.config, kernel_name, kernel_ms) if world == 1 else (None, None, "PMC figures are per single-GPU launch")
        coll = {"none": "none (one GPU)",
                "direct": "rm_frame_submit%s: %s from the C library" % ("_f64" if args.payload == "f64" else "",
                                                                       "ncclAllGather" if args.exchange == "allgather" else "grouped ncclSend / ncclRecv to rank 0"),
                "torch": "torch.distributed %s (%s)" % ("all_gather_into_tensor" if args.exchange == "allgather" else "batch_isend_irecv to rank 0",
                                                       "RCCL" if backend == "nccl" else backend)}[chosen]
        note = note or state["direct_note"]
        out = {
            "metric": "Mpixels/sec at 1920x1080, max-bounce=5" if args.config == "C2"
                      else "Mpixels/sec (%s)" % args.config,
            "value": best["mpx"], "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": best["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s scene %dx%d, depth cap %d, fov 1.5, device-resident f64 RGB frame"
                                   % (args.config, cfg["scene"], w, h, depth),
                       "sharding": ("cyclic patch rows (rank r: rows r, r+N, ...), %d of 32 px per rank, %d rank(s), %s of the %s rows "
                                    "per frame%s"
                                    % (c_rows, world, "one in-place RCCL all-gather" if args.exchange == "allgather" else
                                       "one gather at rank 0 (grouped send / recv, every peer over its own link)",
                                       args.payload, "" if args.payload == "f64" else "; f64 rows stay in each rank's HBM"))
                                   if cyclic else "one GPU renders all %d patch rows" % c_rows,
                       "collective": coll + (" [%s]" % note if note else ""),
                       "frames_in_flight": best.get("frames_in_flight", 1),
                       "warmup_policy": "--warmup steps, then launches until %.1f s have passed" % WARMUP_SECONDS,
                       "outputs": "f64 RGB frame [H][W][3] + u8 display frame (to_vec) per launch",
                       "numerics": ("fast (FMA contraction, Newton rsqrt; exact-incidence pixels may differ)" if args.fast_fp else
                                    "strict: every hit / miss / shadow / side decision computed with the reference's operations, "
                                    "one rounding each, in its order; colour sums after the shadow test use FMAs, the viewer "
                                    "direction is the ray's, integer specular exponents by repeated multiplication "
                                    "(measured max |delta| 4e-14, asserted < 1e-9)"),
                       "build": L.rm_build_info().decode(), "csrc_sha16": csrc_hash()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": kernel_name, "kernel_ms": kernel_ms,
                         "bytes_per_launch": px_launch * BYTES_PER_PIXEL,
                         "fp64_valu": valu,
                         "note": "path is FP64-VALU bound by construction (SURVEY.md 8d); achieved = 24 B x pixels "
                                 "written / kernel time (HIP events on the launch stream around the K steps: a step is the "
                                 "render launch and, where tiles are classified, the classification launch in front of it); " + pmc_note},
        }
        targs = kernel_template_args(kernel_name)                     # ..., EDGES, ORDER, FEEDBACK, HANDON
        if len(targs) > 9 and targs[9] == "true":
            out["config"]["dispatch"] = FEEDBACK_NOTE
        elif len(targs) > 8 and targs[8] == "true":
            out["config"]["dispatch"] = ORDER_NOTE
        if tiles:
            out["config"]["tiles"] = dict(tiles, note=CLASSIFY_NOTE)
        if world > 1 or args.force_dist:
            out["exchange_paths"] = paths
        if checks:
            out.update(checks)
        if host:
            out["end_to_end_host"] = host
        if moving:
            out["camera_on_the_move"] = moving
        if piped:
            out["four_frames_in_flight"] = piped
        if other:
            out["other_frames"] = other
        return out

    def emit(out):
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()

    r = run_workload(args.config, args.steps, args.warmup, extras=not args.no_sizes, motion=not args.no_motion)
    # north star: "Mpixels/sec on synthetic 1080p/4K/8K frames", and the other GPU configs of
    # BASELINE.json (cornell C3, synthetic C5): a short run each (same sharding and collective),
    # reported beside the metric with their own kernel and roofline fraction
    other = []
    if args.config == "C2" and not args.no_sizes:
        for cid in ("C2_4K", "C4", "C3", "C5"):
            k = max(5, min(args.steps, 20 if cid != "C3" else 100))
            o = run_workload(cid, k, 3, motion=(cid == "C3"))
            ach = o["px_launch"] * BYTES_PER_PIXEL / (o["best"]["kernel_ms"] * 1e-3) / 1e9
            other.append({"workload": "%s: %s scene %dx%d, depth cap %d" % (cid, o["cfg"]["scene"], o["cfg"]["width"], o["cfg"]["height"], o["cfg"]["max_depth"]),
                          "value": o["best"]["mpx"], "unit": "Mpixels/s", "steps": k, "ms_per_step": o["best"]["ms_per_step"],
                          "kernel": o["kernel_name"], "kernel_ms": o["best"]["kernel_ms"] if world == 1 else None,
                          "collective": o["chosen"],
                          "roofline_frac": (ach / HBM_PEAK_GBPS) if world == 1 else None})
            if world == 1:
                t_, v_, n_ = pmc_for(cid, o["kernel_name"], o["best"]["kernel_ms"])
                other[-1].update(traffic=t_, fp64_valu=v_, pmc=n_, tiles=o["tiles"])
                if o["moving"]:
                    other[-1]["camera_on_the_move"] = o["moving"]
            targs = kernel_template_args(o["kernel_name"])
            if len(targs) > 9 and targs[9] == "true":
                other[-1]["dispatch"] = FEEDBACK_NOTE
            elif len(targs) > 8 and targs[8] == "true":
                other[-1]["dispatch"] = ORDER_NOTE
            o.clear()

    if rank == 0:
        out = result(r["cfg"], r["paths"], r["chosen"], None, r["kernel_name"], r["n_owned"], r["c_rows"], r["cyclic"],
                     r["host"], r["piped"], r["checks"], other, r["tiles"], r["moving"])
        if world == 1 and not args.no_cpu_baseline:
            O = G.load_oracle()
            out["cpu_baseline"] = cpu_baseline(O, workloads, r["cfg"])
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            # ... and at the seam: the drop-in call itself, frame in host memory, against the same CPU path
            e2e = out.get("end_to_end_host") or {}
            cpu_v = out["cpu_baseline"]["value"]
            at_seam = {"kernel_only_device_resident": out["speedup_vs_cpu_baseline"]}
            if e2e.get("value"):
                at_seam["rm_render_flat_pageable"] = e2e["value"] / cpu_v
            if (e2e.get("rows_of_rows") or {}).get("value"):
                at_seam["rm_render_rows_rows_of_rows"] = e2e["rows_of_rows"]["value"] / cpu_v
            if (e2e.get("display_only") or {}).get("frames_per_s"):
                at_seam["rm_render_display_display_bytes"] = e2e["display_only"]["frames_per_s"] * (r["cfg"]["width"] * r["cfg"]["height"]) / 1e6 / cpu_v
            out["speedup_vs_cpu_baseline_at_the_seam"] = at_seam
        emit(out)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)                       # before anything touches a GPU
    if args.dry_run:
        return dry_main(args)
    return rank_main(args)


if __name__ == "__main__":
    sys.exit(main())
