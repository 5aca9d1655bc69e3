#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the render hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one frame of the workload: BASELINE.json's configs[1] (demo scene,
1920x1080, depth cap 5) unless --config says otherwise.  Each launch writes the f64 RGB
frame (the reference's FrameBuffer) and its display bytes (`to_vec`, what the UI blits)
to HBM; inputs (the scene) are uploaded before the timed region.
N > 1: the frame is sharded by 32-row patch rows (cyclic ownership: rank r
renders patch rows r, r+N, ...), and ONE in-place RCCL all-gather per frame, inside the
timed region, completes the display frame on every rank (--payload u8, default; rank 0 is
the consumer) or the f64 frame (--payload f64, contiguous bands).  The default exchange is
the library's own (rm_frame_submit: four frames in flight, each on a stream of its own,
ncclAllGather issued from C); its first frame is checked byte for byte against
torch.distributed.all_gather_into_tensor, which takes over if they disagree
(--collective torch selects it outright).  Total work is fixed: "strong" scaling.

Rank 0 prints ONE JSON line.  Metric definition follows the reference
(renderer.rs:113-120): frame_width x frame_height pixels per frame / wall time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_PIXEL = 24            # 3 x f64 written per pixel (framebuffer.rs Vec3f), SURVEY.md 8d


def cpu_baseline(O, workloads, cfg, budget_s=12.0):
    """The oracle driven like renderer.rs:63-108 (dynamic 32x32 patches over all host
    threads, per-patch buffers, serial scatter), timed on a bounded sample: whole frames
    of the same workload until ~budget_s of wall time or 12 frames."""
    scene = workloads.oracle_scene(O, cfg["scene"])
    w, h, depth = cfg["width"], cfg["height"], cfg["max_depth"]
    cores = O.lib().orc_online_cpus()
    # large frames: render a band sample instead of whole frames
    n_rows = h // 32
    rows = n_rows
    est_px_per_s = 3.0e6 * cores if cfg["scene"] == "demo" else 2.0e5 * cores
    while rows > 1 and (rows * 32 * w) / est_px_per_s > budget_s / 3:
        rows //= 2
    band = ((n_rows - rows) // 2, (n_rows - rows) // 2 + rows)
    frame = np.zeros((h, w, 3), dtype=np.float64)
    O.render(scene, w, h, max_depth=depth, frame=frame, band=band)      # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 12 and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        O.render(scene, w, h, max_depth=depth, frame=frame, band=band)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    # reference metric: full-width rows; scale the band's pixels by the frame convention
    px = w * h if rows == n_rows else rows * 32 * w
    ms = int(med * 1000)
    return {
        "value": px / med / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "sample": "%d frame(s) of patch rows [%d,%d) of %dx%d %s depth %d, median of %d; "
                  "oracle (-O2, no fast-math) on pthreads, dynamic 32x32 patches + serial scatter"
                  % (len(times), band[0], band[1], w, h, cfg["scene"], depth, len(times)),
        "message": O.status_message(ms, w, h) if rows == n_rows else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", help="workload id from workloads.CONFIGS")
    ap.add_argument("--payload", choices=["u8", "f64"], default="u8",
                    help="what the per-frame all-gather moves at N > 1: the display bytes (to_vec, 3 B/px) "
                         "or the f64 rows (24 B/px)")
    ap.add_argument("--collective", choices=["direct", "torch"], default="direct",
                    help="N > 1, u8 payload: 'direct' = the library's own frame exchange (rm_frame_submit: frames in "
                         "flight on their own streams, ncclAllGather called from C; checked against the torch path "
                         "on the first frame, falls back to it on any disagreement); 'torch' = "
                         "torch.distributed.all_gather_into_tensor per frame")
    ap.add_argument("--fast-fp", action="store_true",
                    help="RM_FLAG_FAST_FP flavour of the kernel (FMA, Newton rsqrt): faster, but may decide "
                         "exact-incidence pixels differently from the reference -- not the parity configuration")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the per-frame collective even at world size 1 (smoke test of the N > 1 code path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sizes", action="store_true", help="skip the extras reported beside the metric (4K / 8K frames, rm_render into host memory)")
    ap.add_argument("--check", action="store_true", help="also compare the frame with the oracle")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: RCCL prints a version banner to fd 1 when its
    # communicator comes up, so everything else in this process goes to stderr.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # Frames in flight run on streams of their own (rm_frame_submit); HIP maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4, shared with every other stream of the
    # process) and two slots on one queue do not overlap.  Must be set before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    import __graft_entry__ as G
    import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    # (RM_BENCH_BACKEND=gloo with every rank on GPU 0 is a dry run of the N > 1 code path on
    # a one-GPU box; the real thing is one GPU per rank over RCCL)
    backend = os.environ.get("RM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29533"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = G.load_package()
    L = pkg.lib()

    # How the frames are exchanged at N > 1 (decided once, the same on every rank).
    mode = {"collective": "none", "note": None}
    if use_dist:
        mode["collective"] = "torch"
        if args.collective == "direct" and args.payload == "u8" and backend == "nccl":
            mode["collective"] = "direct"

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
                ok = 1
            except Exception as e:                     # noqa: BLE001
                sys.stderr.write("rank %d: rm_comm_init failed: %s\n" % (rank, e))
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            ctx.comm_destroy()
            return False
        return True

    def run_workload(cfg_id, steps, warmup):
        """Warm-up, then `steps` frames of one workload between fences; returns the
        timings and what rank 0 holds afterwards."""
        cfg = workloads.CONFIGS[cfg_id]
        w, h, depth = cfg["width"], cfg["height"], cfg["max_depth"]
        n_rows = h // 32
        # N > 1, u8 payload: cyclic row ownership (rank r renders patch rows r, r+N, ...: every
        # rank gets its share of cheap sky and expensive ground rows) with the display bytes
        # packed per rank; f64 payload: contiguous equal bands gathered in place.
        cyclic = use_dist and args.payload == "u8"
        if cyclic:
            c_rows, owned = workloads.cyclic_rows(n_rows, world)
            band = (rank, n_rows, world)
            n_owned = len(owned[rank])
        else:
            c_rows, bands = workloads.equal_bands(n_rows, world)
            band = bands[rank]
            n_owned = band[1] - band[0]

        ctx = pkg.backend.Context(local_rank)
        scene = workloads.product_scene(pkg, cfg["scene"])
        ctx.upload(scene.flatten())
        params = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth, band)
        if args.fast_fp:
            params.flags |= 2                                              # RM_FLAG_FAST_FP
        if cyclic:
            params.flags |= 4                                              # RM_FLAG_U8_COMPACT
        # Per rank: the f64 frame (create_frame_buffer zero-fills; this rank's band of it is
        # rendered, the FrameBuffer is distributed over the ranks' HBM) and the display frame
        # (u8, `to_vec`), padded to world * c patch rows so the bands all-gather in place.
        pad_h = max(h, world * c_rows * 32)
        # Two frames in flight at N > 1: frame k's collective runs on RCCL's stream while frame
        # k+1 renders (double-buffered), so steady-state throughput is 1 / max(render, gather).
        direct = mode["collective"] == "direct" and cyclic
        if direct and not make_comm(ctx):
            direct = False
            mode["collective"], mode["note"] = "torch", "library communicator could not be created on every rank"
        n_buf = (4 if direct else 2) if use_dist else 1   # 3 is slower than 2 or 4 (profiles/r01_slots_cost.txt)
        frames = [torch.zeros((pad_h, w, 3), dtype=torch.float64, device=dev) for _ in range(n_buf)]
        frames8 = [torch.zeros((pad_h, w, 3), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
        frame, frame8 = frames[0], frames8[0]
        gathered = [(f8 if args.payload == "u8" else f)[:world * c_rows * 32] for f, f8 in zip(frames, frames8)]
        my_chunk = [g[rank * c_rows * 32:(rank + 1) * c_rows * 32] for g in gathered]   # views built once
        # A dedicated stream: the kernel, the timing events and the RCCL op are all ordered
        # on it (torch.cuda.Event only sees the stream it is recorded on).
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        torch.cuda.set_stream(stream)
        p_ref = C.byref(params)
        frame_ptrs = [C.c_void_p(f.data_ptr()) for f in frames]
        # cyclic: the kernel packs this rank's display rows straight into its chunk of the gather buffer
        frame8_ptrs = [C.c_void_p((ch if cyclic else f8).data_ptr()) for ch, f8 in zip(my_chunk, frames8)]
        # image-order display frames at the consumer, one per frame in flight
        displays = [torch.zeros_like(g) for g in gathered] if cyclic and rank == 0 else None
        display = displays[0] if displays else None
        stream_ptr = C.c_void_p(stream.cuda_stream)
        has_rows = n_owned > 0
        pending = [None] * n_buf
        counter = [0]

        plain = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth)   # the library derives the rank's rows
        plain.flags = params.flags & 2
        plain_ref = C.byref(plain)
        gather_ptrs = [C.c_void_p(g.data_ptr()) for g in gathered]
        display_ptrs = [C.c_void_p(d.data_ptr()) for d in displays] if displays else [None] * n_buf

        def step_direct():
            """One frame through the library's exchange: render on the slot's own stream, one
            ncclAllGather and the de-interleave at the consumer behind it on the same stream; returns at once."""
            b = counter[0] % n_buf
            counter[0] += 1
            st = L.rm_frame_submit(ctx.ptr, plain_ref, frame_ptrs[b], gather_ptrs[b], display_ptrs[b], b)
            if st != 0:
                raise RuntimeError(L.rm_last_error(ctx.ptr).decode())

        def step_torch():
            """One frame: render this rank's band (f64 rows + their display bytes), then the
            single collective of the frame (asynchronous: the stream is only made to wait for
            it when its buffer is about to be rendered into again)."""
            b = counter[0] % n_buf
            counter[0] += 1
            if pending[b] is not None:
                pending[b].wait()                      # stream-ordered; the host does not block
                pending[b] = None
                if displays:
                    workloads.deinterleave_rows(gathered[b], world, displays[b])
            if has_rows:
                st = L.rm_render_device_u8(ctx.ptr, p_ref, frame_ptrs[b], frame8_ptrs[b], stream_ptr)
                if st != 0:
                    raise RuntimeError(L.rm_last_error(ctx.ptr).decode())
            if use_dist:
                pending[b] = dist.all_gather_into_tensor(gathered[b], my_chunk[b], async_op=True)

        def drain():
            if direct:
                for b in range(n_buf):
                    if L.rm_frame_wait(ctx.ptr, b) != 0:
                        raise RuntimeError(L.rm_last_error(ctx.ptr).decode())
            for i, wk in enumerate(pending):
                if wk is not None:
                    wk.wait()
                    pending[i] = None
                    if displays:
                        workloads.deinterleave_rows(gathered[i], world, displays[i])

        def fence():
            drain()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        if direct:
            # first frame through both paths: the gathered display bytes must be identical
            step_torch()
            fence()
            want = gathered[0].clone()
            gathered[0].zero_()
            counter[0] = 0
            step_direct()
            fence()
            same = torch.tensor([1 if torch.equal(gathered[0], want) else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            counter[0] = 0
            if int(same.item()) != 1:
                direct = False
                ctx.comm_destroy()
                mode["collective"], mode["note"] = "torch", "library exchange disagreed with the torch path on the check frame"
        step = step_direct if direct else step_torch
        for _ in range(warmup):
            step()
        # HIP events on the launch stream bracket the timed region (one pair: an event per
        # launch would put two extra packets between consecutive kernels)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        t0 = time.perf_counter()
        ev0.record(stream)
        for k in range(steps):
            step()
        ev1.record(stream)
        fence()
        elapsed = time.perf_counter() - t0

        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

        # average launch duration of this rank's render kernel over the timed region (for
        # N > 1 the gather's stream time is inside the bracket too, so it is only reported
        # as the kernel's duration at N = 1)
        kernel_ms = ev0.elapsed_time(ev1) / steps
        if direct:
            # the frames render on the slots' own streams and overlap: there is no per-launch
            # duration to bracket; report the per-frame time of the pipeline instead
            kernel_ms = elapsed / steps * 1e3
        px_launch = n_owned * 32 * w                                        # pixels one launch writes

        # the drop-in call itself, rm_render into pageable host memory
        # (what Renderer::render hands back), a few frames outside the timed region
        host = None
        if world == 1 and rank == 0 and cfg_id == args.config and not args.no_sizes:
            torch.cuda.synchronize()
            host_frame = np.zeros((h, w, 3), dtype=np.float64)
            p_host = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth, band)
            p_host.flags = params.flags & 2
            ctx.render(p_host, host_frame)
            ts, tm = [], None
            for _ in range(10):
                t1 = time.perf_counter()
                tm = ctx.render(p_host, host_frame)
                ts.append(time.perf_counter() - t1)
            med = float(np.median(ts))
            host = {"value": w * h / med / 1e6, "unit": "Mpixels/s", "ms_per_frame": med * 1e3,
                    "kernel_ms": tm.kernel_ms, "d2h_ms": tm.d2h_ms,
                    "what": "rm_render(): kernel + device-to-host copy of the f64 frame into pageable memory, median of 10"}
        # the same frames with four in flight, each on a stream of its own (rm_frame_submit with no
        # communicator): what a renderer gets that need not wait for frame k before starting
        # k+1 -- the ramp and drain of one frame fill with the others (four: HIP maps streams onto
        # four hardware queues, and two slots may land on one).  Not the metric: a step
        # of the metric is one frame at a time, like the reference's render().
        piped = None
        if world == 1 and rank == 0 and not use_dist and cfg_id == args.config and not args.no_sizes:
            torch.cuda.synchronize()
            n_slots = 4
            pf = [torch.zeros((pad_h, w, 3), dtype=torch.float64, device=dev) for _ in range(n_slots)]
            pg = [torch.zeros((n_rows * 32, w, 3), dtype=torch.uint8, device=dev) for _ in range(n_slots)]
            p_plain = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth)
            p_plain.flags = params.flags & 2
            n_frames = max(steps, 30)
            for k in range(n_slots * 4 + n_frames):
                if k == n_slots * 4:
                    for b in range(n_slots):
                        ctx.frame_wait(b)
                    t1 = time.perf_counter()
                ctx.frame_submit(p_plain, pf[k % n_slots].data_ptr(), pg[k % n_slots].data_ptr(), None, k % n_slots)
            for b in range(n_slots):
                ctx.frame_wait(b)
            dt = time.perf_counter() - t1
            piped = {"value": w * h * n_frames / dt / 1e6, "unit": "Mpixels/s", "ms_per_step": dt / n_frames * 1e3,
                     "frames_in_flight": n_slots,
                     "what": "rm_frame_submit, %d frames, same outputs per frame as the metric" % n_frames}
            if args.check:
                piped["identical_to_single_stream_frame"] = bool(torch.equal(pf[0][:n_rows * 32], frame[:n_rows * 32])
                                                                 and torch.equal(pg[0], frame8[:n_rows * 32]))
            del pf, pg
        res = dict(cfg=cfg, w=w, h=h, depth=depth, n_rows=n_rows, c_rows=c_rows, cyclic=cyclic, host=host, piped=piped,
                   collective=("direct" if direct else mode["collective"]), frames_in_flight=n_buf,
                   elapsed=elapsed, kernel_ms=kernel_ms, px_launch=px_launch,
                   mpx=(w * h) * steps / elapsed / 1e6,
                   frame=frame, frame8=frame8, display=display)
        ctx.close()
        return res

    r = run_workload(args.config, args.steps, args.warmup)
    cfg, w, h, depth, n_rows, c_rows, cyclic = (r[k] for k in ("cfg", "w", "h", "depth", "n_rows", "c_rows", "cyclic"))
    elapsed, kernel_ms, px_launch = r["elapsed"], r["kernel_ms"], r["px_launch"]
    frame, frame8, display = r["frame"], r["frame8"], r["display"]
    # north star: "Mpixels/sec on synthetic 1080p/4K/8K frames": the larger frames of the
    # same scene, a short run each (same sharding and collective), reported beside the metric
    other = []
    if args.config == "C2" and not args.no_sizes:
        for cid in ("C2_4K", "C4"):
            o = run_workload(cid, max(5, min(args.steps, 20)), 3)
            ach = o["px_launch"] * BYTES_PER_PIXEL / (o["kernel_ms"] * 1e-3) / 1e9
            other.append({"workload": "%s: %s scene %dx%d, depth cap %d" % (cid, o["cfg"]["scene"], o["w"], o["h"], o["depth"]),
                          "value": o["mpx"], "unit": "Mpixels/s", "ms_per_step": o["elapsed"] / max(5, min(args.steps, 20)) * 1e3,
                          "roofline_frac": (ach / HBM_PEAK_GBPS) if world == 1 else None})
            o.clear()
            torch.cuda.empty_cache()

    if rank == 0:
        mpx = r["mpx"]
        achieved = px_launch * BYTES_PER_PIXEL / (kernel_ms * 1e-3) / 1e9
        traffic, valu = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("config") == args.config and pmc.get("n_gpus", 1) == world:
                    traffic = pmc.get("hbm_bytes_per_launch")
                    valu = {k: pmc.get(k) for k in ("valu_busy_frac", "valu_lanes_active_frac", "valu_wave_instructions")}
                    if valu["valu_wave_instructions"]:
                        # the bound that does apply: one 64-lane VALU instruction per SIMD per 4 cycles
                        # (1024 SIMDs, 2.4 GHz peak clock) against this run's kernel time
                        valu["valu_issue_frac_of_peak"] = (valu["valu_wave_instructions"] * 4.0
                                                           / (1024 * 2.4e9 * kernel_ms * 1e-3))
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/sec at 1920x1080, max-bounce=5" if args.config == "C2"
                      else "Mpixels/sec (%s)" % args.config,
            "value": mpx, "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s scene %dx%d, depth cap %d, fov 1.5, device-resident f64 RGB frame"
                                   % (args.config, cfg["scene"], w, h, depth),
                       "sharding": "%s of %d patch rows (32 px) per rank, %d rank(s)%s"
                                   % ("cyclic rows (r, r+N, ...)" if cyclic else "contiguous bands", c_rows, world, (", one in-place RCCL all-gather of the %s rows per frame; "
                                                      "f64 rows stay in each rank's HBM" % args.payload)
                                      if world > 1 else ""),
                       "collective": {"none": "none (one GPU)",
                                      "direct": "rm_frame_submit: ncclAllGather from the C library, %d frames in flight, "
                                                "each rendering on a stream of its own" % r["frames_in_flight"],
                                      "torch": "torch.distributed.all_gather_into_tensor (%s), %d frames in flight"
                                               % ("RCCL" if backend == "nccl" else backend, r["frames_in_flight"])}[r["collective"]]
                                     + (" [%s]" % mode["note"] if mode["note"] else ""),
                       "outputs": "f64 RGB frame [H][W][3] + u8 display frame (to_vec) per launch",
                       "numerics": "fast" if args.fast_fp else "strict (the reference's operations, one rounding each)",
                       "build": L.rm_build_info().decode()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "rmdev_fast::rm_render_static" if args.fast_fp else "rmdev_strict::rm_render_static", "kernel_ms": kernel_ms,
                         "bytes_per_launch": px_launch * BYTES_PER_PIXEL,
                         "fp64_valu": valu,
                         "note": "path is FP64-VALU bound by construction (SURVEY.md 8d); achieved = 24 B x pixels "
                                 "written / kernel time; traffic and fp64_valu are PMC figures of the same command "
                                 "(profiles/pmc_latest.json), not measured in this run"},
        }
        if args.check:
            O = G.load_oracle()
            ref = O.render(workloads.oracle_scene(O, cfg["scene"]), w, h, max_depth=depth)
            # what rank 0 holds after the last frame: the whole f64 frame (N = 1, or f64
            # payload), and / or the whole display frame (N = 1, or u8 payload)
            if world == 1 or args.payload == "f64":
                got = frame[:n_rows * 32].cpu().numpy()
                out["max_abs_delta_vs_oracle"] = float(np.abs(got - ref[:n_rows * 32]).max())
            if world == 1 or args.payload == "u8":
                u8 = (display if display is not None else frame8)[:n_rows * 32].cpu().numpy().reshape(-1)
                out["display_bytes_differing_from_oracle"] = int((u8 != O.to_vec(ref[:n_rows * 32].copy())).sum())
        if world == 1 and not args.no_cpu_baseline:
            O = G.load_oracle()
            out["cpu_baseline"] = cpu_baseline(O, workloads, cfg)
            out["speedup_vs_cpu_baseline"] = mpx / out["cpu_baseline"]["value"]
        if r["host"]:
            out["end_to_end_host"] = r["host"]
        if r["piped"]:
            out["four_frames_in_flight"] = r["piped"]
        if other:
            out["other_frames"] = other
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
