#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the render hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one frame of the workload: BASELINE.json's configs[1] (demo scene,
1920x1080, depth cap 5) unless --config says otherwise.  The frame stays in HBM
(device-resident framebuffer); inputs (the scene) are uploaded before the timed region.
N > 1: the frame is sharded by 32-row patch bands, every rank renders
its band and the bands are gathered into rank 0's framebuffer over RCCL inside the
timed region -- total work is fixed, so scaling is "strong".

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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also compare the frame with the oracle")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    pkg = G.load_package()
    cfg = workloads.CONFIGS[args.config]
    w, h, depth = cfg["width"], cfg["height"], cfg["max_depth"]
    n_rows = h // 32
    band = workloads.patch_rows_for_rank(n_rows, rank, world)

    ctx = pkg.backend.Context(local_rank)
    scene = workloads.product_scene(pkg, cfg["scene"])
    ctx.upload(scene.flatten())
    params = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth, band)
    frame = torch.zeros((h, w, 3), dtype=torch.float64, device=dev)     # create_frame_buffer zero-fills
    # A dedicated stream: the kernel, the timing events and the RCCL ops are all ordered
    # on it (torch.cuda.Event only sees the stream it is recorded on).
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(stream)
    L = pkg.lib()
    p_ref = C.byref(params)
    frame_ptr = C.c_void_p(frame.data_ptr())
    stream_ptr = C.c_void_p(stream.cuda_stream)

    def gather():
        workloads.gather_bands(dist, frame, n_rows, rank, world)

    def step():
        st = L.rm_render_device(ctx.ptr, p_ref, frame_ptr, stream_ptr)
        if st != 0:
            raise RuntimeError(L.rm_last_error(ctx.ptr).decode())
        if world > 1:
            gather()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events on the launch stream bracket the timed region (one pair: an event per
    # launch would put two extra packets between consecutive kernels)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record(stream)
    for k in range(args.steps):
        step()
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # average launch duration of this rank's render kernel over the timed region (for
    # N > 1 the gather's stream time is inside the bracket too, so it is only reported
    # as the kernel's duration at N = 1)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps
    px_launch = (band[1] - band[0]) * 32 * w                            # pixels one launch writes

    if rank == 0:
        mpx = (w * h) * args.steps / elapsed / 1e6
        achieved = px_launch * BYTES_PER_PIXEL / (kernel_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                if pmc.get("config") == args.config and pmc.get("n_gpus", 1) == world:
                    traffic = pmc.get("hbm_bytes_per_launch")
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
                       "sharding": "row bands of 32-px patch rows, %d rank(s)%s"
                                   % (world, ", RCCL send/recv gather to rank 0 per frame" if world > 1 else ""),
                       "build": L.rm_build_info().decode()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "rm_render_kernel", "kernel_ms": kernel_ms,
                         "bytes_per_launch": px_launch * BYTES_PER_PIXEL,
                         "note": "path is FP64-VALU bound by construction (SURVEY.md 8d); "
                                 "achieved = 24 B x pixels written / kernel time"},
        }
        if args.check:
            O = G.load_oracle()
            ref = O.render(workloads.oracle_scene(O, cfg["scene"]), w, h, max_depth=depth)
            out["max_abs_delta_vs_oracle"] = float(np.abs(frame.cpu().numpy() - ref).max())
        if world == 1 and not args.no_cpu_baseline:
            O = G.load_oracle()
            out["cpu_baseline"] = cpu_baseline(O, workloads, cfg)
            out["speedup_vs_cpu_baseline"] = mpx / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
