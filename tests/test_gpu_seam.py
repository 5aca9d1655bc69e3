"""The seam into the reference's real FrameBuffer (-m gpu): framebuffer.rs:6-22 is
`buffer: Vec<Vec<Vec3f>>` -- one heap allocation per scan line, filled by the serial scatter of
renderer.rs:92-108 -- and the window consumes `fb.to_vec()` (main.rs:337-346).

  rm_render_rows     into rows of rows: bit for bit the flat frame, with and without the
                     black-patch elision, for whole frames, bands, strided bands, untouched rows
  rm_render_display  the f64 frame stays on the device, the display bytes come back: the oracle's
                     to_vec() byte for byte, into pageable and page-locked memory, one launch and
                     sub-bands
  rm_fetch_rows      the resident frame's f64 rows on demand
  rm_seam            the same from compiled code holding a std::vector per row (what bench.py runs)
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def make_rows(h, w, fill=0.):
    """One float64 array per scan line, each its own allocation (Vec<Vec<Vec3f>>)."""
    return [np.full((w, 3), fill, dtype=np.float64) for _ in range(h)]


def stack(rows):
    return np.stack(rows)


def upload(pkg, ctx, scene_name):
    scene = workloads.product_scene(pkg, scene_name)
    ctx.upload(scene.flatten())
    return scene


@pytest.fixture(params=["auto", "never", "always", "by-copy-engine"])
def packing(request, monkeypatch):
    """How the frame crosses the link (rm_hostio.inc):
    auto     frames of 2 MB and more leave the black patches behind, smaller ones are copied whole
    never    RM_HOST_PACK=0
    always   RM_HOST_PACK=1
    The patches that are sent are written into page-locked memory by a kernel; RM_HOST_PUSH=0 packs
    them on the device and moves them by copy engine instead."""
    for k in ("RM_HOST_PUSH", "RM_HOST_PACK"):
        monkeypatch.delenv(k, raising=False)
    mode = request.param
    if mode == "never":
        monkeypatch.setenv("RM_HOST_PACK", "0")
    elif mode != "auto":
        monkeypatch.setenv("RM_HOST_PACK", "1")
    if mode == "by-copy-engine":
        monkeypatch.setenv("RM_HOST_PUSH", "0")
    return mode


@pytest.mark.parametrize("cfg", [("demo", 1920, 1080, 5), ("cornell", 1920, 1080, 5), ("demo", 320, 250, 3),
                                 ("synthetic256", 512, 384, 6), ("demo", 64, 64, 2)])
def test_rows_of_rows_equal_the_flat_frame(pkg, O, packing, cfg):
    scene_name, w, h, depth = cfg
    ctx = pkg.backend.Context(0)                       # (reads RM_HOST_PACK when its host-io state is created)
    try:
        upload(pkg, ctx, scene_name)
        p = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth)
        flat = np.full((h, w, 3), -3., dtype=np.float64)
        ctx.render(p, flat)
        rows = make_rows(h, w, fill=-3.)
        t = ctx.render_rows(p, rows)
        got = stack(rows)
        assert np.array_equal(got, flat)
        n_rendered = h // 32 * 32
        assert np.all(got[n_rendered:] == -3.)         # renderer.rs:53: rows below the last patch row untouched
        st = ctx.hostio_stats()
        assert st["patches"] == (h // 32) * (w // 32) and 1 <= st["threads"] <= 64
        big = n_rendered * w * 24 >= (2 << 20)
        if packing == "never" or (packing == "auto" and not big):
            assert st["patches_sent"] == st["patches"] and st["bytes_copied"] == n_rendered * w * 24
        else:
            black = int(sum(1 for py in range(h // 32) for px in range(w // 32)
                            if not flat[py * 32:py * 32 + 32, px * 32:px * 32 + 32].view(np.uint64).any()))
            assert st["patches_sent"] == st["patches"] - black
            # + one flag per patch (a kernel writes the patches into page-locked memory), or + count and index (copy engine)
            assert st["bytes_copied"] - st["patches_sent"] * 24576 in (4 * st["patches"], 4 * (st["patches"] + 1))
            if scene_name in ("demo", "cornell") and big:
                assert st["patches_sent"] < 0.65 * st["patches"]  # sky / the dark around the box is not sent
        assert t.kernel_ms > 0. and t.total_ms >= t.kernel_ms
        ref = O.render(workloads.oracle_scene(O, scene_name), w, h, max_depth=depth)
        assert np.abs(got[:n_rendered] - ref[:n_rendered]).max() < 1e-9
    finally:
        ctx.close()


@pytest.mark.parametrize("band", [(3, 9), (0, 33, 4), (1, 33, 8), (5, 6), (32, 33), (7, 7)])
def test_rows_of_rows_for_bands(pkg, packing, band):
    """Owned patch rows only (a rank's share): the other rows are not touched and need no pointer."""
    w, h, depth = 1920, 1080, 5
    ctx = pkg.backend.Context(0)
    try:
        upload(pkg, ctx, "demo")
        p = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth, band)
        flat = np.full((h, w, 3), -7., dtype=np.float64)
        ctx.render(p, flat)
        stride = band[2] if len(band) > 2 else 1
        owned = set()
        for k in range(band[0], band[1], stride):
            owned.update(range(k * 32, k * 32 + 32))
        rows = [np.full((w, 3), -7., dtype=np.float64) if y in owned else None for y in range(h)]
        ctx.render_rows(p, rows)
        for y in range(h):
            if y in owned:
                assert np.array_equal(rows[y], flat[y]), "row %d" % y
            else:
                assert np.all(flat[y] == -7.)
        if owned:
            rows[min(owned)] = None
            with pytest.raises(pkg.BackendError) as e:
                ctx.render_rows(p, rows)
            assert e.value.status == pkg._lib.RM_ERR_INVALID_ARG
    finally:
        ctx.close()


def test_negative_zero_and_nan_are_not_black(pkg, monkeypatch):
    """The elision goes by BITS: a patch that holds -0.0 or NaN somewhere is sent as it is; only
    +0.0 everywhere is written by the host.  (A frame handed over through rm_buffer_write and
    fetched back through rm_fetch_rows: no render involved.)"""
    monkeypatch.setenv("RM_HOST_PACK", "1")
    w, h = 128, 96
    ctx = pkg.backend.Context(0)
    try:
        upload(pkg, ctx, "demo")
        p = pkg.backend.make_params(workloads.FOV, float(h), float(w), 0)        # depth 0: a fill, allocates the device frame
        ctx.render(p, None)
        dev, nbytes = C.c_void_p(), C.c_size_t(0)
        pkg._lib.check(ctx.L.rm_device_framebuffer(ctx.ptr, C.byref(dev), C.byref(nbytes)), ctx.ptr)
        assert nbytes.value == h * w * 24
        frame = np.zeros((h, w, 3), dtype=np.float64)
        frame[40, 70, 1] = -0.0
        frame[70, 100, 2] = np.nan
        frame[5, 5, 0] = 1e-320                                                 # a denormal
        ctx.buffer_write(dev, frame)
        rows = make_rows(h, w, fill=9.)
        ctx.fetch_rows(rows)
        got = stack(rows)
        assert got.tobytes() == frame.tobytes()
        assert ctx.hostio_stats()["patches_sent"] == 3
    finally:
        ctx.close()


def test_fetch_rows_refuses_a_framebuffer_of_another_size(pkg):
    """rm_fetch_rows writes frame_width * 3 doubles into each of frame_height rows of the caller's
    FrameBuffer -- of the frame it HOLDS.  A FrameBuffer of another size (a window resized since the render) must be
    refused, nothing read or written (ABI 5: the call takes the FrameBuffer's size)."""
    ctx = pkg.backend.Context(0)
    try:
        upload(pkg, ctx, "demo")
        w, h = 256, 160
        ctx.render(pkg.backend.make_params(workloads.FOV, float(h), float(w), 3), None)
        good = make_rows(h, w, fill=3.)
        ctx.fetch_rows(good)
        assert (stack(good)[:h // 32 * 32] != 3.).all()
        for hh, ww in ((h, w - 32), (h - 32, w), (h + 64, w), (h, w + 32)):
            rows = make_rows(hh, ww, fill=4.)
            with pytest.raises(pkg.BackendError) as e:
                ctx.fetch_rows(rows)
            assert e.value.status == pkg._lib.RM_ERR_INVALID_ARG and "resident frame" in str(e.value)
            assert (stack(rows) == 4.).all()
        ragged = make_rows(h, w, fill=4.)
        ragged[7] = np.full(3 * (w - 1), 4.)
        with pytest.raises(ValueError):
            ctx.fetch_rows(ragged)
    finally:
        ctx.close()


@pytest.mark.parametrize("subbands", ["1", "4", "7"])
@pytest.mark.parametrize("cfg", [("demo", 1920, 1080, 5), ("cornell", 800, 600, 3), ("synthetic256", 512, 384, 8), ("demo", 320, 250, 0)])
def test_display_bytes_with_a_device_resident_frame(pkg, O, monkeypatch, subbands, cfg):
    scene_name, w, h, depth = cfg
    monkeypatch.setenv("RM_DISPLAY_SUBBANDS", subbands)
    ctx = pkg.backend.Context(0)
    try:
        upload(pkg, ctx, scene_name)
        p = pkg.backend.make_params(workloads.FOV, float(h), float(w), depth)
        n = h // 32 * 32
        ref = O.render(workloads.oracle_scene(O, scene_name), w, h, max_depth=depth)
        want = O.to_vec(ref[:n].copy()).reshape(n, w, 3)
        out = np.full((h, w, 3), 77, dtype=np.uint8)
        t = ctx.render_display(p, out)
        assert int((out[:n] != want).sum()) <= 2          # a u8 truncation may flip within an ulp of k/255
        assert np.all(out[n:] == 77)
        assert t.total_ms > 0.
        # page-locked destination: written by the copy engine directly
        hp = C.c_void_p()
        pkg._lib.check(ctx.L.rm_host_alloc(ctx.ptr, h * w * 3, C.byref(hp)), ctx.ptr)
        pinned = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_uint8)), shape=(h * w * 3,)).reshape(h, w, 3)
        pinned[:] = 78
        ctx.render_display(p, pinned)
        assert np.array_equal(pinned[:n], out[:n]) and np.all(pinned[n:] == 78)
        del pinned
        ctx.L.rm_host_free(ctx.ptr, hp)
        # the f64 frame stayed on the device: rows on demand, and the device post-process
        rows = make_rows(h, w)
        ctx.fetch_rows(rows)
        assert np.abs(stack(rows)[:n] - ref[:n]).max() < 1e-9
        some = make_rows(h, w, fill=5.)
        ctx.fetch_rows(some, 2, 4)
        got = stack(some)
        assert np.array_equal(got[64:128], stack(rows)[64:128]) and np.all(got[:64] == 5.) and np.all(got[128:] == 5.)
        out8 = np.empty(h * w * 3, dtype=np.uint8)
        pkg._lib.check(ctx.L.rm_postprocess(ctx.ptr, None, w, h, 0, out8.ctypes.data_as(C.POINTER(C.c_uint8)), None), ctx.ptr)
        assert np.array_equal(out8.reshape(h, w, 3)[:n], out[:n])
    finally:
        ctx.close()


def test_display_of_a_strided_band(pkg, O):
    w, h, depth = 640, 352, 4
    ctx = pkg.backend.Context(0)
    try:
        upload(pkg, ctx, "demo")
        ref = O.render(workloads.oracle_scene(O, "demo"), w, h, max_depth=depth)
        want = O.to_vec(ref.copy()).reshape(h, w, 3)
        out = np.full((h, w, 3), 9, dtype=np.uint8)
        ctx.render_display(pkg.backend.make_params(workloads.FOV, float(h), float(w), depth, (1, 11, 3)), out)
        for k in range(11):
            part = out[k * 32:k * 32 + 32]
            if k in (1, 4, 7, 10):
                assert int((part != want[k * 32:k * 32 + 32]).sum()) <= 1
            else:
                assert np.all(part == 9)
    finally:
        ctx.close()


@pytest.mark.parametrize("args", [["--width", "1920", "--height", "1080", "--depth", "5", "--frames", "8"],
                                  ["--width", "800", "--height", "600", "--depth", "3", "--frames", "3"],
                                  ["--scene", "CORNELL", "--width", "1920", "--height", "1080", "--frames", "4"]])
def test_rm_seam_harness(entry, cornell_path, args):
    """Compiled host code holding a std::vector per scan line: rows of rows, flat, display only and
    the fetched rows all hold the same frame (exit code 0), and the line reports what moved."""
    exe = os.path.join(entry.PKG_DIR, "lib", "rm_seam")
    if not os.path.exists(exe):
        entry.build()
    args = [cornell_path if a == "CORNELL" else a for a in args]
    r = subprocess.run([exe] + args, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert d["rows_of_rows"]["identical_to_flat"] and d["display_only"]["identical_to_to_vec_of_rows"]
    assert d["display_only"]["into_page_locked"]["identical"] and d["fetch_rows"]["identical"]
    assert d["rows_of_rows"]["patches_sent"] <= d["rows_of_rows"]["patches"]
    assert d["rows_of_rows"]["ms_per_call"] > 0 and d["display_only"]["frames_per_s"] > 0


def test_rm_demo_device_frame_reproduces_out_ppm(entry, golden_ppm, tmp_path):
    """The UI flow with a device-resident FrameBuffer (render -> display bytes; save -> device
    normalize + quantize): the reference's committed engine/out.ppm again."""
    exe = os.path.join(entry.PKG_DIR, "lib", "rm_demo")
    out = tmp_path / "out.ppm"
    log = subprocess.check_output([exe, "--device-frame", "--out", str(out)]).decode()
    assert "of 450 patches sent" in log
    data = out.read_bytes()
    assert len(data) == len(golden_ppm)
    assert int((np.frombuffer(data, np.uint8) != np.frombuffer(golden_ppm, np.uint8)).sum()) <= 2
