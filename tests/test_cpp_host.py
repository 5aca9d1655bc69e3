"""The C++ host mirror (rusty-marcher_amd/host/rusty_marcher.hpp) and its command-line
harness rm_demo: the reference's UI flow main.rs:119-123, 261-327, 329-357 (default scene /
open .obj -> render -> normalize -> write_ppm) over the C ABI, from compiled code."""
import os
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def rm_demo(entry):
    exe = os.path.join(entry.PKG_DIR, "lib", "rm_demo")
    if not os.path.exists(exe):
        entry.build()
    assert os.path.exists(exe)
    return exe


def test_dump_default_scene_matches_library(pkg, rm_demo):
    out = subprocess.check_output([rm_demo, "--dump-scene"]).decode().splitlines()
    d = pkg.Scene.create_default().flatten().desc()
    assert out[0].startswith("shapes 6 spheres 4 polygons 2 polygon_vertices 7 triangles 0 lights 2")
    for i in range(4):
        s = d.spheres[i]
        assert out[1 + i] == "sphere %d c %.17g %.17g %.17g r2 %.17g glass %d exp %.17g" % (
            i, s.center.x, s.center.y, s.center.z, s.radius_square, s.reflectance.is_glass_like,
            s.reflectance.specular_exponent)
    light = [l for l in out if l.startswith("light 1")][0]
    assert light == "light 1 p 20 20 20 c 1 0.5 0.5 i 0.80000000000000004"


def test_dump_obj_scene_and_camera(rm_demo, cornell_path):
    out = subprocess.check_output([rm_demo, "--obj", cornell_path, "--camera", "0,5,-5", "--dump-scene"]).decode()
    assert out.splitlines()[0] == ("shapes 8 spheres 0 polygons 0 polygon_vertices 0 triangles 36 lights 2 "
                                   "camera 0 5 -5")


def test_failure_is_a_panic_exit(rm_demo, tmp_path):
    import torch
    r = subprocess.run([rm_demo, "--obj", str(tmp_path / "broken.obj"), "--dump-scene"], capture_output=True)
    assert r.returncode == 0                      # obj::load -> None: empty scene, two lights (main.rs:276)
    assert b"shapes 0" in r.stdout
    (tmp_path / "nomtl.obj").write_text("mtllib nothere.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    r = subprocess.run([rm_demo, "--obj", str(tmp_path / "nomtl.obj"), "--dump-scene"], capture_output=True)
    assert r.returncode == 101 and b"WOOPS" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([rm_demo, "--width", "64", "--height", "64"], capture_output=True)
        assert r.returncode == 101 and b"no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--fast-fp"]])
def test_rm_demo_reproduces_out_ppm(rm_demo, golden_ppm, tmp_path, extra):
    out = tmp_path / "out.ppm"
    log = subprocess.check_output([rm_demo, "--out", str(out)] + extra).decode()
    assert "Rendering using patches of size 32, using 450 patches overall" in log
    assert "Dimensions mismatch" in log and "Scene rendered in " in log
    data = out.read_bytes()
    assert len(data) == len(golden_ppm) and data[:15] == golden_ppm[:15]
    n_diff = int((np.frombuffer(data, np.uint8) != np.frombuffer(golden_ppm, np.uint8)).sum())
    assert n_diff <= 2, "%d bytes differ from the reference's out.ppm" % n_diff


@pytest.mark.gpu
def test_rm_demo_width_panic(rm_demo, tmp_path):
    r = subprocess.run([rm_demo, "--width", "100", "--height", "64", "--out", str(tmp_path / "x.ppm")], capture_output=True)
    assert r.returncode == 101 and b"multiple of 32" in r.stderr


@pytest.mark.gpu
def test_repeated_render_calls_upload_the_scene_once(rm_demo, golden_ppm, tmp_path):
    """main.rs:331-333 hands the whole Scene to render() on every call; the library recognises a
    scene whose device image is already resident: three frames = three rm_scene_upload calls,
    ONE copy to the device -- and the third frame is still the reference's image.  At 1080p the
    call is one launch and the device -> host copy of the patches that are not black, scattered
    into the FrameBuffer's rows by a few host threads while the copy runs (its time is printed,
    not asserted: the box is shared)."""
    out = tmp_path / "out.ppm"
    log = subprocess.check_output([rm_demo, "--frames", "3", "--out", str(out)]).decode()
    assert "scene uploads: 3 calls, 1 copies to the device" in log
    data = out.read_bytes()
    assert int((np.frombuffer(data, np.uint8) != np.frombuffer(golden_ppm, np.uint8)).sum()) <= 2
    log = subprocess.check_output([rm_demo, "--frames", "6", "--width", "1920", "--height", "1080", "--depth", "5",
                                   "--out", str(tmp_path / "hd.ppm")]).decode()
    assert "scene uploads: 6 calls, 1 copies to the device" in log
    call_ms = float([ln for ln in log.splitlines() if ln.startswith("kernel ")][-1].split("call ")[1].split(" ms")[0])
    print("rm_demo 1080p: render() call %.3f ms" % call_ms)
    # (the call's host side shares the box with other tenants -- 0.62 to 1.15 ms were read minutes apart -- so the bound is a
    # generous one; the device's part is held tighter)
    kernel_ms = float([ln for ln in log.splitlines() if ln.startswith("kernel ")][-1].split("kernel ")[1].split(" ms")[0])
    assert call_ms < 20.0 and kernel_ms < 0.5, (call_ms, kernel_ms)
    sent, total = [int(x) for x in [ln for ln in log.splitlines() if ln.startswith("last frame:")][-1].split("link, ")[1].split(" patches")[0].split(" of ")]
    assert total == 1980 and 0 < sent < total


@pytest.mark.gpu
@pytest.mark.parametrize("rccl", [True, False])
def test_rm_walk_camera_walk_matches_oracle(entry, O, tmp_path, rccl):
    """rm_walk: the interactive loop from compiled code over the C ABI alone (no HIP, no RCCL
    on the host side) -- camera moves with the scene resident on the device, frames in
    flight over the slots, rank 0 writes every frame's display bytes.  World of one GPU
    (with and without a one-rank RCCL communicator, its id published through a file);
    every frame must be the oracle's to_vec() of the same camera position."""
    import workloads
    exe = os.path.join(entry.PKG_DIR, "lib", "rm_walk")
    if not os.path.exists(exe):
        entry.build()
    w, h, depth, frames = 320, 250, 4, 7
    cmd = [exe, "--rank", "0", "--world", "1", "--frames", str(frames), "--width", str(w), "--height", str(h),
           "--depth", str(depth), "--step", "0.5,0.25,-1", "--out", str(tmp_path / "walk"),
           "--id-file", str(tmp_path / "id.bin")] + ([] if rccl else ["--no-rccl"])
    log = subprocess.check_output(cmd).decode()
    assert "%d frames on 1 GPU(s)" % frames in log
    assert os.path.exists(tmp_path / "id.bin") == rccl and (not rccl or os.path.getsize(tmp_path / "id.bin") == 64 + 128)   # run id + ncclUniqueId
    rows = h // 32 * 32
    so = workloads.oracle_scene(O, "demo")
    for k in range(frames):
        data = (tmp_path / ("walk_%04d.ppm" % k)).read_bytes()
        header = b"P6\n%d %d\n255\n" % (w, rows)
        assert data.startswith(header) and len(data) == len(header) + rows * w * 3
        so.set_camera((0.5 * k, 0.25 * k, -1. * k))
        ref = O.to_vec(O.render(so, w, h, max_depth=depth)[:rows].copy())
        got = np.frombuffer(data[len(header):], np.uint8)
        assert int((got != ref).sum()) <= 2, "frame %d" % k      # a u8 truncation may flip within an ulp of k/255


def test_rm_walk_usage_errors(entry):
    exe = os.path.join(entry.PKG_DIR, "lib", "rm_walk")
    if not os.path.exists(exe):
        entry.build()
    assert subprocess.run([exe, "--rank", "2", "--world", "2"], capture_output=True).returncode == 2
    assert subprocess.run([exe, "--world", "2", "--rank", "0"], capture_output=True).returncode == 2    # no --id-file
    assert subprocess.run([exe, "--bogus"], capture_output=True).returncode == 2
    r = subprocess.run([exe, "--world", "1", "--rank", "0", "--no-rccl"], capture_output=True)          # no GPU here: a panic, not a fallback
    assert r.returncode in (0, 101)
