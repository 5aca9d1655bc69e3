"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/rusty_marcher_amd.h declares, its structs have the declared layout, and
the render entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest


def _declared_functions(header_text):
    text = re.sub(r"/\*.*?\*/", "", header_text, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(pkg, entry):
    header = open(os.path.join(entry.ROOT, "include", "rusty_marcher_amd.h")).read()
    declared = _declared_functions(header)
    assert len(declared) >= 25
    L = pkg.lib()
    for name in declared:
        assert hasattr(L, name), "library does not export %s" % name
    # the ctypes binding covers exactly the declared set
    assert sorted(pkg._lib.SIGNATURES) == declared


def test_abi_version_and_build_info(pkg):
    L = pkg.lib()
    assert L.rm_abi_version() == 5            # 3: rm_render_rows, rm_render_display, rm_fetch_rows, rm_hostio_stats, rm_tile_stats; 4: rm_launch_stats; 5: rm_fetch_rows takes the FrameBuffer's size
    info = L.rm_build_info().decode()
    assert "gfx950" in info


def test_struct_layout(pkg, entry, tmp_path):
    """ctypes mirrors vs the header compiled as plain C (gcc): sizes must agree."""
    import subprocess
    B = pkg._lib
    names = ["rm_vec3", "rm_reflectance", "rm_light", "rm_sphere", "rm_polygon", "rm_triangle",
             "rm_shape_ref", "rm_scene_desc", "rm_params", "rm_timing", "rm_frame_times"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "rusty_marcher_amd.h"\nint main(void){' +
                   "".join('printf("%%zu\\n", sizeof(%s));' % n for n in names) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic",
                           "-I", os.path.join(entry.ROOT, "include"), str(src), "-o", str(exe)])
    c_sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert c_sizes == [C.sizeof(getattr(B, n)) for n in names]
    assert C.sizeof(B.rm_vec3) == 24 and C.sizeof(B.rm_reflectance) == 72
    # C-side view of the same structs: rm_create_renderer fills every field we read
    p = B.rm_params()
    pkg.lib().rm_create_renderer(1.5, 600., 800., C.byref(p))
    assert (p.fov, p.height, p.width) == (1.5, 600., 800.)
    assert p.ratio == 800. / 600.
    assert (p.frame_width, p.frame_height, p.max_depth, p.patch_size) == (800, 600, 3, 32)
    assert (p.background.x, p.background.y, p.background.z) == (0.1, 0.1, 0.1)
    assert (p.patch_row_begin, p.patch_row_end, p.flags) == (0, 0, 0)


def test_create_renderer_matches_oracle(pkg, O):
    # renderer.rs:25-33: half_fov = tan(fov/2)
    p = pkg.backend.make_params(1.5, 1080., 1920.)
    r = O.lib().orc_create_renderer(1.5, 1080., 1920.)
    assert (p.half_fov, p.ratio) == (r.half_fov, r.ratio)


def test_format_status_matches_reference_format(pkg, O):
    L = pkg.lib()
    for ms, w, h in [(92, 1280, 800), (0, 1920, 1080), (1, 32, 32), (1500, 7680, 4320), (7, 0, 0)]:
        buf = C.create_string_buffer(256)
        L.rm_format_status(buf, 256, ms, w, h)
        assert buf.value.decode() == O.status_message(ms, w, h)
    buf = C.create_string_buffer(256)
    L.rm_format_status(buf, 256, 92, 1280, 800)
    assert buf.value.decode() == "Scene rendered in 92 ms (10 fps, 11.13 MP/s)"


def test_render_without_gpu_fails_loudly(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = pkg.lib()
    ctx = C.c_void_p()
    st = L.rm_init(0, C.byref(ctx))
    assert st == pkg._lib.RM_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.rm_last_error(None)
    with pytest.raises(pkg.BackendError):
        pkg.create_renderer(1.5, 64., 64.).render(pkg.create_frame_buffer(64, 64), pkg.Scene.create_default())


def test_null_arguments_return_status_not_crash(pkg):
    L = pkg.lib()
    E = pkg._lib.RM_ERR_INVALID_ARG
    assert L.rm_scene_new(None) == E
    assert L.rm_scene_get_desc(None, None) == E
    assert L.rm_scene_add_light(None, pkg._lib.rm_vec3(), pkg._lib.rm_vec3(), 1.) == E
    assert L.rm_scene_upload(None, None) == E
    assert L.rm_render(None, None, None, None) == E
    assert L.rm_render_device(None, None, None, None) == E
    assert L.rm_postprocess(None, None, 1, 1, 0, None, None) == E
    assert L.rm_last_error(None) is not None


def test_exchange_layout_and_comm_entry_points_without_a_gpu(pkg):
    """The host-only parts of the multi-GPU frame API: the gather-buffer layout for N ranks
    (ceil(patch rows / N) rows of 32 * W * 3 bytes per rank) and the argument checks."""
    L = pkg.lib()
    E = pkg._lib.RM_ERR_INVALID_ARG
    rows, chunk = C.c_uint32(0), C.c_size_t(0)
    for (w, h, world, want_rows) in [(1920, 1080, 1, 33), (1920, 1080, 8, 5), (7680, 4320, 8, 17), (320, 240, 2, 4),
                                     (4096, 4096, 3, 43), (64, 20, 4, 0)]:
        p = pkg.backend.make_params(1.5, float(h), float(w), 5)
        assert L.rm_exchange_layout(C.byref(p), world, C.byref(rows), C.byref(chunk)) == 0
        assert rows.value == want_rows and chunk.value == want_rows * 32 * w * 3
    p = pkg.backend.make_params(1.5, 64., 100., 5)
    assert L.rm_exchange_layout(C.byref(p), 2, None, None) == pkg._lib.RM_ERR_DIMENSIONS
    assert L.rm_exchange_layout(C.byref(p), 0, None, None) == E
    assert L.rm_exchange_layout(None, 2, None, None) == E
    assert L.rm_comm_unique_id(None) == E
    assert L.rm_comm_init(None, None, 0, 1) == E
    assert L.rm_frame_submit(None, None, None, None, None, 0) == E
    assert L.rm_frame_wait(None, 0) == E
    L.rm_comm_destroy(None)                                     # a no-op, like rm_destroy(NULL)
    assert L.rm_buffer_alloc(None, 16, None) == E
    assert L.rm_buffer_read(None, None, None, 0) == E
    L.rm_buffer_free(None, None)
    assert L.rm_host_alloc(None, 16, None) == E
    L.rm_host_free(None, None)
    assert L.rm_frame_submit_to_host(None, None, None, None, None, None, 0) == E
    assert L.rm_frame_submit_f64(None, None, None, None, 0) == E
    assert L.rm_frame_wait_for(None, 0, 10) == E
    assert L.rm_frame_timing(None, 0, None) == E
    assert L.rm_frame_timing_enable(None, 1) == E
    assert L.rm_comm_info(None, None, None, None) == E
    assert L.rm_comm_exchange(None, 1) == E
    assert L.rm_buffer_write(None, None, None, 0) == E
    assert L.rm_render_rows(None, None, None, None) == E
    assert L.rm_render_display(None, None, None, None) == E
    assert L.rm_fetch_rows(None, None, 0, 0, 0, 0) == E
    assert L.rm_hostio_stats(None, None, None, None, None) == E
    assert L.rm_tile_stats(None, None, None, None) == E
    assert L.rm_launch_stats(None, None, None) == E
