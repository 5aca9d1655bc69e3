"""Pins the CPU oracle to the reference: its committed render engine/out.ppm
 must be reproduced byte for byte."""
import hashlib

import numpy as np
import pytest

OUT_PPM_SHA256 = "82d51afaaf4a644547728dde89478e484c245d2e3eb1e40da8b928ebd7584797"


@pytest.fixture(scope="module")
def demo_800x600(O):
    # main.rs:367-368 create_renderer(1.5, h, w); renderer.rs:262 depth cap 3; fresh fb
    return O.render(O.OracleScene.create_default(), 800, 600, fov=1.5, max_depth=3)


def test_fixture_is_the_reference_file(golden_ppm):
    assert hashlib.sha256(golden_ppm).hexdigest() == OUT_PPM_SHA256
    assert golden_ppm[:15] == b"P6\n800 600\n255\n"


def test_oracle_reproduces_out_ppm_byte_exact(O, demo_800x600, golden_ppm):
    frame = demo_800x600.copy()
    O.normalize(frame)                       # main.rs:355 fb.normalize()
    data = b"P6\n800 600\n255\n" + O.to_vec(frame).tobytes()   # framebuffer.rs:26-38
    assert len(data) == len(golden_ppm)
    diff = np.frombuffer(data, np.uint8) != np.frombuffer(golden_ppm, np.uint8)
    assert int(diff.sum()) == 0
    assert hashlib.sha256(data).hexdigest() == OUT_PPM_SHA256


def test_unrendered_rows_stay_zero(demo_800x600):
    # 600 % 32 = 24: rows 576..599 are never written (renderer.rs:53)
    assert np.all(demo_800x600[576:] == 0.)
    assert np.any(demo_800x600[575] != 0.)


def test_spot_values(demo_800x600):
    f = demo_800x600
    assert f.max() == 2.3621674603868565
    assert np.unravel_index(f.argmax(), f.shape) == (278, 312, 0)
    assert f.sum() == pytest.approx(495507.54788627784, rel=1e-14)
    spots = {(300, 400): (0.41865511882860496, 0.4639826782429074, 0.48371263064489023),
             (150, 150): (1.5480871711803066, 1.2660423842077726, 1.2660423842077726),
             (300, 700): (0.87735222556388748, 0.10000000000000228, 0.78014418751122339),
             (330, 510): (0.10000000000000001, 1.2186869026666192, 0.10000000000000001),
             (575, 799): (0.52113527242853053, 0.90982671598748077, 0.90982671598748077),
             (0, 0): (0., 0., 0.)}
    for (y, x), rgb in spots.items():
        assert tuple(f[y, x]) == rgb


def test_width_not_multiple_of_32_is_an_error(O):
    with pytest.raises(RuntimeError):
        O.render(O.OracleScene.create_default(), 100, 64)


def test_band_render_equals_full_render(O):
    s = O.OracleScene.create_default()
    full = O.render(s, 128, 96, max_depth=3)
    parts = np.zeros_like(full)
    for band in [(0, 1), (1, 3)]:
        O.render(s, 128, 96, max_depth=3, frame=parts, band=band)
    assert np.array_equal(full, parts)


def test_threads_do_not_change_the_image(O):
    s = O.OracleScene.create_default()
    a = O.render(s, 160, 96, n_threads=1)
    b = O.render(s, 160, 96, n_threads=5)
    assert np.array_equal(a, b)


def test_depth_caps_above_natural_depth_agree(O):
    # the demo ray tree dies out at depth 4
    s = O.OracleScene.create_default()
    d5 = O.render(s, 320, 224, max_depth=5)
    d8 = O.render(s, 320, 224, max_depth=8)
    d3 = O.render(s, 320, 224, max_depth=3)
    assert np.array_equal(d5, d8)
    assert not np.array_equal(d3, d5)


def test_status_message_format(O):
    # renderer.rs:116-121; README screenshot: 92 ms on 1280x800 -> "10 fps, 11.13 MP/s"
    assert O.status_message(92, 1280, 800) == "Scene rendered in 92 ms (10 fps, 11.13 MP/s)"
    assert O.status_message(0, 1920, 1080) == "Scene rendered in 0 ms (4294967295 fps, inf MP/s)"
