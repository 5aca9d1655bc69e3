"""Host-side scene construction of the product (csrc/rm_scene.cpp through the C ABI
and the Python mirror) against the oracle's independent transcription of the same
reference constructors: scene.rs:28-211, sphere.rs:13-24, polygon.rs:16-42,
lights.rs:10-16, obj.rs:94-138."""
import ctypes as C

import numpy as np
import pytest


def refl_tuple(r):
    return (r.diffusion, r.diffuse_color.x, r.diffuse_color.y, r.diffuse_color.z, r.specular,
            r.specular_exponent, int(bool(r.is_glass_like)), r.reflection, r.refractive_index)


def v(t):
    return (t.x, t.y, t.z)


def assert_scene_equals_oracle(d, o):
    """d: rm_scene_desc (product), o: orc_scene (oracle).  Bit-for-bit equality."""
    assert d.n_shapes == o.n_shapes
    assert d.n_lights == o.n_lights
    assert v(d.camera) == v(o.camera)
    for i in range(d.n_lights):
        a, b = d.lights[i], o.lights[i]
        assert (v(a.position), v(a.color), a.intensity) == (v(b.position), v(b.color), b.intensity)
    for i in range(d.n_shapes):
        ref, s = d.shapes[i], o.shapes[i]
        assert ref.kind == s.kind
        if ref.kind == 0:
            p = d.spheres[ref.first]
            assert (v(p.center), p.radius_square) == (v(s.center), s.radius_square)
            assert refl_tuple(p.reflectance) == refl_tuple(s.reflectance)
        elif ref.kind == 1:
            p = d.polygons[ref.first]
            assert p.n_vertices == s.n_vertices
            assert (v(p.plane_normal), v(p.plane_point)) == (v(s.plane_normal), v(s.plane_point))
            for k in range(p.n_vertices):
                assert v(d.polygon_vertices[p.first_vertex + k]) == v(s.vertices[k])
            assert refl_tuple(p.reflectance) == refl_tuple(s.reflectance)
        else:
            assert ref.count == s.n_triangles
            for t in range(ref.count):
                a, b = d.triangles[ref.first + t], s.triangles[t]
                assert [v(a.vertices[k]) for k in range(3)] == [v(b.vertices[k]) for k in range(3)]
                assert (v(a.normal), v(a.center)) == (v(b.normal), v(b.center))
                assert refl_tuple(a.reflectance) == refl_tuple(s.reflectances[t])


def test_default_scene_constants(pkg, O):
    d = pkg.Scene.create_default().flatten().desc()
    o = O.OracleScene.create_default()
    assert (d.n_shapes, d.n_spheres, d.n_polygons, d.n_polygon_vertices, d.n_lights) == (6, 4, 2, 7, 2)
    assert_scene_equals_oracle(d, o.c)
    # spot checks against scene.rs itself: order blue, green, red, white (scene.rs:201-208)
    assert [d.spheres[i].center.x for i in range(4)] == [-0.5, 6., -5., -10.]
    assert [d.spheres[i].radius_square for i in range(4)] == [4., 9., 16., 16.]
    # carried-over Reflectance fields (one struct mutated top to bottom)
    assert all(d.spheres[i].reflectance.specular_exponent == 100. for i in range(4))
    assert [d.spheres[i].reflectance.is_glass_like for i in range(4)] == [1, 0, 0, 0]
    assert d.spheres[1].reflectance.refractive_index == 1.5      # green: carried from blue
    assert d.polygons[1].reflectance.is_glass_like == 1          # floor
    assert v(d.lights[1].color) == (1., 0.5, 0.5)


def test_python_mirror_builds_the_same_scene(pkg, O):
    """Scene assembled through the mirror's constructors == library's create_default."""
    R = pkg.Reflectance
    V = pkg.Vec3f
    r = R.create_default()
    s = pkg.Scene.new()
    r.diffuse_color = V(0.8, 0., 0.); r.specular_exponent = 100.
    red = pkg.sphere.create(V(-5., 0., -16.), 4., r)
    r.diffuse_color = V(0.6, 0., 0.7)
    tri = pkg.polygon.ConvexPolygon.create([V(7., -4., -8.), V(15., 0., -9.), V(6., 3., -8.)], r)
    r.diffusion = 1.; r.specular = 1.; r.is_glass_like = True; r.refractive_index = 1.5
    r.reflection = 0.5; r.diffuse_color = V(0.3, 0.9, 0.9)
    quad = pkg.polygon.ConvexPolygon.create(
        [V(20., -3., -50.), V(-20., -3., -50.), V(-15., -6., -3.), V(15., -6., -3.)], r)
    r.specular = 1.; r.diffusion = 0.1; r.diffuse_color = V(0., 0., 0.2); r.reflection = 0.2
    blue = pkg.sphere.create(V(-0.5, -1.5, -5.), 2., r)
    r.diffusion = 1.; r.reflection = 1.; r.is_glass_like = False; r.specular = 0.8
    r.diffuse_color = V(0., 1., 0.)
    green = pkg.sphere.create(V(6., -0.5, -18.), 3., r)
    r.diffuse_color = V(0.9, 0.9, 0.9)
    white = pkg.sphere.create(V(-10., 6., -14.), 4., r)
    s.shapes = [blue, green, red, white, tri, quad]
    s.lights = [pkg.create_light(V(0., 0., 0.), V.ones(), 1.),
                pkg.create_light(V(20., 20., 20.), V(1., 0.5, 0.5), 0.8)]
    assert_scene_equals_oracle(s.flatten().desc(), O.OracleScene.create_default().c)


def test_mixed_scene_and_offsets(pkg, O):
    """Interleaved kinds + polygon/mesh offsets applied in sequence."""
    rng = np.random.default_rng(7)
    V = pkg.Vec3f
    ps, os_ = pkg.Scene.new(), O.OracleScene()
    r_p = pkg.Reflectance(0.7, (0.2, 0.4, 0.6), 0.9, 12.5, True, 0.3, 1.33)
    r_o = O.reflectance(0.7, (0.2, 0.4, 0.6), 0.9, 12.5, True, 0.3, 1.33)
    tri = rng.uniform(-5, 5, size=(5, 9))
    verts = [tuple(rng.uniform(-3, 3, 3)) for _ in range(5)]

    m = pkg.obj.Obj(tri); m.offset(V(0.1, 0.2, -30.)); ps.shapes.append(m)
    os_.add_obj(tri, (0.1, 0.2, -30.))
    ps.shapes.append(pkg.sphere.create(V(1., 2., -9.), 1.7, r_p)); os_.add_sphere((1., 2., -9.), 1.7, r_o)
    ps.shapes.append(pkg.polygon.ConvexPolygon.create([V(*p) for p in verts], r_p)); os_.add_polygon(verts, r_o)
    ps.lights.append(pkg.create_light(V(3., 4., 5.), V(0.2, 0.8, 0.4), 0.6)); os_.add_light((3., 4., 5.), (0.2, 0.8, 0.4), 0.6)
    ps.offset_camera(V(0., 0., 5.)); ps.offset_camera(V(-5., 0., 0.)); os_.set_camera((-5., 0., 5.))
    d = ps.flatten().desc()
    assert [d.shapes[i].kind for i in range(3)] == [2, 0, 1]
    assert_scene_equals_oracle(d, os_.c)


def test_polygon_needs_three_vertices(pkg):
    with pytest.raises(AssertionError):
        pkg.polygon.ConvexPolygon.create([pkg.Vec3f(), pkg.Vec3f()], pkg.Reflectance.create_default())
    L = pkg.lib()
    h = C.c_void_p(); L.rm_scene_new(C.byref(h))
    arr = (pkg._lib.rm_vec3 * 2)()
    assert L.rm_scene_add_polygon(h, arr, 2, pkg.Reflectance.create_default().to_c()) == pkg._lib.RM_ERR_INVALID_ARG
    L.rm_scene_free(h)


def test_sphere_has_no_offset(pkg):
    L = pkg.lib()
    h = C.c_void_p(); L.rm_scene_create_default(C.byref(h))
    assert L.rm_scene_offset_shape(h, 0, pkg._lib.rm_vec3(1, 1, 1)) == pkg._lib.RM_ERR_INVALID_ARG
    assert L.rm_scene_offset_shape(h, 99, pkg._lib.rm_vec3(1, 1, 1)) == pkg._lib.RM_ERR_INVALID_ARG
    assert L.rm_scene_offset_shape(h, 5, pkg._lib.rm_vec3(1, 1, 1)) == pkg._lib.RM_OK
    L.rm_scene_free(h)
