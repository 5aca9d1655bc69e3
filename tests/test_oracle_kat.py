"""The reference's own unit tests, restated against the oracle's
functions: geometry.rs:188-408, optics.rs:96-132, triangle.rs:88-138."""
import ctypes as C

import pytest


def V(O, x, y, z):
    return O.v3(x, y, z)


def test_scale(O):                                   # geometry.rs:194-215
    L = O.lib()
    r = L.orc_scaled(V(O, 1., 2., 3.), 1.36)
    assert (r.x, r.y, r.z) == (1.36, 2.72, 4.08)


def test_dot(O):                                     # geometry.rs:217-271
    L = O.lib()
    assert L.orc_dot(V(O, 1, 0, 0), V(O, 0, 1, 0)) == 0.
    assert L.orc_dot(V(O, 1, 2, 3), V(O, 1, 2, 3)) == 14.
    assert L.orc_dot(V(O, 1, 2, 3), V(O, -1, -2, -3)) == -14.
    assert L.orc_dot(V(O, 1, 0, 0), V(O, 1, 0, 0)) == 1.


def test_norm(O):                                    # geometry.rs:272-299
    L = O.lib()
    a = V(O, 42., 1., 0.)
    assert L.orc_squared_norm(a) == 1765.
    assert L.orc_squared_norm(L.orc_normalized(a)) == 1.
    assert L.orc_normalized_l0(a).x == 1.


def test_cross(O):                                   # geometry.rs:300-372
    L = O.lib()
    a, b = V(O, 1., 2., 3.), V(O, -2., 1., 0.)       # a.b == 0
    z = L.orc_cross(a, a)
    assert (z.x, z.y, z.z) == (0., 0., 0.)
    z = L.orc_cross(a, L.orc_neg(a))
    assert (z.x, z.y, z.z) == (0., 0., 0.)
    c = L.orc_cross(a, b)
    assert L.orc_dot(c, a) == 0.
    assert L.orc_squared_norm(c) == L.orc_squared_norm(a) * L.orc_squared_norm(b)


def test_squared_norm_add_sub(O):                    # geometry.rs:374-408
    L = O.lib()
    assert L.orc_squared_norm(V(O, 1., -2., 3.)) == 14.
    s = L.orc_add(V(O, 1., 2., 3.), V(O, 1., 1., 1.))
    assert (s.x, s.y, s.z) == (2., 3., 4.)
    d = L.orc_sub(V(O, 1., 2., 3.), V(O, 1., 1., 1.))
    assert (d.x, d.y, d.z) == (0., 1., 2.)


def test_reflection(O):                              # optics.rs:96-132
    L = O.lib()
    incident = V(O, 0.5, -0.5, 0.)
    isec = O.Intersection(V(O, 0, 0, 0), V(O, 0., 1., 0.), L.orc_reflectance_default())
    r = L.orc_reflect(incident, isec.normal)
    assert (r.x, r.y, r.z) == (0.5, 0.5, 0.)
    o, d = O.Vec3(), O.Vec3()
    assert L.orc_reflect_ray(incident, C.byref(isec), 1.5, C.byref(o), C.byref(d)) == 1
    assert (d.x, d.y, d.z) == (0.5, 0.5, 0.)


def test_triangle_intersect(O):                      # triangle.rs:88-138
    L = O.lib()
    v = [V(O, -1., 3., 2.2), V(O, -3., 0.2, 2.1), V(O, 0., 1., 2.)]
    tris = [L.orc_triangle_create(v[0], v[1], v[2]), L.orc_triangle_create(v[1], v[2], v[0]),
            L.orc_triangle_create(v[2], v[0], v[1])]
    orig = V(O, -1., 2., 5.3)
    d = L.orc_normalized(V(O, 0.1, -0.2, -3.))
    hits = []
    for t in tris:
        isec = O.Intersection()
        assert L.orc_triangle_intersect(C.byref(t), orig, d, C.byref(isec)) == 1
        hits.append(isec)
    for h in hits[1:]:
        assert L.orc_squared_norm(L.orc_sub(hits[0].point, h.point)) < 1e-3
        assert L.orc_squared_norm(L.orc_sub(hits[0].normal, h.normal)) < 1e-3
    assert abs(L.orc_squared_norm(tris[0].normal) - 1.) < 1e-3
    assert L.orc_dot(tris[0].normal, d) < 0.


def test_reflectance_default(O):                     # shapes.rs:49-61
    r = O.lib().orc_reflectance_default()
    assert (r.diffusion, r.specular, r.specular_exponent, r.is_glass_like, r.reflection,
            r.refractive_index) == (1., 1., 30., 0, 0.95, 1.)
    assert r.diffuse_color.tup() == (1., 1., 1.)


def test_quantize(O):                                # framebuffer.rs:80-82
    q = O.lib().orc_quantize
    assert [q(-1.), q(0.), q(0.5), q(1.), q(2.), q(float("nan"))] == [0, 0, 127, 255, 255, 0]
    assert q(0.999999) == 254
