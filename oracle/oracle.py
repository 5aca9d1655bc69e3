"""ctypes view of oracle/liboracle.so -- CPU ORACLE (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package (rusty-marcher_amd/) never does.
The arithmetic lives in rm_oracle.c (which cites the reference lines); this
file only marshals arguments.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile rm_oracle.c -> liboracle.so with the committed Makefile."""
    src = os.path.join(_HERE, "rm_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "rm_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tup(self):
        return (self.x, self.y, self.z)


def v3(x, y=None, z=None):
    if y is None:
        x, y, z = x
    return Vec3(float(x), float(y), float(z))


class Reflectance(C.Structure):
    _fields_ = [("diffusion", C.c_double), ("diffuse_color", Vec3), ("specular", C.c_double),
                ("specular_exponent", C.c_double), ("is_glass_like", C.c_int),
                ("reflection", C.c_double), ("refractive_index", C.c_double)]


class Intersection(C.Structure):
    _fields_ = [("point", Vec3), ("normal", Vec3), ("reflectance", Reflectance)]


class Light(C.Structure):
    _fields_ = [("position", Vec3), ("color", Vec3), ("intensity", C.c_double)]


class Triangle(C.Structure):
    _fields_ = [("vertices", Vec3 * 3), ("normal", Vec3), ("center", Vec3)]


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int), ("center", Vec3), ("radius_square", C.c_double),
                ("vertices", C.POINTER(Vec3)), ("n_vertices", C.c_size_t),
                ("plane_normal", Vec3), ("plane_point", Vec3), ("reflectance", Reflectance),
                ("triangles", C.POINTER(Triangle)), ("reflectances", C.POINTER(Reflectance)),
                ("n_triangles", C.c_size_t)]


class Scene(C.Structure):
    _fields_ = [("lights", C.POINTER(Light)), ("n_lights", C.c_size_t),
                ("shapes", C.POINTER(Shape)), ("n_shapes", C.c_size_t), ("camera", Vec3)]


class Renderer(C.Structure):
    _fields_ = [("fov", C.c_double), ("half_fov", C.c_double), ("height", C.c_double),
                ("width", C.c_double), ("ratio", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("cast_ray", C.c_uint64), ("intersect", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("pow_calls", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    P = C.POINTER
    sig = {
        "orc_add": (Vec3, [Vec3, Vec3]), "orc_sub": (Vec3, [Vec3, Vec3]),
        "orc_mul": (Vec3, [Vec3, Vec3]), "orc_neg": (Vec3, [Vec3]),
        "orc_scaled": (Vec3, [Vec3, C.c_double]), "orc_cross": (Vec3, [Vec3, Vec3]),
        "orc_dot": (C.c_double, [Vec3, Vec3]), "orc_squared_norm": (C.c_double, [Vec3]),
        "orc_normalized": (Vec3, [Vec3]), "orc_normalized_l0": (Vec3, [Vec3]),
        "orc_reflectance_default": (Reflectance, []),
        "orc_triangle_create": (Triangle, [Vec3, Vec3, Vec3]),
        "orc_triangle_offset": (None, [P(Triangle), Vec3]),
        "orc_triangle_intersect": (C.c_int, [P(Triangle), Vec3, Vec3, P(Intersection)]),
        "orc_shape_intersect": (C.c_int, [P(Shape), Vec3, Vec3, P(Intersection)]),
        "orc_intersect_shape_set": (C.c_int, [Vec3, Vec3, P(Shape), C.c_size_t]),
        "orc_find_closest_intersect": (C.c_int, [Vec3, Vec3, P(Shape), C.c_size_t,
                                                 P(Intersection), P(C.c_uint8)]),
        "orc_reflect": (Vec3, [Vec3, Vec3]),
        "orc_reflect_ray": (C.c_int, [Vec3, P(Intersection), C.c_double, P(Vec3), P(Vec3)]),
        "orc_refract_ray": (C.c_int, [Vec3, P(Intersection), C.c_double, P(Vec3), P(Vec3)]),
        "orc_scene_new": (P(Scene), []), "orc_scene_free": (None, [P(Scene)]),
        "orc_scene_add_sphere": (None, [P(Scene), Vec3, C.c_double, Reflectance]),
        "orc_scene_add_polygon": (None, [P(Scene), P(Vec3), C.c_size_t, Reflectance]),
        "orc_scene_add_obj": (None, [P(Scene), P(C.c_double), C.c_size_t, Vec3]),
        "orc_scene_add_light": (None, [P(Scene), Vec3, Vec3, C.c_double]),
        "orc_scene_create_default": (P(Scene), []),
        "orc_create_renderer": (Renderer, [C.c_double, C.c_double, C.c_double]),
        "orc_backproject": (Vec3, [P(Renderer), C.c_size_t, C.c_size_t]),
        "orc_cast_ray": (Vec3, [Vec3, Vec3, P(Scene), Vec3, C.c_uint, C.c_uint]),
        "orc_render": (C.c_int, [P(Renderer), P(Scene), P(C.c_double), C.c_size_t, C.c_size_t,
                                 C.c_uint, C.c_int, P(C.c_double)]),
        "orc_render_band": (C.c_int, [P(Renderer), P(Scene), P(C.c_double), C.c_size_t,
                                      C.c_size_t, C.c_uint, C.c_int, C.c_size_t, C.c_size_t]),
        "orc_status_message": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint64, C.c_size_t,
                                         C.c_size_t]),
        "orc_normalize": (None, [P(C.c_double), C.c_size_t, C.c_size_t]),
        "orc_quantize": (C.c_uint8, [C.c_double]),
        "orc_to_vec": (None, [P(C.c_double), C.c_size_t, C.c_size_t, P(C.c_uint8)]),
        "orc_write_ppm": (C.c_int, [C.c_char_p, P(C.c_double), C.c_size_t, C.c_size_t]),
        "orc_get_stats": (None, [P(Stats)]),
        "orc_online_cpus": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleScene:
    """Owns an orc_scene*; mirrors scene.rs `Scene`."""

    def __init__(self, ptr=None):
        self.L = lib()
        self.ptr = ptr if ptr is not None else self.L.orc_scene_new()

    @classmethod
    def create_default(cls):
        return cls(lib().orc_scene_create_default())

    def __del__(self):
        try:
            if self.ptr:
                self.L.orc_scene_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def add_sphere(self, center, radius, refl):
        self.L.orc_scene_add_sphere(self.ptr, v3(center), float(radius), refl)

    def add_polygon(self, vertices, refl):
        arr = (Vec3 * len(vertices))(*[v3(p) for p in vertices])
        self.L.orc_scene_add_polygon(self.ptr, arr, len(vertices), refl)

    def add_obj(self, tri_xyz, offset=(0., 0., 0.)):
        a = np.ascontiguousarray(tri_xyz, dtype=np.float64).reshape(-1, 9)
        self.L.orc_scene_add_obj(self.ptr, _dptr(a), a.shape[0], v3(offset))

    def add_light(self, position, color, intensity):
        self.L.orc_scene_add_light(self.ptr, v3(position), v3(color), float(intensity))

    def set_camera(self, cam):
        self.ptr.contents.camera = v3(cam)

    @property
    def c(self):
        c = self.ptr.contents
        c._owner = self     # the struct borrows this scene's arrays
        return c


def reflectance(diffusion=1., diffuse_color=(1., 1., 1.), specular=1., specular_exponent=30.,
                is_glass_like=False, reflection=0.95, refractive_index=1.):
    return Reflectance(float(diffusion), v3(diffuse_color), float(specular),
                       float(specular_exponent), int(bool(is_glass_like)), float(reflection),
                       float(refractive_index))


def render(scene, width, height, fov=1.5, max_depth=3, n_threads=0, frame=None,
           band=None, return_ms=False):
    """renderer.rs:36-108 into a [H][W][3] float64 array (fresh zeros unless given)."""
    L = lib()
    r = L.orc_create_renderer(float(fov), float(height), float(width))
    if frame is None:
        frame = np.zeros((height, width, 3), dtype=np.float64)
    assert frame.flags["C_CONTIGUOUS"] and frame.dtype == np.float64
    ms = C.c_double(0.)
    if band is None:
        rc = L.orc_render(C.byref(r), scene.ptr, _dptr(frame), width, height, max_depth,
                          n_threads, C.byref(ms))
    else:
        rc = L.orc_render_band(C.byref(r), scene.ptr, _dptr(frame), width, height, max_depth,
                               n_threads, band[0], band[1])
    if rc != 0:
        raise RuntimeError("oracle: width % 32 != 0 (the reference panics here)")
    return (frame, ms.value) if return_ms else frame


def stats():
    s = Stats()
    lib().orc_get_stats(C.byref(s))
    return {"cast_ray": s.cast_ray, "intersect": s.intersect, "shadow_rays": s.shadow_rays,
            "pow_calls": s.pow_calls}


def normalize(frame):
    h, w, _ = frame.shape
    lib().orc_normalize(_dptr(frame), w, h)
    return frame


def to_vec(frame):
    h, w, _ = frame.shape
    out = np.empty(h * w * 3, dtype=np.uint8)
    lib().orc_to_vec(_dptr(frame), w, h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def status_message(ms, width, height):
    buf = C.create_string_buffer(256)
    lib().orc_status_message(buf, 256, int(ms), width, height)
    return buf.value.decode()
