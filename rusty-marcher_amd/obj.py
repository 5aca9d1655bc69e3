"""`Obj` / `load` of engine/src/obj.rs:13-151.  Parsing (tobj's role) and triangle
set-up run in the library's host half (csrc/rm_scene.cpp); the brute-force
closest-triangle loop (obj.rs:185-221) runs on the GPU."""
import ctypes as C

import numpy as np

from . import _lib
from .geometry import as_vec3f


class Obj:
    """One tobj model: n triangles (9 doubles each, f32 positions widened)."""

    def __init__(self, tri_xyz):
        self.tri_xyz = np.ascontiguousarray(tri_xyz, dtype=np.float64).reshape(-1, 9)
        self._offsets = []

    def offset(self, off):
        """obj.rs:24-29"""
        self._offsets.append(as_vec3f(off))

    def _append_to(self, handle, index):
        L = _lib.lib()
        ptr = self.tri_xyz.ctypes.data_as(C.POINTER(C.c_double))
        _lib.check(L.rm_scene_add_mesh(handle, ptr, self.tri_xyz.shape[0], _lib.rm_vec3(0., 0., 0.)))
        for off in self._offsets:
            _lib.check(L.rm_scene_offset_shape(handle, index, _lib.vec3(off)))


def load(path):
    """obj.rs:44-151: `Some(Vec<Obj>)`, or None when the file cannot be read
    (obj.rs:53-56).  A missing mtllib raises, as the reference panics (obj.rs:64)."""
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.rm_scene_new(C.byref(h)))
    try:
        n = C.c_uint32(0)
        st = L.rm_scene_load_obj(h, str(path).encode(), _lib.rm_vec3(0., 0., 0.), C.byref(n))
        if st == _lib.RM_ERR_IO and L.rm_last_error(None).decode().startswith("Could not load obj"):
            print("Could not load obj from %s" % path)
            return None
        _lib.check(st)
        d = _lib.rm_scene_desc()
        _lib.check(L.rm_scene_get_desc(h, C.byref(d)))
        objs = []
        for i in range(d.n_shapes):
            ref = d.shapes[i]
            tri = np.empty((ref.count, 9), dtype=np.float64)
            for t in range(ref.count):
                v = d.triangles[ref.first + t].vertices
                tri[t] = [v[0].x, v[0].y, v[0].z, v[1].x, v[1].y, v[1].z, v[2].x, v[2].y, v[2].z]
            objs.append(Obj(tri))
        return objs
    finally:
        L.rm_scene_free(h)
