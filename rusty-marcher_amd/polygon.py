"""`ConvexPolygon` of engine/src/polygon.rs:6-49.  Plane normal / plane point are
derived inside the library (rm_scene_add_polygon); the hit test (polygon.rs:60-98)
runs on the GPU."""
from . import _lib
from .geometry import as_vec3f


class ConvexPolygon:
    def __init__(self, vertices, reflectance):
        if len(vertices) <= 2:
            raise AssertionError("vertices.len() > 2 (polygon.rs:18)")
        self.vertices = [as_vec3f(v) for v in vertices]
        self.reflectance = reflectance.copy()
        self._offsets = []

    @staticmethod
    def create(vertices, reflectance):
        return ConvexPolygon(vertices, reflectance)

    def offset(self, off):
        """polygon.rs:44-49"""
        self._offsets.append(as_vec3f(off))

    def _append_to(self, handle, index):
        L = _lib.lib()
        arr = (_lib.rm_vec3 * len(self.vertices))(*[_lib.vec3(v) for v in self.vertices])
        _lib.check(L.rm_scene_add_polygon(handle, arr, len(self.vertices), self.reflectance.to_c()))
        for off in self._offsets:
            _lib.check(L.rm_scene_offset_shape(handle, index, _lib.vec3(off)))
