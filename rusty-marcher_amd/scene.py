"""`Scene` of engine/src/scene.rs:9-211: {lights, shapes, camera}."""
import ctypes as C

from . import _lib
from .geometry import Vec3f, as_vec3f
from .lights import Light
from .obj import Obj
from .polygon import ConvexPolygon
from .shapes import Reflectance
from .sphere import Sphere
import numpy as np


class SceneHandle:
    """Owns an rm_scene* (the flat, uploadable form of a Scene)."""

    def __init__(self, ptr):
        self.ptr = ptr

    def desc(self):
        d = _lib.rm_scene_desc()
        _lib.check(_lib.lib().rm_scene_get_desc(self.ptr, C.byref(d)))
        d._owner = self          # the view borrows this handle's arrays
        return d

    def __del__(self):
        try:
            if self.ptr:
                _lib.lib().rm_scene_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def _shapes_from_desc(d):
    """Rebuild host-side shape objects from a flat description (used by create_default
    and open_obj so that `scene.shapes` can be inspected and edited like the Vec)."""
    shapes = []
    for i in range(d.n_shapes):
        ref = d.shapes[i]
        if ref.kind == _lib.RM_SHAPE_SPHERE:
            s = d.spheres[ref.first]
            # radius is only known squared; sqrt(r*r) == r for the exactly representable
            # radii in use, and Sphere keeps the exact square via _radius_square
            sp = Sphere(Vec3f(s.center.x, s.center.y, s.center.z), s.radius_square ** 0.5,
                        Reflectance.from_c(s.reflectance))
            shapes.append(sp)
        elif ref.kind == _lib.RM_SHAPE_POLYGON:
            p = d.polygons[ref.first]
            verts = [Vec3f(v.x, v.y, v.z) for v in
                     (d.polygon_vertices[p.first_vertex + k] for k in range(p.n_vertices))]
            shapes.append(ConvexPolygon(verts, Reflectance.from_c(p.reflectance)))
        else:
            tri = np.empty((ref.count, 9), dtype=np.float64)
            for t in range(ref.count):
                v = d.triangles[ref.first + t].vertices
                tri[t] = [v[0].x, v[0].y, v[0].z, v[1].x, v[1].y, v[1].z, v[2].x, v[2].y, v[2].z]
            shapes.append(Obj(tri))
    return shapes


class Scene:
    def __init__(self):
        self.lights = []
        self.shapes = []
        self.camera = Vec3f.zero()
        self._prebuilt = None   # SceneHandle built by the library (create_default / open_obj)

    @staticmethod
    def new():
        """scene.rs:16-23"""
        return Scene()

    def offset_camera(self, offset):
        """scene.rs:25-27"""
        self.camera = self.camera + as_vec3f(offset)

    @staticmethod
    def create_default():
        """scene.rs:28-211 -- the demo scene, transcribed once in csrc/rm_scene.cpp."""
        h = C.c_void_p()
        _lib.check(_lib.lib().rm_scene_create_default(C.byref(h)))
        return Scene._from_handle(SceneHandle(h))

    @staticmethod
    def open_obj(path):
        """main.rs:261-327: the scene the UI builds from an .obj file."""
        h = C.c_void_p()
        _lib.check(_lib.lib().rm_scene_open_obj(str(path).encode(), C.byref(h)))
        return Scene._from_handle(SceneHandle(h))

    @staticmethod
    def _from_handle(handle):
        s = Scene()
        d = handle.desc()
        s.camera = Vec3f(d.camera.x, d.camera.y, d.camera.z)
        s.lights = [Light(Vec3f(l.position.x, l.position.y, l.position.z),
                          Vec3f(l.color.x, l.color.y, l.color.z), l.intensity)
                    for l in (d.lights[i] for i in range(d.n_lights))]
        s._prebuilt = handle
        s._prebuilt_lights = list(s.lights)
        s.shapes = _PrebuiltShapes(handle)
        return s

    def flatten(self):
        """-> SceneHandle with the current lights / shapes / camera."""
        L = _lib.lib()
        untouched = (self._prebuilt is not None and isinstance(self.shapes, _PrebuiltShapes)
                     and not self.shapes.touched and len(self.lights) == len(self._prebuilt_lights)
                     and all(a is b for a, b in zip(self.lights, self._prebuilt_lights)))
        if untouched:
            _lib.check(L.rm_scene_set_camera(self._prebuilt.ptr, _lib.vec3(self.camera)))
            return self._prebuilt
        h = C.c_void_p()
        _lib.check(L.rm_scene_new(C.byref(h)))
        handle = SceneHandle(h)
        for i, shape in enumerate(list(self.shapes)):
            if isinstance(shape, Sphere):
                shape._append_to(h)
            else:
                shape._append_to(h, i)
        for l in self.lights:
            # lights built from a flat description are already normalised; L-inf
            # normalisation is idempotent (max component becomes exactly 1)
            _lib.check(L.rm_scene_add_light(h, _lib.vec3(l.position), _lib.vec3(l.color), l.intensity))
        _lib.check(L.rm_scene_set_camera(h, _lib.vec3(self.camera)))
        return handle


class _PrebuiltShapes(list):
    """`scene.shapes` of a library-built scene: materialises host objects lazily and
    remembers whether the caller edited the list (then flatten() rebuilds)."""

    def __init__(self, handle):
        super().__init__(_shapes_from_desc(handle.desc()))
        self.touched = False

    def _touch(self, *a, **k):
        self.touched = True

    def append(self, x):
        self.touched = True
        super().append(x)

    def push(self, x):
        self.append(x)

    def __setitem__(self, i, v):
        self.touched = True
        super().__setitem__(i, v)

    def __delitem__(self, i):
        self.touched = True
        super().__delitem__(i)
