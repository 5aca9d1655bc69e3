"""`sphere::create` of engine/src/sphere.rs:13-24.  The intersection test
(sphere.rs:27-61) runs on the GPU (csrc/rm_trace.hpp)."""
from . import _lib
from .geometry import as_vec3f


class Sphere:
    def __init__(self, center, radius, reflectance):
        self.center, self.radius, self.reflectance = as_vec3f(center), float(radius), reflectance.copy()

    def _append_to(self, handle):
        _lib.check(_lib.lib().rm_scene_add_sphere(handle, _lib.vec3(self.center), self.radius,
                                                  self.reflectance.to_c()))


def create(center, radius, reflectance):
    return Sphere(center, radius, reflectance)
