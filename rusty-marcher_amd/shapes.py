"""`Reflectance` of engine/src/shapes.rs:20-61."""
from . import _lib
from .geometry import Vec3f, as_vec3f


class Reflectance:
    def __init__(self, diffusion=1., diffuse_color=(1., 1., 1.), specular=1., specular_exponent=30.,
                 is_glass_like=False, reflection=0.95, refractive_index=1.):
        self.diffusion = diffusion
        self.diffuse_color = as_vec3f(diffuse_color)
        self.specular = specular
        self.specular_exponent = specular_exponent
        self.is_glass_like = is_glass_like
        self.reflection = reflection
        self.refractive_index = refractive_index

    @staticmethod
    def create_default():
        """shapes.rs:49-61, read back from the library so there is one source of truth."""
        r = _lib.rm_reflectance()
        _lib.lib().rm_reflectance_default(r)
        return Reflectance.from_c(r)

    @staticmethod
    def from_c(r):
        return Reflectance(r.diffusion, Vec3f(r.diffuse_color.x, r.diffuse_color.y, r.diffuse_color.z),
                           r.specular, r.specular_exponent, bool(r.is_glass_like), r.reflection,
                           r.refractive_index)

    def copy(self):
        return Reflectance(self.diffusion, Vec3f(*self.diffuse_color), self.specular,
                           self.specular_exponent, self.is_glass_like, self.reflection,
                           self.refractive_index)

    def to_c(self):
        return _lib.rm_reflectance(float(self.diffusion), _lib.vec3(self.diffuse_color),
                                   float(self.specular), float(self.specular_exponent),
                                   int(bool(self.is_glass_like)), 0, float(self.reflection),
                                   float(self.refractive_index))
