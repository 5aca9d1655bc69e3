"""`Vec3f` of engine/src/geometry.rs:4-182 (f64 x 3) -- host-side value type only;
the arithmetic of the render path runs on the GPU."""


class Vec3f:
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0., y=0., z=0.):
        self.x, self.y, self.z = float(x), float(y), float(z)

    @staticmethod
    def zero():
        return Vec3f(0., 0., 0.)

    @staticmethod
    def ones():
        return Vec3f(1., 1., 1.)

    def __add__(self, o):
        return Vec3f(self.x + o.x, self.y + o.y, self.z + o.z)

    def __sub__(self, o):
        return Vec3f(self.x - o.x, self.y - o.y, self.z - o.z)

    def __neg__(self):
        return Vec3f(-self.x, -self.y, -self.z)

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __eq__(self, o):
        return tuple(self) == tuple(o)

    def __repr__(self):
        return "Vec3f(%r, %r, %r)" % (self.x, self.y, self.z)

    def scaled(self, s):
        return Vec3f(self.x * s, self.y * s, self.z * s)


def as_vec3f(v):
    return v if isinstance(v, Vec3f) else Vec3f(*v)
