"""`Light` / `create_light` of engine/src/lights.rs:4-16.  The L-inf colour
normalisation happens inside the library (rm_scene_add_light)."""
from .geometry import as_vec3f


class Light:
    def __init__(self, position, color, intensity):
        self.position, self.color, self.intensity = as_vec3f(position), as_vec3f(color), float(intensity)


def create_light(position, color, intensity):
    return Light(position, color, intensity)
