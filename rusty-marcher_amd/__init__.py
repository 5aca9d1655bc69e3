"""MI355X-native backend for rusty-marcher's per-pixel render path.

Python mirror of the reference's host-side surface (engine/src/{scene,renderer,
framebuffer,sphere,polygon,lights,obj}.rs) over the C ABI declared in
include/rusty_marcher_amd.h.  Every render call goes through
lib/librusty_marcher_amd.so (hand-written HIP for gfx950); importing the package
without that library built, or rendering without an MI355X, raises.

The directory name carries a hyphen, so load it with __graft_entry__.load_package()
(importlib) under the module name `rusty_marcher_amd`.
"""
from . import _lib, backend, framebuffer, geometry, lights, obj, polygon, renderer, scene, shapes, sphere  # noqa: F401
from ._lib import BackendError, lib  # noqa: F401
from .framebuffer import FrameBuffer, create_frame_buffer  # noqa: F401
from .geometry import Vec3f  # noqa: F401
from .lights import Light, create_light  # noqa: F401
from .renderer import Renderer, create_renderer  # noqa: F401
from .scene import Scene  # noqa: F401
from .shapes import Reflectance  # noqa: F401

lib()   # fail at import time when the extension is missing
