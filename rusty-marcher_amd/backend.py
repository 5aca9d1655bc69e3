"""rm_ctx ownership for the Python host mirror: one context per GPU, created on first
use.  Raises (never falls back) when no MI355X is visible."""
import ctypes as C

from . import _lib


class Context:
    def __init__(self, device=0):
        self.L = _lib.lib()
        p = C.c_void_p()
        _lib.check(self.L.rm_init(int(device), C.byref(p)))
        self.ptr = p
        self.device = int(device)
        self._uploaded = None   # keeps the SceneHandle of the uploaded scene alive

    def close(self):
        if self.ptr:
            self.L.rm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, desc_or_handle):
        desc = desc_or_handle.desc() if hasattr(desc_or_handle, "desc") else desc_or_handle
        _lib.check(self.L.rm_scene_upload(self.ptr, C.byref(desc)), self.ptr)
        self._uploaded = desc_or_handle

    def set_camera(self, cam):
        _lib.check(self.L.rm_camera_update(self.ptr, _lib.vec3(cam)), self.ptr)

    def render(self, params, host_array=None):
        t = _lib.rm_timing()
        ptr = host_array.ctypes.data_as(C.POINTER(C.c_double)) if host_array is not None else None
        _lib.check(self.L.rm_render(self.ptr, C.byref(params), ptr, C.byref(t)), self.ptr)
        return t

    def render_device(self, params, device_ptr, stream=None):
        _lib.check(self.L.rm_render_device(self.ptr, C.byref(params), C.c_void_p(device_ptr),
                                           C.c_void_p(stream) if stream else None), self.ptr)

    def render_device_u8(self, params, device_ptr, device_ptr8, stream=None):
        _lib.check(self.L.rm_render_device_u8(self.ptr, C.byref(params), C.c_void_p(device_ptr),
                                              C.c_void_p(device_ptr8), C.c_void_p(stream) if stream else None),
                   self.ptr)

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, lds = C.c_int(0), C.c_size_t(0)
        _lib.check(self.L.rm_device_info(self.ptr, name, 256, C.byref(cus), C.byref(lds)), self.ptr)
        return {"name": name.value.decode(), "cus": cus.value, "lds_per_block": lds.value}


_contexts = {}


def default_context(device=0):
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


def make_params(fov, height, width, max_depth=3, band=None):
    """rm_create_renderer + optional overrides."""
    p = _lib.rm_params()
    _lib.lib().rm_create_renderer(float(fov), float(height), float(width), C.byref(p))
    p.max_depth = int(max_depth)
    if band is not None:
        p.patch_row_begin, p.patch_row_end = int(band[0]), int(band[1])
        if len(band) > 2:
            p.patch_row_stride = int(band[2])
    return p
