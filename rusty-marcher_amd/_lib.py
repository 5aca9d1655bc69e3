"""ctypes binding of include/rusty_marcher_amd.h (the C ABI) -- loads the in-tree
librusty_marcher_amd.so and fails loudly when it is missing.  No fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RM_LIB_PATH") or os.path.join(_HERE, "lib", "librusty_marcher_amd.so")   # RM_LIB_PATH: dev knob for A/B builds

RM_OK = 0
RM_ERR_INVALID_ARG, RM_ERR_DIMENSIONS, RM_ERR_NO_DEVICE, RM_ERR_HIP = 1, 2, 3, 4
RM_ERR_NO_SCENE, RM_ERR_SCENE_LIMIT, RM_ERR_IO, RM_ERR_PARSE, RM_ERR_DEPTH = 5, 6, 7, 8, 9
RM_ERR_COMM, RM_ERR_TIMEOUT = 10, 11
RM_FLAG_FAST_FP, RM_FLAG_U8_COMPACT, RM_FLAG_F64_COMPACT = 2, 4, 8
RM_MAX_DEPTH = 32
RM_COMM_ID_BYTES, RM_MAX_FRAME_SLOTS = 128, 4
RM_PATCH_SIZE = 32
RM_SHAPE_SPHERE, RM_SHAPE_POLYGON, RM_SHAPE_MESH = 0, 1, 2

STATUS_NAMES = {0: "RM_OK", 1: "RM_ERR_INVALID_ARG", 2: "RM_ERR_DIMENSIONS", 3: "RM_ERR_NO_DEVICE",
                4: "RM_ERR_HIP", 5: "RM_ERR_NO_SCENE", 6: "RM_ERR_SCENE_LIMIT", 7: "RM_ERR_IO",
                8: "RM_ERR_PARSE", 9: "RM_ERR_DEPTH", 10: "RM_ERR_COMM", 11: "RM_ERR_TIMEOUT"}


class rm_vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class rm_reflectance(C.Structure):
    _fields_ = [("diffusion", C.c_double), ("diffuse_color", rm_vec3), ("specular", C.c_double),
                ("specular_exponent", C.c_double), ("is_glass_like", C.c_int32), ("_pad", C.c_int32),
                ("reflection", C.c_double), ("refractive_index", C.c_double)]


class rm_light(C.Structure):
    _fields_ = [("position", rm_vec3), ("color", rm_vec3), ("intensity", C.c_double)]


class rm_sphere(C.Structure):
    _fields_ = [("center", rm_vec3), ("radius_square", C.c_double), ("reflectance", rm_reflectance)]


class rm_polygon(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("n_vertices", C.c_uint32), ("plane_normal", rm_vec3),
                ("plane_point", rm_vec3), ("reflectance", rm_reflectance)]


class rm_triangle(C.Structure):
    _fields_ = [("vertices", rm_vec3 * 3), ("normal", rm_vec3), ("center", rm_vec3),
                ("reflectance", rm_reflectance)]


class rm_shape_ref(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("first", C.c_uint32), ("count", C.c_uint32), ("_pad", C.c_uint32)]


class rm_scene_desc(C.Structure):
    _fields_ = [("shapes", C.POINTER(rm_shape_ref)), ("n_shapes", C.c_uint32),
                ("spheres", C.POINTER(rm_sphere)), ("n_spheres", C.c_uint32),
                ("polygons", C.POINTER(rm_polygon)), ("n_polygons", C.c_uint32),
                ("polygon_vertices", C.POINTER(rm_vec3)), ("n_polygon_vertices", C.c_uint32),
                ("triangles", C.POINTER(rm_triangle)), ("n_triangles", C.c_uint32),
                ("lights", C.POINTER(rm_light)), ("n_lights", C.c_uint32),
                ("camera", rm_vec3)]


class rm_params(C.Structure):
    _fields_ = [("fov", C.c_double), ("half_fov", C.c_double), ("height", C.c_double),
                ("width", C.c_double), ("ratio", C.c_double),
                ("frame_width", C.c_uint32), ("frame_height", C.c_uint32),
                ("max_depth", C.c_uint32), ("patch_size", C.c_uint32),
                ("background", rm_vec3),
                ("patch_row_begin", C.c_uint32), ("patch_row_end", C.c_uint32),
                ("flags", C.c_uint32), ("patch_row_stride", C.c_uint32)]


class rm_timing(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("total_ms", C.c_double)]


class rm_frame_times(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("gather_ms", C.c_double), ("total_ms", C.c_double)]


_P = C.POINTER
_VP = C.c_void_p

# name -> (restype, argtypes); every symbol include/rusty_marcher_amd.h declares
SIGNATURES = {
    "rm_reflectance_default": (None, [_P(rm_reflectance)]),
    "rm_create_renderer": (None, [C.c_double, C.c_double, C.c_double, _P(rm_params)]),
    "rm_scene_new": (C.c_int, [_P(_VP)]),
    "rm_scene_free": (None, [_VP]),
    "rm_scene_create_default": (C.c_int, [_P(_VP)]),
    "rm_scene_add_sphere": (C.c_int, [_VP, rm_vec3, C.c_double, _P(rm_reflectance)]),
    "rm_scene_add_polygon": (C.c_int, [_VP, _P(rm_vec3), C.c_uint32, _P(rm_reflectance)]),
    "rm_scene_add_mesh": (C.c_int, [_VP, _P(C.c_double), C.c_uint32, rm_vec3]),
    "rm_scene_add_light": (C.c_int, [_VP, rm_vec3, rm_vec3, C.c_double]),
    "rm_scene_offset_shape": (C.c_int, [_VP, C.c_uint32, rm_vec3]),
    "rm_scene_set_camera": (C.c_int, [_VP, rm_vec3]),
    "rm_scene_offset_camera": (C.c_int, [_VP, rm_vec3]),
    "rm_scene_load_obj": (C.c_int, [_VP, C.c_char_p, rm_vec3, _P(C.c_uint32)]),
    "rm_scene_open_obj": (C.c_int, [C.c_char_p, _P(_VP)]),
    "rm_scene_get_desc": (C.c_int, [_VP, _P(rm_scene_desc)]),
    "rm_format_status": (C.c_int, [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32]),
    "rm_init": (C.c_int, [C.c_int, _P(_VP)]),
    "rm_destroy": (None, [_VP]),
    "rm_last_error": (C.c_char_p, [_VP]),
    "rm_scene_upload": (C.c_int, [_VP, _P(rm_scene_desc)]),
    "rm_scene_uploads": (C.c_int, [_VP, _P(C.c_uint64), _P(C.c_uint64)]),
    "rm_camera_update": (C.c_int, [_VP, rm_vec3]),
    "rm_render": (C.c_int, [_VP, _P(rm_params), _P(C.c_double), _P(rm_timing)]),
    "rm_render_rows": (C.c_int, [_VP, _P(rm_params), _P(_P(C.c_double)), _P(rm_timing)]),
    "rm_render_display": (C.c_int, [_VP, _P(rm_params), _P(C.c_uint8), _P(rm_timing)]),
    "rm_fetch_rows": (C.c_int, [_VP, _P(_P(C.c_double)), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rm_hostio_stats": (C.c_int, [_VP, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint64), _P(C.c_int)]),
    "rm_render_device": (C.c_int, [_VP, _P(rm_params), _VP, _VP]),
    "rm_render_device_u8": (C.c_int, [_VP, _P(rm_params), _VP, _VP, _VP]),
    "rm_tile_stats": (C.c_int, [_VP, _VP, _P(C.c_uint32), _P(C.c_uint32)]),
    "rm_launch_stats": (C.c_int, [_VP, _P(C.c_uint32), _P(C.c_uint32)]),
    "rm_device_framebuffer": (C.c_int, [_VP, _P(_VP), _P(C.c_size_t)]),
    "rm_postprocess": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint32, C.c_int, _P(C.c_uint8), _P(C.c_double)]),
    "rm_buffer_alloc": (C.c_int, [_VP, C.c_size_t, _P(_VP)]),
    "rm_buffer_free": (None, [_VP, _VP]),
    "rm_buffer_read": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "rm_buffer_write": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "rm_host_alloc": (C.c_int, [_VP, C.c_size_t, _P(_VP)]),
    "rm_host_free": (None, [_VP, _VP]),
    "rm_frame_submit_to_host": (C.c_int, [_VP, _P(rm_params), _VP, _VP, _VP, _VP, C.c_uint32]),
    "rm_comm_unique_id": (C.c_int, [_VP]),
    "rm_comm_init": (C.c_int, [_VP, _VP, C.c_int, C.c_int]),
    "rm_comm_destroy": (None, [_VP]),
    "rm_comm_exchange": (C.c_int, [_VP, C.c_int]),
    "rm_exchange_layout": (C.c_int, [_P(rm_params), C.c_int, _P(C.c_uint32), _P(C.c_size_t)]),
    "rm_frame_submit": (C.c_int, [_VP, _P(rm_params), _VP, _VP, _VP, C.c_uint32]),
    "rm_frame_wait": (C.c_int, [_VP, C.c_uint32]),
    "rm_frame_wait_for": (C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    "rm_frame_submit_f64": (C.c_int, [_VP, _P(rm_params), _VP, _VP, C.c_uint32]),
    "rm_frame_timing_enable": (C.c_int, [_VP, C.c_int]),
    "rm_frame_timing": (C.c_int, [_VP, C.c_uint32, _P(rm_frame_times)]),
    "rm_comm_info": (C.c_int, [_VP, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "rm_abi_version": (C.c_uint32, []),
    "rm_build_info": (C.c_char_p, []),
    "rm_device_info": (C.c_int, [_VP, C.c_char_p, C.c_size_t, _P(C.c_int), _P(C.c_size_t)]),
    "rm_kernel_name": (C.c_int, [_VP, _P(rm_params), C.c_char_p, C.c_size_t]),
}

_lib = None


class BackendError(RuntimeError):
    """Non-zero rm_status.  The reference's failure mode on this path is a panic."""

    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


def lib():
    """The loaded C-ABI library.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `make -C rusty-marcher_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: the PyTorch wheel carries its own libamdhip64.so.7.
        # Importing torch first makes the dynamic loader resolve this library's
        # NEEDED libamdhip64.so.7 to that already-loaded copy (same SONAME), so device
        # pointers and streams are shared with torch.  Loaded the other way round, the
        # process would hold two runtimes and the second finds no device.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, ctx=None):
    if status != RM_OK:
        msg = lib().rm_last_error(ctx)
        raise BackendError(status, msg.decode() if msg else "")


def vec3(v):
    if isinstance(v, rm_vec3):
        return v
    if hasattr(v, "x"):
        return rm_vec3(float(v.x), float(v.y), float(v.z))
    x, y, z = v
    return rm_vec3(float(x), float(y), float(z))
