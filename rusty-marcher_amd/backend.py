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

    def uploads(self):
        """(calls of rm_scene_upload, copies it had to make)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _lib.check(self.L.rm_scene_uploads(self.ptr, C.byref(a), C.byref(b)), self.ptr)
        return a.value, b.value

    def set_camera(self, cam):
        _lib.check(self.L.rm_camera_update(self.ptr, _lib.vec3(cam)), self.ptr)

    def render(self, params, host_array=None):
        t = _lib.rm_timing()
        ptr = host_array.ctypes.data_as(C.POINTER(C.c_double)) if host_array is not None else None
        _lib.check(self.L.rm_render(self.ptr, C.byref(params), ptr, C.byref(t)), self.ptr)
        return t

    @staticmethod
    def row_pointers(rows):
        """double *const * for rm_render_rows / rm_fetch_rows: one pointer per row, each row a
        float64 array of its own (the reference's Vec<Vec<Vec3f>>, framebuffer.rs:6-22); None for
        rows the call must not touch."""
        arr = (C.POINTER(C.c_double) * len(rows))()
        for i, r in enumerate(rows):
            if r is not None:
                arr[i] = r.ctypes.data_as(C.POINTER(C.c_double))
        return arr

    def render_rows(self, params, rows):
        """Renderer::render into rows of rows (rm_render_rows)."""
        t = _lib.rm_timing()
        _lib.check(self.L.rm_render_rows(self.ptr, C.byref(params), self.row_pointers(rows), C.byref(t)), self.ptr)
        return t

    def render_display(self, params, host_u8):
        """Renderer::render with the f64 frame left on the device: only fb.to_vec() comes back."""
        t = _lib.rm_timing()
        _lib.check(self.L.rm_render_display(self.ptr, C.byref(params), host_u8.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            C.byref(t)), self.ptr)
        return t

    def fetch_rows(self, rows, patch_row_begin=0, patch_row_end=0, width=None):
        """`rows`: the FrameBuffer's rows, all of them (None for rows not to be touched): their count is its height,
        a row's length / 3 its width (`width` where every row is None)."""
        w = width if width is not None else next((r.size // 3 for r in rows if r is not None), 0)
        for r in rows:
            if r is not None and r.size != w * 3:
                raise ValueError("fetch_rows: a row of %d doubles in a FrameBuffer %d wide" % (r.size, w))
        _lib.check(self.L.rm_fetch_rows(self.ptr, self.row_pointers(rows), w, len(rows), patch_row_begin, patch_row_end), self.ptr)

    def tile_stats(self, stream=None):
        """(tiles of the last render launch, tiles the classification listed for rendering)"""
        a, b = C.c_uint32(0), C.c_uint32(0)
        _lib.check(self.L.rm_tile_stats(self.ptr, C.c_void_p(stream) if stream else None, C.byref(a), C.byref(b)), self.ptr)
        return a.value, b.value

    def launch_stats(self):
        """(workgroups of the last render launch, patches handed to its sky tail)"""
        a, b = C.c_uint32(0), C.c_uint32(0)
        _lib.check(self.L.rm_launch_stats(self.ptr, C.byref(a), C.byref(b)), self.ptr)
        return a.value, b.value

    def hostio_stats(self):
        """What the last host-bound call moved: bytes over the link, patches, patches sent, threads."""
        b, p, s, t = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        _lib.check(self.L.rm_hostio_stats(self.ptr, C.byref(b), C.byref(p), C.byref(s), C.byref(t)), self.ptr)
        return {"bytes_copied": b.value, "patches": p.value, "patches_sent": s.value, "threads": t.value}

    def render_device(self, params, device_ptr, stream=None):
        _lib.check(self.L.rm_render_device(self.ptr, C.byref(params), C.c_void_p(device_ptr),
                                           C.c_void_p(stream) if stream else None), self.ptr)

    def render_device_u8(self, params, device_ptr, device_ptr8, stream=None):
        _lib.check(self.L.rm_render_device_u8(self.ptr, C.byref(params), C.c_void_p(device_ptr),
                                              C.c_void_p(device_ptr8), C.c_void_p(stream) if stream else None),
                   self.ptr)

    # ---- multi-GPU frames (include/rusty_marcher_amd.h, rm_comm_* / rm_frame_*) ----
    @staticmethod
    def comm_unique_id():
        """RM_COMM_ID_BYTES from RCCL: rank 0 makes one and hands it to the other ranks."""
        buf = C.create_string_buffer(_lib.RM_COMM_ID_BYTES)
        _lib.check(_lib.lib().rm_comm_unique_id(buf), None)
        return buf.raw

    def comm_init(self, rank, world, unique_id=None):
        """unique_id None: the layout of `world` ranks without a transport (tests)."""
        if unique_id is not None and len(unique_id) != _lib.RM_COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _lib.RM_COMM_ID_BYTES)
        _lib.check(self.L.rm_comm_init(self.ptr, unique_id, rank, world), self.ptr)

    def comm_exchange(self, all_ranks):
        """False (default): chunks gathered at rank 0; True: all-gathered on every rank."""
        _lib.check(self.L.rm_comm_exchange(self.ptr, 1 if all_ranks else 0), self.ptr)

    def comm_destroy(self):
        self.L.rm_comm_destroy(self.ptr)

    def exchange_layout(self, params, world):
        rows, chunk = C.c_uint32(0), C.c_size_t(0)
        _lib.check(self.L.rm_exchange_layout(C.byref(params), world, C.byref(rows), C.byref(chunk)), self.ptr)
        return rows.value, chunk.value

    def frame_submit(self, params, device_ptr, gather_ptr, display_ptr=None, slot=0):
        _lib.check(self.L.rm_frame_submit(self.ptr, C.byref(params), C.c_void_p(device_ptr), C.c_void_p(gather_ptr),
                                          C.c_void_p(display_ptr) if display_ptr else None, slot), self.ptr)

    def frame_submit_f64(self, params, gather_ptr, frame_ptr=None, slot=0):
        """The f64 rows themselves: rendered packed into this rank's chunk of gather_ptr,
        all-gathered in place, written in image order to frame_ptr where given."""
        _lib.check(self.L.rm_frame_submit_f64(self.ptr, C.byref(params), C.c_void_p(gather_ptr),
                                              C.c_void_p(frame_ptr) if frame_ptr else None, slot), self.ptr)

    def frame_wait(self, slot=0, timeout_ms=None):
        """Raises BackendError(RM_ERR_TIMEOUT) instead of hanging when a peer never joins."""
        if timeout_ms is None:
            _lib.check(self.L.rm_frame_wait(self.ptr, slot), self.ptr)
        else:
            _lib.check(self.L.rm_frame_wait_for(self.ptr, slot, int(timeout_ms)), self.ptr)

    def frame_timing_enable(self, on=True):
        _lib.check(self.L.rm_frame_timing_enable(self.ptr, 1 if on else 0), self.ptr)

    def frame_timing(self, slot=0):
        t = _lib.rm_frame_times()
        _lib.check(self.L.rm_frame_timing(self.ptr, slot, C.byref(t)), self.ptr)
        return t

    def comm_info(self):
        """(rank, world, communicators in use) as the RCCL communicator itself reports them."""
        r, w, n = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(self.L.rm_comm_info(self.ptr, C.byref(r), C.byref(w), C.byref(n)), self.ptr)
        return r.value, w.value, n.value

    # ---- device buffers without torch (rm_buffer_*) ----
    def buffer_alloc(self, nbytes):
        p = C.c_void_p()
        _lib.check(self.L.rm_buffer_alloc(self.ptr, int(nbytes), C.byref(p)), self.ptr)
        return p

    def buffer_free(self, dptr):
        self.L.rm_buffer_free(self.ptr, dptr)

    def buffer_write(self, dptr, host_array):
        _lib.check(self.L.rm_buffer_write(self.ptr, dptr, host_array.ctypes.data_as(C.c_void_p), host_array.nbytes), self.ptr)

    def buffer_read(self, dptr, host_array):
        _lib.check(self.L.rm_buffer_read(self.ptr, dptr, host_array.ctypes.data_as(C.c_void_p), host_array.nbytes), self.ptr)

    def kernel_name(self, params):
        buf = C.create_string_buffer(256)
        _lib.check(self.L.rm_kernel_name(self.ptr, C.byref(params), buf, 256), self.ptr)
        return buf.value.decode()

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
