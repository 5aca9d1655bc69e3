"""`FrameBuffer` of engine/src/framebuffer.rs:6-82.  `buffer` is one [H][W][3]
float64 array (the reference keeps one Vec per row).  `normalize` / `to_vec` run the
device post-process kernels (rm_postprocess) on the frame."""
import ctypes as C

import numpy as np

from . import _lib, backend


class FrameBuffer:
    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.buffer = np.zeros((self.height, self.width, 3), dtype=np.float64)   # framebuffer.rs:12-22

    def _post(self, normalize, want_u8):
        # The frame goes to the device through the library's own synchronous copies
        # (rm_buffer_write returns when the bytes have landed), so rm_postprocess -- which
        # runs on the context's stream, not on torch's -- can never read a frame still in flight.
        ctx = backend.default_context()
        dev = ctx.buffer_alloc(self.buffer.nbytes)
        try:
            ctx.buffer_write(dev, self.buffer)
            out8 = np.empty(self.height * self.width * 3, dtype=np.uint8) if want_u8 else None
            mx = C.c_double(0.)
            _lib.check(_lib.lib().rm_postprocess(
                ctx.ptr, dev, self.width, self.height, 1 if normalize else 0,
                out8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_u8 else None, C.byref(mx)), ctx.ptr)
            if normalize:
                ctx.buffer_read(dev, self.buffer)
        finally:
            ctx.buffer_free(dev)
        return out8

    def normalize(self):
        """framebuffer.rs:58-77"""
        self._post(True, False)

    def to_vec(self):
        """framebuffer.rs:40-55"""
        return self._post(False, True)

    def write_ppm(self, filename):
        """framebuffer.rs:26-38"""
        with open(filename, "wb") as f:
            f.write(("P6\n%d %d\n255\n" % (self.width, self.height)).encode())
            f.write(self.to_vec().tobytes())
        return 0


def create_frame_buffer(width, height):
    return FrameBuffer(width, height)
