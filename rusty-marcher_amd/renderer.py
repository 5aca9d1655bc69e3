"""`Renderer` / `create_renderer` of engine/src/renderer.rs:17-126.  render() keeps
the reference's contract -- synchronous, fills the caller's FrameBuffer, returns the
status string -- and runs the patch loop on the MI355X through the C ABI."""
import ctypes as C
import time

from . import _lib, backend


class Renderer:
    def __init__(self, fov, height, width):
        p = backend.make_params(fov, height, width)
        self.fov, self.half_fov, self.height, self.width, self.ratio = (p.fov, p.half_fov, p.height,
                                                                        p.width, p.ratio)
        # The reference hard-codes the recursion cap (renderer.rs:262); exposed here.
        self.max_depth = 3
        self.device = 0
        self.last_timing = None

    def render(self, frame, scene):
        t0 = time.perf_counter()
        ctx = backend.default_context(self.device)
        if frame.height % 32 != 0 or frame.width % 32 != 0:
            print("Dimensions mismatch")                                   # renderer.rs:49-51
        n_patches = (frame.height // 32) * (frame.width // 32)
        print("Rendering using patches of size %d, using %d patches overall" % (32, n_patches))

        p = backend.make_params(self.fov, self.height, self.width, self.max_depth)
        p.frame_width, p.frame_height = frame.width, frame.height         # renderer.rs:53-54
        handle = scene.flatten()
        ctx.upload(handle)
        self.last_timing = ctx.render(p, frame.buffer)

        ms = int((time.perf_counter() - t0) * 1000.)                      # renderer.rs:111-112
        buf = C.create_string_buffer(256)
        _lib.lib().rm_format_status(buf, 256, ms, frame.width, frame.height)
        message = buf.value.decode()
        print(message)
        info = ctx.device_info()
        print("%d compute units used" % info["cus"])                     # renderer.rs:124 prints the thread count
        return message


def create_renderer(fov, height, width):
    """renderer.rs:25-33 -- argument order (fov, height, width)."""
    return Renderer(fov, height, width)
