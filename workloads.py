"""Benchmark / test workloads: the five configurations of BASELINE.json, each built
twice from ONE neutral recipe -- once through the product's host API (-> GPU) and once
through the oracle's (-> CPU check).  Harness code: used by bench.py, tests/ and
__graft_entry__.smoke(); not part of the product package.

  C1 demo scene 320x240,  depth 2   (CPU-runnable plumbing case)
  C2 demo scene 1920x1080, depth 5  (the headline metric's configuration)
  C3 cornell_box.obj 1920x1080, depth 5 (main.rs:261-327 recipe)
  C4 demo scene 7680x4320, depth 8
  C5 256-sphere synthetic 4096x4096, depth 10 (defined HERE; not in the reference)
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(GOLDEN, "cornell_box.obj")
SYNTH_FIXTURE = os.path.join(GOLDEN, "synthetic256.json")

FOV = 1.5   # main.rs:368

CONFIGS = {
    "C1": dict(scene="demo", width=320, height=240, max_depth=2),
    "C2": dict(scene="demo", width=1920, height=1080, max_depth=5),
    "C3": dict(scene="cornell", width=1920, height=1080, max_depth=5),
    "C4": dict(scene="demo", width=7680, height=4320, max_depth=8),
    "C2_4K": dict(scene="demo", width=3840, height=2160, max_depth=5),     # north star's 4K point (C2 at 4K)
    "C5": dict(scene="synthetic256", width=4096, height=4096, max_depth=10),
    "QHD": dict(scene="demo", width=2560, height=1440, max_depth=5),          # between 1080p and 4K
    "VGA": dict(scene="demo", width=640, height=480, max_depth=5),            # 4,800 tiles: a little more than one round of wave slots
    "SVGA": dict(scene="demo", width=800, height=608, max_depth=5),           # 7,600 tiles
    "C2_5K": dict(scene="demo", width=5120, height=2880, max_depth=5),
    # the reference's own operating points (depth cap 3, renderer.rs:262)
    "REF800": dict(scene="demo", width=800, height=600, max_depth=3),      # engine/out.ppm
    "UI": dict(scene="demo", width=1600, height=1280, max_depth=3),        # main.rs:240 framebuffer
    "SHOT": dict(scene="demo", width=1280, height=800, max_depth=3),       # README screenshot: 92 ms, 11.13 MP/s
}


# --------------------------------------------------------------------------
# A camera on the move.  The reference renders only after the camera has moved or the scene has
# changed (main.rs:74-78 `update_camera` -> `update_raytrace_image`): its six buttons offset the
# camera by +-5 on one axis (main.rs:119-170).  The walk below is that and nothing else: every
# position differs from its predecessor by one button press.
# --------------------------------------------------------------------------
CAMERA_MOVES = [(5., 0., 0.), (-5., 0., 0.), (0., 5., 0.), (0., -5., 0.), (0., 0., 5.), (0., 0., -5.)]


def camera_walk(n=240, seed=0xCA3E7A, box=((-10., 10.), (0., 10.), (0., 15.))):
    """-> n + (a few) camera positions, a closed walk from (0,0,0) by the reference's own offsets
    inside `box` (x, y, z ranges: the demo scene stays in front of the camera; above the floor);
    the last position is one press away from the first, so the list can be cycled."""
    u = _splitmix64(seed)
    pos, out = [0., 0., 0.], []
    while len(out) < n:
        m = CAMERA_MOVES[int(next(u) * 6) % 6]
        q = [pos[i] + m[i] for i in range(3)]
        if all(box[i][0] <= q[i] <= box[i][1] for i in range(3)):
            pos = q
            out.append(tuple(pos))
    while pos != [0., 0., 0.]:                           # ... and home again, a press at a time
        for i in range(3):
            if pos[i] != 0.:
                pos[i] -= 5. if pos[i] > 0. else -5.
                out.append(tuple(pos))
                break
    return out


# --------------------------------------------------------------------------
# C5: seeded synthetic scene.  SplitMix64, seed 0xC0FFEE.
# --------------------------------------------------------------------------
def _splitmix64(seed):
    state = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        yield (z >> 11) * (1.0 / 9007199254740992.0)     # uniform [0, 1)


def generate_synthetic(n_spheres=256, seed=0xC0FFEE):
    """Neutral description of the synthetic scene: list of sphere dicts.
    Draw order per sphere: cx, cy, cz, radius, r, g, b."""
    u = _splitmix64(seed)
    spheres = []
    for i in range(n_spheres):
        cx = -24. + 48. * next(u)
        cy = -10. + 24. * next(u)
        cz = -70. + 58. * next(u)
        radius = 0.4 + 1.2 * next(u)
        col = (next(u), next(u), next(u))
        glass = (i % 4 == 0)
        spheres.append(dict(center=(cx, cy, cz), radius=radius, color=col, glass=glass))
    return spheres


def load_synthetic():
    """The frozen fixture (tests/golden/synthetic256.json, written by
    tests/golden/make_synthetic.py); falls back to regenerating it."""
    if os.path.exists(SYNTH_FIXTURE):
        with open(SYNTH_FIXTURE) as f:
            return json.load(f)["spheres"]
    return generate_synthetic()


def _synthetic_material(sp):
    # every 4th sphere glass (ri 1.5, reflection 0.3, diffusion 0.2); the others
    # Reflectance::create_default with exponent 50
    if sp["glass"]:
        return dict(diffusion=0.2, diffuse_color=tuple(sp["color"]), specular=1., specular_exponent=50.,
                    is_glass_like=True, reflection=0.3, refractive_index=1.5)
    return dict(diffusion=1., diffuse_color=tuple(sp["color"]), specular=1., specular_exponent=50.,
                is_glass_like=False, reflection=0.95, refractive_index=1.)


FLOOR_QUAD = [(20., -3., -50.), (-20., -3., -50.), (-15., -6., -3.), (15., -6., -3.)]   # scene.rs:87-110
FLOOR_MATERIAL = dict(diffusion=1., diffuse_color=(0.3, 0.9, 0.9), specular=1., specular_exponent=100.,
                      is_glass_like=True, reflection=0.5, refractive_index=1.5)
DEMO_LIGHTS = [((0., 0., 0.), (1., 1., 1.), 1.), ((20., 20., 20.), (1., 0.5, 0.5), 0.8)]   # scene.rs:173-200


# --------------------------------------------------------------------------
# builders
# --------------------------------------------------------------------------
def product_scene(pkg, name, n_spheres=None):
    """-> rusty_marcher_amd.Scene for workload scene `name`."""
    if name == "demo":
        return pkg.Scene.create_default()
    if name == "cornell":
        return pkg.Scene.open_obj(CORNELL)
    if name == "synthetic256":
        s = pkg.Scene.new()
        spheres = load_synthetic()
        for sp in spheres[:n_spheres]:
            s.shapes.append(pkg.sphere.create(pkg.Vec3f(*sp["center"]), sp["radius"],
                                              pkg.Reflectance(**_synthetic_material(sp))))
        s.shapes.append(pkg.polygon.ConvexPolygon.create([pkg.Vec3f(*p) for p in FLOOR_QUAD],
                                                         pkg.Reflectance(**FLOOR_MATERIAL)))
        for pos, col, inten in DEMO_LIGHTS:
            s.lights.append(pkg.create_light(pkg.Vec3f(*pos), pkg.Vec3f(*col), inten))
        return s
    raise KeyError(name)


def oracle_scene(O, name, n_spheres=None):
    """-> oracle.OracleScene for workload scene `name` (independent code path)."""
    if name == "demo":
        return O.OracleScene.create_default()
    if name == "cornell":
        from oracle import obj_oracle
        s = O.OracleScene()
        for _, tris in obj_oracle.load_models(CORNELL):
            s.add_obj(np.array(tris, dtype=np.float64), (0., 0., -500.))       # main.rs:278-286
        for pos, col, inten in DEMO_LIGHTS:                                     # main.rs:293-315
            s.add_light(pos, col, inten)
        return s
    if name == "synthetic256":
        s = O.OracleScene()
        for sp in load_synthetic()[:n_spheres]:
            s.add_sphere(sp["center"], sp["radius"], O.reflectance(**_synthetic_material(sp)))
        s.add_polygon(FLOOR_QUAD, O.reflectance(**FLOOR_MATERIAL))
        for pos, col, inten in DEMO_LIGHTS:
            s.add_light(pos, col, inten)
        return s
    raise KeyError(name)


def patch_rows_for_rank(n_patch_rows, rank, world):
    """rank r owns patch rows [r*P/N, (r+1)*P/N)."""
    return (rank * n_patch_rows) // world, ((rank + 1) * n_patch_rows) // world


def gather_bands(dist, frame, n_patch_rows, rank, world, dst=0):
    """The one collective of a row-sharded frame: every peer sends its band
    of `frame` ([H][W][3] tensor, full-frame layout on every rank) straight into the same
    rows of rank `dst`'s frame.  Bands may be unequal or empty, hence grouped send/recv
    rather than an all-gather.  Returns the list of requests already waited on."""
    ops = []
    if rank == dst:
        for r in range(world):
            if r == dst:
                continue
            b, e = patch_rows_for_rank(n_patch_rows, r, world)
            if e > b:
                ops.append(dist.P2POp(dist.irecv, frame[b * 32:e * 32], r))
    else:
        b, e = patch_rows_for_rank(n_patch_rows, rank, world)
        if e > b:
            ops.append(dist.P2POp(dist.isend, frame[b * 32:e * 32], dst))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    for req in reqs:
        req.wait()          # on CUDA streams this orders the stream, it does not block the host
    return reqs


def equal_bands(n_patch_rows, world):
    """Equal-size contiguous bands for an all-gather: every rank owns c = ceil(P/N) patch
    rows' worth of frame rows, of which [begin, end) are real (trailing ranks may own
    fewer, or none: P = 33, N = 8 -> 5,5,5,5,5,5,3,0).  The makespan equals that of the
    balanced split (max band = c either way).  Returns (c, [(begin, end)] per rank)."""
    c = -(-n_patch_rows // world) if n_patch_rows else 0
    return c, [(min(r * c, n_patch_rows), min((r + 1) * c, n_patch_rows)) for r in range(world)]


def allgather_bands(dist, padded, rank, world):
    """The one collective per frame: `padded` is a [world*c*32, W, 3] tensor (any dtype) in
    which this rank has filled rows [rank*c*32, (rank+1)*c*32); an in-place all-gather
    (ncclAllGather on GPUs) completes it on every rank -- rank 0 is the consumer."""
    rows = padded.shape[0] // world
    dist.all_gather_into_tensor(padded, padded[rank * rows:(rank + 1) * rows])


def gather_chunks(dist, gathered, rank, world, dst=0):
    """The one exchange of a cyclically sharded frame with ONE consumer: `gathered` is the
    [world * c * 32, W, 3] buffer of rank-major chunks in which this rank has filled its own chunk;
    every peer sends that chunk straight to rank `dst` (grouped isend / irecv: on GPUs each peer
    over its own xGMI link, all at once).  Returns the requests (wait() orders the stream)."""
    rows = gathered.shape[0] // world
    ops = []
    if rank == dst:
        for r in range(world):
            if r != dst:
                ops.append(dist.P2POp(dist.irecv, gathered[r * rows:(r + 1) * rows], r))
    else:
        ops.append(dist.P2POp(dist.isend, gathered[rank * rows:(rank + 1) * rows], dst))
    return dist.batch_isend_irecv(ops) if ops else []


def cyclic_rows(n_patch_rows, world):
    """Cyclic ownership for load balance (sky rows are cheap, ground rows
    expensive): rank r owns patch rows r, r + N, r + 2N, ... -- rm_params band
    (begin = r, end = P, stride = N).  Returns (c, [rows of each rank]) with
    c = ceil(P / N) the chunk size in patch rows every rank's packed buffer is padded to."""
    c = -(-n_patch_rows // world) if n_patch_rows else 0
    return c, [list(range(r, n_patch_rows, world)) for r in range(world)]


def deinterleave_rows(gathered, world, out):
    """`gathered` is the all-gathered [world * c * 32, W, 3] buffer of packed chunks (rank
    major: chunk r holds rank r's rows r, r+N, ... in order); writes the frame in image
    order into `out` (same shape; rows past the last real patch row are padding)."""
    rows, w, ch = gathered.shape
    c = rows // (world * 32)
    out.view(c, world, 32, w, ch).copy_(gathered.view(world, c, 32, w, ch).permute(1, 0, 2, 3, 4))
    return out
