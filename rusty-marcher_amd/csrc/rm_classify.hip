// rm_classify.hip -- which tiles can a primary ray hit anything in, and what?
//
// The reference computes a BoundingBox per shape and never consults it (shapes.rs:34-86): every
// pixel walks every shape (shapes.rs:110-143), and a primary ray that leaves the scene yields
// zero (renderer.rs:305).  In the reference's scenes that is most of the frame: 45 % of the demo
// scene's tiles, 86 % of the Cornell frame's -- and each such tile cost a whole wave: launch,
// backproject, a walk (or a cull step) that finds nothing, 1.7 KB of zero stores.
//
// This launch runs ahead of the render launch with A FEW LANES PER TILE (2,000 waves at 1080p instead
// of 31,680): they enclose the tile's 64 primary rays in a cone -- they leave one point, the
// camera, and their unnormalised directions (bp_x[x], bp_y[y], -1) fill a rectangle, so the cone
// around the rectangle's centre through its farthest corner holds them all -- and tests every
// primitive against it with the very tests of the ray-bundle cull (rm_trace.inc: bounding sphere
// against the solid cone, edge planes of triangles and quads; margins of 1e-7, NaN keeps) and, for
// planar primitives, the plane itself (a cone that looks away from it misses it).
// It leaves a mask per tile:
//   * no primitive left (mask 0): the tile's wave in the render launch stores the zeros and leaves
//     at once -- no scene copy, no ray, no walk;
//   * else (scenes of up to 64 primitives) the primitives left are what its primary rays walk.
// Conservative by construction -- a primitive some ray of the tile hits is never dropped -- so the
// frame is bit-identical with the classification off (RM_TILE_CLASSIFY=0;
// tests/test_gpu_parity.py::test_tile_classification_is_bitwise_invisible).
#define RM_KERNEL_FAST 0
#include "rm_render_kernel.hpp"

using namespace rmdev;
using namespace rmdev_strict;

// Two steps, coarse to fine, sixteen lanes to a 32x32 patch (a wave takes four patches):
//   1. the PATCH: the cone around its 1,024 primary rays against every primitive -- lane s of the group
//      takes the primitives s, s + 16, ... -- and the lanes combine what they kept;
//   2. its sixteen TILES, lane s the s-th: the tile's own cone against the primitives the patch kept.
// Most patches keep nothing or one or two primitives, so step 2 is short; a lane's chain of tests is what
// the launch lasts (495 waves at 1080p cannot fill the chip, only get done sooner).
__device__ __forceinline__ V3 unit(V3 a) {
    // (1/sqrt from the hardware estimate and two Newton steps: relative error ~1e-16 against margins of 1e-9)
    const double x = dot(a, a);
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return scaled(a, y);
}

// The cone around the rays whose unnormalised directions fill the rectangle [bx0, bx1] x [by0, by1] x {-1}; the
// lanes of a group that shares the rectangle each take a corner (`sub`).  A cone is convex: it holds the rectangle when it
// holds its corners.  The margins are far beyond the rounding of the kernel's own normalisation of the same
// table values.  False: the cone tests do not hold for it (half-angle towards 90 degrees -- a frame of a few
// tiles -- or a camera / table value that is not a number): everything is kept.
template <bool SHARED>
__device__ __forceinline__ bool cone_of(Bundle &b, const KernelArgs &a, double bx0, double bx1, double by0, double by1, uint32_t sub) {
    const V3 u = unit(mk(0.5 * (bx0 + bx1), 0.5 * (by0 + by1), -1.));
    double cm;
    if (SHARED) {                                                  // the lanes of a group share the rectangle: a corner each
        cm = dot(unit(mk((sub & 1u) ? bx1 : bx0, (sub & 2u) ? by1 : by0, -1.)), u);
        cm = __builtin_fmin(cm, __shfl_xor(cm, 1, 64));
        cm = __builtin_fmin(cm, __shfl_xor(cm, 2, 64));
    } else {
        cm = __builtin_fmin(__builtin_fmin(dot(unit(mk(bx0, by0, -1.)), u), dot(unit(mk(bx1, by0, -1.)), u)),
                            __builtin_fmin(dot(unit(mk(bx0, by1, -1.)), u), dot(unit(mk(bx1, by1, -1.)), u)));
    }
    cm -= 1e-9;
    b.ax = a.cam_x; b.ay = a.cam_y; b.az = a.cam_z;
    b.ux = u.x; b.uy = u.y; b.uz = u.z;
    b.cos2 = cm * cm;
    // (single-precision roots, rounded towards the wider cone, as bundle_cone does)
    b.sin_t = (double)__builtin_sqrtf((float)__builtin_fmax((1. - cm) * (1. + cm), 0.)) * (1. + 1e-6) + 1e-7;
    b.chord = (double)__builtin_sqrtf((float)__builtin_fmax(2. * (1. - cm), 0.)) * (1. + 1e-6) + 1e-7;
    b.rho = 0.;                                                   // every primary ray starts AT the camera
    b.two_sided = false;
    b.narrow = true;
    return cm > 0.5;
}

// One primitive against a cone of primary rays: true = keep.
template <bool EDGES>
__device__ __forceinline__ bool keeps(const double *__restrict__ scene_blob, const KernelArgs &a, const Bundle &b, uint32_t pid, bool in) {
    const uint32_t ns = a.H.n_spheres, np = a.H.n_polygons;
    const double2 *bp = reinterpret_cast<const double2 *>(scene_blob + a.H.off_bounds) + 2u * (in ? pid : 0u);
    const SphereCull sph = cull_sphere(bp, b);
    bool out = sph.out;
    const bool planar = in & (pid >= ns);
    if (EDGES && __any(planar)) {
        const double2 *pp = reinterpret_cast<const double2 *>(scene_blob + a.H.off_planar) + 8u * (planar ? pid - ns : 0u);
        out = out | cull_edges(pp, b, sph, planar & !out);
        // The plane itself: a hit needs dist = num / dotprod >= 0 (polygon.rs:71-76, triangle.rs:62-67), num =
        // (plane point - origin) . normal the same for every primary ray, dotprod = dir . normal within
        // chord |normal| of axis . normal.  Signs that differ for the whole cone: no ray of it hits (what bounds
        // a primitive whose lifted hull no sphere holds -- a wall along z).
        const uint32_t word = !planar ? 0u : pid < ns + np ? a.H.off_polygons + RM_POLYGON_WORDS * (pid - ns)
                                                           : a.H.off_triangles + RM_TRIANGLE_WORDS * (pid - ns - np);
        const double2 *rec = reinterpret_cast<const double2 *>(scene_blob + word);
        const double2 r0 = rec[0], r1 = rec[1], r2 = rec[2];      // nx ny | nz px | py pz
        const double nn = (double)__builtin_sqrtf((float)(r0.x * r0.x + r0.y * r0.y + r1.x * r1.x)) * (1. + 1e-6);   // >= |normal|
        const double num = (r1.y - b.ax) * r0.x + (r2.x - b.ay) * r0.y + (r2.y - b.az) * r1.x;
        const double un = b.ux * r0.x + b.uy * r0.y + b.uz * r1.x, spread = (b.chord + 1e-7) * nn;
        const double tiny = 1e-9 * nn * (__builtin_fabs(r1.y - b.ax) + __builtin_fabs(r2.x - b.ay) + __builtin_fabs(r2.y - b.az)) + 1e-290;
        const bool away = ((num > tiny) & (un + spread < 0.)) | ((num < -tiny) & (un - spread > 0.));
        out = out | (planar & away);
    }
    return in & !out;
}

template <bool EDGES>
__global__ __launch_bounds__(64) void rm_classify_tiles_kernel(const double *__restrict__ scene_blob, KernelArgs a, ClassifyArgs o) {
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 15u;
    const uint32_t n_patches = a.n_tiles >> 4;
    const uint32_t patch = blockIdx.x * 4u + (lane >> 4);
    const bool valid = patch < n_patches;
    const uint32_t tile = (valid ? patch : 0u) * 16u + sub;       // lane s: tile s of the patch
    uint32_t tx0, ty0, tyf, ty8;
    tile_origin(a, tile, tx0, ty0, tyf, ty8);
    const uint32_t px0 = tx0 & ~31u, py0 = ty0 & ~31u;            // the patch's corner
    const uint32_t n = o.n_prims;

    // ---- 1. the patch
    Bundle b;
    const bool patch_usable = cone_of<true>(b, a, a.bp_x[px0], a.bp_x[px0 + 31u], a.bp_y[py0], a.bp_y[py0 + 31u], sub);
    unsigned long long pm = 0ull;
    for (uint32_t base = 0; base < n; base += 16u) {               // wave-uniform trip count
        const uint32_t pid = base + sub;
        const bool in = pid < n;
        const bool keep = in & (keeps<EDGES>(scene_blob, a, b, pid, in) | !patch_usable);
        pm |= keep ? (1ull << (pid & 63u)) : 0ull;
    }
    for (uint32_t off = 1u; off < 16u; off <<= 1) {                // the group's lanes combine what they kept
        const uint32_t lo = (uint32_t)pm | (uint32_t)__shfl_xor((int)(uint32_t)pm, (int)off, 64);
        const uint32_t hi = (uint32_t)(pm >> 32) | (uint32_t)__shfl_xor((int)(uint32_t)(pm >> 32), (int)off, 64);
        pm = ((unsigned long long)hi << 32) | lo;
    }

    // ---- 2. its tiles, against what the patch kept (scenes of more than 64 primitives stop at the patch:
    // a mask cannot name them, it only says whether there is anything)
    unsigned long long mask = 0ull;
    if (n > 64u) {
        mask = pm != 0ull ? ~0ull : 0ull;
    } else if (__any(pm != 0ull)) {
        const bool tile_usable = cone_of<false>(b, a, a.bp_x[tx0], a.bp_x[tx0 + TILE_W - 1u], a.bp_y[ty0], a.bp_y[ty0 + TILE_H - 1u], sub);
        unsigned long long m = pm;
        while (__any(m != 0ull)) {
            const bool in = m != 0ull;
            const uint32_t pid = in ? (uint32_t)__builtin_ctzll(m) : 0u;
            m &= m - 1ull;
            const bool keep = in & (keeps<EDGES>(scene_blob, a, b, pid, in) | !tile_usable);
            mask |= keep ? (1ull << pid) : 0ull;
        }
    }
    if (valid) o.tile_mask[tile] = mask;
}

const void *rmdev::rm_classify_kernel(bool edges) {
    return edges ? (const void *)rm_classify_tiles_kernel<true> : (const void *)rm_classify_tiles_kernel<false>;
}
