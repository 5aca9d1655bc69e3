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

template <bool EDGES>
__global__ __launch_bounds__(64) void rm_classify_tiles_kernel(const double *__restrict__ scene_blob, KernelArgs a, ClassifyArgs o) {
    const uint32_t patch = blockIdx.x * 4u + ((threadIdx.x & 63u) >> 4);
    unsigned long long sig, own;
    const bool valid = patch < (a.n_tiles >> 4);
    (void)classify_patches<EDGES>(cls_view_of_blob(scene_blob, a.H), a, patch, valid, classify_ask(a, patch, valid), o.tile_mask, o.n_prims, 0u, sig, own);
}

const void *rmdev::rm_classify_kernel(bool edges) {
    return edges ? (const void *)rm_classify_tiles_kernel<true> : (const void *)rm_classify_tiles_kernel<false>;
}
