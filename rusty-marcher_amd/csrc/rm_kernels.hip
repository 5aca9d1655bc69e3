// rm_kernels.hip -- the render-kernel instantiations.  Compiled once per numeric flavour
// (-DRM_KERNEL_FAST=0|1) and kernel group (-DRM_KERNEL_GROUP=0..4) into objects of their own:
// 72 instantiations per flavour in one translation unit take minutes, ten units build in parallel.
//
//   group 0  small scenes (LDS copy for the per-lane gathers), plain walk        -- demo scene: C2, C4
//   group 1  small scenes with the bundle cull (with / without its edge test)
//   group 2  scenes read from global memory, flat walk + bundle cull            -- Cornell box: C3
//   group 3  hierarchy walk + bundle cull
//   group 4  hierarchy walk + bundle cull + frame-to-frame feedback             -- 256 spheres: C5
#include "rm_render_kernel.hpp"

#if RM_KERNEL_FAST
namespace flavour = rmdev_fast;
#else
namespace flavour = rmdev_strict;
#endif

using namespace rmdev;

namespace {

template <bool STAGED, bool BVH, bool CULL, bool EDGES, bool FB>
const void *pick(int stack, int pow_mode) {
#define RM_ROW(S)                                                                                                            \
    if (stack == S)                                                                                                          \
        return pow_mode == POW_INTEGER ? (const void *)flavour::rm_render_static<S, POW_INTEGER, 1, 1, STAGED, BVH, CULL, EDGES, FB> \
                                       : (const void *)flavour::rm_render_static<S, POW_GENERIC, 1, 1, STAGED, BVH, CULL, EDGES, FB>;
    RM_ROW(4) RM_ROW(8) RM_ROW(16) RM_ROW(32)
#undef RM_ROW
    return nullptr;
}

}  // namespace

#define RM_PICK_CAT2(a, b, c) rm_pick_kernel_##a##_g##b
#define RM_PICK_CAT(a, b) RM_PICK_CAT2(a, b, )
#if RM_KERNEL_FAST
#define RM_PICK_NAME RM_PICK_CAT(fast, RM_KERNEL_GROUP)
#else
#define RM_PICK_NAME RM_PICK_CAT(strict, RM_KERNEL_GROUP)
#endif

const void *RM_PICK_NAME(bool edges, int stack, int pow_mode) {
#if RM_KERNEL_GROUP == 0
    (void)edges;
    return pick<true, false, false, false, false>(stack, pow_mode);
#elif RM_KERNEL_GROUP == 1
    return edges ? pick<true, false, true, true, false>(stack, pow_mode) : pick<true, false, true, false, false>(stack, pow_mode);
#elif RM_KERNEL_GROUP == 2
    return edges ? pick<false, false, true, true, false>(stack, pow_mode) : pick<false, false, true, false, false>(stack, pow_mode);
#elif RM_KERNEL_GROUP == 3
    return edges ? pick<false, true, true, true, false>(stack, pow_mode) : pick<false, true, true, false, false>(stack, pow_mode);
#elif RM_KERNEL_GROUP == 4
    return edges ? pick<false, true, true, true, true>(stack, pow_mode) : pick<false, true, true, false, true>(stack, pow_mode);
#else
#error "RM_KERNEL_GROUP must be 0..4"
#endif
}
