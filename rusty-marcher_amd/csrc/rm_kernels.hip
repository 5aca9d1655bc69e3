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

template <bool STAGED, bool BVH, bool CULL, bool EDGES, bool ORDER, bool FB>
const void *pick(int stack, int pow_mode) {
#define RM_ROW(S)                                                                                                            \
    if (stack == S)                                                                                                          \
        return pow_mode == POW_INTEGER ? (const void *)flavour::rm_render_static<S, POW_INTEGER, 1, 1, STAGED, BVH, CULL, EDGES, ORDER, FB> \
                                       : (const void *)flavour::rm_render_static<S, POW_GENERIC, 1, 1, STAGED, BVH, CULL, EDGES, ORDER, FB>;
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

// (ORDER -- the dispatch order from the launch's own classification -- in every group but the one with the tile-level feedback)
template <bool STAGED, bool BVH, bool CULL, bool FB>
static const void *pick_eo(bool edges, int order, int stack, int pow_mode) {     // order: 0 off, 1 on
    if constexpr (FB) {
        return edges ? pick<STAGED, BVH, CULL, true, false, true>(stack, pow_mode) : pick<STAGED, BVH, CULL, false, false, true>(stack, pow_mode);
    } else {
        if constexpr (CULL) {
            if (edges) return order ? pick<STAGED, BVH, CULL, true, true, false>(stack, pow_mode) : pick<STAGED, BVH, CULL, true, false, false>(stack, pow_mode);
        }
        return order ? pick<STAGED, BVH, CULL, false, true, false>(stack, pow_mode) : pick<STAGED, BVH, CULL, false, false, false>(stack, pow_mode);
    }
}

const void *RM_PICK_NAME(bool edges, int order, int stack, int pow_mode) {
#if RM_KERNEL_GROUP == 0
    return pick_eo<true, false, false, false>(false, order, stack, pow_mode);
#elif RM_KERNEL_GROUP == 1
    return pick_eo<true, false, true, false>(edges, order, stack, pow_mode);
#elif RM_KERNEL_GROUP == 2
    return pick_eo<false, false, true, false>(edges, order, stack, pow_mode);
#elif RM_KERNEL_GROUP == 3
    return pick_eo<false, true, true, false>(edges, order, stack, pow_mode);
#elif RM_KERNEL_GROUP == 4
    return pick_eo<false, true, true, true>(edges, 0, stack, pow_mode);
#else
#error "RM_KERNEL_GROUP must be 0..4"
#endif
}
