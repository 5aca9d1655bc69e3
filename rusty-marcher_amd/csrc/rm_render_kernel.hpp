// rm_render_kernel.hpp -- the render kernels (gfx950, FP64).
//
// Replaces Renderer::render's Rayon patch loop (renderer.rs:63-89), the serial
// scatter (renderer.rs:92-108) and cast_ray's recursion (renderer.rs:254-309).
//
// Unit of work: a 64-pixel TILE (16x4) rendered by one 64-lane wave, one lane per
// pixel (the 64 rays of a wave stay spatially coherent: same primitives hit,
// same branches; measured VALU lane utilisation 87-94 %).  Sixteen tiles make one of
// the reference's 32x32 patches (renderer.rs:47); tile ids are patch-major so a
// run of 16 consecutive ids is exactly one reference patch.
//
// cast_ray's recursion becomes a per-lane depth-first walk of the ray tree with
// an explicit stack: radiance is linear in the children (renderer.rs:219,249),
// so each ray carries the product of the reflection factors above it.
#ifndef RM_RENDER_KERNEL_HPP
#define RM_RENDER_KERNEL_HPP

#include <hip/hip_runtime.h>

#include "rm_kernel_args.hpp"

namespace rmdev {

__device__ __forceinline__ uint32_t fb_bucket(uint32_t ticks) {
    const uint32_t t = ticks < 4u ? 4u : ticks;
    const uint32_t e = 31u - (uint32_t)__builtin_clz(t);            // t in [2^e, 2^(e+1)), e >= 2
    const uint32_t b = 4u * (e - 2u) + ((t >> (e - 2u)) & 3u);
    return b < RM_FB_BUCKETS ? b : RM_FB_BUCKETS - 1u;
}
__device__ __forceinline__ uint32_t fb_edge(uint32_t b) { return (4u + (b & 3u)) << (b >> 2); }   // lower edge of bucket b

// Copies a wave-uniform value into a scalar register of its own (see the kernel's header copy).
// (a real move: an empty asm with a tied operand is coalesced back into the tuple)
__device__ __forceinline__ double own_sgpr(double v) { double r; asm("s_mov_b64 %0, %1" : "=s"(r) : "s"(v)); return r; }
__device__ __forceinline__ uint32_t own_sgpr(uint32_t v) { uint32_t r; asm("s_mov_b32 %0, %1" : "=s"(r) : "s"(v)); return r; }

// A value every lane holds alike (read from one address, or handed back by a function that is not inlined), moved to scalar
// registers: what is computed from it stays on the scalar unit, loops over it are scalar loops.
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    return ((unsigned long long)uniform_u32((uint32_t)(v >> 32)) << 32) | uniform_u32((uint32_t)v);
}

// The first round, the index and its inverse a launch came with (KernelArgs::static_list / dyn_index / dyn_inv) are there only if the
// launch that was to write them laid out an order at all (it may have fallen back on the order of last resort): every wave of
// this launch reads the same word of a launch that is over, so all agree.
__device__ __forceinline__ bool lists_ok(const KernelArgs &a) {
    return a.static_list && a.lists_done && uniform_u32(*a.lists_done) == a.lists_tag;
}

// The kernel's argument segment through a pointer the compiler cannot see through: what is read from it is read then and
// there (scalar loads from the constant address space), not carried in registers from wherever it was first needed.
__device__ __forceinline__ const char *reread_kernargs(const char *kargs) {
    unsigned long long p = uniform_u64((unsigned long long)kargs);   // (inside a function that is not inlined the pointer arrives in vector registers)
    asm volatile("s_mov_b64 %0, %1" : "=s"(p) : "s"(p));
    return (const char *)reinterpret_cast<const __attribute__((address_space(4))) char *>(p);
}

extern __shared__ double rm_lds[];

// Every workgroup keeps its own copy of the scene in LDS; all later reads are
// wave-uniform broadcasts.
__device__ __forceinline__ void stage_scene(const double *__restrict__ scene_blob, const rm_dev_header &H) {
    const double2 *src = reinterpret_cast<const double2 *>(scene_blob);
    double2 *dst = reinterpret_cast<double2 *>(rm_lds);
    const uint32_t n2 = H.total_words / 2;
    for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// The same copy in two halves: asked for when the wave starts, put into LDS once the wave knows it renders a tile at all -- the
// loads travel while the wave finds its place in the order and its tile's classification word (measured: a wave of the demo
// frame knew its tile after 1.07 us, its word 0.76 us later, and only then spent 0.72 us on the copy).  Scenes of the staged
// kernels are at most RM_LDS_SCENE_LIMIT_WORDS long (rm_device.hip): four 16-byte pieces a lane.
struct SceneAsked { double2 v[RM_LDS_SCENE_LIMIT_WORDS / 128u]; };
__device__ __forceinline__ SceneAsked stage_scene_ask(const double *__restrict__ scene_blob, const rm_dev_header &H) {
    const double2 *src = reinterpret_cast<const double2 *>(scene_blob);
    const uint32_t n2 = H.total_words / 2, lane = threadIdx.x & 63u;
    SceneAsked p;
#pragma unroll
    for (uint32_t k = 0; k < RM_LDS_SCENE_LIMIT_WORDS / 128u; k++) {
        p.v[k] = double2{0., 0.};
        if (lane + 64u * k < n2) p.v[k] = src[lane + 64u * k];
    }
    return p;
}
__device__ __forceinline__ void stage_scene_put(const SceneAsked &p, const rm_dev_header &H) {
    double2 *dst = reinterpret_cast<double2 *>(rm_lds);
    const uint32_t n2 = H.total_words / 2, lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t k = 0; k < RM_LDS_SCENE_LIMIT_WORDS / 128u; k++)
        if (lane + 64u * k < n2) dst[lane + 64u * k] = p.v[k];
    __syncthreads();
}

// tile id -> pixel origin.  Patch-major: patch = id / 16 walks the band row by row
// (renderer.rs:69-70), sub = id % 16 walks the 4x4 tiles of the patch.
// dispatch id -> tile.  Workgroups are dispatched in id order; the affine map (a bijection:
// order_mul is coprime with n_tiles) decides which part of the image is rendered when.
// (natural and bottom-up order, the two that ship, without the 64-bit modulo)
__device__ __forceinline__ uint32_t tile_of_id(const KernelArgs &a, uint32_t id) {
    const uint32_t last = a.n_tiles - 1u;
    return (a.order_mul == 1u && a.order_add == 0u) ? id
         : (a.order_mul == last && a.order_add == last) ? last - id
         : (uint32_t)(((unsigned long long)id * a.order_mul + a.order_add) % a.n_tiles);
}

// tile -> pixel origin.  Patch-major: patch = tile / 16 walks the band row by row
// (renderer.rs:69-70), sub = tile % 16 walks the 4x4 tiles of the patch.
__device__ __forceinline__ void tile_origin(const KernelArgs &a, uint32_t tile, uint32_t &tx0, uint32_t &ty0, uint32_t &tyf,
                                            uint32_t &ty8) {
    const uint32_t patch = tile >> 4, sub = tile & 15u;
    const uint32_t pcol = patch % a.n_width, prow = patch / a.n_width;
    tx0 = pcol * 32u + (sub & (32u / TILE_W - 1u)) * TILE_W;
    const uint32_t in_patch = (sub / (32u / TILE_W)) * TILE_H;
    ty0 = (a.patch_row_begin + prow * a.patch_row_stride) * 32u + in_patch;
    const uint32_t packed = prow * 32u + in_patch;               // row of the tile among the owned rows
    tyf = a.f64_compact ? packed : ty0;                          // ... in the f64 frame
    ty8 = a.u8_compact ? packed : ty0;                           // ... in the u8 frame
}

}  // namespace rmdev

// ---- the two numeric flavours of the same source --------------------------------------
#if !defined(RM_KERNEL_FAST) || !RM_KERNEL_FAST
#define RM_FLAVOR_NS rmdev_strict
#define RM_FAST 0
#pragma clang fp contract(off)
#include "rm_trace.inc"
#include "rm_classify.inc"
#include "rm_render_kernel.inc"
#undef RM_FLAVOR_NS
#undef RM_FAST
#endif

#if !defined(RM_KERNEL_FAST) || RM_KERNEL_FAST
#define RM_FLAVOR_NS rmdev_fast
#define RM_FAST 1
#pragma clang fp contract(fast)
#include "rm_trace.inc"
#include "rm_classify.inc"
#include "rm_render_kernel.inc"
#undef RM_FLAVOR_NS
#undef RM_FAST
#pragma clang fp contract(off)
#endif

#endif
