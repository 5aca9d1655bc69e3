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

#include "rm_internal.h"

#define RM_BVH_NODE_WORDS 16u

// waves per SIMD the integer-power kernels are compiled for (register budget 512 / this)
#ifndef RM_MIN_WAVES
#define RM_MIN_WAVES 4
#endif

namespace rmdev {

// specular power flavours (see specular_pow in rm_trace.inc)
enum { POW_GENERIC = 0, POW_INTEGER = 1 };


// Tile shape (pixels).  RM_TILE_W x (64 / RM_TILE_W): 16x4 makes every tile row 384
// bytes = three whole 128-byte lines owned by one wave (8x8 rows are 192 bytes and
// share lines between waves: measured HBM write traffic 1.84x the frame's bytes).
#ifndef RM_TILE_W
#define RM_TILE_W 16
#endif
constexpr uint32_t TILE_W = RM_TILE_W, TILE_H = 64u / RM_TILE_W;
static_assert(TILE_W == 8 || TILE_W == 16 || TILE_W == 32, "tile width");

struct KernelArgs {
    rm_dev_header H;
    double half_fov, height, width, ratio;   // Renderer (renderer.rs:17-23)
    double cam_x, cam_y, cam_z;              // Scene.camera
    double bg_x, bg_y, bg_z;                 // renderer.rs:40-44
    uint32_t frame_width;                    // FrameBuffer.width
    uint32_t patch_row_begin;                // first owned patch row
    uint32_t patch_row_stride;               // owned rows: begin, begin + stride, ...
    uint32_t u8_compact;                     // RM_FLAG_U8_COMPACT: frame8 holds only the owned rows, packed
    uint32_t max_depth;                      // renderer.rs:262
    uint32_t n_width;                        // patches per row, renderer.rs:54
    uint32_t n_tiles;                        // 16 * patches in the band
    uint32_t order_mul;                      // dispatch order: tile = (id * order_mul + order_add) % n_tiles
    uint32_t order_add;
    uint32_t f64_compact;                    // RM_FLAG_F64_COMPACT: frame holds only the owned rows, packed
    double cull_cos;                         // bundles at least this narrow cull primitives (> 1: never; RM_DISABLE_CULL)
    uint8_t *frame8;                         // optional [H][W][3] u8 display frame (NULL: not written)
    // backproject (renderer.rs:128-135) tabulated per column and per row by the host with the
    // kernel's own operations: bp_x[x] = 2 (x / width - 0.5) half_fov ratio, bp_y[y] = -2 (y / height - 0.5) half_fov
    const double *bp_x, *bp_y;
    unsigned long long *debug_stamps;        // RM_EXP_STAMPS diagnostic build only
    // Frame-to-frame feedback (rm_device.hip, rm_feedback): the tiles that took longest in the
    // previous frame on this stream are dispatched first.  fb_flag == NULL: off.
    const uint32_t *fb_list;                 // previous frame: its long tiles, fb_count[0] of them (at most fb_cap)
    const uint32_t *fb_count;
    const uint8_t *fb_flag;                  // previous frame: 1 per tile that is on the list
    const uint32_t *fb_hist;                 // previous frame: its tiles by time, RM_FB_BUCKETS buckets (sampled)
    uint32_t *fb_next_list, *fb_next_count, *fb_next_hist;   // what this frame leaves for the next
    uint8_t *fb_next_flag;
    uint32_t *fb_zero;                       // counter and histogram of the frame after next: cleared by this one
    uint32_t *fb_threshold;                  // ticks from which a tile is long: set by this frame's first wave
    uint32_t fb_cap;                         // workgroups [0, fb_cap) take the list, the rest the tiles in order
    uint32_t fb_long_ticks;                  // the threshold while there is no histogram (100 MHz ticks)
    uint32_t fb_target;                      // tiles the list should hold (0: fb_long_ticks is the threshold)
};

// Feedback histogram: tile times in 100 MHz ticks, four buckets per octave (bucket b holds
// [edge(b), edge(b+1)), edge(b) = (4 + b % 4) << (b / 4) >> 2: 1, 1, 1, 1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 10, ...);
// every RM_FB_SAMPLE-th tile is counted.
#define RM_FB_BUCKETS 64u
#define RM_FB_SAMPLE 8u
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

struct StackEntry {
    double ox, oy, oz, dx, dy, dz, w;
    uint32_t depth, _pad;
};

// Per-wave LDS block behind the scene copy (8-byte words):
//   [0, 448)    level 0 of the lanes' ray stacks: 7 f64 fields, one lane-contiguous column each
//   [448, 480)  its u32 depth column
//   [480, 512)  the wave's hierarchy stack: 64 u32 entries
//   [512, 704)  the tile's pixel sums, [pixel][channel] -- read as they lie by the store phase
//   [704, 736)  pairing table of the ray hand-over: 64 u32 entries
#define RM_WAVE_L0_DEPTH_WORDS 448u
#define RM_WAVE_BVH_STACK_WORDS 480u
#define RM_WAVE_SUM_WORDS 512u
#define RM_WAVE_PAIR_WORDS 704u
#define RM_WAVE_LDS_WORDS 736u

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
#define RM_FLAVOR_NS rmdev_strict
#define RM_FAST 0
#pragma clang fp contract(off)
#include "rm_trace.inc"
#include "rm_render_kernel.inc"
#undef RM_FLAVOR_NS
#undef RM_FAST

#define RM_FLAVOR_NS rmdev_fast
#define RM_FAST 1
#pragma clang fp contract(fast)
#include "rm_trace.inc"
#include "rm_render_kernel.inc"
#undef RM_FLAVOR_NS
#undef RM_FAST
#pragma clang fp contract(off)

#endif
