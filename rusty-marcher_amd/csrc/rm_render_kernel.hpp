// rm_render_kernel.hpp -- the render kernels (gfx950, FP64).
//
// Replaces Renderer::render's Rayon patch loop (renderer.rs:63-89), the serial
// scatter (renderer.rs:92-108) and cast_ray's recursion (renderer.rs:254-309).
//
// Unit of work: an 8x8 pixel TILE rendered by one 64-lane wave, one lane per
// pixel (the 64 rays of a wave stay spatially coherent: same primitives hit,
// same branches; measured VALU lane utilisation 94 %).  Sixteen tiles make one of
// the reference's 32x32 patches (renderer.rs:47); tile ids are patch-major so a
// run of 16 consecutive ids is exactly one reference patch.
//
// cast_ray's recursion becomes a per-lane depth-first walk of the ray tree with
// an explicit stack: radiance is linear in the children (renderer.rs:219,249),
// so each ray carries the product of the reflection factors above it.
#ifndef RM_RENDER_KERNEL_HPP
#define RM_RENDER_KERNEL_HPP

#include "rm_trace.hpp"

namespace rmdev {

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
    uint32_t patch_row_begin;                // first patch row of the band
    uint32_t max_depth;                      // renderer.rs:262
    uint32_t n_width;                        // patches per row, renderer.rs:54
    uint32_t n_tiles;                        // 16 * patches in the band
    uint32_t order_mul;                      // dispatch order: tile = (id * order_mul + order_add) % n_tiles
    uint32_t order_add;
    uint32_t _pad;
    unsigned long long *debug_stamps;        // RM_EXP_STAMPS diagnostic build only
};

struct StackEntry {
    double ox, oy, oz, dx, dy, dz, w;
    uint32_t depth, _pad;
};

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
__device__ __forceinline__ void tile_origin(const KernelArgs &a, uint32_t id, uint32_t &tx0, uint32_t &ty0) {
    // Workgroups are dispatched in id order; the affine map (a bijection: order_mul is
    // coprime with n_tiles) decides which part of the image is rendered when.
    const uint32_t tile = (uint32_t)(((unsigned long long)id * a.order_mul + a.order_add) % a.n_tiles);
    const uint32_t patch = tile >> 4, sub = tile & 15u;
    const uint32_t pcol = patch % a.n_width, prow = patch / a.n_width;
    tx0 = pcol * 32u + (sub & (32u / TILE_W - 1u)) * TILE_W;
    ty0 = (a.patch_row_begin + prow) * 32u + (sub / (32u / TILE_W)) * TILE_H;
}

template <int STACK, int POW>
__device__ __forceinline__ void render_tile(const SceneView &sc, const KernelArgs &a, uint32_t tx0, uint32_t ty0,
                                            double *slab, double *__restrict__ frame) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t x = tx0 + (lane % TILE_W);
    const uint32_t y = ty0 + (lane / TILE_W);
    const V3 bg = mk(a.bg_x, a.bg_y, a.bg_z);

    // backproject, renderer.rs:128-135 (no pixel-centre offset)
    V3 dir = normalized(mk(2. * ((double)x / a.width - 0.5) * a.half_fov * a.ratio,
                           -2. * ((double)y / a.height - 0.5) * a.half_fov, -1.));
    V3 orig = mk(a.cam_x, a.cam_y, a.cam_z);
    double weight = 1.;
    uint32_t depth = 1;                                       // renderer.rs:83
    V3 acc = mk(0., 0., 0.);

    StackEntry stack[STACK];
    int sp = 0;

#ifdef RM_EXP_PHASES   // diagnostic build: shader-clock cycles per phase, summed per wave
    unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_closest = 0, ph_shade = 0, ph_child = 0, ph_setup = 0;
#define RM_PHASE(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n__ = __builtin_amdgcn_s_memtime(); acc += n__ - ph_t; ph_t = n__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RM_PHASE(acc) do {} while (0)
#endif
    RM_PHASE(ph_setup);
    for (;;) {
        Hit h;
        bool descend = false;
        const bool got = closest_hit(sc, orig, dir, h);
        RM_PHASE(ph_closest);
        if (got) {
            const Surface s = surface_at(sc, orig, dir, h);
            // renderer.rs:272-275: background + direct lighting
            const V3 L = bg + shade_direct<POW>(sc, orig, s);
            RM_PHASE(ph_shade);
            acc = acc + scaled(L, weight);
            if (s.mat[8] != 0.) {                             // is_glass_like, renderer.rs:277
                const double reflection = s.mat[6], ri = s.mat[7];
                const V3 incident = dir;
                const double w_here = weight;
                const uint32_t child_depth = depth + 1u;
                // a child beyond the cap returns the background (renderer.rs:262-264)
                const bool child_capped = child_depth > a.max_depth;
                V3 co, cd;
                if (reflect_child(incident, s, ri, co, cd)) { // renderer.rs:195-222
                    const double cw = w_here * reflection;
                    if (child_capped) {
                        acc = acc + scaled(bg, cw);
                    } else {                                  // pending sibling: at most one per level
                        StackEntry &e = stack[sp++];
                        e.ox = co.x; e.oy = co.y; e.oz = co.z;
                        e.dx = cd.x; e.dy = cd.y; e.dz = cd.z;
                        e.w = cw; e.depth = child_depth;
                    }
                }
                if (refract_child(incident, s, ri, co, cd)) { // renderer.rs:225-252
                    const double cw = w_here * (1. - reflection);
                    if (child_capped) {
                        acc = acc + scaled(bg, cw);
                    } else {                                  // walk into this child directly
                        orig = co; dir = cd; weight = cw; depth = child_depth;
                        descend = true;
                    }
                }
            }
        } else if (depth > 1u) {
            acc = acc + scaled(bg, weight);                   // renderer.rs:302-303
        }                                                     // primary miss: zero, :305
        RM_PHASE(ph_child);
        if (descend) continue;
        if (sp == 0) break;
        const StackEntry &e = stack[--sp];
        orig = mk(e.ox, e.oy, e.oz);
        dir = mk(e.dx, e.dy, e.dz);
        weight = e.w;
        depth = e.depth;
    }

    // ---- store the 8x8 tile: transpose through LDS so that each of the 8 rows
    // leaves as 192 contiguous bytes in 16-byte pieces (frame.buffer[y][x], :103)
    slab[lane * 3 + 0] = acc.x;
    slab[lane * 3 + 1] = acc.y;
    slab[lane * 3 + 2] = acc.z;
    __builtin_amdgcn_wave_barrier();
    const double2 *slab2 = reinterpret_cast<const double2 *>(slab);
    constexpr uint32_t PIECES = TILE_W * 3u / 2u;             // 16-byte pieces per tile row
    for (uint32_t q = lane; q < 96u; q += 64u) {
        const uint32_t row = q / PIECES, piece = q % PIECES;
        double2 *dst = reinterpret_cast<double2 *>(frame + ((size_t)(ty0 + row) * a.frame_width + tx0) * 3u);
        dst[piece] = slab2[row * PIECES + piece];
    }
    __builtin_amdgcn_wave_barrier();
#ifdef RM_EXP_PHASES
    RM_PHASE(ph_setup);
    if (lane == 0 && a.debug_stamps) {
        unsigned long long *o = a.debug_stamps + 4ull * blockIdx.x;
        o[0] = ph_closest; o[1] = ph_shade; o[2] = ph_child; o[3] = ph_setup;
    }
#endif
}

// ---- static mode ---------------------------------------------------------------
// Workgroup of WAVES waves renders WAVES*TPW consecutive tile ids; wave w takes ids
// base + w, base + w + WAVES, ...  (WAVES = 4, TPW = 4 is one reference patch per
// workgroup.)  The hardware dispatcher balances the load across workgroups.
// Register budget: with the integer specular power the kernel fits 128 VGPRs (4 waves per
// SIMD) at the cost of two spilled doubles; measured 131.6 -> 120.6 us at 1080p.  The
// generic pow flavour needs ~168 (its constants, see specular_pow) and is left alone.
template <int STACK, int POW, int WAVES, int TPW>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(POW == POW_INTEGER ? 4 : 1, 8))) void rm_render_static(const double *__restrict__ scene_blob, KernelArgs a,
                                                              double *__restrict__ frame) {
#ifdef RM_EXP_STAMPS   // diagnostic build: wave start / staged / end times + hardware slot, to a buffer of their own
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    stage_scene(scene_blob, a.H);
    const uint32_t wave = threadIdx.x >> 6;
#ifdef RM_EXP_STAMPS
    const unsigned long long t_staged = __builtin_amdgcn_s_memrealtime();
#endif
    double *slab = rm_lds + a.H.total_words + wave * (64 * 3);
    SceneView sc;
    sc.S = rm_lds;
    sc.G = scene_blob;
    sc.H = a.H;
    const uint32_t base = blockIdx.x * (uint32_t)(WAVES * TPW);
    for (uint32_t k = 0; k < (uint32_t)TPW; k++) {
        const uint32_t tile = base + k * WAVES + wave;
        if (tile >= a.n_tiles) break;                          // wave-uniform
        uint32_t tx0, ty0;
        tile_origin(a, tile, tx0, ty0);
        render_tile<STACK, POW>(sc, a, tx0, ty0, slab, frame);
    }
#ifdef RM_EXP_STAMPS
    if ((threadIdx.x & 63u) == 0 && a.debug_stamps) {
        unsigned long long *o = a.debug_stamps + 4ull * (blockIdx.x * WAVES + wave);
        o[0] = t_start; o[1] = t_staged; o[2] = __builtin_amdgcn_s_memrealtime();
        o[3] = ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) << 32) |   // HW_REG_HW_ID
               __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));                                  // HW_REG_XCC_ID
    }
#endif
}

}  // namespace rmdev
#endif
