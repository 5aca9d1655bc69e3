// rm_device.hip -- device half of the C ABI: context, scene upload, the render
// kernel and the framebuffer post-process kernels.  gfx950 (MI355X) only.
//
// Replaces Renderer::render's Rayon patch loop (renderer.rs:63-89), the serial
// scatter (renderer.rs:92-108) and cast_ray's recursion (renderer.rs:254-309).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "rm_bvh.hpp"
#include "rm_internal.h"
#include "rm_kernel_args.hpp"

using namespace rmdev;

#ifndef RM_BUILD_FLAVOR
#define RM_BUILD_FLAVOR "fp64 strict+fast"
#endif

// With max_depth == 0 the primary ray itself is capped: every pixel is the
// background (renderer.rs:262-264 with n_recursion = 1 > 0).
__global__ void rm_fill_band_kernel(double *frame, size_t first_px, size_t n_px, double r, double g, double b) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_px) {
        double *p = frame + (first_px + i) * 3;
        p[0] = r; p[1] = g; p[2] = b;
    }
}

// ---------------------------------------------------------------------------
// Post-process kernels: framebuffer.rs:58-77 (normalize) and :40-55,:80-82
// (to_vec / quantize).
// ---------------------------------------------------------------------------

// Global max over all channel values, starting from 0 like the reference's
// `max = Vec3f::zero()`; f64::max ignores NaN, as fmax does.  Values are >= 0
// after max(0, .), so their bit patterns order like unsigned integers.
__global__ __launch_bounds__(256) void rm_max_kernel(const double *__restrict__ v, size_t n, unsigned long long *out_bits) {
    double m = 0.;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = __builtin_fmax(m, v[i]);
    for (int off = 32; off > 0; off >>= 1) m = __builtin_fmax(m, __shfl_down(m, off));
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = __builtin_fmax(__builtin_fmax(part[0], part[1]), __builtin_fmax(part[2], part[3]));
        atomicMax(out_bits, (unsigned long long)__double_as_longlong(m));
    }
}

// Optional in-place scale by 1/max (framebuffer.rs:69-76) and optional u8 output
// `(255. * f.max(0.).min(1.)) as u8` (truncating, NaN -> 0).
__global__ __launch_bounds__(256) void rm_scale_quantize_kernel(double *__restrict__ v, size_t n, const unsigned long long *max_bits,
                                                               int do_scale, uint8_t *__restrict__ out8) {
    double s = 1.;
    bool scale = false;
    if (do_scale) {
        const double max_val = __longlong_as_double((long long)*max_bits);
        if (max_val > 0.) { s = 1. / max_val; scale = true; }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double f = v[i];
        if (scale) { f = f * s; v[i] = f; }
        if (out8) {
            const double c = 255. * __builtin_fmin(__builtin_fmax(f, 0.), 1.);
            out8[i] = (uint8_t)c;
        }
    }
}

// ---------------------------------------------------------------------------
// Context
// ---------------------------------------------------------------------------

// How the 64-pixel tiles of a band are handed to waves (see rm_render_kernel.hpp): `waves`
// waves per workgroup, `per_wave` tiles per wave.  Measured at 1080p on the demo scene
// (profiles/r01_ab_launch_modes.txt): one tile per wave wins (finer units for the
// hardware dispatcher: 4 tiles per wave 193 us vs 1 tile 122 us); 1 or 4 waves per
// workgroup differ by ~2 % there -- but not where tile costs differ widely: a slot freed by a
// workgroup of four is handed on only when a whole workgroup fits (256 spheres: 1,777 us with
// four waves, 1,420 with one).
// A persistent variant (waves pulling tiles from a global counter) was measured too and
// dropped: one atomic word serves ~70 claims/us, a 1080p frame needs >300 tiles/us.
// This build instantiates one tile per wave, one wave per workgroup only; the other geometries
// were measured with earlier builds.
// (3,584: both share sizes of a 1080p frame at N = 8 -- 3,840 and 4,800 tiles -- take the same path; measured no difference in time)
static constexpr uint32_t RM_CLASSIFY_MIN_TILES_DEFAULT = 3584;
static constexpr uint32_t RM_CULL_MIN_PRIMS = 12, RM_CULL_EDGES_MIN_PLANAR = 4, RM_CULL_MAX_COST = 250;
static inline uint32_t k_planar_from(const rm_dev_header &H) { return H.n_spheres; }   // pid of the first planar primitive

struct rm_launch_mode {
    int waves = 0;      // waves per workgroup
    int per_wave = 1;   // tiles per wave
};

// dispatch order of the tiles (tile_origin): RM_TILE_ORDER = natural | reverse | hash
enum { TILE_ORDER_NATURAL = 0, TILE_ORDER_REVERSE = 1, TILE_ORDER_HASH = 2 };

// A frame in flight (rm_frame_submit): its own render stream, so that consecutive frames overlap.
struct rm_frame_slot {
    hipStream_t render = nullptr;
    hipEvent_t begun = nullptr, rendered = nullptr, gathered = nullptr;   // stamps of the frame on the slot's stream
    hipEvent_t exchanged = nullptr;   // recorded after the last operation of the slot's frame
    bool used = false;
    bool stamped = false;             // the slot's last frame carries the stamps of rm_frame_timing
};

// Frame-to-frame feedback.  The cost of a tile is known only once it has been rendered -- 1 to
// 18 ray steps on the 256-sphere scene, the incoherent ones thirty times the price of a coherent
// one -- and a launch ends with whatever long tile was dispatched late: there the last 150 of
// 260,000 waves ran for 200 of the launch's 1,600 us on an empty chip.  Frames of a render loop
// resemble their predecessors, so every wave times its tile, the tiles that took long are put on
// a list (one atomic each: they are few) and the next frame ON THE SAME STREAM dispatches the
// list first.  Nothing is carried over but the order of dispatch: every tile of every frame is
// rendered in full, by the same code.  Three rotating sets (list, count, one flag per tile): a
// frame reads one, fills the next and clears the counter of the third.  Kept per stream --
// launches on one stream are ordered, so a set is never read and written at once.
struct rm_feedback {
    hipStream_t stream = nullptr;
    uint64_t key[3] = {0, 0, 0};      // launch geometry and scene the sets belong to
    uint32_t n_tiles = 0, cap = 0;
    void *block = nullptr;            // one allocation: 3 x (hist[64] | count | 3 pad | list[cap] u32 | flag[n_tiles] u8) + threshold
    size_t set_bytes = 0;
    int cur = 0;                      // the set the next frame reads
    uint64_t used = 0;
    uint32_t *hist(int j) const { return reinterpret_cast<uint32_t *>(static_cast<char *>(block) + (size_t)j * set_bytes); }
    uint32_t *count(int j) const { return hist(j) + RM_FB_BUCKETS; }          // (cleared together with the histogram)
    uint32_t *list(int j) const { return hist(j) + RM_FB_BUCKETS + 4u; }
    uint8_t *flag(int j) const { return reinterpret_cast<uint8_t *>(list(j) + cap); }
    uint32_t *threshold() const { return hist(3); }
};

struct rm_hostio;   // rm_hostio.inc: staging buffer, row-scatter threads, display frame

// The classification's output for the render launches on one stream: a mask per tile.  (Launches on a
// stream are ordered: a render launch reads what the classification launch in front of it wrote, and the
// next classification overwrites it only after that render launch is over.  Classifying a frame ahead on
// a stream of its own was measured and dropped: the two cross-stream events cost more than the 3-8 us of
// the classification they hid -- demo scene 1080p 99.8 against 81.2 us per frame.)
struct rm_tile_lists {
    hipStream_t stream = nullptr;
    uint32_t cap = 0;                 // tiles the masks have room for
    void *block = nullptr;            // mask[cap] u64
    uint64_t used = 0;
    // Dispatch order from the launch's own classification (KernelArgs::ord_*), one block per stream:
    //   cost[3][cap] | ctab[3][128] | cnt[2][RM_ORD_CNT_WORDS] | flat[2][cap] | first[2][cap] | index[2][cap] | inv[2][cap] | rec[3][cap + 4096] | done[2]      (u32; cap: patches)
    // the sets take turns from launch to launch (a launch reads what its predecessor on the stream wrote, and clears what
    // its successor will count into)
    void *order_block = nullptr;
    uint32_t order_cap = 0, order_frames = 0;
    uint64_t order_key[3] = {0, 0, 0};
    uint32_t *words() const { return static_cast<uint32_t *>(order_block); }
    uint32_t *cost(uint32_t j) const { return words() + (size_t)j * order_cap; }
    uint32_t *ctab(uint32_t j) const { return words() + 3u * (size_t)order_cap + j * RM_CTAB_WORDS; }
    uint32_t *cnt(uint32_t j) const { return words() + 3u * (size_t)order_cap + 3u * RM_CTAB_WORDS + j * RM_ORD_CNT_WORDS; }
    uint32_t *flat(uint32_t j) const { return cnt(2) + (size_t)j * order_cap; }
    uint32_t *first(uint32_t j) const { return flat(2u + j); }
    uint32_t *index(uint32_t j) const { return flat(4u + j); }
    uint32_t *inv(uint32_t j) const { return flat(6u + j); }
    uint32_t *rec() const { return flat(8u); }
    uint32_t *done() const { return rec() + 3u * ((size_t)order_cap + 4096u); }
    uint32_t list_tag[2] = {0, 0};    // the tag of the launch that was to write first[] / index[] / inv[] of that number
    static size_t order_bytes(uint32_t cap) { return ((size_t)cap * 11u + 64u + 3u * ((size_t)cap + 4096u) + 3u * RM_CTAB_WORDS + 2u * RM_ORD_CNT_WORDS) * 4u; }
    int static_read = -1, static_written = -1;   // the first[] / index[] pair the previous launch's first round came from / the one it wrote for a successor (-1: none)
    uint32_t last_tag = 0;            // the tag of the order the previous launch laid out (0: none to dispatch by)
    // sky tail: the first classifying workgroup's word in page-locked memory -- (launch seq << 32) | ordered patches with something
    // to hit --, behind it the word a wave writes when it gives up a wait that cannot fail; the launches on this stream counted, the
    // first launch of the view being rendered, the first of this geometry and scene, and that view
    unsigned long long *hint = nullptr;
    uint32_t seq = 0, view_seq0 = 0, key_seq0 = 0, ord_tag = 0;
    double view[7] = {0., 0., 0., 0., 0., 0., 0.};
    // classification at the head of the render launch: the words carry the launch's tag (1..255)
    uint32_t tag = 0, tagged_tiles = 0;
    uint64_t tagged_scene = 0;
    bool tagged = false;
    uint32_t frozen_run = 0;          // launches in a row that took everything from their predecessor (order_freeze)
    unsigned long long *mask() const { return static_cast<unsigned long long *>(block); }
};

struct rm_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string error;
    hipDeviceProp_t prop{};

    // uploaded scene
    bool have_scene = false;
    rm_dev_header H{};
    double *d_scene = nullptr;
    size_t d_scene_words = 0;
    rm_vec3 camera{0., 0., 0.};
    std::vector<double> host_blob;    // the device image of the resident scene (rm_scene_upload skips identical ones)
    uint64_t upload_calls = 0, upload_copies = 0;
    std::vector<unsigned char> desc_bytes;   // the description arrays the resident image was built from, back to back
    size_t desc_sizes[6] = {0, 0, 0, 0, 0, 0};
    bool integer_exponents = false;   // every material's specular_exponent is a small non-negative integer
    bool force_generic_pow = false;   // RM_FORCE_GENERIC_POW=1 (A/B knob)
    bool force_fast_fp = false;       // RM_FORCE_FAST_FP=1 (A/B knob; same as RM_FLAG_FAST_FP on every call)
    // Bottom-up by default: workgroups are dispatched in id order and the drain at the end
    // of a launch runs at low occupancy, so the rows that are expensive in the
    // reference's scenes (ground, objects resting on it) go first and the cheap sky rows
    // drain (1080p demo: 124 -> 116 us; hashed order 131 us).
    int tile_order = TILE_ORDER_REVERSE;
    bool disable_bvh = false;         // RM_DISABLE_BVH=1: brute-force walk (A/B knob)
    // Ray bundles whose half-angle has at least this cosine cull primitives before a walk
    // (rm_trace.inc); wider ones take the plain walk / the hierarchy.  RM_DISABLE_CULL=1 sets 2
    // (never), RM_CULL_COS overrides (A/B knobs).
    double cull_cos = 0.975;          // (synthetic-256: 0.9 1,911 us, 0.95 1,825, 0.97-0.98 1,777, 0.99 1,791, 0.999 1,847; cornell flat)
    uint32_t cull_min_prims = RM_CULL_MIN_PRIMS;   // RM_CULL_MIN (A/B knob)
    bool cull_edges = true;           // RM_CULL_EDGES=0: the bundle cull without its edge test (A/B knob)
    bool force_unstaged = false;      // RM_FORCE_UNSTAGED=1 (A/B knob)
    int force_stack = 0;              // RM_FORCE_STACK=4|8|16|32: a deeper ray stack than the depth cap needs (A/B knob)
    bool debug_empty = false;         // RM_DEBUG_EMPTY=1: measure the dispatch floor of a launch geometry
    // frame-to-frame feedback (rm_feedback): RM_FEEDBACK=0 never, 1 always, unset: launches of
    // RM_FEEDBACK_MIN_TILES tiles and more with a depth cap of 6 and more (below, no tile is long)
    int feedback_mode = -1;
    uint32_t feedback_us = 50;        // RM_FEEDBACK_US: a tile is long from here on, until a frame's histogram says better
    int feedback_target = -1;         // RM_FEEDBACK_TARGET: tiles the list should hold (-1: two per wave slot; 0: fixed threshold)
    std::vector<rm_feedback> feedback;
    uint64_t feedback_clock = 0, scene_epoch = 0;
    // tile classification (rm_classify.hip): RM_TILE_CLASSIFY=0 never, 1 whenever the scene allows; unset:
    // launches of RM_CLASSIFY_MIN_TILES tiles and more
    int classify_mode = -1;
    bool classify_in_launch = true;      // RM_CLASSIFY_IN_LAUNCH=0: always a launch of its own in front (A/B knob)
    bool sky_tail = true;                // RM_SKY_TAIL=0: every patch gets its sixteen waves
    uint32_t patch_order_max = 4096;     // RM_PATCH_ORDER_MAX: launches of up to this many patches take the kernels with the patch order
    uint32_t patch_order_max_deep = 65536;   // RM_PATCH_ORDER_MAX_DEEP: ... in scenes with a hierarchy (tile times with a long tail)
    uint32_t sky_tail_big_min = 16384;   // RM_SKY_TAIL_BIG_MIN (patches; see RM_SKY_TAIL_BIG_MIN_PATCHES)
    uint32_t sky_tail_room_div = 16;     // RM_SKY_TAIL_ROOM_DIV: a guessed tail's room in launches of many patches: patches / this (A/B knob)
    bool sky_tail_big = true;             // RM_SKY_TAIL_BIG=0: launches of more than patch_order_max patches keep the kernels without the patch order
    bool sky_tail_motion = true;         // RM_SKY_TAIL_MOTION=0: no tail in a frame whose view differs from the frames the hint came from
    int sky_tail_place = 0;              // RM_SKY_TAIL_PLACE=even|end: the tail's waves dealt out among the tile waves / behind them (unset: behind them in launches of up to patch_order_max patches)
    int sky_tail_cap = -1;               // RM_SKY_TAIL_CAP=n: places a guessed tail can hand on to waves behind the grid's end (unset: max(512, patches / 16))
    bool cull_lds = false;               // RM_CULL_LDS=1: the render waves of scenes without an LDS copy pack the bundle cull's arrays into their LDS block (A/B knob: measured slower)
    bool classify_lds = true;            // RM_CLASSIFY_LDS=0: the classifying workgroups of scenes without an LDS copy read their tests' data from memory (A/B knob)
    uint32_t classify_in_launch_prims = 56;   // RM_CLASSIFY_IN_LAUNCH_PRIMS: scenes of up to this many primitives are classified at the head of the render launch
    uint32_t classify_min_tiles = 0;     // RM_CLASSIFY_MIN_TILES: launches of this many tiles and more are classified (and ordered); 0: RM_CLASSIFY_MIN_TILES, the built-in
    bool mask_reuse = true;              // RM_MASK_REUSE=0: a launch waits for its own classification even where its predecessor's is as good (A/B knob)
    uint32_t static_rounds = 2;          // RM_STATIC_ROUNDS=n: n rounds of waves take their patches from the previous ranking without waiting for the order (A/B knob)
    uint32_t cls_max_blocks = RM_ORD_MAX_CLS;   // RM_CLS_MAX_BLOCKS=n: at most n classifying workgroups (each then takes more groups of four patches; A/B knob)
    int order_freeze = 7;                // RM_ORDER_FREEZE=n: of n + 1 launches of a standing view only one classifies and lays out an order (0: every launch)
    bool order_late_places = true;       // RM_ORDER_LATE_PLACES=0: the classifying workgroups always write the order's places themselves (A/B knob)
    bool order_reuse = true;             // RM_ORDER_REUSE=0: every launch dispatches by its own order, standing view or not (A/B knob)
    bool first_round_from_order = true;  // RM_FIRST_ROUND_FROM_ORDER=0: the first round is the bottom rows by place (A/B knob)
    int first_round = -1;                // RM_FIRST_ROUND=n: the waves that neither wait for their tiles' classification nor take a place in the order (unset: what is resident at once)
    int order_keys = -1;                 // RM_ORDER_KEYS=0 by place only, 1 the previous frame's times by place only, 2 cost by content only (A/B knob; unset: times while the view stands, content once it has moved)
    uint32_t ord_tag_wrap = 0;           // RM_ORD_TAG_WRAP=n (test hook): the order's tags start afresh after n launches instead of 4,095
    int test_stall_order = 0;            // RM_TEST_STALL_ORDER (test hook) = 1: the launch's order is never laid out (the frame is void); = 2: its classifying workgroups never say they have arrived (the order of last resort)
    int sky_tail_force = -1;             // RM_SKY_TAIL_FORCE=n (test hook): the last n patches of the order are taken for sky, whatever the hint says
    int patch_order_mode = -1;           // RM_PATCH_ORDER=0 never, 1 whenever possible; unset: launches of RM_CLASSIFY_MIN_TILES tiles and more
    std::vector<rm_tile_lists> tile_lists;
    uint32_t last_launch_grid = 0, last_launch_tail = 0;   // rm_launch_stats
    uint32_t last_launch_tiles = 0;   // rm_tile_stats: the last render launch's tiles, and whether they were classified
    bool last_launch_classified = false;
    hipStream_t last_launch_stream = nullptr;

    // device framebuffer of rm_render
    double *d_frame = nullptr;
    size_t frame_bytes = 0;
    uint32_t frame_w = 0, frame_h = 0;

    // multi-GPU frames (rm_exchange.inc)
    void *comm = nullptr;             // ncclComm_t
    void *slot_comm[RM_MAX_FRAME_SLOTS] = {};   // per frame slot: `comm` itself, or (RM_SLOT_COMMS=1) one split off it
    int n_comms = 0;                  // distinct communicators in use
    bool frame_stamps = false;        // rm_frame_timing_enable
    bool exchange_all_ranks = false;  // rm_comm_exchange: all-gather instead of the gather at rank 0
    bool comm_failed = false;         // a frame wait timed out: the communicator is abandoned, not destroyed
    bool comm_local = false;          // rank/world set without a transport (rm_comm_init with id == NULL)
    int rank = 0, world = 1;
    rm_frame_slot slots[RM_MAX_FRAME_SLOTS];

    // backproject tables (per column, per row) of the current frame geometry
    double *d_backproject = nullptr;
    size_t backproject_words = 0;
    double backproject_key[6] = {};

    // the seam into the reference's FrameBuffer (rm_hostio.inc)
    rm_hostio *hostio = nullptr;
    bool comm_stuck = false;          // a timed-out collective could not be aborted: nothing that waits for the device may run

    // post-process scratch
    unsigned long long *d_max = nullptr;
    uint8_t *d_rgb8 = nullptr;
    size_t rgb8_bytes = 0;
};

static void hostio_destroy(rm_ctx *ctx, bool device_ok);
struct rm_band;
static bool hostio_packs(rm_ctx *ctx, size_t band_bytes);

static rm_status ctx_fail(rm_ctx *ctx, rm_status st, const std::string &msg) {
    if (ctx) ctx->error = msg;
    else rm_set_host_error(msg);
    return st;
}

// Nothing unwinds across the C ABI: what an entry point's body may throw (std::vector / std::function growing: bad_alloc)
// comes back as a status like every other failure.
template <class F>
static rm_status guarded(rm_ctx *ctx, const char *who, F &&body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return ctx_fail(ctx, RM_ERR_INVALID_ARG, std::string(who) + ": out of host memory");
    } catch (const std::exception &e) {
        return ctx_fail(ctx, RM_ERR_INVALID_ARG, std::string(who) + ": " + e.what());
    } catch (...) {
        return ctx_fail(ctx, RM_ERR_INVALID_ARG, std::string(who) + ": unexpected exception");
    }
}

#define RM_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ctx_fail(ctx, RM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Scenes up to this size get a copy in every workgroup's LDS for the per-lane gathers;
// larger ones are read from global memory only (the primitive loops always are, through
// scalar loads).  Measured: the LDS copy is worth 2 % on the 1.3 KB demo scene (81.3 vs
// 82.8 us), nothing on the 7.6 KB cornell box (153 vs 150 us) and costs 20 % on the 29 KB
// synthetic scene (10.5 vs 8.7 ms: every workgroup re-stages the blob).
// RM_ERR_SCENE_LIMIT is left for what the blob's 32-bit word offsets cannot address.
static constexpr size_t RM_LDS_SCENE_LIMIT_BYTES = RM_LDS_SCENE_LIMIT_WORDS * sizeof(double);   // (4 KB)
static constexpr uint64_t RM_SCENE_MAX_WORDS = 0xFFFFFFF0ull;
// A hierarchy is built over a kind once it has this many primitives (below, the flat walk
// is as fast: the demo scene has 4 spheres).
static constexpr size_t RM_BVH_MIN_SPHERES = 16, RM_BVH_MIN_TRIANGLES = 12;


// The kernel instantiations live in rm_kernels.hip, one object per numeric flavour and kernel
// group (STAGED: LDS copy of the scene for the per-lane gathers; BVH: hierarchy walk for wide
// bundles; CULL: bundle culling, EDGES: its edge test for planar primitives, rm_trace.inc;
// FEEDBACK: longest tiles of the previous frame first).
#define RM_DECLARE_GROUP(g) \
    const void *rm_pick_kernel_strict_g##g(bool edges, int order, int stack, int pow_mode); \
    const void *rm_pick_kernel_fast_g##g(bool edges, int order, int stack, int pow_mode);
RM_DECLARE_GROUP(0) RM_DECLARE_GROUP(1) RM_DECLARE_GROUP(2) RM_DECLARE_GROUP(3) RM_DECLARE_GROUP(4)
#undef RM_DECLARE_GROUP

const void *rm_pick_kernel(bool fast, bool staged, bool bvh, bool cull, bool edges, int order, bool feedback, int stack, int pow_mode) {
    const int group = staged ? (cull ? 1 : 0) : !bvh ? 2 : feedback ? 4 : 3;
    if (staged && (bvh || feedback)) return nullptr;         // no such kernel: small scenes have no hierarchy
    if (!staged && !cull) return nullptr;                    // scenes in global memory always cull
    switch (group) {
    case 0: return fast ? rm_pick_kernel_fast_g0(edges, order, stack, pow_mode) : rm_pick_kernel_strict_g0(edges, order, stack, pow_mode);
    case 1: return fast ? rm_pick_kernel_fast_g1(edges, order, stack, pow_mode) : rm_pick_kernel_strict_g1(edges, order, stack, pow_mode);
    case 2: return fast ? rm_pick_kernel_fast_g2(edges, order, stack, pow_mode) : rm_pick_kernel_strict_g2(edges, order, stack, pow_mode);
    case 3: return fast ? rm_pick_kernel_fast_g3(edges, order, stack, pow_mode) : rm_pick_kernel_strict_g3(edges, order, stack, pow_mode);
    default: return fast ? rm_pick_kernel_fast_g4(edges, order, stack, pow_mode) : rm_pick_kernel_strict_g4(edges, order, stack, pow_mode);
    }
}

extern "C" {

const char *rm_build_info(void) {
    return "rusty-marcher_amd " RM_BUILD_FLAVOR " gfx950 abi5";
}

const char *rm_last_error(const rm_ctx *ctx) {
    return ctx ? ctx->error.c_str() : rm_get_host_error();
}

rm_status rm_init(int device_ordinal, rm_ctx **out) {
    if (!out) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_init: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return ctx_fail(nullptr, RM_ERR_NO_DEVICE,
                        std::string("rm_init: no HIP device visible (") + hipGetErrorString(e) +
                            "); this backend has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n)
        return ctx_fail(nullptr, RM_ERR_NO_DEVICE, "rm_init: device ordinal out of range");
    rm_ctx *ctx = new rm_ctx();
    ctx->device = device_ordinal;
    auto bail = [&](const char *what, hipError_t err) {
        rm_set_host_error(std::string("rm_init: ") + what + ": " + hipGetErrorString(err));
        delete ctx;
        return RM_ERR_HIP;
    };
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipGetDeviceProperties(&ctx->prop, device_ordinal)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    if (std::string(ctx->prop.gcnArchName).rfind("gfx950", 0) != 0) {
        rm_set_host_error(std::string("rm_init: device is ") + ctx->prop.gcnArchName +
                          ", this library carries gfx950 code only");
        delete ctx;
        return RM_ERR_NO_DEVICE;
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc(&ctx->d_max, sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if (const char *env = std::getenv("RM_FORCE_GENERIC_POW")) ctx->force_generic_pow = env[0] == '1';
    if (const char *env = std::getenv("RM_FORCE_FAST_FP")) ctx->force_fast_fp = env[0] == '1';
    if (const char *env = std::getenv("RM_FORCE_UNSTAGED")) ctx->force_unstaged = env[0] == '1';
    if (const char *env = std::getenv("RM_DISABLE_BVH")) ctx->disable_bvh = env[0] == '1';
    if (const char *env = std::getenv("RM_FORCE_STACK")) ctx->force_stack = std::atoi(env);
    if (const char *env = std::getenv("RM_FEEDBACK")) ctx->feedback_mode = env[0] == '1' ? 1 : 0;
    if (const char *env = std::getenv("RM_FEEDBACK_US")) ctx->feedback_us = (uint32_t)std::max(1, std::atoi(env));
    if (const char *env = std::getenv("RM_FEEDBACK_TARGET")) ctx->feedback_target = std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_DEBUG_EMPTY")) ctx->debug_empty = env[0] == '1';
    if (const char *env = std::getenv("RM_TILE_CLASSIFY")) ctx->classify_mode = env[0] == '1' ? 1 : 0;
    if (const char *env = std::getenv("RM_CLASSIFY_IN_LAUNCH")) ctx->classify_in_launch = env[0] == '1';
    if (const char *env = std::getenv("RM_PATCH_ORDER")) ctx->patch_order_mode = env[0] == '1' ? 1 : 0;
    if (const char *env = std::getenv("RM_SKY_TAIL")) ctx->sky_tail = env[0] != '0';
    if (const char *env = std::getenv("RM_SKY_TAIL_FORCE")) ctx->sky_tail_force = std::atoi(env);
    if (const char *env = std::getenv("RM_PATCH_ORDER_MAX")) ctx->patch_order_max = (uint32_t)std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_SKY_TAIL_BIG")) ctx->sky_tail_big = env[0] != '0';
    if (const char *env = std::getenv("RM_PATCH_ORDER_MAX_DEEP")) ctx->patch_order_max_deep = (uint32_t)std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_SKY_TAIL_ROOM_DIV")) ctx->sky_tail_room_div = (uint32_t)std::max(1, std::atoi(env));
    if (const char *env = std::getenv("RM_SKY_TAIL_BIG_MIN")) ctx->sky_tail_big_min = (uint32_t)std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_SKY_TAIL_PLACE")) ctx->sky_tail_place = env[0] == 'e' && env[1] == 'v' ? 1 : env[0] == 'e' ? 2 : 0;
    if (const char *env = std::getenv("RM_SKY_TAIL_MOTION")) ctx->sky_tail_motion = env[0] != '0';
    if (const char *env = std::getenv("RM_SKY_TAIL_CAP")) ctx->sky_tail_cap = std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_ORDER_KEYS")) ctx->order_keys = std::atoi(env);
    if (const char *env = std::getenv("RM_FIRST_ROUND")) ctx->first_round = std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_FIRST_ROUND_FROM_ORDER")) ctx->first_round_from_order = env[0] != '0';
    if (const char *env = std::getenv("RM_ORDER_REUSE")) ctx->order_reuse = env[0] != '0';
    if (const char *env = std::getenv("RM_STATIC_ROUNDS")) ctx->static_rounds = (uint32_t)std::max(1, std::atoi(env));
    if (const char *env = std::getenv("RM_CLS_MAX_BLOCKS")) ctx->cls_max_blocks = (uint32_t)std::max(1, std::atoi(env));
    if (const char *env = std::getenv("RM_ORDER_FREEZE")) ctx->order_freeze = std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_ORDER_LATE_PLACES")) ctx->order_late_places = env[0] != '0';
    if (const char *env = std::getenv("RM_MASK_REUSE")) ctx->mask_reuse = env[0] != '0';
    if (const char *env = std::getenv("RM_CULL_LDS")) ctx->cull_lds = env[0] == '1';
    if (const char *env = std::getenv("RM_CLASSIFY_LDS")) ctx->classify_lds = env[0] != '0';
    if (const char *env = std::getenv("RM_CLASSIFY_IN_LAUNCH_PRIMS")) ctx->classify_in_launch_prims = (uint32_t)std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_CLASSIFY_MIN_TILES")) ctx->classify_min_tiles = (uint32_t)std::max(0, std::atoi(env));
    if (const char *env = std::getenv("RM_ORD_TAG_WRAP")) ctx->ord_tag_wrap = (uint32_t)std::max(1, std::atoi(env));
    if (const char *env = std::getenv("RM_TEST_STALL_ORDER")) ctx->test_stall_order = std::atoi(env);
    if (const char *env = std::getenv("RM_TILE_ORDER"))
        ctx->tile_order = !std::strcmp(env, "reverse") ? TILE_ORDER_REVERSE
                        : !std::strcmp(env, "hash") ? TILE_ORDER_HASH : TILE_ORDER_NATURAL;
    if (const char *env = std::getenv("RM_DISABLE_CULL")) ctx->cull_cos = env[0] == '1' ? 2. : ctx->cull_cos;
    if (const char *env = std::getenv("RM_CULL_COS"))   // (the cone tests hold for half-angles below 90 degrees)
        ctx->cull_cos = std::max(0.05, std::atof(env));
    if (const char *env = std::getenv("RM_CULL_MIN")) ctx->cull_min_prims = (uint32_t)std::strtoul(env, nullptr, 10);
    if (const char *env = std::getenv("RM_CULL_EDGES")) ctx->cull_edges = env[0] != '0';
    if (ctx->classify_min_tiles == 0u) ctx->classify_min_tiles = RM_CLASSIFY_MIN_TILES_DEFAULT;
    *out = ctx;
    return RM_OK;
}

void rm_destroy(rm_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream && !ctx->comm_failed) (void)hipStreamSynchronize(ctx->stream);
    rm_comm_destroy(ctx);
    // A frame wait that timed out left a collective in flight.  Where RCCL could abort its
    // communicator the device drains and everything below is safe; where it could not
    // (ncclCommAbort absent or failing) the collective's kernel never ends, and hipFree /
    // hipStreamSynchronize / hipStreamDestroy -- each waits for the device -- would hang for
    // ever: the hang the timeout was there to remove.  Then nothing device-side is released:
    // the process is about to exit (RM_ERR_TIMEOUT: "report and exit") and takes it along.
    const bool device_ok = !ctx->comm_stuck;
    hostio_destroy(ctx, device_ok);
    if (device_ok) {
        for (rm_frame_slot &s : ctx->slots) {
            // after a timed-out collective the slot's stream may never drain: leave it to process exit
            if (s.render && !ctx->comm_failed) { (void)hipStreamSynchronize(s.render); (void)hipStreamDestroy(s.render); }
            for (hipEvent_t e : {s.begun, s.rendered, s.gathered, s.exchanged})
                if (e) (void)hipEventDestroy(e);
        }
        for (rm_feedback &f : ctx->feedback)
            if (f.block) (void)hipFree(f.block);
        for (rm_tile_lists &t : ctx->tile_lists) {
            if (t.block) (void)hipFree(t.block);
            if (t.order_block) (void)hipFree(t.order_block);
            if (t.hint) (void)hipHostFree(t.hint);
        }
        if (ctx->d_scene) (void)hipFree(ctx->d_scene);
        if (ctx->d_frame) (void)hipFree(ctx->d_frame);
        if (ctx->d_backproject) (void)hipFree(ctx->d_backproject);
        if (ctx->d_max) (void)hipFree(ctx->d_max);
        if (ctx->d_rgb8) (void)hipFree(ctx->d_rgb8);
        if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
        if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

rm_status rm_device_info(rm_ctx *ctx, char *name_buf, size_t buflen, int *n_cus, size_t *lds_bytes) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_device_info: NULL ctx");
    if (name_buf && buflen) std::snprintf(name_buf, buflen, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (n_cus) *n_cus = ctx->prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = ctx->prop.sharedMemPerBlock;
    return RM_OK;
}

static uint64_t pack_u32x2(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)hi << 32); }

static rm_status rm_scene_upload_impl(rm_ctx *ctx, const rm_scene_desc *d) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_scene_upload: NULL ctx");
    if (!d) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_scene_upload: NULL desc");
    if ((d->n_shapes && !d->shapes) || (d->n_spheres && !d->spheres) || (d->n_polygons && !d->polygons) ||
        (d->n_polygon_vertices && !d->polygon_vertices) || (d->n_triangles && !d->triangles) ||
        (d->n_lights && !d->lights))
        return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_scene_upload: NULL array with non-zero count");

    // The reference's hosts hand the whole Scene to every render() call (main.rs:331-333).  A
    // description byte-identical to the one the resident image was built from (cameras apart:
    // the camera travels as a kernel argument) needs no work at all.  Identity is decided on
    // the bytes themselves -- the context keeps a copy of the arrays it last built from and
    // compares in place, one pass, no hashing: a digest can collide, and a collision would
    // silently render the old scene.
    ctx->upload_calls++;
    const struct { const void *p; size_t bytes; } parts[6] = {
        {d->shapes, (size_t)d->n_shapes * sizeof(rm_shape_ref)},   {d->spheres, (size_t)d->n_spheres * sizeof(rm_sphere)},
        {d->polygons, (size_t)d->n_polygons * sizeof(rm_polygon)}, {d->polygon_vertices, (size_t)d->n_polygon_vertices * sizeof(rm_vec3)},
        {d->triangles, (size_t)d->n_triangles * sizeof(rm_triangle)}, {d->lights, (size_t)d->n_lights * sizeof(rm_light)}};
    auto same_description = [&]() {
        size_t off = 0;
        for (int i = 0; i < 6; i++) {
            if (ctx->desc_sizes[i] != parts[i].bytes) return false;
            if (parts[i].bytes && std::memcmp(ctx->desc_bytes.data() + off, parts[i].p, parts[i].bytes) != 0) return false;
            off += parts[i].bytes;
        }
        return true;
    };
    auto keep_description = [&]() {
        size_t total = 0;
        for (int i = 0; i < 6; i++) total += parts[i].bytes;
        ctx->desc_bytes.resize(total);
        size_t off = 0;
        for (int i = 0; i < 6; i++) {
            if (parts[i].bytes) std::memcpy(ctx->desc_bytes.data() + off, parts[i].p, parts[i].bytes);
            ctx->desc_sizes[i] = parts[i].bytes;
            off += parts[i].bytes;
        }
    };
    if (ctx->have_scene && same_description()) {
        ctx->camera = d->camera;
        return RM_OK;
    }

    // ---- regroup Scene.shapes by kind, remembering list order for ties ----
    std::vector<uint32_t> sphere_src, polygon_src, tri_src;   // indices into desc arrays
    std::vector<uint32_t> sphere_key, polygon_key, tri_key;   // ordinal in flattened list order
    uint32_t ordinal = 0;
    for (uint32_t i = 0; i < d->n_shapes; i++) {
        const rm_shape_ref &r = d->shapes[i];
        switch (r.kind) {
        case RM_SHAPE_SPHERE:
            if (r.first >= d->n_spheres || r.count != 1) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_scene_upload: bad sphere ref");
            sphere_src.push_back(r.first); sphere_key.push_back(ordinal++);
            break;
        case RM_SHAPE_POLYGON:
            if (r.first >= d->n_polygons || r.count != 1) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_scene_upload: bad polygon ref");
            polygon_src.push_back(r.first); polygon_key.push_back(ordinal++);
            break;
        case RM_SHAPE_MESH:
            if ((uint64_t)r.first + r.count > d->n_triangles) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_scene_upload: bad mesh ref");
            for (uint32_t t = 0; t < r.count; t++) { tri_src.push_back(r.first + t); tri_key.push_back(ordinal++); }
            break;
        default:
            return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_scene_upload: unknown shape kind");
        }
    }
    // ---- hierarchies over the spheres and the mesh triangles (rm_bvh.hpp): primitives of
    // a kind are re-ordered into leaf order; their list-order keys travel with them
    rm_bvh bvh_s, bvh_t;
    const bool use_bvh = !ctx->disable_bvh;
    if (use_bvh && sphere_src.size() >= RM_BVH_MIN_SPHERES) {
        std::vector<rm_aabb> boxes(sphere_src.size());
        for (size_t i = 0; i < sphere_src.size(); i++) {
            const rm_sphere &sp = d->spheres[sphere_src[i]];
            const double r = std::sqrt(sp.radius_square) * (1. + 1e-12);
            const double c[3] = {sp.center.x, sp.center.y, sp.center.z};
            for (int a = 0; a < 3; a++) { boxes[i].lo[a] = c[a] - r; boxes[i].hi[a] = c[a] + r; }
        }
        bvh_s = rm_build_bvh(boxes, 4);
        std::vector<uint32_t> src(sphere_src.size()), key(sphere_src.size());
        for (size_t k = 0; k < src.size(); k++) { src[k] = sphere_src[bvh_s.order[k]]; key[k] = sphere_key[bvh_s.order[k]]; }
        sphere_src.swap(src);
        sphere_key.swap(key);
    }
    if (use_bvh && tri_src.size() >= RM_BVH_MIN_TRIANGLES) {
        std::vector<rm_aabb> boxes(tri_src.size());
        for (size_t i = 0; i < tri_src.size(); i++) {
            const rm_triangle &t = d->triangles[tri_src[i]];
            boxes[i].reset();
            for (const rm_vec3 &v : t.vertices) {
                const double c[3] = {v.x, v.y, v.z};
                for (int a = 0; a < 3; a++) { boxes[i].lo[a] = std::min(boxes[i].lo[a], c[a]); boxes[i].hi[a] = std::max(boxes[i].hi[a], c[a]); }
            }
        }
        bvh_t = rm_build_bvh(boxes, 2);
        std::vector<uint32_t> src(tri_src.size()), key(tri_src.size());
        for (size_t k = 0; k < src.size(); k++) { src[k] = tri_src[bvh_t.order[k]]; key[k] = tri_key[bvh_t.order[k]]; }
        tri_src.swap(src);
        tri_key.swap(key);
    }

    std::vector<uint32_t> keys;
    keys.insert(keys.end(), sphere_key.begin(), sphere_key.end());
    keys.insert(keys.end(), polygon_key.begin(), polygon_key.end());
    keys.insert(keys.end(), tri_key.begin(), tri_key.end());
    bool ordered = true;
    for (size_t i = 1; i < keys.size(); i++) ordered = ordered && keys[i - 1] < keys[i];

    rm_dev_header H{};
    H.n_spheres = (uint32_t)sphere_src.size();
    H.n_polygons = (uint32_t)polygon_src.size();
    H.n_triangles = (uint32_t)tri_src.size();
    H.n_lights = d->n_lights;
    const uint32_t n_prims = H.n_spheres + H.n_polygons + H.n_triangles;
    H.list_ordered = ordered ? 1u : 0u;

    uint32_t n_pverts = 0;
    for (uint32_t src : polygon_src) {
        const rm_polygon &p = d->polygons[src];
        if (p.n_vertices < 3 || (uint64_t)p.first_vertex + p.n_vertices > d->n_polygon_vertices)
            return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_scene_upload: bad polygon vertex range");
        n_pverts += p.n_vertices;
    }

    // 32-bit word offsets: refuse scenes they cannot address
    const uint64_t need_words = (uint64_t)H.n_spheres * RM_SPHERE_WORDS + (uint64_t)H.n_polygons * RM_POLYGON_WORDS +
                                ((uint64_t)n_pverts + 1u) * RM_PVERT_WORDS + (uint64_t)H.n_triangles * RM_TRIANGLE_WORDS +
                                (uint64_t)n_prims * (RM_MATERIAL_WORDS + 1u + 4u + 16u + 1u) + (uint64_t)H.n_lights * RM_LIGHT_WORDS +
                                bvh_s.nodes.size() + bvh_t.nodes.size() + 256u;
    if (need_words > RM_SCENE_MAX_WORDS)
        return ctx_fail(ctx, RM_ERR_SCENE_LIMIT, "rm_scene_upload: scene exceeds the 32 GiB the device layout can address");
    uint32_t off = 0;
    auto take = [&](uint32_t words) { uint32_t o = off; off += (words + 1u) & ~1u; return o; };
    H.off_spheres = take(H.n_spheres * RM_SPHERE_WORDS);
    H.off_polygons = take(H.n_polygons * RM_POLYGON_WORDS);
    H.off_pverts = take((n_pverts + 1u) * RM_PVERT_WORDS);   // +1: the loops fetch four vertices at a time
    H.off_triangles = take(H.n_triangles * RM_TRIANGLE_WORDS);
    H.off_materials = take(n_prims * RM_MATERIAL_WORDS);
    H.off_lights = take(H.n_lights * RM_LIGHT_WORDS);
    H.off_keys = take((n_prims + 1u) / 2u);
    H.off_bounds = take(n_prims * 4u);
    H.off_planar = take((H.n_polygons + H.n_triangles) * 16u);
    const uint32_t n_groups = (n_prims + 63u) / 64u;
    H.off_groups = (n_groups >= 3u && n_groups <= 64u) ? take(n_groups * 4u) : 0u;
    // The wave's hierarchy stack holds 64 entries, one parked sibling per level: the builder
    // keeps every tree under RM_BVH_MAX_DEPTH levels (rm_bvh.hpp); a tree that is deeper all
    // the same is not walked (its primitives keep their leaf order and are walked flat).
    if (bvh_s.depth > RM_BVH_MAX_DEPTH) bvh_s.nodes.clear();
    if (bvh_t.depth > RM_BVH_MAX_DEPTH) bvh_t.nodes.clear();
    H.off_bvh_spheres = bvh_s.nodes.empty() ? 0u : take((uint32_t)bvh_s.nodes.size());
    H.off_bvh_triangles = bvh_t.nodes.empty() ? 0u : take((uint32_t)bvh_t.nodes.size());
    take(64u);                                               // batch loads may read past the last record
    H.total_words = off;

    std::vector<double> blob(H.total_words ? H.total_words : 2, 0.);
    auto put_material = [&](uint32_t pid, const rm_reflectance &r) {
        double *m = &blob[H.off_materials + RM_MATERIAL_WORDS * pid];
        m[0] = r.diffusion;
        m[1] = r.diffuse_color.x; m[2] = r.diffuse_color.y; m[3] = r.diffuse_color.z;
        m[4] = r.specular; m[5] = r.specular_exponent;
        m[6] = r.reflection; m[7] = r.refractive_index;
        m[8] = r.is_glass_like ? 1. : 0.;
        m[9] = 0.;
    };
    // Bounding sphere of everything of primitive `pid` a ray can hit, for the bundle cull
    // (rm_trace.inc): inflated by 1e-7 relative + 1e-9 of the coordinates' magnitude -- far
    // beyond the rounding of any hit test, so a primitive some ray hits is never culled.
    // Anything that is not a finite number makes the primitive a candidate for every bundle.
    double max_normal = 1.;                                   // sphere normals are unit (sphere.rs:58)
    auto put_bounds = [&](uint32_t pid, double cx, double cy, double cz, double r) {
        double *w = &blob[H.off_bounds + 4u * pid];
        const double mag = std::fabs(cx) + std::fabs(cy) + std::fabs(cz);
        double rr = r * (1. + 1e-7) + 1e-9 * (1. + mag);
        if (!(rr >= 0.) || !std::isfinite(rr) || !std::isfinite(mag)) { cx = cy = cz = 0.; rr = std::numeric_limits<double>::infinity(); }
        w[0] = cx; w[1] = cy; w[2] = cz; w[3] = rr;
    };
    // A planar primitive is hit where the ray meets the plane (point, normal) AND the x, y of
    // that point pass the 2-D edge tests (polygon.rs:54-56, triangle.rs:69-77), i.e. lie in
    // the convex hull of the vertices' x, y: the hit points are the hull of the vertices
    // LIFTED onto that plane along z -- the vertices themselves when they are coplanar with
    // it, as they are for everything the reference's constructors build.
    auto planar_bounds = [&](uint32_t pid, const rm_vec3 &n, const rm_vec3 &pp, const rm_vec3 *v, uint32_t nv) {
        max_normal = std::max(max_normal, std::sqrt(n.x * n.x + n.y * n.y + n.z * n.z));
        double *pl = &blob[H.off_planar + 16u * (pid - H.n_spheres)];   // zero-filled: count 0 = no edge test
        // The inside test reads only x and y (polygon.rs:54-56): with every vertex at the SAME x
        // (or the same y) its cross products are differences of the same rounded products, sum to
        // zero exactly and can never all be positive -- the primitive is never hit (the floor and
        // ceiling of the Cornell box, any wall along z).  Radius -1: the cull drops it outright.
        bool same_x = true, same_y = true;
        for (uint32_t i = 1; i < nv; i++) { same_x = same_x && v[i].x == v[0].x; same_y = same_y && v[i].y == v[0].y; }
        // Likewise two CONSECUTIVE vertices with the same x and the same y (a wall along z cut into triangles:
        // the red wall of the Cornell box): the cross product of that edge is x y' - y x' with (x, y) == (x', y')
        // bit for bit -- the same product twice, exactly zero, never > 0 -- for every hit point.
        bool twin_edge = false;
        for (uint32_t i = 0; i < nv; i++) {
            const rm_vec3 &p = v[i], &q = v[(i + 1u) % nv];
            twin_edge = twin_edge || (p.x == q.x && p.y == q.y);
        }
        if (same_x || same_y || twin_edge) {
            double *w = &blob[H.off_bounds + 4u * pid];
            w[0] = w[1] = w[2] = 0.; w[3] = -1.;
            return;
        }
        if (!(std::fabs(n.z) > 1e-12 * (std::fabs(n.x) + std::fabs(n.y) + std::fabs(n.z)))) {
            // plane along z: the x, y of its points are a line; no finite bound holds the lifted hull
            put_bounds(pid, 0., 0., 0., std::numeric_limits<double>::infinity());
            return;
        }
        std::vector<rm_vec3> lifted(nv);
        double cx = 0., cy = 0., cz = 0.;
        for (uint32_t i = 0; i < nv; i++) {
            const double z = pp.z - (n.x * (v[i].x - pp.x) + n.y * (v[i].y - pp.y)) / n.z;
            lifted[i] = rm_vec3{v[i].x, v[i].y, z};
            cx += v[i].x; cy += v[i].y; cz += z;
        }
        cx /= nv; cy /= nv; cz /= nv;
        double r2 = 0.;
        for (const rm_vec3 &q : lifted) r2 = std::max(r2, (q.x - cx) * (q.x - cx) + (q.y - cy) * (q.y - cy) + (q.z - cz) * (q.z - cz));
        put_bounds(pid, cx, cy, cz, std::sqrt(r2));
        // the lifted vertices for the cull's edge test (triangles and quads; a triangle
        // repeats its first vertex so that edge 2-3 closes it)
        bool finite = true;
        for (const rm_vec3 &q : lifted) finite = finite && std::isfinite(q.x) && std::isfinite(q.y) && std::isfinite(q.z);
        if (finite && (nv == 3u || nv == 4u)) {
            for (uint32_t i = 0; i < 4u; i++) {
                const rm_vec3 &q = lifted[i < nv ? i : 0u];
                pl[3 * i] = q.x; pl[3 * i + 1] = q.y; pl[3 * i + 2] = q.z;
            }
            pl[12] = (double)nv;
        }
    };
    uint32_t pid = 0;
    for (uint32_t i = 0; i < H.n_spheres; i++, pid++) {
        const rm_sphere &s = d->spheres[sphere_src[i]];
        double *w = &blob[H.off_spheres + RM_SPHERE_WORDS * i];
        w[0] = s.center.x; w[1] = s.center.y; w[2] = s.center.z; w[3] = s.radius_square;
        put_material(pid, s.reflectance);
        put_bounds(pid, s.center.x, s.center.y, s.center.z, std::sqrt(s.radius_square));
    }
    uint32_t pv = 0;
    for (uint32_t i = 0; i < H.n_polygons; i++, pid++) {
        const rm_polygon &p = d->polygons[polygon_src[i]];
        double *w = &blob[H.off_polygons + RM_POLYGON_WORDS * i];
        w[0] = p.plane_normal.x; w[1] = p.plane_normal.y; w[2] = p.plane_normal.z;
        w[3] = p.plane_point.x; w[4] = p.plane_point.y; w[5] = p.plane_point.z;
        const uint64_t packed = pack_u32x2(pv, p.n_vertices);
        std::memcpy(&w[6], &packed, sizeof packed);
        w[7] = 0.;
        for (uint32_t v = 0; v < p.n_vertices; v++, pv++) {
            const rm_vec3 &q = d->polygon_vertices[p.first_vertex + v];
            blob[H.off_pverts + RM_PVERT_WORDS * pv] = q.x;
            blob[H.off_pverts + RM_PVERT_WORDS * pv + 1] = q.y;
            // the first four also travel in the record: one fetch per polygon, not two dependent ones
            if (v < 4) { w[8 + 2 * v] = q.x; w[9 + 2 * v] = q.y; }
        }
        put_material(pid, p.reflectance);
        planar_bounds(pid, p.plane_normal, p.plane_point, &d->polygon_vertices[p.first_vertex], p.n_vertices);
    }
    for (uint32_t i = 0; i < H.n_triangles; i++, pid++) {
        const rm_triangle &t = d->triangles[tri_src[i]];
        double *w = &blob[H.off_triangles + RM_TRIANGLE_WORDS * i];
        w[0] = t.normal.x; w[1] = t.normal.y; w[2] = t.normal.z;
        w[3] = t.center.x; w[4] = t.center.y; w[5] = t.center.z;
        for (int v = 0; v < 3; v++) { w[6 + 2 * v] = t.vertices[v].x; w[7 + 2 * v] = t.vertices[v].y; }
        put_material(pid, t.reflectance);
        planar_bounds(pid, t.normal, t.center, t.vertices, 3u);
    }
    // The cull's first step in scenes of 3+ steps: a sphere around the bounding spheres of each 64
    // consecutive pids (box centre of the members; primitives that can never be hit -- radius -1 --
    // do not count, a group of nothing else is never visited).
    for (uint32_t g = 0; H.off_groups && g < n_groups; g++) {
        const uint32_t first = g * 64u, last = std::min(n_prims, first + 64u);
        double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        bool any = false, unbounded = false;
        for (uint32_t q = first; q < last; q++) {
            const double *w = &blob[H.off_bounds + 4u * q];
            if (w[3] < 0.) continue;
            any = true;
            if (!std::isfinite(w[3])) { unbounded = true; continue; }
            for (int c = 0; c < 3; c++) { lo[c] = std::min(lo[c], w[c] - w[3]); hi[c] = std::max(hi[c], w[c] + w[3]); }
        }
        double *o = &blob[H.off_groups + 4u * g];
        if (!any) { o[0] = o[1] = o[2] = 0.; o[3] = -1.; continue; }
        if (unbounded) { o[0] = o[1] = o[2] = 0.; o[3] = std::numeric_limits<double>::infinity(); continue; }
        const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
        double r = 0.;
        for (uint32_t q = first; q < last; q++) {
            const double *w = &blob[H.off_bounds + 4u * q];
            if (w[3] < 0.) continue;
            const double dx = w[0] - cx, dy = w[1] - cy, dz = w[2] - cz;
            r = std::max(r, std::sqrt(dx * dx + dy * dy + dz * dz) + w[3]);
        }
        const double mag = std::fabs(cx) + std::fabs(cy) + std::fabs(cz);
        double rr = r * (1. + 1e-9) + 1e-12 * (1. + mag);
        if (!std::isfinite(rr) || !std::isfinite(mag)) { o[0] = o[1] = o[2] = 0.; rr = std::numeric_limits<double>::infinity(); }
        else { o[0] = cx; o[1] = cy; o[2] = cz; }
        o[3] = rr;
    }
    // renderer.rs:168-172: a shadow ray starts 1e-3 of the normal off the hit point and runs
    // along normalize(light - point): it passes within 1e-3 |normal| of the light
    H.shadow_rho = 1e-3 * max_normal * (1. + 1e-6) + 1e-12;
    for (uint32_t l = 0; l < H.n_lights; l++) {
        const rm_light &lt = d->lights[l];
        double *w = &blob[H.off_lights + RM_LIGHT_WORDS * l];
        w[0] = lt.position.x; w[1] = lt.position.y; w[2] = lt.position.z;
        w[3] = lt.color.x; w[4] = lt.color.y; w[5] = lt.color.z;
        w[6] = lt.intensity; w[7] = 0.;
    }
    if (!keys.empty()) std::memcpy(&blob[H.off_keys], keys.data(), keys.size() * sizeof(uint32_t));
    if (H.off_bvh_spheres) std::memcpy(&blob[H.off_bvh_spheres], bvh_s.nodes.data(), bvh_s.nodes.size() * sizeof(double));
    if (H.off_bvh_triangles) std::memcpy(&blob[H.off_bvh_triangles], bvh_t.nodes.data(), bvh_t.nodes.size() * sizeof(double));

    // (a different description that builds the same device image -- an edit undone -- is not copied either)
    if (ctx->have_scene && blob == ctx->host_blob && std::memcmp(&H, &ctx->H, sizeof H) == 0) {
        ctx->camera = d->camera;
        keep_description();
        return RM_OK;
    }
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());   // a render (on any stream: the caller's, a frame slot's) may still be reading the old blob
    ctx->upload_copies++;
    if (ctx->d_scene_words < blob.size()) {
        if (ctx->d_scene) RM_HIP(ctx, hipFree(ctx->d_scene));
        ctx->d_scene = nullptr;
        RM_HIP(ctx, hipMalloc(&ctx->d_scene, blob.size() * sizeof(double)));
        ctx->d_scene_words = blob.size();
    }
    RM_HIP(ctx, hipMemcpy(ctx->d_scene, blob.data(), blob.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->H = H;
    ctx->camera = d->camera;
    ctx->have_scene = true;
    ctx->scene_epoch++;                                      // (the feedback of another scene's frames is void)
    // specular_pow<POW_INTEGER> applies when pow(x, y) is a plain integer power for every material
    bool int_exp = true;
    for (uint32_t q = 0; q < n_prims; q++) {
        const double y = blob[H.off_materials + RM_MATERIAL_WORDS * q + 5];
        int_exp = int_exp && (y >= 0. && y <= 1048576. && y == std::floor(y));
    }
    ctx->integer_exponents = int_exp;
    ctx->host_blob.swap(blob);
    keep_description();
    return RM_OK;
}

rm_status rm_scene_upload(rm_ctx *ctx, const rm_scene_desc *d) {
    return guarded(ctx, "rm_scene_upload", [&]() { return rm_scene_upload_impl(ctx, d); });
}

rm_status rm_scene_uploads(rm_ctx *ctx, uint64_t *calls, uint64_t *copies) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_scene_uploads: NULL ctx");
    if (calls) *calls = ctx->upload_calls;
    if (copies) *copies = ctx->upload_copies;
    return RM_OK;
}

rm_status rm_camera_update(rm_ctx *ctx, rm_vec3 camera) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_camera_update: NULL ctx");
    if (!ctx->have_scene) return ctx_fail(ctx, RM_ERR_NO_SCENE, "rm_camera_update: no scene uploaded");
    ctx->camera = camera;   // the camera travels as a kernel argument
    return RM_OK;
}

// The patch rows a call owns: begin, begin + stride, ... < end.
struct rm_band {
    uint32_t begin = 0, end = 0, stride = 1;
    uint32_t count() const { return end > begin ? (end - begin + stride - 1) / stride : 0; }
};

// Validates params and computes the band; shared by the render entry points.
static rm_status check_params(rm_ctx *ctx, const rm_params *p, rm_band *band) {
    if (!p) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "render: NULL params");
    if (!ctx->have_scene) return ctx_fail(ctx, RM_ERR_NO_SCENE, "render: no scene uploaded (rm_scene_upload)");
    if (p->patch_size != RM_PATCH_SIZE) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "render: patch_size must be 32 (renderer.rs:47)");
    if (p->max_depth > RM_MAX_DEPTH) return ctx_fail(ctx, RM_ERR_DEPTH, "render: max_depth above RM_MAX_DEPTH");
    if (p->frame_width == 0 || p->frame_height == 0) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "render: empty frame");
    if (p->frame_width % RM_PATCH_SIZE != 0)
        return ctx_fail(ctx, RM_ERR_DIMENSIONS,
                        "render: frame width is not a multiple of 32; the reference's scatter "
                        "(renderer.rs:92-108) indexes out of bounds and panics");
    const uint32_t n_height = p->frame_height / RM_PATCH_SIZE;   // renderer.rs:53: bottom H%32 rows never rendered
    uint32_t b = p->patch_row_begin, e = p->patch_row_end == 0 ? n_height : p->patch_row_end;
    if (e > n_height) e = n_height;
    if (b > e) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "render: patch_row_begin > patch_row_end");
    band->begin = b;
    band->end = e;
    band->stride = p->patch_row_stride == 0 ? 1u : p->patch_row_stride;
    return RM_OK;
}

// backproject (renderer.rs:128-135): `2 (x / width - 0.5) half_fov ratio` per column and
// `-2 (y / height - 0.5) half_fov` per row, tabulated with the operations the kernel used to
// perform per pixel, in their order (this file is compiled with -ffp-contract=off): the same
// IEEE results.  Rebuilt when the Renderer or the frame geometry changes.
static rm_status backproject_tables(rm_ctx *ctx, const rm_params *p) {
    const double key[6] = {p->width, p->height, p->half_fov, p->ratio, (double)p->frame_width, (double)p->frame_height};
    if (ctx->d_backproject && std::memcmp(key, ctx->backproject_key, sizeof key) == 0) return RM_OK;
    std::vector<double> t((size_t)p->frame_width + p->frame_height);
    for (uint32_t x = 0; x < p->frame_width; x++) t[x] = 2. * ((double)x / p->width - 0.5) * p->half_fov * p->ratio;
    for (uint32_t y = 0; y < p->frame_height; y++) t[p->frame_width + y] = -2. * ((double)y / p->height - 0.5) * p->half_fov;
    RM_HIP(ctx, hipDeviceSynchronize());                       // a launch in flight may still read the old tables
    if (ctx->backproject_words < t.size()) {
        if (ctx->d_backproject) RM_HIP(ctx, hipFree(ctx->d_backproject));
        ctx->d_backproject = nullptr;
        RM_HIP(ctx, hipMalloc(&ctx->d_backproject, t.size() * sizeof(double)));
        ctx->backproject_words = t.size();
    }
    RM_HIP(ctx, hipMemcpy(ctx->d_backproject, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    std::memcpy(ctx->backproject_key, key, sizeof key);
    return RM_OK;
}

static constexpr uint32_t RM_FEEDBACK_MIN_TILES = 32768, RM_FEEDBACK_STREAMS = 8;

// The feedback sets of launches of this geometry on this stream (created, or cleared when the
// geometry or the scene changed: the first frame then has no list).
static rm_status feedback_for(rm_ctx *ctx, hipStream_t stream, const uint64_t key[3], uint32_t n_tiles, rm_feedback **out) {
    rm_feedback *f = nullptr;
    for (rm_feedback &g : ctx->feedback)
        if (g.stream == stream) f = &g;
    if (!f) {
        if (ctx->feedback.size() >= RM_FEEDBACK_STREAMS) {         // forget the stream used longest ago
            size_t old = 0;
            for (size_t i = 1; i < ctx->feedback.size(); i++)
                if (ctx->feedback[i].used < ctx->feedback[old].used) old = i;
            if (ctx->feedback[old].block) RM_HIP(ctx, hipFree(ctx->feedback[old].block));   // (waits for the device)
            ctx->feedback.erase(ctx->feedback.begin() + (long)old);
        }
        ctx->feedback.emplace_back();
        f = &ctx->feedback.back();
        f->stream = stream;
    }
    f->used = ++ctx->feedback_clock;
    const bool same = f->block && f->n_tiles == n_tiles && std::memcmp(f->key, key, sizeof f->key) == 0;
    if (!same) {
        const uint32_t cap = std::max(64u, (n_tiles + 7u) / 8u);
        const size_t set_bytes = (((size_t)cap + RM_FB_BUCKETS + 4u) * sizeof(uint32_t) + n_tiles + 255u) & ~(size_t)255u;
        if (!f->block || f->set_bytes != set_bytes) {
            if (f->block) RM_HIP(ctx, hipFree(f->block));
            f->block = nullptr;
            RM_HIP(ctx, hipMalloc(&f->block, 3u * set_bytes + 256u));
        }
        f->set_bytes = set_bytes; f->cap = cap; f->n_tiles = n_tiles; f->cur = 0;
        std::memcpy(f->key, key, sizeof f->key);
        RM_HIP(ctx, hipMemsetAsync(f->block, 0, 3u * set_bytes + 256u, stream));
    }
    *out = f;
    return RM_OK;
}

// (launches of more than one round of wave slots -- 4,096 -- and a little: with the patches sorted by their longest tile the order
// pays from there on: 640x480, 4,800 tiles, 35.9 -> 31.9 us, 800x600 34.1 -> 31.0, a quarter of a 1080p frame 44.4 -> 34.8;
// 320x240, 1,120 tiles, which all start at once: 18.0 -> 19.3, left alone)
static constexpr uint32_t RM_CLASSIFY_STREAMS = 8;
// Launches of this many patches and more take the kernels with the patch order for the sky tail alone (order_by_place):
// measured 8K 987 -> 960 us; at 4K (8,100 patches) the sorting workgroup and the order's indirection cost what the tail saves
// (245.3 against 243.8 us).
static constexpr uint32_t RM_SKY_TAIL_BIG_MIN_PATCHES = 16384;   // (rm_ctx::sky_tail_big_min)
// What a lane of the classification spends on its share of a patch's primitives, in vector instructions: ~22
// per bounding sphere, ~110 more for the edge and plane tests of a planar primitive.  Beyond this the launch
// is not worth its time.
static constexpr uint32_t RM_CLASSIFY_MAX_COST = 4000;
// ... and at the head of the render launch itself, where every wave behind the first round may wait for it: a quarter of that
static constexpr uint32_t RM_CLASSIFY_IN_LAUNCH_MAX_COST = 1000;

static rm_status tile_lists_for(rm_ctx *ctx, hipStream_t stream, uint32_t n_tiles, rm_tile_lists **out) {
    rm_tile_lists *t = nullptr;
    for (rm_tile_lists &g : ctx->tile_lists)
        if (g.stream == stream) t = &g;
    if (!t) {
        if (ctx->tile_lists.size() >= RM_CLASSIFY_STREAMS) {       // forget the stream used longest ago
            size_t old = 0;
            for (size_t i = 1; i < ctx->tile_lists.size(); i++)
                if (ctx->tile_lists[i].used < ctx->tile_lists[old].used) old = i;
            if (ctx->tile_lists[old].block) RM_HIP(ctx, hipFree(ctx->tile_lists[old].block));   // (waits for the device)
            if (ctx->tile_lists[old].order_block) RM_HIP(ctx, hipFree(ctx->tile_lists[old].order_block));
            if (ctx->tile_lists[old].hint) RM_HIP(ctx, hipHostFree(ctx->tile_lists[old].hint));
            ctx->tile_lists.erase(ctx->tile_lists.begin() + (long)old);
        }
        ctx->tile_lists.emplace_back();
        t = &ctx->tile_lists.back();
        t->stream = stream;
    }
    t->used = ++ctx->feedback_clock;
    if (!t->block || t->cap < n_tiles) {
        if (t->block) RM_HIP(ctx, hipFree(t->block));             // (waits for the device: nothing reads the old masks any more)
        t->block = nullptr;
        t->cap = n_tiles;
        RM_HIP(ctx, hipMalloc(&t->block, (size_t)n_tiles * sizeof(unsigned long long)));
    }
    *out = t;
    return RM_OK;
}

// A wave that gives up a wait that cannot fail (the dispatch order of its own launch, 10 ms) says so in page-locked memory
// and renders nothing: that frame is void.  Reported once, by the first call that has waited for the device since -- the
// synchronous render calls and rm_frame_wait report the void frame itself -- or by the next launch on the stream.
static rm_status void_frame_check(rm_ctx *ctx, const char *who) {
    for (rm_tile_lists &t : ctx->tile_lists)
        if (t.hint) {
            const unsigned long long e = *(volatile unsigned long long *)(t.hint + 1);
            if (e) {
                *(volatile unsigned long long *)(t.hint + 1) = 0ull;
                return ctx_fail(ctx, RM_ERR_HIP, std::string(who) + ": launch " + std::to_string((uint32_t)(e >> 32)) +
                                                     " of its stream gave up waiting for its own classification (that frame is void)");
            }
        }
    return RM_OK;
}

// Which kernel instantiation a render with these params launches, and how.
struct rm_kernel_choice {
    const void *fn = nullptr;
    rm_launch_mode mode;
    size_t lds_bytes = 0;
    int stack = 0, pow_mode = 0;
    bool fast = false, staged = false, bvh = false, cull = false, edges = false, order = false, feedback = false;
    bool order_in_big_scene = false;  // a scene with a hierarchy whose launches are classified at their own head and dispatched by that
};

static rm_status choose_kernel(rm_ctx *ctx, const rm_params *p, uint32_t tiles, rm_kernel_choice *k) {
    // launch geometry: one tile per wave, one wave per workgroup; small scenes get an LDS copy
    // of the scene for the per-lane gathers, larger ones none
    const size_t scene_bytes = (size_t)ctx->H.total_words * sizeof(double);
    const uint32_t n_prims = ctx->H.n_spheres + ctx->H.n_polygons + ctx->H.n_triangles;
    k->bvh = ctx->H.off_bvh_spheres != 0 || ctx->H.off_bvh_triangles != 0;
    k->staged = scene_bytes <= RM_LDS_SCENE_LIMIT_BYTES && !ctx->force_unstaged && !k->bvh;
    // Bundle culling pays from about a dozen primitives on (a cull step costs about what two
    // primitive tests cost); the six primitives of the demo scene are walked as they are.
    k->cull = n_prims >= ctx->cull_min_prims || !k->staged;
    // One wave per workgroup in every kernel: the waves of a workgroup share nothing but the LDS
    // scene copy (which only small scenes get), and a wave slot a workgroup of four has freed is
    // handed on only when the whole workgroup fits -- with tiles of 1 to 18 ray steps that kept
    // 2.6 of a SIMD's 4 slots filled on the 256-sphere scene (1,777 -> 1,425 us with one wave).
    k->mode.waves = 1;
    k->mode.per_wave = 1;
    k->lds_bytes = ((k->staged ? (size_t)ctx->H.total_words : 0u) + (size_t)k->mode.waves * RM_WAVE_LDS_WORDS) * sizeof(double);

    // A lane parks at most one sibling per level below the cap: max_depth - 1 entries.
    k->stack = p->max_depth <= 5 ? 4 : p->max_depth <= 9 ? 8 : p->max_depth <= 17 ? 16 : 32;
    if (ctx->force_stack > k->stack && (ctx->force_stack == 8 || ctx->force_stack == 16 || ctx->force_stack == 32)) k->stack = ctx->force_stack;
    k->pow_mode = (ctx->integer_exponents && !ctx->force_generic_pow) ? POW_INTEGER : POW_GENERIC;
    k->fast = (p->flags & RM_FLAG_FAST_FP) != 0 || ctx->force_fast_fp;
    // the cull's edge test for planar primitives where there are several of them
    k->edges = k->cull && ctx->cull_edges && ctx->H.n_polygons + ctx->H.n_triangles >= RM_CULL_EDGES_MIN_PLANAR;
    const int st = k->stack, pw = k->pow_mode;
    const bool f = k->fast;
    // Feedback where tile costs have a long tail: deep ray trees in scenes with a hierarchy (a
    // step of incoherent rays through it costs thirty coherent ones) and launches long enough for
    // a tail to matter.  Elsewhere a tile costs its ray steps, the expensive rows are known (the
    // ground: dispatched first) and the bookkeeping only costs -- measured with it forced on: demo
    // scene 1080p 85.0 -> 87.7 us, 4K 306 -> 328, 8K depth 8 1,205 -> 1,375, Cornell box 72 -> 77.
    // RM_FEEDBACK=1 forces it for every launch of a kernel with the hierarchy walk, =0 switches it off.
    // r4: scenes with a hierarchy take the dispatch order from the launch's own classification instead where that can run
    // at the launch's head (patches timed by their longest tile, the sky tail on top): 256 spheres 4096x4096 1,174-1,185 ->
    // 1,160 us.  The tile-level feedback stays for what is left (RM_FEEDBACK=1 forces it).
    const uint32_t n_planar = ctx->H.n_polygons + ctx->H.n_triangles;
    k->order_in_big_scene = k->bvh && ctx->patch_order_mode != 0 && ctx->feedback_mode != 1 && !ctx->debug_empty && ctx->classify_mode != 0 &&
                            ctx->classify_in_launch && ctx->tile_order == TILE_ORDER_REVERSE && (22u * n_prims + 110u * n_planar) / 16u <= RM_CLASSIFY_IN_LAUNCH_MAX_COST &&
                            tiles / 16u <= ctx->patch_order_max_deep && (ctx->patch_order_mode == 1 || tiles >= ctx->classify_min_tiles);
    k->feedback = k->bvh && !k->order_in_big_scene && ctx->feedback_mode != 0 && !ctx->debug_empty &&
                  (ctx->feedback_mode == 1 || (p->max_depth >= 6u && tiles >= RM_FEEDBACK_MIN_TILES));
    // the dispatch order: launches of up to 4,096 patches that do not carry the tile-level feedback (launches of more patches
    // than that may take the same kernels for the sky tail alone -- RM_SKY_TAIL_BIG=1: by place, below; r4: measured to buy
    // nothing any more, 8K 988.9 against 985.6 us for the kernels without)
    k->order = k->order_in_big_scene ||
               (ctx->patch_order_mode != 0 && !k->feedback && !ctx->debug_empty && ctx->tile_order == TILE_ORDER_REVERSE &&
                (tiles / 16u <= ctx->patch_order_max || (ctx->sky_tail && ctx->sky_tail_big && n_prims <= 56u && tiles / 16u >= ctx->sky_tail_big_min)) &&
                (ctx->patch_order_mode == 1 || tiles >= ctx->classify_min_tiles));
    k->fn = rm_pick_kernel(f, k->staged, k->bvh, k->cull, k->edges, k->order ? 1 : 0, k->feedback, st, pw);
    if (!k->fn) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "render: no kernel for this scene / depth combination");
    return RM_OK;
}

static rm_status launch_render(rm_ctx *ctx, const rm_params *p, const rm_band &band, double *d_frame,
                               uint8_t *d_frame8, hipStream_t stream) {
    const uint32_t n_rows = band.count();
    if (n_rows == 0) return RM_OK;
    const uint32_t row_begin = band.begin;
    const uint32_t n_width = p->frame_width / RM_PATCH_SIZE;
    if (p->max_depth == 0) {
        // one fill per run of consecutive owned rows (a single run unless the band is strided)
        const uint32_t run = band.stride == 1 ? n_rows : 1u;
        for (uint32_t k = 0; k < n_rows; k += run) {
            const size_t first_px = (size_t)((p->flags & RM_FLAG_F64_COMPACT) ? k : row_begin + k * band.stride) * 32u * p->frame_width;
            const size_t n_px = (size_t)run * 32u * p->frame_width;
            hipLaunchKernelGGL(rm_fill_band_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, stream, d_frame,
                               first_px, n_px, p->background.x, p->background.y, p->background.z);
            RM_HIP(ctx, hipGetLastError());
        }
        return RM_OK;
    }
    rm_status bst = backproject_tables(ctx, p);
    if (bst != RM_OK) return bst;
    KernelArgs a{};
    a.H = ctx->H;
    a.bp_x = ctx->d_backproject;
    a.bp_y = ctx->d_backproject + p->frame_width;
    a.half_fov = p->half_fov; a.height = p->height; a.width = p->width; a.ratio = p->ratio;
    a.cam_x = ctx->camera.x; a.cam_y = ctx->camera.y; a.cam_z = ctx->camera.z;
    a.bg_x = p->background.x; a.bg_y = p->background.y; a.bg_z = p->background.z;
    a.frame_width = p->frame_width;
    a.patch_row_begin = row_begin;
    a.patch_row_stride = band.stride;
    a.u8_compact = (p->flags & RM_FLAG_U8_COMPACT) ? 1u : 0u;
    a.f64_compact = (p->flags & RM_FLAG_F64_COMPACT) ? 1u : 0u;
    a.max_depth = p->max_depth;
    a.n_width = n_width;
    a.n_tiles = n_rows * n_width * 16u;
    a.debug_stamps = nullptr;
    a.frame8 = d_frame8;
    // A cull step handles 64 primitives for ~25 vector instructions, ~110 when it holds planar
    // primitives and the edge test runs; a bundle pays for every step.  The hierarchy walk finds a
    // bundle's few primitives in a few hundred instructions whatever their number, so scenes whose
    // cull would cost more than that take the hierarchy only (2 = no bundle is ever narrow
    // enough).  Measured at 1080p: 36 triangles cull 67 us / hierarchy 110; 320 triangles
    // 190 / 119; 1,280 triangles 389 / 155; 256 spheres + 1 quad 1.76 ms / 2.77.
    const uint32_t n_prims_all = ctx->H.n_spheres + ctx->H.n_polygons + ctx->H.n_triangles;
    const uint32_t cull_steps = (n_prims_all + 63u) / 64u;
    const uint32_t planar_steps = (k_planar_from(ctx->H) < n_prims_all) ? cull_steps - k_planar_from(ctx->H) / 64u : 0u;
    const bool edges_on = ctx->H.n_polygons + ctx->H.n_triangles >= RM_CULL_EDGES_MIN_PLANAR;
    const uint32_t cull_cost = 25u * cull_steps + (edges_on ? 85u * planar_steps : 0u);
    a.cull_cos = cull_cost > RM_CULL_MAX_COST ? 2. : ctx->cull_cos;
    // dispatch order: tile = (id * order_mul + order_add) % n_tiles, a bijection
    a.order_mul = 1; a.order_add = 0;
    if (ctx->tile_order == TILE_ORDER_REVERSE && a.n_tiles > 1) {             // id -> n-1-id
        a.order_mul = a.n_tiles - 1; a.order_add = a.n_tiles - 1;
    } else if (ctx->tile_order == TILE_ORDER_HASH && a.n_tiles > 2) {
        uint32_t mul = (uint32_t)(a.n_tiles * 0.6180339887) | 1u;
        auto gcd = [](uint32_t x, uint32_t y) { while (y) { uint32_t t = x % y; x = y; y = t; } return x; };
        while (gcd(mul, a.n_tiles) != 1) mul += 2;
        a.order_mul = mul % a.n_tiles;
    }

    rm_kernel_choice k;
    rm_status st = choose_kernel(ctx, p, a.n_tiles, &k);
    if (st != RM_OK) return st;
    // the launch's first round -- the waves resident at once -- does not wait for its tiles' classification (Cornell, whose
    // edge-test kernel holds three waves to a SIMD: 31.5 -> 30.8 us with 3,072 instead of 4,096)
    a.first_round = (uint32_t)ctx->prop.multiProcessorCount * 4u * (k.edges ? RM_EDGES_WAVES : RM_MIN_WAVES);
    if (ctx->first_round >= 0) a.first_round = (uint32_t)ctx->first_round;     // (RM_FIRST_ROUND: A/B knob, and how the tests reach the order in frames of a few thousand tiles)
    const rm_launch_mode m = k.mode;
    size_t lds = k.lds_bytes;
    uint32_t cls_words_wanted = 0u;
    const dim3 block(m.waves * 64);
    const void *fn = k.fn;
    const uint32_t per_wg = (uint32_t)(m.waves * m.per_wave);
    dim3 grid((a.n_tiles + per_wg - 1) / per_wg);
    rm_feedback *fb = nullptr;
    const bool want_feedback = k.feedback;
    if (want_feedback && per_wg == 1u && !ctx->debug_empty) {
        const uint64_t key[3] = {(uint64_t)a.n_tiles | ((uint64_t)p->frame_width << 32),
                                 (uint64_t)row_begin | ((uint64_t)band.stride << 32), ctx->scene_epoch};
        rm_status fst = feedback_for(ctx, stream, key, a.n_tiles, &fb);
        if (fst != RM_OK) return fst;
        const int r = fb->cur, w = (fb->cur + 1) % 3, z = (fb->cur + 2) % 3;
        a.fb_list = fb->list(r); a.fb_count = fb->count(r); a.fb_flag = fb->flag(r); a.fb_hist = fb->hist(r);
        a.fb_next_list = fb->list(w); a.fb_next_count = fb->count(w); a.fb_next_flag = fb->flag(w); a.fb_next_hist = fb->hist(w);
        a.fb_zero = fb->hist(z);
        a.fb_threshold = fb->threshold();
        a.fb_cap = fb->cap;
        a.fb_long_ticks = ctx->feedback_us * 100u;            // s_memrealtime: 100 MHz
        // longest-first scheduling needs the longest tiles first, a few per wave slot: the rest fill in
        const uint32_t slots = (uint32_t)ctx->prop.multiProcessorCount * 16u;
        a.fb_target = ctx->feedback_target >= 0 ? (uint32_t)ctx->feedback_target : 4u * slots;
        a.fb_target = std::min(a.fb_target, fb->cap / 2u);
        grid.x = a.n_tiles + fb->cap;                         // ids [0, cap): the list; the rest: the tiles in order
    }
    if (ctx->debug_empty) a.n_tiles = 0;   // RM_DEBUG_EMPTY=1: same grid, every wave exits after staging
    rm_tile_lists tl_before{};           // the stream's lists as this launch found them (order_freeze)
    rm_tile_lists *tl_snapped = nullptr;
    bool frozen_launch = false;
    uint32_t mask_tag_before_ = 0u;      // the tag the previous launch on this stream gave its tiles' words (0: none that this launch could take)
    // Tile classification in front of the render launch (rm_classify.hip): tiles whose primary rays can hit
    // nothing are filled there and never get a wave; the others are listed, with the primitives their primary
    // rays can reach.  Worth a launch of its own from a few thousand tiles on, in scenes whose primitives a
    // lane can get through.  Only the primary rays are concerned: frames are bit-identical with it off.
    {
        const uint32_t n_planar = ctx->H.n_polygons + ctx->H.n_triangles;
        const uint32_t cost = (22u * n_prims_all + 110u * n_planar) / 16u;   // a lane's share of the patch step
        uint32_t &mask_tag_before = mask_tag_before_;
        const bool classify = ctx->classify_mode != 0 && !ctx->debug_empty && per_wg == 1u && n_prims_all > 0u &&
                              cost <= RM_CLASSIFY_MAX_COST && (ctx->classify_mode == 1 || a.n_tiles >= ctx->classify_min_tiles);
        ctx->last_launch_tiles = a.n_tiles;
        ctx->last_launch_classified = classify;
        ctx->last_launch_stream = stream;
        if (classify) {
            rm_tile_lists *tl = nullptr;
            rm_status cst = tile_lists_for(ctx, stream, a.n_tiles, &tl);
            if (cst != RM_OK) return cst;
            tl_before = *tl;
            tl_snapped = tl;
            // Scenes of up to 56 primitives are classified at the head of the render launch itself (its first
            // workgroups; the words carry the launch's tag): no launch of its own, no gap, and the classification
            // runs while the first round of tiles renders.  Measured at 1080p, demo scene: 82.6 us with the launch
            // in front, against 82.1 without any classification.  Larger scenes, and launches that carry the
            // frame-to-frame feedback, get the launch in front.
            // (RM_CLASSIFY_IN_LAUNCH_PRIMS: larger scenes too -- their words then only say whether there is anything to hit)
            const bool in_launch = ctx->classify_in_launch && (n_prims_all <= ctx->classify_in_launch_prims || k.order_in_big_scene) && !want_feedback;
            if (in_launch) {
                if (!tl->tagged || tl->tagged_tiles != a.n_tiles || tl->tagged_scene != ctx->scene_epoch || tl->tag >= 255u) {
                    // (a word is taken by its tag: after anything that could leave an old word with a tag in use, start afresh)
                    RM_HIP(ctx, hipMemsetAsync(tl->block, 0, (size_t)tl->cap * sizeof(unsigned long long), stream));
                    tl->tag = 0;
                    tl->tagged = true; tl->tagged_tiles = a.n_tiles; tl->tagged_scene = ctx->scene_epoch;
                }
                a.mask_tag = ++tl->tag;
                mask_tag_before = a.mask_tag > 1u ? a.mask_tag - 1u : 0u;     // (the previous launch's words are still there, under this tag)
                a.cls_blocks = (a.n_tiles / 16u + 3u) / 4u;
                a.cls_prims = n_prims_all;
                grid.x += a.cls_blocks;
                // Scenes too long for an LDS copy (the Cornell box: 36 triangles, 15 KB): the classifying workgroups pack what their
                // tests read -- bounds, lifted vertices, plane records: 4 n + 22 n_planar words -- into their LDS block, where there is
                // room for it at the kernel's occupancy (every workgroup of the launch is given the block: 16 to a CU, 12 in the
                // edge-test kernels, of 160 KB).  RM_CLASSIFY_LDS=0: from memory.
                // (decided where the launch's geometry is final, below: the workgroups' records lie behind the packed data)
                cls_words_wanted = (!k.staged && ctx->classify_lds) ? 4u * n_prims_all + 22u * n_planar : 0u;
            } else {
                tl->tagged = false;
                ClassifyArgs o{};
                o.tile_mask = tl->mask();
                o.n_prims = n_prims_all;
                // sixteen lanes to a 32x32 patch, four patches to a wave
                void *cargs[] = {(void *)&ctx->d_scene, (void *)&a, (void *)&o};
                RM_HIP(ctx, hipLaunchKernel(rm_classify_kernel(n_planar > 0u), dim3((a.n_tiles / 16u + 3u) / 4u), dim3(64), cargs, 0, stream));
            }
            a.tile_mask = tl->mask();
            a.mask_exact = n_prims_all <= (in_launch ? 56u : 64u) ? 1u : 0u;      // (a tagged word names 56 primitives, rm_classify.inc)
        }
    }
    // Dispatch order from the launch's own classification, and the sky tail (KernelArgs::ord_*; rm_classify.inc place_patch /
    // order_slot / order_places, rm_render_kernel.inc order_entry / sky_tail_patch).  The reference renders only after the camera has moved
    // (main.rs:74-78): an order by place from earlier frames is stale exactly then (r3: demo 1080p 68.5 us standing, 78.7 with a
    // press before every frame).  The classifying workgroups at the launch's head give every patch behind the first round
    // one of sixteen keys -- by its longest tile's time in the previous frame while the view stands still, by the cost of what
    // it can reach (learned per primitive from earlier frames: it moves with the picture) once it has moved, the sky last.
    // Only the order of dispatch and the launch's geometry depend on any of it: every tile of every frame is rendered in full by
    // the same code, exactly once.
    {
        const uint32_t n_patches = a.n_tiles / 16u;
        // (the waves that take their patches from the previous ranking and wait for no order: static_rounds times what is resident at
        // once -- those behind the first of them start when its tiles are done, their classification words are there by then)
        // (two rounds where the launch is at least four deep: a rank's share of a frame keeps one and its order.  Measured with a
        // press before every frame, 1 / 2 / 3 rounds: Cornell 38.9-40.0 / 37.2 / 38.8-38.9 us, standing 30.0-30.4 / 29.8-29.9 / 31.4;
        // demo within its noise)
        const uint64_t rounds = (uint64_t)a.first_round * ctx->static_rounds * 2u <= a.n_tiles ? ctx->static_rounds : 1u;
        a.n_static = (uint32_t)std::min<uint64_t>(((uint64_t)a.first_round * rounds) & ~15ull, a.n_tiles);
        const uint32_t n_dyn = n_patches - a.n_static / 16u;
        // (at most sixteen turns a classifying workgroup: its records -- 48 bytes a turn -- lie in its LDS block)
        const bool ordered = k.order && per_wg == 1u && a.cls_blocks != 0u && n_dyn > 0u && n_patches < (1u << RM_ORD_PATCH_BITS) &&
                             n_patches <= 16u * 4u * std::min<uint32_t>(RM_ORD_MAX_CLS, std::max(1u, ctx->cls_max_blocks));
        if (ordered) {
            rm_tile_lists *tl = nullptr;
            rm_status ost = tile_lists_for(ctx, stream, a.n_tiles, &tl);
            if (ost != RM_OK) return ost;
            const uint64_t key[3] = {(uint64_t)a.n_tiles | ((uint64_t)p->frame_width << 32), (uint64_t)row_begin | ((uint64_t)band.stride << 32),
                                     ctx->scene_epoch ^ ((uint64_t)a.n_static << 40)};
            if (!tl->order_block || tl->order_cap < n_patches) {
                if (tl->order_block) RM_HIP(ctx, hipFree(tl->order_block));     // (waits for the device)
                tl->order_block = nullptr;
                tl->order_cap = n_patches;
                RM_HIP(ctx, hipMalloc(&tl->order_block, rm_tile_lists::order_bytes(n_patches)));
                tl->order_key[0] = ~0ull;
            }
            // (an entry of the order is taken by its tag: another geometry or scene, or the tags used up -> start afresh)
            const uint32_t tag_wrap = ctx->ord_tag_wrap ? ctx->ord_tag_wrap : (1u << RM_ORD_TAG_BITS) - 1u;
            const bool fresh = std::memcmp(tl->order_key, key, sizeof key) != 0;
            if (fresh || tl->ord_tag >= tag_wrap) {
                if (fresh) {
                    RM_HIP(ctx, hipMemsetAsync(tl->order_block, 0, rm_tile_lists::order_bytes(tl->order_cap), stream));
                    std::memcpy(tl->order_key, key, sizeof key);
                    tl->order_frames = 0;
                } else {
                    RM_HIP(ctx, hipMemsetAsync(tl->flat(0), 0, 2u * (size_t)tl->order_cap * sizeof(uint32_t), stream));
                }
                if (fresh) { tl->static_read = tl->static_written = -1; tl->list_tag[0] = tl->list_tag[1] = 0u; }
                tl->ord_tag = 0;
                tl->last_tag = 0;
            }
            if (!tl->hint) {
                RM_HIP(ctx, hipHostMalloc((void **)&tl->hint, 2u * sizeof(unsigned long long), hipHostMallocDefault));
                tl->hint[0] = tl->hint[1] = 0ull;
            }
            // (a wave of an earlier launch gave up a wait that cannot fail: that frame is void -- said once)
            if (rm_status vst = void_frame_check(ctx, "render")) return vst;
            const uint32_t f = tl->order_frames++;
            const uint32_t seq = ++tl->seq;
            if (f == 0u) tl->key_seq0 = seq;
            const double view[7] = {ctx->camera.x, ctx->camera.y, ctx->camera.z, p->half_fov, p->height, p->width, p->ratio};
            if (f == 0u || std::memcmp(tl->view, view, sizeof view) != 0) {
                std::memcpy(tl->view, view, sizeof view);
                tl->view_seq0 = seq;
            }
            const bool timed = n_patches <= (k.order_in_big_scene ? ctx->patch_order_max_deep : ctx->patch_order_max);             // (larger launches are many rounds deep: by place, for the sky tail alone)
            // the classifying workgroups wait for each other: they must all be resident, whatever the kernel's occupancy --
            // at most 1,024 of them, each taking as many groups of four patches, one after the other, as that needs
            grid.x -= a.cls_blocks;
            const uint32_t groups4 = (n_patches + 3u) / 4u;
            const uint32_t cls_max = std::min<uint32_t>(RM_ORD_MAX_CLS, std::max(1u, ctx->cls_max_blocks));
            a.cls_iters = (groups4 + cls_max - 1u) / cls_max;
            a.cls_blocks = (groups4 + a.cls_iters - 1u) / a.cls_iters;
            a.ord_cnt = tl->cnt(f & 1u);
            a.ord_cnt_next = tl->cnt((f + 1u) & 1u);
            // While the view stands still a launch dispatches by the order its predecessor laid out -- same view, same
            // classification, the same first round: nothing to wait for -- and lays out the next launch's, from tile times a
            // frame fresher.  A view that has moved dispatches by its own order (the waves behind the first round wait for it).
            // (from the view's FOURTH launch on.  A launch that dispatches by its predecessor's order keeps its predecessor's first
            // round, and so do all after it: that first round had better be the view's dearest patches -- the first places of an
            // order sorted by this view's own tile times, which the view's second launch is the first to lay out and its third
            // the first to take its first round from.  Measured with the first round frozen a launch earlier, by place: a
            // quarter of the 1080p frame 45 us a frame against 34.5.)
            const bool reuse = ctx->order_reuse && f >= 3u && seq >= tl->view_seq0 + 3u && tl->last_tag != 0u;
            a.ord_flat = tl->flat(f & 1u);
            a.ord_cap = tl->order_cap;
            a.ord_tag = ++tl->ord_tag;
            a.ord_read = reuse ? tl->flat((f + 1u) & 1u) : a.ord_flat;
            a.ord_read_tag = reuse ? tl->last_tag : a.ord_tag;
            tl->last_tag = a.ord_tag;
            a.ord_rec = reuse && ctx->order_late_places ? tl->rec() : nullptr;
            // (same view as the launch before: its classification is this launch's)
            a.mask_tag_prev = reuse && ctx->mask_reuse ? mask_tag_before_ : 0u;
            a.patch_cost = timed ? tl->cost(f % 3u) : nullptr;
            a.cost_prev = timed ? tl->cost((f + 2u) % 3u) : nullptr;
            a.cost_zero = timed ? tl->cost((f + 1u) % 3u) : nullptr;
            a.ctab = tl->ctab((f + 2u) % 3u);
            a.ctab_cur = timed && ctx->order_keys != 1 ? tl->ctab(f % 3u) : nullptr;
            a.ctab_zero = tl->ctab((f + 1u) % 3u);
            // The first round: the first places of the order the previous launch laid out (its classifying workgroups wrote them
            // down) -- unless this launch dispatches by that very order: then it keeps its predecessor's first round, which that
            // order leaves out.  Every launch writes the first places of the order it lays out for whoever comes next.
            if (ctx->first_round_from_order) {
                const int read = reuse ? tl->static_read : tl->static_written;
                const uint32_t write = read == 0 ? 1u : 0u;
                a.static_list = read >= 0 ? tl->first((uint32_t)read) : nullptr;
                a.dyn_index = read >= 0 ? tl->index((uint32_t)read) : nullptr;
                a.dyn_inv = read >= 0 ? tl->inv((uint32_t)read) : nullptr;
                a.lists_done = read >= 0 ? tl->done() + read : nullptr;
                a.lists_tag = read >= 0 ? tl->list_tag[read] : 0u;
                a.static_next = tl->first(write);
                a.dyn_index_next = tl->index(write);
                a.dyn_inv_next = tl->inv(write);
                a.lists_done_next = tl->done() + write;
                tl->list_tag[write] = a.ord_tag;
                tl->static_read = read;
                tl->static_written = (int)write;
            } else {
                tl->static_read = tl->static_written = -1;
            }
            // what the patches are ordered by: the previous frame's times by place while the view is the one that frame had; else
            // the cost of what a patch can reach, once a table exists (written by the launch before from the launch before that)
            a.key_mode = !timed || f == 0u || ctx->order_keys == 0 ? RM_KEY_PLACE
                       : (seq > tl->view_seq0 && ctx->order_keys != 2) ? RM_KEY_COST
                       : (a.mask_exact && a.mask_tag && ctx->order_keys != 1) ? RM_KEY_CONTENT : RM_KEY_PLACE;
            a.ord_hint = tl->hint;
            a.err_word = tl->hint + 1;
            a.launch_seq = seq;
            a.test_stall = (uint32_t)ctx->test_stall_order;                     // (test hooks)
            // Sky tail.  The first classifying workgroup of every launch tells the host how many of the ordered patches had
            // something to hit (page-locked memory, read here without a wait).  The places behind them -- the sky -- get one wave
            // each instead of sixteen (the dispatcher takes ~0.7 ns per wave that finds out that its tile is sky: half of a
            // Cornell launch).  From a frame of THIS view the count is exact; from an earlier view it is a guess, and the places
            // it gets wrong -- the tail's first -- are rendered by sixteen waves each behind the grid's end.
            uint32_t tail = 0, cap = 0;
            if (ctx->sky_tail) {
                bool guess = false;
                if (ctx->sky_tail_force >= 0) {                                 // (test hook: a hint that is wrong)
                    tail = std::min((uint32_t)ctx->sky_tail_force, n_dyn);
                    guess = true;
                } else {
                    const unsigned long long h = *(volatile unsigned long long *)tl->hint;
                    const uint32_t h_seq = (uint32_t)(h >> 32), n_lit = (uint32_t)h;
                    const bool valid = h_seq >= tl->key_seq0 && h_seq < seq && h != 0ull && n_lit <= n_dyn;
                    guess = h_seq < tl->view_seq0 + 1u;                 // (the view's first launch may have had another first round)
                    if (valid && (!guess || ctx->sky_tail_motion)) tail = n_dyn - n_lit;
                    if (tail < 8u) tail = 0u;
                }
                // (room to hand on: a press of the reference's buttons turns a few hundred of a 1080p frame's 1,980 patches)
                // (a count from this very view is exact -- which patches went first does not change it -- but a place too many in the
                // tail with nobody to hand it to costs sixteen tiles one after the other: a little room all the same)
                // (a guess's room, a press before every frame, 512 / 768 / 1,024 places: demo 50.2 / 47.9 / 47.8 us, Cornell 37.1 / - / 38.2 --
                // the walk turns up to 700 of the demo's patches at a press; a place beyond the room costs sixteen tiles one after the
                // other, an empty place sixteen waves that look and leave.  Sizing the room from how wrong the stream's recent guesses
                // were was tried and is worse, 60-80 us: the shortfall is mostly small and now and then 500)
                if (tail) cap = std::min(tail, ctx->sky_tail_cap >= 0 ? (uint32_t)ctx->sky_tail_cap : guess ? std::max(768u, n_patches / ctx->sky_tail_room_div) : 32u);
            }
            a.tail_patches = tail;
            a.ov_cap = cap;
            grid.x = a.cls_blocks + a.n_static + 16u * (n_dyn - tail) + tail + 16u * cap + (a.ord_rec ? a.cls_blocks : 0u);
            // (dealt out evenly among the tile waves behind the launch's first round: rm_render_kernel.inc)
            const uint64_t behind = 16ull * (n_dyn - tail);
            a.tail_q = (tail && behind) ? (uint32_t)((((uint64_t)tail << 32) + behind + tail - 1u) / (behind + tail)) : 0u;
            // Launches of up to 4,096 patches: the tail BEHIND every tile wave instead (same box: Cornell 32.7 -> 31.4 us, demo
            // 69.2 -> 68.5 -- a tile with something to hit never waits for a slot behind a wave that only stores, and the
            // tail's stores, 24-43 MB, overlap the drain); an 8K launch ends with 380 MB of them if they wait: 960 -> 1,020 us.
            if (a.tail_q > 1u && (ctx->sky_tail_place == 2 || (ctx->sky_tail_place == 0 && timed))) a.tail_q = 1u;
            // A standing view, further: a launch that dispatches by its predecessor's order and takes its predecessor's
            // classification words computes, at its head, the very words and (but for a frame's noise in the tile times) the very
            // order its predecessor did.  Of order_freeze + 1 such launches only one does: the others are that launch less its
            // classifying workgroups and the workgroups that write the places -- same order read, same first round, same words
            // (all of them there: the predecessor is over), same tail -- and leave the stream's lists as they found them, so the
            // next launch that does classify is set up exactly as if they had not been.  Their tile times go into the same
            // counters (a maximum, a sum and a count: of two frames then).  What it spares: ~500 waves that hold a slot for 5-13 us
            // at the launch's start and ~500 short workgroups at its end.
            if (ctx->order_freeze > 0 && reuse && f >= 4u && tl_snapped == tl && tl_before.frozen_run < (uint32_t)ctx->order_freeze &&
                a.mask_tag > 1u && a.mask_tag_prev != 0u && a.ord_rec != nullptr && ctx->test_stall_order == 0 && ctx->sky_tail_force < 0 &&
                !std::getenv("RM_DEBUG_TAIL")) {
                frozen_launch = true;
                a.cls_blocks = 0u; a.cls_iters = 0u; a.cls_prims = 0u;
                a.ord_rec = nullptr; a.ord_cnt_next = nullptr;
                a.cost_zero = nullptr; a.ctab_zero = nullptr;
                a.static_next = nullptr; a.dyn_index_next = nullptr; a.dyn_inv_next = nullptr; a.lists_done_next = nullptr;
                a.mask_tag = tl_before.tag;                                     // (the words as the predecessor left them)
                a.mask_tag_prev = tl_before.tag;
                cls_words_wanted = 0u;
                grid.x = a.n_static + 16u * (n_dyn - tail) + tail + 16u * cap;
                // (the lists as they were; only the count of launches like this one moves)
                const uint32_t run = tl_before.frozen_run + 1u;
                tl->order_frames = tl_before.order_frames; tl->seq = tl_before.seq; tl->key_seq0 = tl_before.key_seq0;
                tl->view_seq0 = tl_before.view_seq0; tl->ord_tag = tl_before.ord_tag; tl->last_tag = tl_before.last_tag;
                tl->list_tag[0] = tl_before.list_tag[0]; tl->list_tag[1] = tl_before.list_tag[1];
                tl->static_read = tl_before.static_read; tl->static_written = tl_before.static_written;
                tl->tag = tl_before.tag; tl->tagged = tl_before.tagged; tl->tagged_tiles = tl_before.tagged_tiles; tl->tagged_scene = tl_before.tagged_scene;
                tl->frozen_run = run;
            } else {
                tl->frozen_run = 0u;
            }
        }
    }
    // RM_CULL_LDS=1: the bundle cull's arrays in every render wave's LDS block (scenes too long for a copy: the Cornell box), where
    // the block still lets all the kernel's waves be resident.  Measured (r4, as in r2): SLOWER -- Cornell 33.7-33.8 against
    // 32.9-33.1 us standing, 41.4-41.9 against 38.8-40.2 with the camera on the move: a wave packs 5.8 KB to spare three
    // round trips that its SIMD's other waves cover anyway.  Off.
    if (!k.staged && k.cull && ctx->cull_lds) {
        const uint32_t n_planar_ = ctx->H.n_polygons + ctx->H.n_triangles;
        const uint32_t cull_words = 4u * (ctx->H.n_spheres + n_planar_) + 16u * n_planar_;
        if (cull_words + (uint32_t)m.waves * RM_WAVE_LDS_WORDS <= (k.edges ? 1664u : 1248u)) {
            a.cull_lds_words = cull_words;
            lds = std::max(lds, (size_t)(cull_words + (uint32_t)m.waves * RM_WAVE_LDS_WORDS) * sizeof(double));
        }
    }
    if (cls_words_wanted) {
        // a record of three words per group of four patches and turn (OrdRec), behind the packed data; the block every workgroup
        // of the launch is given must still let the kernel's waves all be resident: 16 workgroups to a CU, 12 in the edge-test kernels
        const uint32_t rec_words = (std::max(a.cls_iters, 1u) * 4u * 3u + 1u) / 2u;
        if (cls_words_wanted + rec_words <= (k.edges ? 1664u : 1248u)) {
            a.cls_lds_words = cls_words_wanted;
            lds = std::max(lds, (size_t)(cls_words_wanted + rec_words) * sizeof(double));
        }
    }
#if defined(RM_EXP_STAMPS) || defined(RM_EXP_PHASES)
    unsigned long long *d_stamps = nullptr;
    const size_t n_waves = (size_t)grid.x * m.waves;
    RM_HIP(ctx, hipMalloc(&d_stamps, n_waves * 48));
    RM_HIP(ctx, hipMemsetAsync(d_stamps, 0, n_waves * 48, stream));
    a.debug_stamps = d_stamps;
#endif
    void *args[] = {(void *)&ctx->d_scene, (void *)&a, (void *)&d_frame};
    RM_HIP(ctx, hipLaunchKernel(fn, grid, block, args, lds, stream));
    ctx->last_launch_grid = grid.x;
    ctx->last_launch_tail = a.tail_patches;
    if (a.ord_cnt && std::getenv("RM_DEBUG_TAIL")) {                    // (diagnostic: waits for the launch)
        uint32_t c[RM_ORD_BUCKETS] = {}, c1[RM_ORD_BUCKETS] = {}, raw[RM_ORD_ARRIVE] = {};
        RM_HIP(ctx, hipStreamSynchronize(stream));
        RM_HIP(ctx, hipMemcpy(raw, a.ord_cnt, sizeof raw, hipMemcpyDeviceToHost));
        for (uint32_t b = 0; b < RM_ORD_BUCKETS; b++)
            for (uint32_t u = 0; u < RM_ORD_SUBS; u++) { c[b] += raw[(b * RM_ORD_SUBS + u) * RM_ORD_LINE]; c1[b] += raw[RM_ORD_FIRST + (b * RM_ORD_SUBS + u) * RM_ORD_LINE]; }
        uint32_t lit = 0;
        for (uint32_t b = 0; b < RM_ORD_SKY; b++) lit += c[b];
        std::fprintf(stderr, "[rm_order] launch %u keys %u: %u classifying workgroups x %u, first round %u waves, %u + %u sky places; tail %u, room to hand on %u, handed on %u; buckets", a.launch_seq, a.key_mode,
                     a.cls_blocks, a.cls_iters, a.n_static, lit, c[RM_ORD_SKY], a.tail_patches, a.ov_cap, a.tail_patches > c[RM_ORD_SKY] ? std::min(a.ov_cap, a.tail_patches - c[RM_ORD_SKY]) : 0u);
        for (uint32_t b = 0; b < RM_ORD_BUCKETS; b++) std::fprintf(stderr, " %u", c[b]);
        std::fprintf(stderr, " | first round's");
        for (uint32_t b = 0; b < RM_ORD_BUCKETS; b++) std::fprintf(stderr, " %u", c1[b]);
        // (the order this launch laid out: every place taken, by a patch of its own)
        const uint32_t n_pat = a.n_tiles / 16u, n_dyn_ = n_pat - a.n_static / 16u;
        std::vector<uint32_t> fl(n_dyn_);
        RM_HIP(ctx, hipMemcpy(fl.data(), a.ord_flat, n_dyn_ * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::vector<uint8_t> seen(n_pat, 0);
        uint32_t untagged = 0, twice = 0, bad = 0;
        for (uint32_t v : fl) {
            if ((v >> RM_ORD_TAG_SHIFT) != a.ord_tag) { untagged++; continue; }
            const uint32_t pch = v & (RM_ORD_SKY_BIT - 1u);
            if (pch >= n_pat) { bad++; continue; }
            twice += seen[pch]; seen[pch] = 1;
        }
        std::fprintf(stderr, " | order laid out: %u places, %u without the tag, %u patches twice, %u out of range; dispatched by %s order; places written %s\n", n_dyn_, untagged, twice, bad,
                     a.ord_read == a.ord_flat ? "its own" : "its predecessor's", a.ord_rec ? "at the grid's end" : "by the classifying workgroups");
    }
    if (fb) fb->cur = (fb->cur + 1) % 3;
#if defined(RM_EXP_STAMPS) || defined(RM_EXP_PHASES)
    RM_HIP(ctx, hipStreamSynchronize(stream));
    if (const char *path = std::getenv("RM_DEBUG_STAMPS")) {
        std::fprintf(stderr, "stamps: grid %u cls_blocks %u n_static %u n_tiles %u tail_patches %u tail_q %u ov_cap %u key_mode %u\n", grid.x, a.cls_blocks, a.ord_cnt ? a.n_static : 0u, a.n_tiles, a.tail_patches, a.tail_q, a.ov_cap, a.key_mode);
        std::vector<unsigned long long> h(n_waves * 4);
        RM_HIP(ctx, hipMemcpy(h.data(), d_stamps, n_waves * 32, hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(path, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
        // (<path>.ext: per wave the tile it rendered and the tile's classification word)
        std::vector<unsigned long long> x(n_waves * 2);
        RM_HIP(ctx, hipMemcpy(x.data(), d_stamps + n_waves * 4, n_waves * 16, hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen((std::string(path) + ".ext").c_str(), "wb")) { std::fwrite(x.data(), 8, x.size(), f); std::fclose(f); }
    }
    RM_HIP(ctx, hipFree(d_stamps));
#endif
    return RM_OK;
}

static rm_status rm_render_device_impl(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *hip_stream) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_render_device: NULL ctx");
    if (!device_rgb) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_render_device: NULL device buffer");
    rm_band band;
    rm_status st = check_params(ctx, params, &band);
    if (st != RM_OK) return st;
    RM_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;   // NULL is HIP's default stream, as in any HIP API
    return launch_render(ctx, params, band, (double *)device_rgb, nullptr, s);
}

rm_status rm_render_device(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *hip_stream) {
    return guarded(ctx, "rm_render_device", [&]() { return rm_render_device_impl(ctx, params, device_rgb, hip_stream); });
}

static rm_status rm_render_device_u8_impl(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *device_rgb8,
                              void *hip_stream) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_render_device_u8: NULL ctx");
    if (!device_rgb || !device_rgb8) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_render_device_u8: NULL device buffer");
    rm_band band;
    rm_status st = check_params(ctx, params, &band);
    if (st != RM_OK) return st;
    if (params->max_depth == 0)
        return ctx_fail(ctx, RM_ERR_DEPTH, "rm_render_device_u8: max_depth 0 renders no ray; use rm_render_device + rm_postprocess");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    return launch_render(ctx, params, band, (double *)device_rgb, (uint8_t *)device_rgb8, (hipStream_t)hip_stream);
}

rm_status rm_render_device_u8(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *device_rgb8,
                              void *hip_stream) {
    return guarded(ctx, "rm_render_device_u8", [&]() { return rm_render_device_u8_impl(ctx, params, device_rgb, device_rgb8, hip_stream); });
}

static rm_status rm_render_impl(rm_ctx *ctx, const rm_params *params, double *host_rgb, rm_timing *timing) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_render: NULL ctx");
    const auto t_begin = std::chrono::steady_clock::now();
    rm_band band;
    rm_status st = check_params(ctx, params, &band);
    if (st != RM_OK) return st;
    if (params->flags & (RM_FLAG_F64_COMPACT | RM_FLAG_U8_COMPACT))
        return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_render: the packed layouts are for rm_render_device* (caller-owned device buffers)");
    RM_HIP(ctx, hipSetDevice(ctx->device));

    const size_t need = (size_t)params->frame_width * params->frame_height * 3u * sizeof(double);
    if (ctx->frame_bytes != need || ctx->frame_w != params->frame_width || ctx->frame_h != params->frame_height) {
        RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_frame) RM_HIP(ctx, hipFree(ctx->d_frame));
        ctx->d_frame = nullptr;
        ctx->frame_bytes = 0;
        RM_HIP(ctx, hipMalloc(&ctx->d_frame, need));
        // create_frame_buffer zero-fills (framebuffer.rs:12-22)
        RM_HIP(ctx, hipMemsetAsync(ctx->d_frame, 0, need, ctx->stream));
        ctx->frame_bytes = need;
        ctx->frame_w = params->frame_width;
        ctx->frame_h = params->frame_height;
    }

    // One launch, then the copy.  The copy IS the call: 48.7 MB of f64 at 1080p take 0.87-0.93 ms
    // at the 52-56 GB/s this PCIe link delivers device -> host, the kernel 0.08 ms.  Rendering
    // in sub-bands and copying each while the next renders was measured and dropped
    // (profiles/r02_host_copy.txt: 0.97-1.04 ms either way).  What does help is not sending the
    // black patches (rm_hostio.inc): frames of 2 MB and more take that path, into flat memory as
    // into rows of rows; smaller ones are copied as they are.
    const size_t row_bytes = (size_t)params->frame_width * 3u * sizeof(double);
    const uint32_t n_rows = band.count();
    if (host_rgb && n_rows > 0 && hostio_packs(ctx, (size_t)n_rows * 32u * row_bytes)) {
        std::vector<double *> rows(params->frame_height);
        for (uint32_t y = 0; y < params->frame_height; y++) rows[y] = host_rgb + (size_t)y * params->frame_width * 3u;
        return rm_render_rows(ctx, params, rows.data(), timing);
    }
    RM_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    st = launch_render(ctx, params, band, ctx->d_frame, nullptr, ctx->stream);
    if (st != RM_OK) return st;
    RM_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));

    double d2h_ms = 0.;
    if (host_rgb && n_rows > 0) {
        // Only the owned rows are copied: rows below the last whole patch row keep the
        // caller's previous contents, as in the reference (renderer.rs:53).
        RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const auto t0 = std::chrono::steady_clock::now();
        const uint32_t run = band.stride == 1 ? n_rows : 1u;               // consecutive owned patch rows per copy
        for (uint32_t k = 0; k < n_rows; k += run) {
            const size_t off = (size_t)(band.begin + k * band.stride) * 32u * row_bytes;
            RM_HIP(ctx, hipMemcpy((char *)host_rgb + off, (const char *)ctx->d_frame + off, (size_t)run * 32u * row_bytes,
                                  hipMemcpyDeviceToHost));
        }
        d2h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else {
        RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (rm_status vst = void_frame_check(ctx, "rm_render")) return vst;
    if (timing) {
        float ms = 0.f;
        RM_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        timing->kernel_ms = ms;
        timing->d2h_ms = d2h_ms;
        timing->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    }
    return RM_OK;
}

rm_status rm_render(rm_ctx *ctx, const rm_params *params, double *host_rgb, rm_timing *timing) {
    return guarded(ctx, "rm_render", [&]() { return rm_render_impl(ctx, params, host_rgb, timing); });
}

rm_status rm_kernel_name(rm_ctx *ctx, const rm_params *params, char *buf, size_t buflen) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_kernel_name: NULL ctx");
    if (!buf || buflen == 0) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_kernel_name: NULL buffer");
    rm_band band;
    rm_status st = check_params(ctx, params, &band);
    if (st != RM_OK) return st;
    if (params->max_depth == 0) { std::snprintf(buf, buflen, "rm_fill_band_kernel"); return RM_OK; }
    rm_kernel_choice k;
    st = choose_kernel(ctx, params, band.count() * (params->frame_width / RM_PATCH_SIZE) * 16u, &k);
    if (st != RM_OK) return st;
    // the name rocprofv3's kernel trace shows (template arguments in declaration order)
    std::snprintf(buf, buflen, "%s::rm_render_static<%d, %d, %d, %d, %s, %s, %s, %s, %s, %s>", k.fast ? "rmdev_fast" : "rmdev_strict", k.stack,
                  k.pow_mode, k.mode.waves, k.mode.per_wave, k.staged ? "true" : "false", k.bvh ? "true" : "false",
                  k.cull ? "true" : "false", k.edges ? "true" : "false", k.order ? "true" : "false", k.feedback ? "true" : "false");
    return RM_OK;
}

rm_status rm_launch_stats(rm_ctx *ctx, uint32_t *workgroups, uint32_t *tail_patches) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_launch_stats: NULL ctx");
    if (workgroups) *workgroups = ctx->last_launch_grid;
    if (tail_patches) *tail_patches = ctx->last_launch_tail;
    return RM_OK;
}

static rm_status rm_tile_stats_impl(rm_ctx *ctx, void *hip_stream, uint32_t *tiles, uint32_t *tiles_listed) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_tile_stats: NULL ctx");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    RM_HIP(ctx, hipDeviceSynchronize());
    const uint32_t n = ctx->last_launch_tiles;
    uint32_t listed = n;
    for (const rm_tile_lists &t : ctx->tile_lists)
        if (t.stream == (hip_stream ? (hipStream_t)hip_stream : ctx->last_launch_stream) && t.block && ctx->last_launch_classified && n <= t.cap) {
            std::vector<unsigned long long> m(n);
            RM_HIP(ctx, hipMemcpy(m.data(), t.mask(), (size_t)n * 8u, hipMemcpyDeviceToHost));
            listed = 0;
            uint32_t hist[64] = {};
            uint64_t bits = 0;
            for (unsigned long long v : m) {
                if (t.tagged) v &= 0x00FFFFFFFFFFFFFFull;            // (classified at the head of the launch: the top byte is its tag)
                listed += v != 0ull;
                for (int b = 0; b < 64 && v != ~0ull; b++)
                    if (v >> b & 1ull) { hist[b]++; bits++; }
            }
            if (std::getenv("RM_DEBUG_CLASSIFY")) {
                std::fprintf(stderr, "[rm_classify] %u of %u tiles have something to hit, %.2f primitives each; tiles per pid:", listed, n,
                             listed ? (double)bits / listed : 0.);
                for (int b = 0; b < 64; b++)
                    if (hist[b]) std::fprintf(stderr, " %d:%u", b, hist[b]);
                std::fprintf(stderr, "\n");
            }
        }
    if (tiles) *tiles = n;
    if (tiles_listed) *tiles_listed = listed;
    return RM_OK;
}

rm_status rm_tile_stats(rm_ctx *ctx, void *hip_stream, uint32_t *tiles, uint32_t *tiles_listed) {
    return guarded(ctx, "rm_tile_stats", [&]() { return rm_tile_stats_impl(ctx, hip_stream, tiles, tiles_listed); });
}

rm_status rm_device_framebuffer(rm_ctx *ctx, void **device_rgb, size_t *bytes) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_device_framebuffer: NULL ctx");
    if (!ctx->d_frame) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_device_framebuffer: nothing rendered yet");
    if (device_rgb) *device_rgb = ctx->d_frame;
    if (bytes) *bytes = ctx->frame_bytes;
    return RM_OK;
}

static rm_status rm_postprocess_impl(rm_ctx *ctx, void *device_rgb, uint32_t w, uint32_t h, int normalize, uint8_t *host_rgb8,
                         double *max_out) {
    if (!ctx) return ctx_fail(nullptr, RM_ERR_INVALID_ARG, "rm_postprocess: NULL ctx");
    RM_HIP(ctx, hipSetDevice(ctx->device));
    double *v = (double *)device_rgb;
    if (!v) {
        if (!ctx->d_frame) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_postprocess: nothing rendered yet");
        if (w != ctx->frame_w || h != ctx->frame_h)
            return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_postprocess: size differs from the rendered frame");
        v = ctx->d_frame;
    }
    const size_t n = (size_t)w * h * 3u;
    if (n == 0) return ctx_fail(ctx, RM_ERR_INVALID_ARG, "rm_postprocess: empty frame");
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    RM_HIP(ctx, hipMemsetAsync(ctx->d_max, 0, sizeof(unsigned long long), ctx->stream));
    if (normalize) {
        hipLaunchKernelGGL(rm_max_kernel, dim3(blocks), dim3(256), 0, ctx->stream, v, n, ctx->d_max);
        RM_HIP(ctx, hipGetLastError());
    }
    uint8_t *d8 = nullptr;
    if (host_rgb8) {
        if (ctx->rgb8_bytes < n) {
            RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_rgb8) RM_HIP(ctx, hipFree(ctx->d_rgb8));
            ctx->d_rgb8 = nullptr;
            ctx->rgb8_bytes = 0;
            RM_HIP(ctx, hipMalloc(&ctx->d_rgb8, n));
            ctx->rgb8_bytes = n;
        }
        d8 = ctx->d_rgb8;
    }
    if (normalize || d8) {
        hipLaunchKernelGGL(rm_scale_quantize_kernel, dim3(blocks), dim3(256), 0, ctx->stream, v, n, ctx->d_max,
                           normalize ? 1 : 0, d8);
        RM_HIP(ctx, hipGetLastError());
    }
    RM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (host_rgb8) RM_HIP(ctx, hipMemcpy(host_rgb8, d8, n, hipMemcpyDeviceToHost));
    if (max_out) {
        unsigned long long bits = 0;
        RM_HIP(ctx, hipMemcpy(&bits, ctx->d_max, sizeof bits, hipMemcpyDeviceToHost));
        std::memcpy(max_out, &bits, sizeof bits);
    }
    return RM_OK;
}

rm_status rm_postprocess(rm_ctx *ctx, void *device_rgb, uint32_t w, uint32_t h, int normalize, uint8_t *host_rgb8,
                         double *max_out) {
    return guarded(ctx, "rm_postprocess", [&]() { return rm_postprocess_impl(ctx, device_rgb, w, h, normalize, host_rgb8, max_out); });
}

}  // extern "C"

#include "rm_exchange.inc"
#include "rm_hostio.inc"
