// rm_kernel_args.hpp -- what the host side needs to know about the render kernels: their
// argument block, the per-wave LDS layout, the feedback histogram's shape, and the function that
// hands out a kernel.  No device code (rm_render_kernel.hpp has that).
#ifndef RM_KERNEL_ARGS_HPP
#define RM_KERNEL_ARGS_HPP

#include <stdint.h>

#include "rm_internal.h"

#ifndef RM_BVH4
#define RM_BVH4 0
#endif
#if RM_BVH4
#define RM_BVH_NODE_WORDS 32u
#else
#define RM_BVH_NODE_WORDS 16u
#endif
// dispatch order built at the launch's head (KernelArgs::ord_*): sixteen buckets, the last one the sky
#define RM_ORD_BUCKETS 16u
#define RM_ORD_SKY 15u
#define RM_ORD_MAX_CLS 1024u            /* classifying workgroups of a launch with an order: all resident, whatever the kernel's occupancy */
/* the launch's counters, each in a 128-byte line of its own (waves that add to or ask for one word must not queue behind those of
   another): [b] ordered patches in bucket b | [16 + b] first-round patches in bucket b | [32 + g] classifying workgroups of the g-th 32 that
   have arrived | groups complete | go */
#define RM_ORD_LINE 32u
/* every bucket counts into RM_ORD_SUBS counters, a classifying workgroup into the one of its number: the sky's counter took ~400
   atomics of a launch's 495 workgroups, which take their turns at ~70 a microsecond; four lines a bucket, one lane a counter */
#define RM_ORD_SUBS 4u
#define RM_ORD_COUNTERS (RM_ORD_BUCKETS * RM_ORD_SUBS)
#define RM_ORD_FIRST (RM_ORD_COUNTERS * RM_ORD_LINE)         /* the same again: patches of the FIRST ROUND by bucket (for the next launch's first round) */
#define RM_ORD_ARRIVE (2u * RM_ORD_COUNTERS * RM_ORD_LINE)
#define RM_ORD_GROUPS (RM_ORD_ARRIVE + (RM_ORD_MAX_CLS / 32u) * RM_ORD_LINE)
#define RM_ORD_GO (RM_ORD_GROUPS + RM_ORD_LINE)
#define RM_ORD_CNT_WORDS (RM_ORD_GO + RM_ORD_LINE)
#define RM_ORD_STATIC_BIT 0x80000000u
/* the launch's decision word (RM_ORD_GO): 0 undecided | its tag: every classifying workgroup has arrived, the places are being written |
   tag + this bit: a wave gave up waiting before that -- nobody writes places, every wave takes its patch from the index instead */
#define RM_ORD_FALLBACK_BIT 0x80000000u
#define RM_ORD_PATIENCE 4096u           /* polls (~1 us each) before a waiting wave asks for the order of last resort */
#define RM_ORD_PATCH_BITS 20u           /* an entry: patch | sky << 20 | tag << 21 */
#define RM_ORD_SKY_BIT (1u << RM_ORD_PATCH_BITS)
#define RM_ORD_TAG_SHIFT (RM_ORD_PATCH_BITS + 1u)
#define RM_ORD_TAG_BITS 11u
#define RM_CTAB_WORDS 128u             /* sums | counts */
enum { RM_KEY_PLACE = 0, RM_KEY_COST = 1, RM_KEY_CONTENT = 2 };

// waves per SIMD the integer-power kernels are compiled for (register budget 512 / this)
#ifndef RM_MIN_WAVES
#define RM_MIN_WAVES 4
#endif
// ... and the kernels with the cull's edge test (at 4 they spill 19 values around the cull step)
#ifndef RM_EDGES_WAVES
#define RM_EDGES_WAVES 3
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
    // Tile classification (rm_classify.hip): a launch of one lane per tile has tested the cone of each tile's
    // primary rays against the primitives' bounds.  tile_mask[tile] == 0: no primary ray of the tile can hit
    // anything -- its wave stores the primary-miss value (zeros, renderer.rs:305) and leaves; else, in scenes
    // of up to 64 primitives (mask_exact), the primitives (bit = pid) the tile's primary rays can reach.
    // NULL: off.
    const unsigned long long *tile_mask;
    uint32_t mask_exact;
    // The classification at the head of THIS launch (scenes of up to 56 primitives): its first cls_blocks
    // workgroups classify -- four patches each, in the order the tiles are dispatched -- and write
    // (mask_tag << 56) | mask; a render wave takes its tile's word once it carries this launch's tag.
    // mask_tag == 0: the masks come from a launch of their own, ahead of this one.
    uint32_t mask_tag;
    // While the view stands still the previous launch's words are as good (same view, same classification): mask_tag_prev != 0
    // says so, and every wave -- the first round's too -- takes its tile's word as it finds it.
    uint32_t mask_tag_prev;
    uint32_t cls_blocks;
    uint32_t cls_prims;
    uint32_t cls_lds_words;                  // != 0: the classifying workgroups pack what their tests read into their LDS block first (rm_classify.inc cls_stage), this many words
    uint32_t cull_lds_words;                 // != 0: every render wave packs the bundle cull's arrays (bounds, lifted vertices: 4 n + 16 n_planar words) into its LDS block, in front of its own words
    // Dispatch order from THIS launch's classification (rm_classify.inc `order_patches`, rm_render_kernel.inc `order_entry`).
    // The reference renders only after the camera has moved (main.rs:74-78), so an order kept from earlier frames BY PLACE is
    // stale exactly when it is needed.  Instead the classifying workgroups at the launch's head, which know what each patch's
    // primary rays can reach, put every patch behind the launch's first round into one of sixteen buckets -- by what the
    // patch's longest tile cost in the previous frame while the view stands still, else by what tiles reaching the same
    // primitives cost in the previous frame (cost by content: it moves with the picture), the sky last: one atomic per wave
    // and bucket gives its patches their slots.  The classifying workgroups -- at most 1,024, all resident -- then wait for
    // each other (the last of each 32 to arrive tells the launch, the last of those says go), read the sixteen totals and write their own patches' places into
    // THE ORDER (ord_flat); a render wave behind the first round takes the k-th entry of that: one load, the entry says
    // itself when it is there.  Nothing serial anywhere.  The first round -- the waves resident at once -- renders the first
    // places of the PREVIOUS launch's order (its dearest patches; the bottom rows where there is none) and waits for nobody.
    // Only the order of dispatch depends on any of it.  ord_cnt == NULL: off.
    uint32_t *ord_cnt;                       // this launch's counters (RM_ORD_CNT_WORDS: patches per bucket, classifying workgroups arrived, go)
    uint32_t *ord_cnt_next;                  // the next launch's block: cleared by this launch's first classifying workgroup
    uint32_t *ord_flat;                      // ord_cap entries, the order this launch lays out: patch | sky << 20 | ord_tag << 21
    // ... and the order it DISPATCHES by: its own (a view that has moved: the waves behind the first round wait for it, 10-20 us
    // into the launch) -- or, while the view stands still, the one the previous launch laid out (same view, same classification,
    // the same first round: nothing to wait for), this launch's then being the next one's, from tile times a frame fresher
    const uint32_t *ord_read;
    uint32_t ord_read_tag;
    // (... in which case nobody waits for the order being laid out: the classifying workgroups leave their patches' slots in
    // memory and go -- 495 wave slots held for the slowest of them are 3 us of a Cornell launch --, and as many workgroups
    // at the grid's very end, long after, write the places)
    uint32_t *ord_rec;                       // 3 words per patch group slot: patch, bucket << 24 | slot, dyn_index's word (NULL: the classifying workgroups write the places themselves)
    uint32_t ord_cap;                        // entries (the patches behind the first round, at most)
    uint32_t ord_tag;                        // 1..2047: an entry is there once it carries this launch's tag
    uint32_t cls_iters;                      // groups of four patches a classifying workgroup takes, one after the other (so that they are at most 1,024)
    const uint32_t *static_list;             // n_static / 16 patches: the first round (NULL: the bottom rows by place)
    const uint32_t *dyn_index;               // per patch: bit 31 -- a patch of the first round, its index there below (NULL: the bottom rows)
    const uint32_t *dyn_inv;                 // the other way round: the patch at place r of the index (NULL: bottom-up) -- the order of last resort
    uint32_t *static_next, *dyn_index_next, *dyn_inv_next;   // ... for the next launch: the first places of this launch's order
    // (a launch that fell back on the order of last resort writes none of the three: whoever reads them checks that the launch
    // that was to write them did -- its tag in lists_done -- and goes by the bottom rows otherwise)
    const uint32_t *lists_done;
    uint32_t lists_tag;
    uint32_t *lists_done_next;
    uint32_t n_static;                       // waves of the first round (a multiple of 16)
    uint32_t key_mode;                       // RM_KEY_PLACE / RM_KEY_COST / RM_KEY_CONTENT
    uint32_t *patch_cost;                    // this launch's waves: their tile's time -> max per patch (100 MHz ticks); NULL: tiles are not timed
    const uint32_t *cost_prev;               // the previous launch's (same geometry, same scene)
    uint32_t *cost_zero;                     // the next launch's: cleared by this launch's classifying workgroups, patch by patch
    const uint32_t *ctab;                    // cost by content, the previous launch's: [pid] sum of the times (64-tick units) of the tiles that could reach the primitive, [64 + pid] how many
    uint32_t *ctab_cur;                      // this launch's waves add theirs (one tile in sixteen)
    uint32_t *ctab_zero;                     // the next launch's: cleared by this launch's first classifying workgroup
    // Sky tail: the last tail_patches places of the order get ONE wave each instead of sixteen -- sized by the host from a HINT
    // (how many patches the previous frames found nothing to hit in: ord_hint, page-locked, read without a wait).  Such a wave
    // finds its place's bucket: sky -> 24 KB of zeros.  Where the hint was a guess (the view has moved) and more patches have
    // something to hit than it said, the first of the tail's places hold such patches: place tail_first + j is rendered by the
    // sixteen waves j * 16 .. j * 16 + 15 behind the grid's end while j < ov_cap, by its own wave, tile by tile, beyond that.
    uint32_t tail_patches;
    uint32_t ov_cap;
    uint32_t first_round;                    // workgroups resident at once: they do not wait for their tiles' classification
    uint32_t tail_q;                         // ceil(tail_patches 2^32 / (tail_patches + tile waves behind the first round)); 1: the tail behind every tile wave; 0: no such waves
    unsigned long long *ord_hint;            // (launch_seq << 32) | patches behind the first round with something to hit
    unsigned long long *err_word;            // page-locked: non-zero once a wave of a launch on this stream gave up a wait that cannot fail (the frame is void)
    uint32_t launch_seq;
    uint32_t test_stall;                     // test hook: the order is never laid out (the waves behind the first round give up)
};

// What the classification launch gets besides the render launch's own arguments.
struct ClassifyArgs {
    unsigned long long *tile_mask;
    uint32_t n_prims;
    uint32_t _pad;
};

// rm_classify.hip
const void *rm_classify_kernel(bool edges);

// Feedback histogram: tile times in 100 MHz ticks, four buckets per octave (bucket b holds
// [edge(b), edge(b+1)), edge(b) = (4 + b % 4) << (b / 4) >> 2: 1, 1, 1, 1, 2, 2, 3, 3, 4, 5, 6, 7, 8, 10, ...);
// every RM_FB_SAMPLE-th tile is counted.
#define RM_FB_BUCKETS 64u
#define RM_FB_SAMPLE 8u
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
//   [736, 738)  the wave's start time (patch order)
#define RM_WAVE_L0_DEPTH_WORDS 448u
#define RM_WAVE_BVH_STACK_WORDS 480u
#define RM_WAVE_SUM_WORDS 512u
#define RM_WAVE_PAIR_WORDS 704u
#define RM_WAVE_T0_WORDS 736u      /* the wave's start time (patch order) */
#define RM_WAVE_LDS_WORDS 738u
// Scenes up to this long are copied into every workgroup's LDS block (the STAGED kernels): 4 KB.
#define RM_LDS_SCENE_LIMIT_WORDS 512u

}  // namespace rmdev

// The kernel of a launch (rm_kernels.hip, compiled once per numeric flavour and kernel group):
// stack 4 / 8 / 16 / 32, pow_mode POW_GENERIC / POW_INTEGER.  NULL: no such instantiation.
const void *rm_pick_kernel(bool fast, bool staged, bool bvh, bool cull, bool edges, int order, bool feedback, int stack, int pow_mode);   // order: 0 off, 1 the dispatch order from the launch's own classification

#endif
