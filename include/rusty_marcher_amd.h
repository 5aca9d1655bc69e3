/*
 * rusty_marcher_amd.h -- C ABI of the MI355X render backend.
 *
 * Drop-in boundary for ONE hot path of blefaudeux/rusty-marcher: the call
 *     raymarcher.render(&mut self.fb, &self.scene)      (engine/src/main.rs:331-333)
 * i.e. Renderer::render (renderer.rs:36-126) and everything below it
 * (cast_ray, the Shape::intersect implementations, optics, lighting).
 * The reference has no FFI of its own; these are the entry points a Rust
 * `extern "C"` block for that path binds (see INTEGRATION.md for the shim).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; every struct below is `#[repr(C)]`-able.
 *   - all arithmetic types are IEEE-754 binary64 (geometry.rs:4-8).
 *   - every call returns rm_status; nothing unwinds across the boundary.  The
 *     reference's failure mode is a panic, so the Rust shim turns a non-zero
 *     status into `panic!("{}", rm_last_error(..))`.
 *   - one caller at a time per rm_ctx / rm_scene (the reference calls render on
 *     the GTK main thread and blocks until the frame is complete).
 *   - the render entry points FAIL (RM_ERR_NO_DEVICE / RM_ERR_HIP) when no
 *     gfx950 device or kernel image is available; there is no CPU fallback.
 */
#ifndef RUSTY_MARCHER_AMD_H
#define RUSTY_MARCHER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_ABI_VERSION 5u

typedef enum rm_status {
    RM_OK = 0,
    RM_ERR_INVALID_ARG = 1,
    RM_ERR_DIMENSIONS = 2,   /* width % 32 != 0: the reference's scatter indexes out of
                                bounds and panics (renderer.rs:92-108) */
    RM_ERR_NO_DEVICE = 3,
    RM_ERR_HIP = 4,
    RM_ERR_NO_SCENE = 5,     /* rm_render before rm_scene_upload */
    RM_ERR_SCENE_LIMIT = 6,  /* scene exceeds a documented device limit */
    RM_ERR_IO = 7,           /* file missing (obj.rs:53-56 returns None; a missing mtllib
                                panics "WOOPS", obj.rs:64) */
    RM_ERR_PARSE = 8,
    RM_ERR_DEPTH = 9,        /* max_depth outside [0, RM_MAX_DEPTH] */
    RM_ERR_COMM = 10,        /* RCCL unavailable or a collective failed (rm_comm_*, rm_frame_*) */
    RM_ERR_TIMEOUT = 11      /* rm_frame_wait*: the slot's frame did not complete in time (a peer is
                                missing or stuck in the collective); the context is not usable for
                                further frames -- report and exit (rm_destroy then releases what it
                                can without waiting for the device) */
} rm_status;

/* Recursion cap accepted by rm_render.  The reference hard-codes 3
 * (renderer.rs:262) and counts in a u8. */
#define RM_MAX_DEPTH 32u
/* The reference's patch edge (renderer.rs:47); the only value accepted. */
#define RM_PATCH_SIZE 32u

/* geometry.rs:4-8 `Vec3f` */
typedef struct rm_vec3 { double x, y, z; } rm_vec3;

/* shapes.rs:20-32 `Reflectance` (bool widened to a 32-bit int + padding) */
typedef struct rm_reflectance {
    double  diffusion;
    rm_vec3 diffuse_color;
    double  specular;
    double  specular_exponent;
    int32_t is_glass_like;
    int32_t _pad;
    double  reflection;
    double  refractive_index;
} rm_reflectance;

/* lights.rs:4-8 `Light`; colour already L-inf normalised (lights.rs:10-16) */
typedef struct rm_light { rm_vec3 position, color; double intensity; } rm_light;

/* sphere.rs:6-11 `Sphere` (bounding box dropped: never consulted, shapes.rs:34-86) */
typedef struct rm_sphere { rm_vec3 center; double radius_square; rm_reflectance reflectance; } rm_sphere;

/* polygon.rs:6-12 `ConvexPolygon`; vertices live in rm_scene_desc.polygon_vertices */
typedef struct rm_polygon {
    uint32_t first_vertex, n_vertices;
    rm_vec3  plane_normal, plane_point;
    rm_reflectance reflectance;
} rm_polygon;

/* triangle.rs:6-10 `Triangle` + its entry of Obj.reflectances (obj.rs:16) */
typedef struct rm_triangle {
    rm_vec3 vertices[3];
    rm_vec3 normal, center;
    rm_reflectance reflectance;
} rm_triangle;

typedef enum rm_shape_kind { RM_SHAPE_SPHERE = 0, RM_SHAPE_POLYGON = 1, RM_SHAPE_MESH = 2 } rm_shape_kind;

/* One element of Scene.shapes: Vec<Box<dyn Shape + Sync>> (scene.rs:11), in list
 * order (the order decides ties, shapes.rs:130 and obj.rs:198).
 * SPHERE/POLYGON: `first` indexes spheres[]/polygons[], count == 1.
 * MESH (an obj.rs `Obj`): triangles[first .. first+count). */
typedef struct rm_shape_ref { uint32_t kind, first, count, _pad; } rm_shape_ref;

/* Flat, pointer-and-size view of scene.rs:9-13 `Scene`. */
typedef struct rm_scene_desc {
    const rm_shape_ref *shapes;            uint32_t n_shapes;
    const rm_sphere    *spheres;           uint32_t n_spheres;
    const rm_polygon   *polygons;          uint32_t n_polygons;
    const rm_vec3      *polygon_vertices;  uint32_t n_polygon_vertices;
    const rm_triangle  *triangles;         uint32_t n_triangles;
    const rm_light     *lights;            uint32_t n_lights;
    rm_vec3 camera;
} rm_scene_desc;

/* renderer.rs:17-23 `Renderer` + the frame geometry render() reads from the
 * FrameBuffer (renderer.rs:49-55) + the constants it hard-codes. */
typedef struct rm_params {
    /* Renderer, as create_renderer(fov, height, width) fills it (renderer.rs:25-33) */
    double fov, half_fov, height, width, ratio;
    /* FrameBuffer.width / .height (framebuffer.rs:6-10) */
    uint32_t frame_width, frame_height;
    /* renderer.rs:262 (reference: 3) */
    uint32_t max_depth;
    /* renderer.rs:47 (must be RM_PATCH_SIZE) */
    uint32_t patch_size;
    /* renderer.rs:40-44 (reference: 0.1, 0.1, 0.1) */
    rm_vec3 background;
    /* Patch rows owned by this caller: begin, begin + stride, begin + 2 stride, ... < end.
     * end == 0 means "frame_height/32"; stride 0 or 1 means every row of [begin, end).
     * A stride of N with begin = rank deals the rows out cyclically to N GPUs (sky rows
     * are cheap, ground rows expensive).  Pixels of rows not owned are untouched. */
    uint32_t patch_row_begin, patch_row_end;
    uint32_t flags;        /* RM_FLAG_* */
    uint32_t patch_row_stride;
} rm_params;

#define RM_FLAG_NONE 0u
/* Numeric flavour of the kernel.
 * Default (flag clear): every binary64 operation separately rounded, in the reference's
 * order (sqrt then divide, hits ordered by the squared distance of shapes.rs:128): the
 * hit / miss / shadow decisions are the reference's bit for bit, including where a ray
 * lies exactly on a polygon edge and the decision is the reference's rounding noise.
 * RM_FLAG_FAST_FP (opt-in, ~20 % faster): fused multiply-add, 1/sqrt by Newton iteration,
 * hits ordered by ray parameter.  Values move by ~1e-12; decisions can differ from the
 * reference ONLY at such exact-incidence pixels (the demo scene has a handful per frame
 * for some camera positions: its triangle has small-integer coordinates). */
#define RM_FLAG_FAST_FP 2u
/* rm_render_device_u8 only: device_rgb8 holds just the OWNED patch rows, packed -- the
 * k-th owned patch row occupies byte rows [32k, 32k + 32) of a [n_owned*32][frame_width][3]
 * buffer -- so that a rank's display bytes are one contiguous chunk for a gather. */
#define RM_FLAG_U8_COMPACT 4u
/* rm_render_device / rm_render_device_u8: device_rgb holds just the OWNED patch rows, packed
 * the same way ([n_owned*32][frame_width][3] doubles): a rank's f64 rows as one contiguous
 * chunk (what rm_frame_submit_f64 gathers). */
#define RM_FLAG_F64_COMPACT 8u

typedef struct rm_timing {
    double kernel_ms;   /* HIP-event time of the render kernel on its stream */
    double d2h_ms;      /* device -> host copy (0 when host_rgb == NULL) */
    double total_ms;    /* host wall time of the call */
} rm_timing;

typedef struct rm_scene rm_scene;   /* host-side scene builder (opaque) */
typedef struct rm_ctx rm_ctx;       /* one GPU: stream, device scene, framebuffer (opaque) */

/* ---------------------------------------------------------------------- */
/* Host side: scene construction.  Replaces the constructors the reference  */
/* runs before render(): nothing here touches the GPU.                      */
/* ---------------------------------------------------------------------- */

/* shapes.rs:49-61 Reflectance::create_default */
void rm_reflectance_default(rm_reflectance *out);

/* renderer.rs:25-33 create_renderer(fov, height, width) -- note the argument order.
 * Fills fov/half_fov/height/width/ratio, sets frame_* from (width, height) truncated,
 * max_depth = 3, patch_size = 32, background = 0.1, full band, flags = 0. */
void rm_create_renderer(double fov, double height, double width, rm_params *out);

rm_status rm_scene_new(rm_scene **out);                                   /* scene.rs:16-23 */
void      rm_scene_free(rm_scene *scene);
rm_status rm_scene_create_default(rm_scene **out);                        /* scene.rs:28-211 */
/* sphere.rs:13-24 */
rm_status rm_scene_add_sphere(rm_scene *scene, rm_vec3 center, double radius, const rm_reflectance *r);
/* polygon.rs:16-42; n_vertices < 3 -> RM_ERR_INVALID_ARG (reference asserts) */
rm_status rm_scene_add_polygon(rm_scene *scene, const rm_vec3 *vertices, uint32_t n_vertices,
                               const rm_reflectance *r);
/* One obj.rs `Obj`: n triangles as 9 doubles each; per-triangle colour ramp of
 * obj.rs:125-138; then Obj::offset(offset) (obj.rs:24-29, triangle.rs:19-24). */
rm_status rm_scene_add_mesh(rm_scene *scene, const double *tri_xyz, uint32_t n_triangles, rm_vec3 offset);
/* lights.rs:10-16 */
rm_status rm_scene_add_light(rm_scene *scene, rm_vec3 position, rm_vec3 color, double intensity);
/* ConvexPolygon::offset (polygon.rs:44-49) / Obj::offset (obj.rs:24-29): moves shape
 * `shape_index` of the list (plane point or triangle centres, and vertices; normals
 * are kept).  Spheres have no offset method in the reference -> RM_ERR_INVALID_ARG. */
rm_status rm_scene_offset_shape(rm_scene *scene, uint32_t shape_index, rm_vec3 offset);
rm_status rm_scene_set_camera(rm_scene *scene, rm_vec3 camera);
rm_status rm_scene_offset_camera(rm_scene *scene, rm_vec3 offset);       /* scene.rs:25-27 */
/* obj.rs:44-151 obj::load: appends one MESH shape per model of the file, each
 * moved by `offset`; *n_models_out (optional) receives the model count. */
rm_status rm_scene_load_obj(rm_scene *scene, const char *path, rm_vec3 offset, uint32_t *n_models_out);
/* main.rs:261-327 Win::open_obj: new scene = load(path) offset by (0,0,-500) + the
 * two hard-coded lights. */
rm_status rm_scene_open_obj(const char *path, rm_scene **out);
/* Flat view; pointers stay valid until the scene is next modified or freed. */
rm_status rm_scene_get_desc(const rm_scene *scene, rm_scene_desc *out);

/* renderer.rs:111-121: "Scene rendered in {} ms ({} fps, {:.2} MP/s)".
 * Returns the length written (snprintf semantics). */
int rm_format_status(char *buf, size_t buflen, uint64_t ms, uint32_t frame_width, uint32_t frame_height);

/* ---------------------------------------------------------------------- */
/* Device side: the render path.                                            */
/* ---------------------------------------------------------------------- */

/* Binds HIP device `device_ordinal` (one process per GPU), creates its stream. */
rm_status rm_init(int device_ordinal, rm_ctx **out);
void      rm_destroy(rm_ctx *ctx);
/* Last error text of `ctx`; ctx == NULL -> last error of a host-side call or of a
 * failed rm_init on this thread.  Never NULL. */
const char *rm_last_error(const rm_ctx *ctx);

/* Copies the scene into device memory (arrays are copied; caller keeps ownership).  The
 * reference hands its whole Scene to every render() call (main.rs:331-333); a scene whose
 * device image equals the resident one is recognised and not copied again (cameras apart:
 * the camera is not part of the image), so a host may simply upload before every frame. */
rm_status rm_scene_upload(rm_ctx *ctx, const rm_scene_desc *desc);
/* How often rm_scene_upload was called on this context, and how often it had to copy. */
rm_status rm_scene_uploads(rm_ctx *ctx, uint64_t *calls, uint64_t *copies);
/* scene.rs:25-27 without re-upload: replaces the camera of the uploaded scene. */
rm_status rm_camera_update(rm_ctx *ctx, rm_vec3 camera);

/*
 * Renderer::render (renderer.rs:36-126) for the patch rows of params' band.
 * host_rgb: [frame_height][frame_width][3] doubles row-major (framebuffer.rs:12-22
 * flattened), caller-allocated; only the band's rows are written, so rows
 * >= frame_height - frame_height%32 keep their previous contents exactly as in the
 * reference.  NULL leaves the result in the context's device framebuffer.
 * Blocks until host_rgb is filled (or, for NULL, until the kernel has finished).
 * The device -> host copy dominates the call (48.7 MB at 1080p: ~0.9 ms over PCIe against
 * 0.08 ms of kernel; black patches are not sent, see rm_render_rows); a window wants
 * rm_render_display (3 B/pixel), a render loop rm_frame_submit_to_host.
 */
rm_status rm_render(rm_ctx *ctx, const rm_params *params, double *host_rgb, rm_timing *timing);

/*
 * Renderer::render into the reference's ACTUAL render target: framebuffer.rs:6-22 is
 * `buffer: Vec<Vec<Vec3f>>`, one heap allocation per scan line, filled by the serial scatter of
 * renderer.rs:92-108.  rows[y] points at row y: frame_width * 3 doubles (x, y, z of each Vec3f --
 * `#[repr(C)]` on geometry.rs:4-8); frame_height pointers, of which only
 * those of the band's rows are read (the others may be NULL: rows >= frame_height -
 * frame_height % 32 keep their contents, renderer.rs:53).  Blocks until the rows are filled, bit
 * for bit what rm_render writes into a flat array.
 * Inside: one launch, then the frame crosses the PCIe link into a page-locked staging buffer of
 * the context in chunks while a few host threads (RM_HOST_THREADS, default min(8, cores))
 * scatter the chunks that have arrived into the rows.  Patches that are black -- +0.0 in every
 * channel of all 1,024 pixels: what primary rays that leave the scene produce, renderer.rs:305 --
 * are not sent: a kernel packs the others, the host writes the zeros itself (frames of 2 MB and
 * more; RM_HOST_PACK=0|1 forces).  rm_render takes the same path into its flat array.
 */
rm_status rm_render_rows(rm_ctx *ctx, const rm_params *params, double *const *rows, rm_timing *timing);

/*
 * Renderer::render with a DEVICE-RESIDENT FrameBuffer: the f64 frame stays in the context's
 * device framebuffer (rm_device_framebuffer, rm_fetch_rows, rm_postprocess(ctx, NULL, ..)) and
 * only `fb.to_vec()` of the band -- the bytes update_raytrace_image hands to the pixbuf,
 * main.rs:337-346; (255 * clamp(f, 0, 1)) as u8, framebuffer.rs:40-55,80-82 -- comes back:
 * host_rgb8 is [frame_height][frame_width][3] bytes, only the band's rows are written.  3 B/pixel
 * instead of 24: one launch, then one copy-engine transfer (RM_DISPLAY_SUBBANDS=n renders n sub-bands, the bytes
 * of one crossing the link while the next renders: measured and slower, default 1).  Blocks until host_rgb8 is filled.
 * host_rgb8 from rm_host_alloc is written by the copy engine directly; any other memory is
 * reached through the context's staging buffer.
 */
rm_status rm_render_display(rm_ctx *ctx, const rm_params *params, uint8_t *host_rgb8, rm_timing *timing);

/*
 * f64 rows of the device-resident frame on demand (save_to_file, main.rs:353-357: normalize +
 * write_ppm read the f64 values): patch rows [patch_row_begin, patch_row_end) -- end 0 = all
 * whole patch rows -- of the frame the last rm_render / rm_render_rows / rm_render_display left
 * on the device, into rows[y] (as rm_render_rows).  frame_width x frame_height is the size of the
 * FrameBuffer `rows` belongs to -- frame_height pointers to frame_width * 3 doubles each: it must be
 * the size of the resident frame (a window that has been resized since holds another), else
 * RM_ERR_INVALID_ARG and nothing is read or written (ABI 5; ABI 4 took no size and trusted the caller).
 */
rm_status rm_fetch_rows(rm_ctx *ctx, double *const *rows, uint32_t frame_width, uint32_t frame_height,
                        uint32_t patch_row_begin, uint32_t patch_row_end);

/* What the last rm_render / rm_render_rows / rm_fetch_rows / rm_render_display of this context
 * moved: bytes that crossed the link, 32x32 patches of the band, patches among them that were
 * sent (the others were black and written by the host), and the host threads that scatter. */
rm_status rm_hostio_stats(rm_ctx *ctx, uint64_t *bytes_copied, uint64_t *patches, uint64_t *patches_sent, int *threads);

/*
 * Same kernel, asynchronous, into a caller-owned DEVICE buffer with the same
 * [frame_height][frame_width][3] layout, enqueued on `hip_stream` (a hipStream_t;
 * NULL = HIP's default stream).  Returns once enqueued; ordering with the caller's
 * other work is that stream's.
 */
rm_status rm_render_device(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *hip_stream);

/*
 * rm_render_device that ALSO writes the band's display bytes: device_rgb8 is a
 * [frame_height][frame_width][3] byte buffer receiving `fb.to_vec()` of the rendered
 * pixels (framebuffer.rs:40-55: (255 * clamp(f, 0, 1)) as u8, no normalisation -- what
 * update_raytrace_image hands to the pixbuf, main.rs:337-346), fused into the kernel's
 * epilogue.  3 B/pixel instead of 24: the payload a multi-GPU gather or a D2H copy for
 * display needs.  max_depth must be >= 1.
 */
rm_status rm_render_device_u8(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *device_rgb8,
                              void *hip_stream);

/* Tile classification (the launch in front of the render launch: one lane per 16x4 tile tests the cone of
 * the tile's primary rays against every primitive's bounds -- the BoundingBox the reference computes and never
 * consults, shapes.rs:34-86; tiles nothing can be hit in are filled with the primary-miss value there,
 * renderer.rs:305, and never get a wave): the tiles of the last render launch of this context and how
 * many of them the classification listed for rendering (all of them when it was off: RM_TILE_CLASSIFY=0,
 * small frames, scenes of many primitives).  hip_stream: the stream that launch was enqueued on (NULL: whichever
 * the context's last render launch went to).  Waits for the device. */
rm_status rm_tile_stats(rm_ctx *ctx, void *hip_stream, uint32_t *tiles, uint32_t *tiles_listed);

/* The geometry of the context's last render launch: its workgroups (one wave each), and how many 32x32 patches of
 * it were handed to its sky tail -- patches that the previous frames of the same view on the stream found nothing to
 * hit in get ONE wave instead of sixteen (it looks at this launch's own classification of the patch, stores the
 * primary-miss value of renderer.rs:305 over the patch when that still says sky, and renders the patch itself when
 * it does not).  Only the launch's geometry is carried from frame to frame; RM_SKY_TAIL=0 switches it off.
 * Does not wait for the device. */
rm_status rm_launch_stats(rm_ctx *ctx, uint32_t *workgroups, uint32_t *tail_patches);

/* Device framebuffer of the last rm_render(.., NULL, ..) and its size in bytes. */
rm_status rm_device_framebuffer(rm_ctx *ctx, void **device_rgb, size_t *bytes);

/*
 * framebuffer.rs:58-77 normalize + :40-55 to_vec/:80-82 quantize, on the device.
 * device_rgb: [h][w][3] doubles (NULL = the context's framebuffer).  The global max
 * is taken over ALL h*w pixels (unrendered rows included, as the reference does).
 *   normalize != 0 : scales device_rgb in place by 1/max (if max > 0), as
 *                    FrameBuffer::normalize does before write_ppm (main.rs:355-356)
 *   host_rgb8      : optional [h][w][3] bytes, receives to_vec() of the result
 *   max_out        : optional, receives the global max found (0 when !normalize)
 */
rm_status rm_postprocess(rm_ctx *ctx, void *device_rgb, uint32_t frame_width, uint32_t frame_height,
                         int normalize, uint8_t *host_rgb8, double *max_out);

/* ---- device buffers for hosts without HIP bindings -------------------------------------
 * The device-pointer entry points (rm_render_device*, rm_frame_submit) take plain
 * hipMalloc'ed pointers; a host that links no HIP runtime of its own (the Rust shim) gets
 * them here.  rm_buffer_alloc zero-fills (create_frame_buffer does, framebuffer.rs:12-22);
 * rm_buffer_read is synchronous and does not order itself after work in flight: call it
 * after rm_frame_wait / on buffers no launch is still writing. */
rm_status rm_buffer_alloc(rm_ctx *ctx, size_t bytes, void **device_ptr);
void rm_buffer_free(rm_ctx *ctx, void *device_ptr);
rm_status rm_buffer_read(rm_ctx *ctx, const void *device_ptr, void *host_dst, size_t bytes);
/* Synchronous host -> device copy (e.g. a FrameBuffer's contents for rm_postprocess). */
rm_status rm_buffer_write(rm_ctx *ctx, void *device_ptr, const void *host_src, size_t bytes);
/* Page-locked host memory: the destination rm_frame_submit can copy a finished display
 * frame into asynchronously (a pageable destination would make the copy synchronous). */
rm_status rm_host_alloc(rm_ctx *ctx, size_t bytes, void **host_ptr);
void rm_host_free(rm_ctx *ctx, void *host_ptr);

/* ---- multi-GPU frames: one process per GPU, RCCL over xGMI ----------------------------
 * Replaces, for N GPUs, what renderer.rs:63-108 does with N Rayon workers: the patch rows
 * of a frame are owned cyclically (rank r renders patch rows r, r+N, r+2N, ... so that
 * every rank gets its share of cheap sky and expensive ground rows), each rank's f64 rows
 * stay in its own `device_rgb` (a distributed FrameBuffer), and ONE RCCL exchange per frame
 * completes the display frame (`to_vec` bytes, framebuffer.rs:40-55) at the consumer, RANK 0
 * (the reference has one window, main.rs:337-346): every peer sends its chunk straight to rank 0
 * over its own xGMI link, all at once (grouped ncclSend / ncclRecv).  rm_comm_exchange(ctx, 1)
 * on every rank selects an in-place ncclAllGather instead (the frame on every rank).  Up to RM_MAX_FRAME_SLOTS frames are in flight; a slot has a stream of its
 * own (render, gather, de-interleave in order on the stream); every rank must submit the
 * same sequence of (frame, slot) pairs.  All slots share ONE communicator unless
 * RM_SLOT_COMMS=1 is exported on EVERY rank (then each slot gets one split off the first;
 * the ranks agree by an all-reduce whether every split succeeded, else all keep the one).
 *
 * Bootstrap: rank 0 calls rm_comm_unique_id and hands the RM_COMM_ID_BYTES to the other
 * ranks by any channel (a file, a socket, torch.distributed's store); every rank then
 * calls rm_comm_init.  id == NULL sets rank/world without a transport: the layout is that
 * of `world` ranks but nothing is exchanged (single-GPU tests of the N-GPU layout).
 */
#define RM_COMM_ID_BYTES 128u
#define RM_MAX_FRAME_SLOTS 4u
rm_status rm_comm_unique_id(void *id_out /* RM_COMM_ID_BYTES */);
rm_status rm_comm_init(rm_ctx *ctx, const void *id, int rank, int world);
void rm_comm_destroy(rm_ctx *ctx);   /* also done by rm_destroy */
/* all_ranks = 0 (default): the chunks are gathered at rank 0; 1: all-gathered, every rank ends
 * up with the whole gather buffer.  The same on every rank, before the first submit. */
rm_status rm_comm_exchange(rm_ctx *ctx, int all_ranks);

/* Layout of the gather buffer for `world` ranks: world chunks of rows_per_rank patch rows
 * (32 * frame_width * 3 bytes each); chunk k holds patch rows k, k+world, ... packed. */
rm_status rm_exchange_layout(const rm_params *params, int world, uint32_t *rows_per_rank, size_t *chunk_bytes);

/*
 * One frame, asynchronously: renders this rank's rows into device_rgb ([H][W][3] f64,
 * only the owned rows are written) and their display bytes into this rank's chunk of
 * device_gather8 (world * chunk_bytes), gathers the chunks there (at rank 0; with
 * rm_comm_exchange(ctx, 1) on every rank), and -- where device_display8 is not NULL (the
 * consumer: rank 0) -- writes the [32*n_patch_rows][W][3] image-order display frame.
 * params->patch_row_* must be zero.
 * All four buffers belong to `slot` until rm_frame_wait(slot) returns or the slot is
 * submitted again (a slot's frames are ordered); two slots must not share a buffer.
 */
rm_status rm_frame_submit(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *device_gather8,
                          void *device_display8, uint32_t slot);
/* The same, and the display frame is then copied into host_display8 (rm_host_alloc'ed,
 * 32 * n_patch_rows * W * 3 bytes) on the slot's stream: after rm_frame_wait(slot) the bytes
 * of fb.to_vec() are in host memory (main.rs:337-346 hands them to the pixbuf). */
rm_status rm_frame_submit_to_host(rm_ctx *ctx, const rm_params *params, void *device_rgb, void *device_gather8,
                                  void *device_display8, void *host_display8, uint32_t slot);
/* Blocks the host until the slot's last frame is complete on this rank.  With a
 * communicator the wait is bounded: RM_ERR_TIMEOUT after RM_FRAME_TIMEOUT_MS (environment,
 * default 60000; 0 = wait for ever) -- a peer that never joins the collective must not hang
 * the host. */
rm_status rm_frame_wait(rm_ctx *ctx, uint32_t slot);
/* The same with an explicit bound in milliseconds (0 = wait for ever). */
rm_status rm_frame_wait_for(rm_ctx *ctx, uint32_t slot, uint32_t timeout_ms);

/*
 * The f64 frame itself (framebuffer.rs:6-22: the reference's render target is f64) to the
 * consumer: this rank's rows are rendered packed (RM_FLAG_F64_COMPACT) straight into its
 * chunk of device_gather64 (world * 8 * chunk_bytes of rm_exchange_layout), ONE exchange
 * completes the buffer at rank 0 (on every rank with rm_comm_exchange(ctx, 1)), and -- where
 * device_frame64 is not NULL --
 * the [32*n_patch_rows][W][3] f64 frame is written in image order, bit-identical to the
 * single-GPU frame.  8x the bytes of the display path: at 1080p ~6 MB per peer per frame.
 * Frames in flight (slots) overlap one frame's gather with the next frame's render.
 */
rm_status rm_frame_submit_f64(rm_ctx *ctx, const rm_params *params, void *device_gather64, void *device_frame64,
                              uint32_t slot);

/* Device times of the slot's last completed frame (call after rm_frame_wait): render kernel,
 * collective, and everything of the frame on its stream.  The stamps are three more events
 * per frame in the slot's stream and are recorded only after rm_frame_timing_enable(ctx, 1)
 * (a diagnostic: they cost a frame in flight ~15 us). */
typedef struct rm_frame_times { double kernel_ms, gather_ms, total_ms; } rm_frame_times;
rm_status rm_frame_timing_enable(rm_ctx *ctx, int on);
rm_status rm_frame_timing(rm_ctx *ctx, uint32_t slot, rm_frame_times *out);

/* What the communicator itself reports (ncclCommUserRank / ncclCommCount) and how many
 * communicators the slots use; world 1 / rank 0 / 0 communicators without one. */
rm_status rm_comm_info(rm_ctx *ctx, int *rank, int *world, int *n_communicators);


/* Library / device introspection for harnesses. */
uint32_t    rm_abi_version(void);
const char *rm_build_info(void);
rm_status   rm_device_info(rm_ctx *ctx, char *name_buf, size_t buflen, int *n_cus, size_t *lds_bytes);
/* Name of the kernel a render of the uploaded scene with `params` launches, as a kernel trace
 * (rocprofv3) prints it: ties a measured launch to its profile. */
rm_status   rm_kernel_name(rm_ctx *ctx, const rm_params *params, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* RUSTY_MARCHER_AMD_H */
