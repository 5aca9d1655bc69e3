/*
 * rm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the render path of blefaudeux/rusty-marcher
 * (engine/src/{renderer,shapes,sphere,polygon,triangle,obj,optics,geometry,
 * lights,scene,framebuffer}.rs).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (rusty-marcher_amd/) never links or calls it.
 *
 * Parity status: PINNED for the demo scene -- orc_render + orc_normalize +
 * orc_to_vec reproduce the reference's committed engine/out.ppm (800x600,
 * sha256 82d51afa...84797) byte for byte (tests/test_oracle_golden.py), and
 * the reference's unit known-answer tests are restated in
 * tests/test_oracle_kat.py.  UNPINNED for the OBJ/cornell recipe (the only
 * reference test is `load(..).is_some()`, obj.rs:229-233; tobj's source is not
 * in /root/reference) and for the build's own synthetic scene.
 *
 * Compile with -ffp-contract=off: every operation below is one IEEE-754
 * double operation in the order the Rust source performs it.
 */
#ifndef RM_ORACLE_H
#define RM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* geometry.rs:4-8 */
typedef struct { double x, y, z; } orc_vec3;

/* shapes.rs:20-32 */
typedef struct {
    double   diffusion;
    orc_vec3 diffuse_color;
    double   specular;
    double   specular_exponent;
    int      is_glass_like;
    double   reflection;
    double   refractive_index;
} orc_reflectance;

/* shapes.rs:3-8 */
typedef struct {
    orc_vec3        point;
    orc_vec3        normal;
    orc_reflectance reflectance;
} orc_intersection;

/* lights.rs:4-8 */
typedef struct { orc_vec3 position, color; double intensity; } orc_light;

/* triangle.rs:6-10 */
typedef struct { orc_vec3 vertices[3]; orc_vec3 normal; orc_vec3 center; } orc_triangle;

enum { ORC_SPHERE = 0, ORC_POLYGON = 1, ORC_OBJ = 2 };

/* One `Box<dyn Shape + Sync>` (scene.rs:11): tagged union of the three
 * implementors of `trait Shape` (sphere.rs:26, polygon.rs:59, obj.rs:185). */
typedef struct {
    int kind;
    /* sphere.rs:6-11 */
    orc_vec3 center;
    double   radius_square;
    /* polygon.rs:6-12 */
    orc_vec3 *vertices;
    size_t    n_vertices;
    orc_vec3  plane_normal, plane_point;
    /* sphere + polygon */
    orc_reflectance reflectance;
    /* obj.rs:13-20 */
    orc_triangle    *triangles;
    orc_reflectance *reflectances;
    size_t           n_triangles;
} orc_shape;

/* scene.rs:9-13 */
typedef struct {
    orc_light *lights;  size_t n_lights;
    orc_shape *shapes;  size_t n_shapes;
    orc_vec3   camera;
} orc_scene;

/* renderer.rs:17-23 */
typedef struct { double fov, half_fov, height, width, ratio; } orc_renderer;

/* ---- geometry.rs ---- */
orc_vec3 orc_v(double x, double y, double z);
orc_vec3 orc_add(orc_vec3 a, orc_vec3 b);
orc_vec3 orc_sub(orc_vec3 a, orc_vec3 b);
orc_vec3 orc_mul(orc_vec3 a, orc_vec3 b);
orc_vec3 orc_neg(orc_vec3 a);
orc_vec3 orc_scaled(orc_vec3 a, double s);
orc_vec3 orc_cross(orc_vec3 a, orc_vec3 b);
double   orc_dot(orc_vec3 a, orc_vec3 b);
double   orc_squared_norm(orc_vec3 a);
orc_vec3 orc_normalized(orc_vec3 a);
orc_vec3 orc_normalized_l0(orc_vec3 a);

/* ---- shapes / primitives ---- */
orc_reflectance orc_reflectance_default(void);
orc_triangle    orc_triangle_create(orc_vec3 v0, orc_vec3 v1, orc_vec3 v2);
void            orc_triangle_offset(orc_triangle *t, orc_vec3 off);
int orc_triangle_intersect(const orc_triangle *t, orc_vec3 orig, orc_vec3 dir, orc_intersection *out);
int orc_shape_intersect(const orc_shape *s, orc_vec3 orig, orc_vec3 dir, orc_intersection *out);
int orc_intersect_shape_set(orc_vec3 orig, orc_vec3 dir, const orc_shape *shapes, size_t n);
int orc_find_closest_intersect(orc_vec3 orig, orc_vec3 dir, const orc_shape *shapes, size_t n,
                               orc_intersection *out, uint8_t *shape_hit);

/* ---- optics.rs ---- */
orc_vec3 orc_reflect(orc_vec3 incident, orc_vec3 normal);
int orc_reflect_ray(orc_vec3 incident, const orc_intersection *is, double refractive_index,
                    orc_vec3 *out_orig, orc_vec3 *out_dir);
int orc_refract_ray(orc_vec3 incident, const orc_intersection *is, double refractive_index,
                    orc_vec3 *out_orig, orc_vec3 *out_dir);

/* ---- scene construction ---- */
orc_scene *orc_scene_new(void);
void       orc_scene_free(orc_scene *s);
void orc_scene_add_sphere(orc_scene *s, orc_vec3 center, double radius, orc_reflectance r);
void orc_scene_add_polygon(orc_scene *s, const orc_vec3 *vertices, size_t n, orc_reflectance r);
/* One obj.rs `Obj`: n triangles given as 9 doubles each (already widened from
 * f32), colour ramp of obj.rs:125-138, then Obj::offset(off) (obj.rs:24-29). */
void orc_scene_add_obj(orc_scene *s, const double *tri_xyz, size_t n_triangles, orc_vec3 offset);
void orc_scene_add_light(orc_scene *s, orc_vec3 position, orc_vec3 color, double intensity);
orc_scene *orc_scene_create_default(void);          /* scene.rs:28-211 */

/* ---- renderer.rs ---- */
orc_renderer orc_create_renderer(double fov, double height, double width);
orc_vec3 orc_backproject(const orc_renderer *r, size_t i, size_t j);
/* max_depth generalises the constant 3 of renderer.rs:262. */
orc_vec3 orc_cast_ray(orc_vec3 orig, orc_vec3 dir, const orc_scene *scene,
                      orc_vec3 background, unsigned n_recursion, unsigned max_depth);
/*
 * renderer.rs:36-108.  frame = [height][width][3] doubles, row-major; rows
 * >= height - height%32 are left untouched.  n_threads <= 0 -> one per online
 * CPU.  Returns 0, or -1 when width%32 != 0 (the reference panics there).
 * ms_out (optional) receives the wall time of patches + scatter in ms.
 */
int orc_render(const orc_renderer *r, const orc_scene *scene, double *frame,
               size_t width, size_t height, unsigned max_depth, int n_threads,
               double *ms_out);
/* Sub-band form used to restate a row-sharded job: renders only patch rows
 * [patch_row_begin, patch_row_end). */
int orc_render_band(const orc_renderer *r, const orc_scene *scene, double *frame,
                    size_t width, size_t height, unsigned max_depth, int n_threads,
                    size_t patch_row_begin, size_t patch_row_end);
/* renderer.rs:111-121: formats the status string; returns chars written. */
int orc_status_message(char *buf, size_t buflen, uint64_t ms, size_t width, size_t height);

/* ---- framebuffer.rs ---- */
void    orc_normalize(double *frame, size_t width, size_t height);   /* :58-77 */
uint8_t orc_quantize(double f);                                      /* :80-82 */
void    orc_to_vec(const double *frame, size_t width, size_t height, uint8_t *out); /* :40-55 */
int     orc_write_ppm(const char *filename, const double *frame, size_t width, size_t height); /* :26-38 */

/* ray statistics of the last orc_render call (summed over threads) */
typedef struct { uint64_t cast_ray, intersect, shadow_rays, pow_calls; } orc_stats;
void orc_get_stats(orc_stats *out);

int orc_online_cpus(void);

#ifdef __cplusplus
}
#endif
#endif
