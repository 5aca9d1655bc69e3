/*
 * rm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See rm_oracle.h for scope and parity status.  Every function cites the
 * reference lines it restates (paths relative to /root/reference/engine/src).
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#define _GNU_SOURCE
#include "rm_oracle.h"

#include <assert.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------ */
/* geometry.rs                                                         */
/* ------------------------------------------------------------------ */

orc_vec3 orc_v(double x, double y, double z) { orc_vec3 v = {x, y, z}; return v; }

/* geometry.rs:118-128 */
orc_vec3 orc_add(orc_vec3 a, orc_vec3 b) { return orc_v(a.x + b.x, a.y + b.y, a.z + b.z); }
/* geometry.rs:163-172 */
orc_vec3 orc_sub(orc_vec3 a, orc_vec3 b) { return orc_v(a.x - b.x, a.y - b.y, a.z - b.z); }
/* geometry.rs:152-161: component-wise product */
orc_vec3 orc_mul(orc_vec3 a, orc_vec3 b) { return orc_v(a.x * b.x, a.y * b.y, a.z * b.z); }
/* geometry.rs:130-140 */
orc_vec3 orc_neg(orc_vec3 a) { return orc_v(-a.x, -a.y, -a.z); }
/* geometry.rs:36-40,49-53 */
orc_vec3 orc_scaled(orc_vec3 a, double s) { return orc_v(a.x * s, a.y * s, a.z * s); }
/* geometry.rs:59-65 */
orc_vec3 orc_cross(orc_vec3 a, orc_vec3 b)
{
    return orc_v(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* geometry.rs:180-182: left-to-right sum */
double orc_dot(orc_vec3 a, orc_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* geometry.rs:67-69 */
double orc_squared_norm(orc_vec3 a) { return orc_dot(a, a); }
/* geometry.rs:19-23,104-109: reciprocal, then multiply */
orc_vec3 orc_normalized(orc_vec3 a)
{
    double norm = sqrt(orc_dot(a, a));
    if (norm > 0.) return orc_scaled(a, 1. / norm);
    return a;
}
/* Rust f64::max / f64::min: a NaN operand is ignored. */
static double rs_max(double a, double b) { return fmax(a, b); }
static double rs_min(double a, double b) { return fmin(a, b); }
/* geometry.rs:29-33,111-116: divide by the max component */
orc_vec3 orc_normalized_l0(orc_vec3 a)
{
    double norm = rs_max(rs_max(a.x, a.y), a.z);
    if (norm > 0.) return orc_scaled(a, 1. / norm);
    return a;
}

/* ------------------------------------------------------------------ */
/* shapes.rs / sphere.rs / polygon.rs / triangle.rs / obj.rs           */
/* ------------------------------------------------------------------ */

/* shapes.rs:49-61 */
orc_reflectance orc_reflectance_default(void)
{
    orc_reflectance r;
    r.diffusion = 1.;
    r.diffuse_color = orc_v(1., 1., 1.);
    r.specular = 1.;
    r.specular_exponent = 30.;
    r.is_glass_like = 0;
    r.reflection = 0.95;
    r.refractive_index = 1.;
    return r;
}

static __thread orc_stats tls_stats;

/* sphere.rs:31, polygon.rs:62, triangle.rs:53: `assert!` -> panic */
static void assert_normalized(orc_vec3 dir)
{
    if (!(fabs(orc_squared_norm(dir) - 1.) < 1e-4)) {
        fprintf(stderr, "oracle: direction not normalized (reference would panic)\n");
        abort();
    }
}

/* sphere.rs:27-61 */
static int sphere_intersect(const orc_shape *s, orc_vec3 orig, orc_vec3 dir, orc_intersection *out)
{
    orc_vec3 line = orc_sub(s->center, orig);
    assert_normalized(dir);

    double tca = orc_dot(line, dir);
    double d2 = orc_dot(line, line) - tca * tca;
    if (d2 > s->radius_square) return 0;

    double thc = sqrt(s->radius_square - d2);
    double t0 = tca - thc;
    double t1 = tca + thc;
    if (t0 < 0.) t0 = t1;
    if (t0 < 0.) return 0;

    orc_vec3 p = orc_add(orig, orc_scaled(dir, t0));
    out->point = p;
    out->normal = orc_normalized(orc_sub(p, s->center));
    out->reflectance = s->reflectance;
    return 1;
}

/* polygon.rs:54-56, triangle.rs:13-15 */
static int inside(orc_vec3 a, orc_vec3 p1, orc_vec3 p2)
{
    return orc_cross(orc_sub(p1, a), orc_sub(p2, a)).z > 0.;
}

/* polygon.rs:60-98 */
static int polygon_intersect(const orc_shape *s, orc_vec3 orig, orc_vec3 dir, orc_intersection *out)
{
    assert_normalized(dir);
    double dotprod = orc_dot(dir, s->plane_normal);
    if (dotprod == 0.) return 0;
    double dist = orc_dot(orc_sub(s->plane_point, orig), s->plane_normal) / dotprod;
    if (dist < 0.) return 0;
    orc_vec3 p = orc_add(orig, orc_scaled(dir, dist));
    size_t n = s->n_vertices;
    for (size_t i = 0; i < n; i++)
        if (!inside(p, s->vertices[i], s->vertices[(i + 1) % n])) return 0;
    out->point = p;
    out->normal = s->plane_normal;
    out->reflectance = s->reflectance;
    return 1;
}

/* triangle.rs:33-47 */
orc_triangle orc_triangle_create(orc_vec3 v0, orc_vec3 v1, orc_vec3 v2)
{
    orc_triangle t;
    t.vertices[0] = v0; t.vertices[1] = v1; t.vertices[2] = v2;
    orc_vec3 mean = orc_scaled(orc_add(orc_add(v0, v1), v2), 1. / 3.);
    orc_vec3 edge_1 = orc_sub(v1, v0);
    orc_vec3 edge_2 = orc_sub(v2, v1);
    t.normal = orc_normalized(orc_cross(edge_1, edge_2));
    t.center = mean;
    return t;
}

/* triangle.rs:19-24: moves centre and vertices, keeps the normal */
void orc_triangle_offset(orc_triangle *t, orc_vec3 off)
{
    t->center = orc_add(t->center, off);
    for (int i = 0; i < 3; i++) t->vertices[i] = orc_add(t->vertices[i], off);
}

/* triangle.rs:49-83 */
int orc_triangle_intersect(const orc_triangle *t, orc_vec3 orig, orc_vec3 dir, orc_intersection *out)
{
    assert_normalized(dir);
    double dot_product = orc_dot(dir, t->normal);
    if (fabs(dot_product) < 1e-6) return 0;
    double dist = orc_dot(orc_sub(t->center, orig), t->normal) / dot_product;
    if (dist < 0.) return 0;
    orc_vec3 p = orc_add(orig, orc_scaled(dir, dist));
    for (int i = 0; i < 3; i++)
        if (!inside(p, t->vertices[i], t->vertices[(i + 1) % 3])) return 0;
    out->point = p;
    out->normal = t->normal;
    out->reflectance = orc_reflectance_default();
    return 1;
}

/* obj.rs:185-221 */
static int obj_intersect(const orc_shape *s, orc_vec3 orig, orc_vec3 dir, orc_intersection *out)
{
    int hit_triangle = 0;
    double dist_closest = 0.;
    orc_intersection fin;
    memset(&fin, 0, sizeof fin);
    for (size_t i = 0; i < s->n_triangles; i++) {
        orc_intersection is;
        if (orc_triangle_intersect(&s->triangles[i], orig, dir, &is)) {
            double dist_hit = orc_squared_norm(orc_sub(is.point, orig));
            if (!hit_triangle || dist_hit < dist_closest) {
                fin.point = is.point;
                fin.normal = is.normal;
                fin.reflectance = s->reflectances[i];
                hit_triangle = 1;
                dist_closest = dist_hit;
            }
        }
    }
    if (hit_triangle) { *out = fin; return 1; }
    return 0;
}

/* `shape.intersect(..)` dyn dispatch, shapes.rs:40-47 */
int orc_shape_intersect(const orc_shape *s, orc_vec3 orig, orc_vec3 dir, orc_intersection *out)
{
    switch (s->kind) {
    case ORC_SPHERE:  return sphere_intersect(s, orig, dir, out);
    case ORC_POLYGON: return polygon_intersect(s, orig, dir, out);
    default:          return obj_intersect(s, orig, dir, out);
    }
}

/* shapes.rs:92-108: any-hit, first hit returns */
int orc_intersect_shape_set(orc_vec3 orig, orc_vec3 dir, const orc_shape *shapes, size_t n)
{
    orc_intersection tmp;
    for (size_t i = 0; i < n; i++) {
        if (orc_shape_intersect(&shapes[i], orig, dir, &tmp)) {
            tls_stats.intersect += i + 1;
            return 1;
        }
    }
    tls_stats.intersect += n;
    return 0;
}

/* shapes.rs:110-143: closest by squared distance hit-point -> origin, strict < */
int orc_find_closest_intersect(orc_vec3 orig, orc_vec3 dir, const orc_shape *shapes, size_t n,
                               orc_intersection *out, uint8_t *shape_hit_out)
{
    orc_intersection fin;
    memset(&fin, 0, sizeof fin);
    fin.reflectance = orc_reflectance_default();
    int hit = 0;
    size_t shape_hit = 0;
    double dist_closest = 0.;
    for (size_t i = 0; i < n; i++) {
        orc_intersection is;
        if (orc_shape_intersect(&shapes[i], orig, dir, &is)) {
            double dist_hit = orc_squared_norm(orc_sub(is.point, orig));
            if (!hit || dist_hit < dist_closest) {
                fin = is;
                hit = 1;
                shape_hit = i;
                dist_closest = dist_hit;
            }
        }
    }
    tls_stats.intersect += n;
    if (hit) {
        *out = fin;
        if (shape_hit_out) *shape_hit_out = (uint8_t)shape_hit; /* `as u8` wraps, shapes.rs:140 */
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* optics.rs                                                           */
/* ------------------------------------------------------------------ */

/* optics.rs:4-6 */
orc_vec3 orc_reflect(orc_vec3 incident, orc_vec3 normal)
{
    return orc_sub(incident, orc_scaled(normal, 2. * orc_dot(incident, normal)));
}

/* optics.rs:8-48 */
int orc_reflect_ray(orc_vec3 incident, const orc_intersection *is, double refractive_index,
                    orc_vec3 *out_orig, orc_vec3 *out_dir)
{
    orc_vec3 normal = is->normal;
    double c = orc_dot(normal, incident);
    double r = (c < 0.) ? refractive_index : 1. / refractive_index;
    if (c < 0.) { c = -c; normal = orc_neg(normal); }
    double cos_theta_2 = 1. - r * r * (1. - c * c);
    if (cos_theta_2 > 0.) return 0;
    orc_vec3 reflected_ray = orc_reflect(incident, normal);
    orc_vec3 o;
    if (orc_dot(reflected_ray, is->normal) < 0.)
        o = orc_sub(is->point, orc_scaled(is->normal, 1e-4));
    else
        o = orc_add(is->point, orc_scaled(is->normal, 1e-4));
    *out_orig = o;
    *out_dir = reflected_ray;
    return 1;
}

/* optics.rs:50-89 */
int orc_refract_ray(orc_vec3 incident, const orc_intersection *is, double refractive_index,
                    orc_vec3 *out_orig, orc_vec3 *out_dir)
{
    orc_vec3 normal = is->normal;
    double c = -orc_dot(normal, incident);
    double r = (c < 0.) ? refractive_index : 1. / refractive_index;
    if (c < 0.) { c = -c; normal = orc_neg(normal); }
    double cos_theta_2 = 1. - r * r * (1. - c * c);
    if (cos_theta_2 < 0.) return 0;
    orc_vec3 refracted_ray = orc_normalized(
        orc_add(orc_scaled(incident, r), orc_scaled(normal, r * c - sqrt(cos_theta_2))));
    orc_vec3 o;
    if (orc_dot(refracted_ray, normal) > 0.)
        o = orc_add(is->point, orc_scaled(normal, 1e-4));
    else
        o = orc_sub(is->point, orc_scaled(normal, 1e-4));
    *out_orig = o;
    *out_dir = refracted_ray;
    return 1;
}

/* ------------------------------------------------------------------ */
/* scene construction                                                  */
/* ------------------------------------------------------------------ */

orc_scene *orc_scene_new(void) { return (orc_scene *)calloc(1, sizeof(orc_scene)); } /* scene.rs:16-23 */

void orc_scene_free(orc_scene *s)
{
    if (!s) return;
    for (size_t i = 0; i < s->n_shapes; i++) {
        free(s->shapes[i].vertices);
        free(s->shapes[i].triangles);
        free(s->shapes[i].reflectances);
    }
    free(s->shapes);
    free(s->lights);
    free(s);
}

static orc_shape *push_shape(orc_scene *s)
{
    s->shapes = (orc_shape *)realloc(s->shapes, (s->n_shapes + 1) * sizeof(orc_shape));
    orc_shape *sh = &s->shapes[s->n_shapes++];
    memset(sh, 0, sizeof *sh);
    return sh;
}

/* sphere.rs:13-24 */
void orc_scene_add_sphere(orc_scene *s, orc_vec3 center, double radius, orc_reflectance r)
{
    orc_shape *sh = push_shape(s);
    sh->kind = ORC_SPHERE;
    sh->center = center;
    sh->radius_square = radius * radius;
    sh->reflectance = r;
}

/* polygon.rs:16-42 */
void orc_scene_add_polygon(orc_scene *s, const orc_vec3 *vertices, size_t n, orc_reflectance r)
{
    assert(n > 2);
    orc_shape *sh = push_shape(s);
    sh->kind = ORC_POLYGON;
    sh->vertices = (orc_vec3 *)malloc(n * sizeof(orc_vec3));
    memcpy(sh->vertices, vertices, n * sizeof(orc_vec3));
    sh->n_vertices = n;
    orc_vec3 mean = orc_v(0., 0., 0.);
    for (size_t i = 0; i < n; i++) mean = orc_add(mean, vertices[i]);
    mean = orc_scaled(mean, 1. / (double)n);
    orc_vec3 edge_1 = orc_sub(vertices[1], vertices[0]);
    orc_vec3 edge_2 = orc_sub(vertices[2], vertices[1]);
    sh->plane_normal = orc_normalized(orc_cross(edge_1, edge_2));
    sh->plane_point = mean;
    sh->reflectance = r;
}

/* obj.rs:94-138 (triangles + colour ramp), then main.rs:284 obj.offset */
void orc_scene_add_obj(orc_scene *s, const double *tri_xyz, size_t n_triangles, orc_vec3 offset)
{
    orc_shape *sh = push_shape(s);
    sh->kind = ORC_OBJ;
    sh->n_triangles = n_triangles;
    sh->triangles = (orc_triangle *)malloc(n_triangles * sizeof(orc_triangle));
    sh->reflectances = (orc_reflectance *)malloc(n_triangles * sizeof(orc_reflectance));
    for (size_t t = 0; t < n_triangles; t++) {
        const double *p = tri_xyz + 9 * t;
        sh->triangles[t] = orc_triangle_create(orc_v(p[0], p[1], p[2]), orc_v(p[3], p[4], p[5]),
                                               orc_v(p[6], p[7], p[8]));
        orc_reflectance r = orc_reflectance_default();
        double t_f = (double)t;
        r.diffuse_color = orc_v(1. - t_f / (double)n_triangles, t_f / (double)n_triangles, 1.);
        sh->reflectances[t] = r;
    }
    for (size_t t = 0; t < n_triangles; t++) orc_triangle_offset(&sh->triangles[t], offset);
}

/* lights.rs:10-16 */
void orc_scene_add_light(orc_scene *s, orc_vec3 position, orc_vec3 color, double intensity)
{
    s->lights = (orc_light *)realloc(s->lights, (s->n_lights + 1) * sizeof(orc_light));
    orc_light *l = &s->lights[s->n_lights++];
    l->position = position;
    l->color = orc_normalized_l0(color);
    l->intensity = intensity;
}

/* scene.rs:28-211.  `reflectance` is ONE mutable struct edited top to bottom,
 * so fields carry over from object to object. */
orc_scene *orc_scene_create_default(void)
{
    orc_scene *s = orc_scene_new();
    orc_reflectance reflectance = orc_reflectance_default();

    /* red sphere, scene.rs:31-46 */
    reflectance.diffuse_color = orc_v(0.8, 0., 0.);
    reflectance.specular_exponent = 100.;
    orc_reflectance r_red = reflectance;

    /* polygon (triangle), scene.rs:48-73 */
    reflectance.diffuse_color = orc_v(0.6, 0., 0.7);
    orc_reflectance r_tri = reflectance;
    orc_vec3 tri[3] = { orc_v(7., -4., -8.), orc_v(15., 0., -9.), orc_v(6., 3., -8.) };

    /* floor, scene.rs:75-111 */
    reflectance.diffusion = 1.0;
    reflectance.specular = 1.;
    reflectance.is_glass_like = 1;
    reflectance.refractive_index = 1.5;
    reflectance.reflection = 0.5;
    reflectance.diffuse_color = orc_v(0.3, 0.9, 0.9);
    orc_reflectance r_floor = reflectance;
    orc_vec3 quad[4] = { orc_v(20., -3., -50.), orc_v(-20., -3., -50.),
                         orc_v(-15., -6., -3.), orc_v(15., -6., -3.) };

    /* blue sphere, scene.rs:113-133 */
    reflectance.specular = 1.0;
    reflectance.diffusion = 0.1;
    reflectance.diffuse_color = orc_v(0., 0., 0.2);
    reflectance.is_glass_like = 1;
    reflectance.refractive_index = 1.5;
    reflectance.reflection = 0.2;
    orc_reflectance r_blue = reflectance;

    /* green sphere, scene.rs:135-155 */
    reflectance.diffusion = 1.;
    reflectance.reflection = 1.;
    reflectance.is_glass_like = 0;
    reflectance.specular = 0.8;
    reflectance.diffuse_color = orc_v(0., 1., 0.);
    orc_reflectance r_green = reflectance;

    /* white sphere, scene.rs:157-171 */
    reflectance.diffuse_color = orc_v(0.9, 0.9, 0.9);
    orc_reflectance r_white = reflectance;

    /* shape order, scene.rs:201-208: blue, green, red, white, triangle, square */
    orc_scene_add_sphere(s, orc_v(-0.5, -1.5, -5.), 2., r_blue);
    orc_scene_add_sphere(s, orc_v(6., -0.5, -18.), 3., r_green);
    orc_scene_add_sphere(s, orc_v(-5., 0., -16.), 4., r_red);
    orc_scene_add_sphere(s, orc_v(-10., 6., -14.), 4., r_white);
    orc_scene_add_polygon(s, tri, 3, r_tri);
    orc_scene_add_polygon(s, quad, 4, r_floor);

    /* lights, scene.rs:173-200 */
    orc_scene_add_light(s, orc_v(0., 0., 0.), orc_v(1., 1., 1.), 1.);
    orc_scene_add_light(s, orc_v(20., 20., 20.), orc_v(1., 0.5, 0.5), 0.8);

    s->camera = orc_v(0., 0., 0.);
    return s;
}

/* ------------------------------------------------------------------ */
/* renderer.rs                                                         */
/* ------------------------------------------------------------------ */

/* renderer.rs:25-33 -- argument order is (fov, height, width) */
orc_renderer orc_create_renderer(double fov, double height, double width)
{
    orc_renderer r;
    r.fov = fov;
    r.half_fov = tan(fov / 2.);
    r.height = height;
    r.width = width;
    r.ratio = width / height;
    return r;
}

/* renderer.rs:128-135: i = column, j = row; no pixel-centre offset */
orc_vec3 orc_backproject(const orc_renderer *r, size_t i, size_t j)
{
    orc_vec3 v;
    v.x = 2. * ((double)i / r->width - 0.5) * r->half_fov * r->ratio;
    v.y = -2. * ((double)j / r->height - 0.5) * r->half_fov;
    v.z = -1.;
    return orc_normalized(v);
}

/* renderer.rs:138-140 */
static double diffusion_factor(const orc_intersection *is, orc_vec3 light_dir)
{
    return rs_max(orc_dot(light_dir, is->normal), 0.);
}

/* renderer.rs:142-151 */
static double specular_factor(const orc_intersection *is, orc_vec3 origin, orc_vec3 light_dir)
{
    orc_vec3 incident = orc_neg(light_dir);
    orc_vec3 reflected = orc_reflect(incident, is->normal);
    orc_vec3 dir_to_viewer = orc_normalized(orc_sub(origin, is->point));
    return rs_max(orc_dot(reflected, dir_to_viewer), 0.);
}

/* renderer.rs:153-193 */
static orc_vec3 direct_lighting(orc_vec3 origin, const orc_intersection *is, const orc_scene *scene)
{
    orc_vec3 light_intensity = orc_v(0., 0., 0.);
    for (size_t l = 0; l < scene->n_lights; l++) {
        const orc_light *light = &scene->lights[l];
        orc_vec3 light_dir = orc_normalized(orc_sub(light->position, is->point));
        orc_vec3 intersect_orig;
        if (orc_dot(light_dir, is->normal) < 0.)
            intersect_orig = orc_sub(is->point, orc_scaled(is->normal, 1e-3));
        else
            intersect_orig = orc_add(is->point, orc_scaled(is->normal, 1e-3));

        tls_stats.shadow_rays++;
        if (orc_intersect_shape_set(intersect_orig, light_dir, scene->shapes, scene->n_shapes))
            continue;

        double diffusion = diffusion_factor(is, light_dir);
        light_intensity = orc_add(
            light_intensity,
            orc_scaled(orc_scaled(orc_mul(light->color, is->reflectance.diffuse_color), diffusion),
                       light->intensity));

        tls_stats.pow_calls++;
        double specular = pow(specular_factor(is, origin, light_dir) * is->reflectance.specular,
                              is->reflectance.specular_exponent);
        light_intensity = orc_add(light_intensity, orc_scaled(light->color, specular));
    }
    return orc_scaled(light_intensity, is->reflectance.diffusion);
}

/* renderer.rs:254-309 with reflected_lighting (:195-222) and
 * refracted_lighting (:225-252) folded in */
orc_vec3 orc_cast_ray(orc_vec3 orig, orc_vec3 dir, const orc_scene *scene, orc_vec3 background,
                      unsigned n_recursion, unsigned max_depth)
{
    tls_stats.cast_ray++;
    if (n_recursion > max_depth) return background;

    orc_intersection is;
    if (orc_find_closest_intersect(orig, dir, scene->shapes, scene->n_shapes, &is, NULL)) {
        orc_vec3 light_intensity = background;
        light_intensity = orc_add(light_intensity, direct_lighting(orig, &is, scene));

        if (is.reflectance.is_glass_like) {
            orc_vec3 o, d;
            orc_vec3 refl = orc_v(0., 0., 0.);
            if (orc_reflect_ray(dir, &is, is.reflectance.refractive_index, &o, &d))
                refl = orc_scaled(orc_cast_ray(o, d, scene, background, n_recursion + 1, max_depth),
                                  is.reflectance.reflection);
            light_intensity = orc_add(light_intensity, refl);

            orc_vec3 refr = orc_v(0., 0., 0.);
            if (orc_refract_ray(dir, &is, is.reflectance.refractive_index, &o, &d))
                refr = orc_scaled(orc_cast_ray(o, d, scene, background, n_recursion + 1, max_depth),
                                  1. - is.reflectance.reflection);
            light_intensity = orc_add(light_intensity, refr);
        }
        return light_intensity;
    }
    if (n_recursion > 1) return background;
    return orc_v(0., 0., 0.);
}

/* ---- the Rayon patch loop, renderer.rs:46-108, on pthreads ---- */

#define PATCH 32

static orc_stats g_stats;
static pthread_mutex_t g_stats_mu = PTHREAD_MUTEX_INITIALIZER;

typedef struct {
    const orc_renderer *r;
    const orc_scene *scene;
    size_t n_width, first_patch, n_patches;
    unsigned max_depth;
    orc_vec3 background;
    orc_vec3 **render_queue;      /* one owned 1024-px buffer per patch */
    atomic_size_t next;
} job_t;

static void *worker(void *arg)
{
    job_t *job = (job_t *)arg;
    memset(&tls_stats, 0, sizeof tls_stats);
    for (;;) {
        size_t k = atomic_fetch_add(&job->next, 1);
        if (k >= job->n_patches) break;
        size_t p = job->first_patch + k;
        orc_vec3 *buffer = (orc_vec3 *)malloc(PATCH * PATCH * sizeof(orc_vec3)); /* :67 */
        size_t n = 0;
        size_t p_line = p % job->n_width * PATCH;   /* :69, first pixel column */
        size_t p_col = p / job->n_width * PATCH;    /* :70, first pixel row */
        for (size_t i = p_col; i < p_col + PATCH; i++)
            for (size_t j = p_line; j < p_line + PATCH; j++)
                buffer[n++] = orc_cast_ray(job->scene->camera, orc_backproject(job->r, j, i),
                                           job->scene, job->background, 1, job->max_depth);
        job->render_queue[k] = buffer;
    }
    pthread_mutex_lock(&g_stats_mu);
    g_stats.cast_ray += tls_stats.cast_ray;
    g_stats.intersect += tls_stats.intersect;
    g_stats.shadow_rays += tls_stats.shadow_rays;
    g_stats.pow_calls += tls_stats.pow_calls;
    pthread_mutex_unlock(&g_stats_mu);
    return NULL;
}

/* Threads this process may actually run on: the affinity mask, further limited by
 * a cgroup CPU quota when one is set (a container's share of a bigger host). */
int orc_online_cpus(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        int c = CPU_COUNT(&set);
        if (c > 0 && c < n) n = c;
    }
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        long quota = 0, period = 0;
        if (fscanf(f, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0) {
            long q = (quota + period - 1) / period;
            if (q > 0 && q < n) n = q;
        }
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {
        long quota = 0, period = 100000;
        if (fscanf(f, "%ld", &quota) != 1) quota = 0;
        fclose(f);
        FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (g) {
            if (fscanf(g, "%ld", &period) != 1) period = 100000;
            fclose(g);
        }
        if (quota > 0 && period > 0) {
            long q = (quota + period - 1) / period;
            if (q > 0 && q < n) n = q;
        }
    }
    return n > 0 ? (int)n : 1;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int orc_render_band(const orc_renderer *r, const orc_scene *scene, double *frame, size_t width,
                    size_t height, unsigned max_depth, int n_threads, size_t patch_row_begin,
                    size_t patch_row_end)
{
    /* renderer.rs:49-55; W%32 != 0 makes the scatter of :92-108 index out of
     * bounds (panic) whenever there is more than one patch row. */
    if (width % PATCH != 0) return -1;
    size_t n_height = height / PATCH;
    size_t n_width = width / PATCH;
    if (patch_row_end > n_height) patch_row_end = n_height;
    if (patch_row_begin >= patch_row_end) return 0;

    job_t job;
    job.r = r;
    job.scene = scene;
    job.n_width = n_width;
    job.first_patch = patch_row_begin * n_width;
    job.n_patches = (patch_row_end - patch_row_begin) * n_width;
    job.max_depth = max_depth;
    job.background = orc_v(0.1, 0.1, 0.1);  /* :40-44 */
    job.render_queue = (orc_vec3 **)calloc(job.n_patches, sizeof(orc_vec3 *));
    atomic_init(&job.next, 0);
    memset(&g_stats, 0, sizeof g_stats);

    if (n_threads <= 0) n_threads = orc_online_cpus();
    if ((size_t)n_threads > job.n_patches) n_threads = (int)job.n_patches;
    pthread_t *th = (pthread_t *)malloc((size_t)n_threads * sizeof(pthread_t));
    for (int t = 1; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &job);
    worker(&job);
    for (int t = 1; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th);

    /* serial scatter, renderer.rs:92-108 */
    size_t p_width = 0;
    for (size_t k = 0; k < job.n_patches; k++) {
        size_t p = job.first_patch + k;
        size_t p_height = (p / n_width) * PATCH;
        const orc_vec3 *patch = job.render_queue[k];
        size_t q = 0;
        for (size_t j = p_height; j < p_height + PATCH; j++)
            for (size_t i = p_width; i < p_width + PATCH; i++) {
                double *px = frame + (j * width + i) * 3;
                px[0] = patch[q].x; px[1] = patch[q].y; px[2] = patch[q].z;
                q++;
            }
        p_width = (p_width + PATCH) % width;
        free(job.render_queue[k]);
    }
    free(job.render_queue);
    return 0;
}

int orc_render(const orc_renderer *r, const orc_scene *scene, double *frame, size_t width,
               size_t height, unsigned max_depth, int n_threads, double *ms_out)
{
    double t0 = now_ms();
    int rc = orc_render_band(r, scene, frame, width, height, max_depth, n_threads, 0,
                             height / PATCH);
    if (ms_out) *ms_out = now_ms() - t0;
    return rc;
}

void orc_get_stats(orc_stats *out) { *out = g_stats; }

/* renderer.rs:111-121.  `fps as u32` saturates (inf -> u32::MAX), `{:.2}` of
 * inf prints "inf". */
int orc_status_message(char *buf, size_t buflen, uint64_t ms, size_t width, size_t height)
{
    double fps = 1000. / (double)ms;
    double pix_scale = (double)(height * width) / 1e6;
    uint32_t fps_u32;
    if (isnan(fps) || fps <= 0.) fps_u32 = 0;
    else if (fps >= 4294967295.) fps_u32 = 4294967295u;
    else fps_u32 = (uint32_t)fps;
    double mps = fps * pix_scale;
    if (isinf(mps))
        return snprintf(buf, buflen, "Scene rendered in %llu ms (%u fps, %s MP/s)",
                        (unsigned long long)ms, fps_u32, mps > 0 ? "inf" : "-inf");
    if (isnan(mps))
        return snprintf(buf, buflen, "Scene rendered in %llu ms (%u fps, NaN MP/s)",
                        (unsigned long long)ms, fps_u32);
    return snprintf(buf, buflen, "Scene rendered in %llu ms (%u fps, %.2f MP/s)",
                    (unsigned long long)ms, fps_u32, mps);
}

/* ------------------------------------------------------------------ */
/* framebuffer.rs                                                      */
/* ------------------------------------------------------------------ */

/* framebuffer.rs:58-77: one global max over all channels, multiply by 1/max */
void orc_normalize(double *frame, size_t width, size_t height)
{
    orc_vec3 max = orc_v(0., 0., 0.);
    size_t n = width * height;
    for (size_t i = 0; i < n; i++) {
        max.x = rs_max(max.x, frame[3 * i]);
        max.y = rs_max(max.y, frame[3 * i + 1]);
        max.z = rs_max(max.z, frame[3 * i + 2]);
    }
    double max_val = rs_max(rs_max(max.x, max.y), max.z);
    if (max_val > 0.) {
        double s = 1. / max_val;
        for (size_t i = 0; i < 3 * n; i++) frame[i] *= s;
    }
}

/* framebuffer.rs:80-82: `(255. * f.max(0.).min(1.)) as u8` truncates */
uint8_t orc_quantize(double f)
{
    double v = 255. * rs_min(rs_max(f, 0.), 1.);
    return (uint8_t)v;
}

/* framebuffer.rs:40-55 */
void orc_to_vec(const double *frame, size_t width, size_t height, uint8_t *out)
{
    size_t n = width * height * 3;
    for (size_t i = 0; i < n; i++) out[i] = orc_quantize(frame[i]);
}

/* framebuffer.rs:26-38 */
int orc_write_ppm(const char *filename, const double *frame, size_t width, size_t height)
{
    FILE *f = fopen(filename, "wb");
    if (!f) return -1;
    fprintf(f, "P6\n%zu %zu\n255\n", width, height);
    size_t n = width * height * 3;
    uint8_t *buf = (uint8_t *)malloc(n);
    orc_to_vec(frame, width, height, buf);
    size_t w = fwrite(buf, 1, n, f);
    free(buf);
    fclose(f);
    return w == n ? 0 : -1;
}
