// rm_trace.hpp -- device functions of the render path (gfx950, FP64).
//
// What each function replaces in the reference (engine/src):
//   closest_hit   : find_closest_intersect (shapes.rs:110-143) over
//                   Sphere::intersect (sphere.rs:27-61), ConvexPolygon::intersect
//                   (polygon.rs:60-98), Obj/Triangle::intersect (obj.rs:185-221,
//                   triangle.rs:49-83)
//   any_hit       : intersect_shape_set (shapes.rs:92-108)
//   shade_direct  : direct_lighting + diffusion_factor + specular_factor
//                   (renderer.rs:138-193)
//   reflect_child / refract_child : reflect_ray / refract_ray (optics.rs:8-89)
//
// Every lane owns one ray.  All lanes of a wave walk the same primitive list
// (wave-uniform loop counters, scene words read from LDS as broadcasts); the
// per-lane state is the ray, the best hit so far and the lane's EXEC bit.
// Arithmetic is IEEE binary64 in the reference's operation order; the file is
// compiled without FMA contraction unless the build says otherwise.
#ifndef RM_TRACE_HPP
#define RM_TRACE_HPP

#include <hip/hip_runtime.h>

#include "rm_internal.h"

namespace rmdev {

struct V3 { double x, y, z; };

__device__ __forceinline__ V3 mk(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 scaled(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
// geometry.rs:180-182: (x*x' + y*y') + z*z'
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// geometry.rs:104-109: multiply by the reciprocal of the norm
__device__ __forceinline__ V3 normalized(V3 a) {
    double norm = __builtin_sqrt(dot(a, a));
    if (norm > 0.) {
        double inv = 1. / norm;
        return scaled(a, inv);
    }
    return a;
}

// Scene view: S points at the blob (LDS copy), H carries counts and offsets in SGPRs.
struct SceneView {
    const double *S;
    const double *__restrict__ G;
    rm_dev_header H;
};

__device__ __forceinline__ uint32_t prim_key(const SceneView &sc, uint32_t pid) {
    const uint32_t *keys = reinterpret_cast<const uint32_t *>(sc.S + sc.H.off_keys);
    return keys[pid];
}

// shapes.rs:130 / obj.rs:198: `!hit || dist_hit < dist_closest`, first in list
// order wins an exact tie.  Primitives are visited grouped by kind, so an exact
// tie consults the list-order keys unless grouping preserved the list order.
__device__ __forceinline__ bool closer(const SceneView &sc, bool hit, double d, double best, uint32_t pid,
                                       uint32_t best_pid) {
    if (!hit || d < best) return true;
    if (d == best && !sc.H.list_ordered) return prim_key(sc, pid) < prim_key(sc, best_pid);
    return false;
}

struct Hit {
    double t;       // ray parameter of the closest hit (point = orig + dir * t)
    uint32_t pid;   // device primitive id
};

// ---- closest hit ---------------------------------------------------------
__device__ __forceinline__ bool closest_hit(const SceneView &sc, V3 o, V3 d, Hit &out) {
    const double *__restrict__ S = sc.G;
    bool hit = false;
    double best = 0., best_t = 0.;
    uint32_t best_pid = 0;

    // sphere.rs:27-61
    for (uint32_t i = 0; i < sc.H.n_spheres; i++) {
        const double *sp = S + sc.H.off_spheres + RM_SPHERE_WORDS * i;
        V3 line = mk(sp[0] - o.x, sp[1] - o.y, sp[2] - o.z);
        double r2 = sp[3];
        double tca = dot(line, d);
        double d2 = dot(line, line) - tca * tca;
        if (d2 > r2) continue;
        double thc = __builtin_sqrt(r2 - d2);
        double t0 = tca - thc;
        double t1 = tca + thc;
        if (t0 < 0.) t0 = t1;
        if (t0 < 0.) continue;
        V3 p = o + scaled(d, t0);
        V3 dp = p - o;
        double dist = dot(dp, dp);                       // shapes.rs:128
        if (closer(sc, hit, dist, best, i, best_pid)) { hit = true; best = dist; best_t = t0; best_pid = i; }
    }

    // polygon.rs:60-98
    for (uint32_t i = 0; i < sc.H.n_polygons; i++) {
        const double *pg = S + sc.H.off_polygons + RM_POLYGON_WORDS * i;
        V3 n = mk(pg[0], pg[1], pg[2]);
        V3 pp = mk(pg[3], pg[4], pg[5]);
        const uint32_t first = reinterpret_cast<const uint32_t *>(pg + 6)[0];
        const uint32_t nv = reinterpret_cast<const uint32_t *>(pg + 6)[1];
        double dotprod = dot(d, n);
        bool alive = !(dotprod == 0.);
        double dist = dot(pp - o, n) / dotprod;
        alive = alive && !(dist < 0.);
        V3 p = o + scaled(d, dist);
        const double *pv = S + sc.H.off_pverts + RM_PVERT_WORDS * first;
        double ax = pv[0] - p.x, ay = pv[1] - p.y;       // vertex 0 relative to the hit point
        const double ax0 = ax, ay0 = ay;
        for (uint32_t e = 0; e < nv; e++) {
            double bx, by;
            if (e + 1 < nv) { bx = pv[2 * (e + 1)] - p.x; by = pv[2 * (e + 1) + 1] - p.y; }
            else            { bx = ax0; by = ay0; }
            // polygon.rs:54-56: ((v_i - p) x (v_{i+1} - p)).z > 0
            alive = alive && (ax * by - ay * bx > 0.);
            ax = bx; ay = by;
        }
        if (!alive) continue;
        V3 dp = p - o;
        double dh = dot(dp, dp);
        const uint32_t pid = sc.H.n_spheres + i;
        if (closer(sc, hit, dh, best, pid, best_pid)) { hit = true; best = dh; best_t = dist; best_pid = pid; }
    }

    // triangle.rs:49-83 inside obj.rs:193-210
    for (uint32_t i = 0; i < sc.H.n_triangles; i++) {
        const double *tr = S + sc.H.off_triangles + RM_TRIANGLE_WORDS * i;
        V3 n = mk(tr[0], tr[1], tr[2]);
        V3 c = mk(tr[3], tr[4], tr[5]);
        double dotprod = dot(d, n);
        bool alive = !(__builtin_fabs(dotprod) < 1e-6);
        double dist = dot(c - o, n) / dotprod;
        alive = alive && !(dist < 0.);
        V3 p = o + scaled(d, dist);
        double ax = tr[6] - p.x, ay = tr[7] - p.y;
        double bx = tr[8] - p.x, by = tr[9] - p.y;
        double cx = tr[10] - p.x, cy = tr[11] - p.y;
        alive = alive && (ax * by - ay * bx > 0.);
        alive = alive && (bx * cy - by * cx > 0.);
        alive = alive && (cx * ay - cy * ax > 0.);
        if (!alive) continue;
        V3 dp = p - o;
        double dh = dot(dp, dp);
        const uint32_t pid = sc.H.n_spheres + sc.H.n_polygons + i;
        if (closer(sc, hit, dh, best, pid, best_pid)) { hit = true; best = dh; best_t = dist; best_pid = pid; }
    }

    out.t = best_t;
    out.pid = best_pid;
    return hit;
}

// ---- any hit (shadow rays) -------------------------------------------------
// shapes.rs:92-108.  The answer does not depend on visiting order.  A lane that
// has found an occluder keeps walking with its result latched; the wave leaves
// a loop early once every active lane is occluded.
__device__ __forceinline__ bool any_hit(const SceneView &sc, V3 o, V3 d) {
    const double *__restrict__ S = sc.G;
    bool occ = false;

    for (uint32_t i = 0; i < sc.H.n_spheres; i++) {
        const double *sp = S + sc.H.off_spheres + RM_SPHERE_WORDS * i;
        V3 line = mk(sp[0] - o.x, sp[1] - o.y, sp[2] - o.z);
        double r2 = sp[3];
        double tca = dot(line, d);
        double d2 = dot(line, line) - tca * tca;
        if (!occ && !(d2 > r2)) {
            // sphere.rs:43-52: hit unless both roots are negative.  thc >= 0, so
            // tca >= 0 already gives t1 = tca + thc >= 0 without the square root.
            if (!(tca < 0.)) {
                occ = true;
            } else {
                double thc = __builtin_sqrt(r2 - d2);
                double t0 = tca - thc;
                double t1 = tca + thc;
                if (t0 < 0.) t0 = t1;
                if (!(t0 < 0.)) occ = true;
            }
        }
        if (__all(occ)) return true;
    }

    for (uint32_t i = 0; i < sc.H.n_polygons; i++) {
        const double *pg = S + sc.H.off_polygons + RM_POLYGON_WORDS * i;
        V3 n = mk(pg[0], pg[1], pg[2]);
        V3 pp = mk(pg[3], pg[4], pg[5]);
        const uint32_t first = reinterpret_cast<const uint32_t *>(pg + 6)[0];
        const uint32_t nv = reinterpret_cast<const uint32_t *>(pg + 6)[1];
        double dotprod = dot(d, n);
        bool alive = !(dotprod == 0.);
        double dist = dot(pp - o, n) / dotprod;
        alive = alive && !(dist < 0.);
        V3 p = o + scaled(d, dist);
        const double *pv = S + sc.H.off_pverts + RM_PVERT_WORDS * first;
        double ax = pv[0] - p.x, ay = pv[1] - p.y;
        const double ax0 = ax, ay0 = ay;
        for (uint32_t e = 0; e < nv; e++) {
            double bx, by;
            if (e + 1 < nv) { bx = pv[2 * (e + 1)] - p.x; by = pv[2 * (e + 1) + 1] - p.y; }
            else            { bx = ax0; by = ay0; }
            alive = alive && (ax * by - ay * bx > 0.);
            ax = bx; ay = by;
        }
        occ = occ || alive;
        if (__all(occ)) return true;
    }

    for (uint32_t i = 0; i < sc.H.n_triangles; i++) {
        const double *tr = S + sc.H.off_triangles + RM_TRIANGLE_WORDS * i;
        V3 n = mk(tr[0], tr[1], tr[2]);
        V3 c = mk(tr[3], tr[4], tr[5]);
        double dotprod = dot(d, n);
        bool alive = !(__builtin_fabs(dotprod) < 1e-6);
        double dist = dot(c - o, n) / dotprod;
        alive = alive && !(dist < 0.);
        V3 p = o + scaled(d, dist);
        double ax = tr[6] - p.x, ay = tr[7] - p.y;
        double bx = tr[8] - p.x, by = tr[9] - p.y;
        double cx = tr[10] - p.x, cy = tr[11] - p.y;
        alive = alive && (ax * by - ay * bx > 0.);
        alive = alive && (bx * cy - by * cx > 0.);
        alive = alive && (cx * ay - cy * ax > 0.);
        occ = occ || alive;
        if (__all(occ)) return true;
    }
    return occ;
}

// ---- surface record ----------------------------------------------------------
struct Surface {
    V3 point, normal;
    const double *mat;   // RM_MATERIAL_WORDS words in the scene blob
};

__device__ __forceinline__ Surface surface_at(const SceneView &sc, V3 o, V3 d, const Hit &h) {
    Surface s;
    s.point = o + scaled(d, h.t);
    const uint32_t ns = sc.H.n_spheres, np = sc.H.n_polygons;
    if (h.pid < ns) {
        const double *sp = sc.S + sc.H.off_spheres + RM_SPHERE_WORDS * h.pid;
        s.normal = normalized(mk(s.point.x - sp[0], s.point.y - sp[1], s.point.z - sp[2]));   // sphere.rs:58
    } else if (h.pid < ns + np) {
        const double *pg = sc.S + sc.H.off_polygons + RM_POLYGON_WORDS * (h.pid - ns);
        s.normal = mk(pg[0], pg[1], pg[2]);                                                    // polygon.rs:95
    } else {
        const double *tr = sc.S + sc.H.off_triangles + RM_TRIANGLE_WORDS * (h.pid - ns - np);
        s.normal = mk(tr[0], tr[1], tr[2]);                                                    // triangle.rs:80
    }
    s.mat = sc.S + sc.H.off_materials + RM_MATERIAL_WORDS * h.pid;
    return s;
}

// optics.rs:4-6
__device__ __forceinline__ V3 reflect(V3 incident, V3 normal) {
    return incident - scaled(normal, 2. * dot(incident, normal));
}

// ---- specular power: f64::powf (renderer.rs:186-188) ---------------------------------
// POW_GENERIC: the device libm pow (<= 1 ulp), any exponent.  Its ~40 polynomial
// constants cannot be f64 literals on gfx950, so the compiler parks them in VGPRs for
// the whole kernel (+38 VGPRs, one wave per SIMD less).
// POW_INTEGER: exponent is a non-negative integer (checked for every material at
// scene upload): binary powering, bit-for-bit the same special cases as pow for such
// exponents (x^0 = 1 also for NaN; 0^n = 0; sign of negative bases), <= ~10 roundings
// instead of 1, no constants.
enum { POW_GENERIC = 0, POW_INTEGER = 1 };

template <int POW>
__device__ __forceinline__ double specular_pow(double x, double y) {
    if (POW == POW_GENERIC) return pow(x, y);
    uint32_t n = (uint32_t)y;
    double r = 1., b = x;
    while (__any(n != 0u)) {
        if (n & 1u) r = r * b;
        b = b * b;
        n >>= 1;
    }
    return r;
}

// ---- direct lighting: renderer.rs:153-193 ---------------------------------------
// `origin` is the origin of the ray that produced the hit (renderer.rs:275), so
// for secondary rays the "viewer" is the previous hit point.
template <int POW>
__device__ __forceinline__ V3 shade_direct(const SceneView &sc, V3 origin, const Surface &s) {
    const double *m = s.mat;
    const V3 diffuse_color = mk(m[1], m[2], m[3]);
    const double specular_k = m[4], exponent = m[5];
    const V3 dir_to_viewer = normalized(origin - s.point);            // renderer.rs:149
    V3 li = mk(0., 0., 0.);
    for (uint32_t l = 0; l < sc.H.n_lights; l++) {
        const double *lt = sc.G + sc.H.off_lights + RM_LIGHT_WORDS * l;
        const V3 lpos = mk(lt[0], lt[1], lt[2]);
        const V3 lcol = mk(lt[3], lt[4], lt[5]);
        const double intensity = lt[6];
        const V3 light_dir = normalized(lpos - s.point);              // :166
        const double ldn = dot(light_dir, s.normal);
        const V3 off = scaled(s.normal, 1e-3);
        const V3 so = (ldn < 0.) ? (s.point - off) : (s.point + off); // :168-172
        if (any_hit(sc, so, light_dir)) continue;                     // :174
        const double diffusion = __builtin_fmax(ldn, 0.);             // :139
        li = li + scaled(scaled(lcol * diffuse_color, diffusion), intensity);   // :181-183
        const V3 reflected = reflect(neg(light_dir), s.normal);       // :144-145
        const double sf = __builtin_fmax(dot(reflected, dir_to_viewer), 0.);    // :150
        const double spec = specular_pow<POW>(sf * specular_k, exponent);   // :186-188
        li = li + scaled(lcol, spec);                                 // :189
    }
    return scaled(li, m[0]);                                          // :192
}

// ---- secondary rays: optics.rs:8-89 ------------------------------------------------
__device__ __forceinline__ bool reflect_child(V3 incident, const Surface &s, double ri, V3 &o, V3 &d) {
    V3 normal = s.normal;
    double c = dot(normal, incident);                                 // optics.rs:16 (no sign flip)
    const double r = (c < 0.) ? ri : 1. / ri;
    if (c < 0.) { c = -c; normal = neg(normal); }
    const double cos_theta_2 = 1. - r * r * (1. - c * c);
    if (cos_theta_2 > 0.) return false;                               // :33-35
    d = reflect(incident, normal);
    const V3 off = scaled(s.normal, 1e-4);
    o = (dot(d, s.normal) < 0.) ? (s.point - off) : (s.point + off);  // :41-45, un-flipped normal
    return true;
}

__device__ __forceinline__ bool refract_child(V3 incident, const Surface &s, double ri, V3 &o, V3 &d) {
    V3 normal = s.normal;
    double c = -dot(normal, incident);                                // optics.rs:57
    const double r = (c < 0.) ? ri : 1. / ri;
    if (c < 0.) { c = -c; normal = neg(normal); }
    const double cos_theta_2 = 1. - r * r * (1. - c * c);
    if (cos_theta_2 < 0.) return false;                               // :74
    d = normalized(scaled(incident, r) + scaled(normal, r * c - __builtin_sqrt(cos_theta_2)));   // :78-79
    const V3 off = scaled(normal, 1e-4);
    o = (dot(d, normal) > 0.) ? (s.point + off) : (s.point - off);    // :82-86, flipped normal
    return true;
}

}  // namespace rmdev
#endif
