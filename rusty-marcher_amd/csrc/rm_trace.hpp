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

// ---- primitive records as the loops read them ------------------------------------
// The loops below are wave-uniform: the records are fetched with scalar loads into
// SGPRs and feed the FP64 ops as scalar operands.  A scalar load costs a full round
// trip to the scalar cache per dependent use, so records are fetched in BATCHES -- four
// spheres, a polygon header together with its first four vertices, two triangles --
// and one wait covers the whole batch.
struct SphereRec { double cx, cy, cz, r2; };
struct PolyRec { double nx, ny, nz, px, py, pz; uint32_t first, nv; };
struct Vert2 { double x, y; };
struct TriRec { double nx, ny, nz, cx, cy, cz, ax, ay, bx, by, ex, ey; };

__device__ __forceinline__ SphereRec load_sphere(const double *__restrict__ S, const rm_dev_header &H, uint32_t i) {
    const double *sp = S + H.off_spheres + RM_SPHERE_WORDS * i;
    return SphereRec{sp[0], sp[1], sp[2], sp[3]};
}
__device__ __forceinline__ PolyRec load_polygon(const double *__restrict__ S, const rm_dev_header &H, uint32_t i) {
    const double *pg = S + H.off_polygons + RM_POLYGON_WORDS * i;
    const uint32_t *u = reinterpret_cast<const uint32_t *>(pg + 6);
    return PolyRec{pg[0], pg[1], pg[2], pg[3], pg[4], pg[5], u[0], u[1]};
}
__device__ __forceinline__ Vert2 load_vertex(const double *__restrict__ S, const rm_dev_header &H, uint32_t v) {
    const double *pv = S + H.off_pverts + RM_PVERT_WORDS * v;
    return Vert2{pv[0], pv[1]};
}
__device__ __forceinline__ TriRec load_triangle(const double *__restrict__ S, const rm_dev_header &H, uint32_t i) {
    const double *t = S + H.off_triangles + RM_TRIANGLE_WORDS * i;
    return TriRec{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11]};
}

// sphere.rs:27-61 up to the discriminant: tca and d2 for one sphere
__device__ __forceinline__ void sphere_setup(const SphereRec &s, V3 o, V3 d, double &tca, double &d2) {
    const V3 line = mk(s.cx - o.x, s.cy - o.y, s.cz - o.z);
    tca = dot(line, d);
    d2 = dot(line, line) - tca * tca;
}

// polygon.rs:60-98: plane hit + 2-D inside test (polygon.rs:54-56: only the z of
// (v_i - p) x (v_{i+1} - p), i.e. x and y of the vertices).  `q` holds the first four
// vertices of the polygon (uploaded scenes pad the pool so the read is in bounds).
__device__ __forceinline__ bool polygon_hit(const double *__restrict__ S, const rm_dev_header &H, const PolyRec &g,
                                            const Vert2 (&q)[4], V3 o, V3 d, double &dist, V3 &p) {
    const V3 n = mk(g.nx, g.ny, g.nz);
    const double dotprod = dot(d, n);
    const double num = dot(mk(g.px - o.x, g.py - o.y, g.pz - o.z), n);
    // :66 parallel, :71-76 `dist = num / dotprod; if dist < 0 -> None`.  The quotient is
    // negative exactly when the signs differ (|dotprod| <= 1, so no underflow to -0 while
    // |num| > 1e-300): such lanes are rejected before the division and the edge tests,
    // and a wave whose lanes are all rejected skips them altogether.
    bool alive = !(dotprod == 0.) && !((num < -1e-300 && dotprod > 0.) || (num > 1e-300 && dotprod < 0.));
    if (!__any(alive)) return false;
    dist = num / dotprod;
    alive = alive && !(dist < 0.);
    p = o + scaled(d, dist);
    const double x0 = q[0].x - p.x, y0 = q[0].y - p.y;
    const double x1 = q[1].x - p.x, y1 = q[1].y - p.y;
    const double x2 = q[2].x - p.x, y2 = q[2].y - p.y;
    alive = alive && (x0 * y1 - y0 * x1 > 0.);
    alive = alive && (x1 * y2 - y1 * x2 > 0.);
    if (g.nv == 3u) {
        alive = alive && (x2 * y0 - y2 * x0 > 0.);
    } else {
        double ax = q[3].x - p.x, ay = q[3].y - p.y;
        alive = alive && (x2 * ay - y2 * ax > 0.);
        for (uint32_t e = 4; e < g.nv; e++) {                         // pentagons and up
            const Vert2 v = load_vertex(S, H, g.first + e);
            const double bx = v.x - p.x, by = v.y - p.y;
            alive = alive && (ax * by - ay * bx > 0.);
            ax = bx; ay = by;
        }
        alive = alive && (ax * y0 - ay * x0 > 0.);
    }
    return alive;
}

// triangle.rs:49-83
__device__ __forceinline__ bool triangle_hit(const TriRec &t, V3 o, V3 d, double &dist, V3 &p) {
    const V3 n = mk(t.nx, t.ny, t.nz);
    const double dotprod = dot(d, n);
    const double num = dot(mk(t.cx - o.x, t.cy - o.y, t.cz - o.z), n);
    // :57 parallel, :62-67 going away -- same early sign test as polygon_hit
    bool alive = !(__builtin_fabs(dotprod) < 1e-6) &&
                 !((num < -1e-300 && dotprod > 0.) || (num > 1e-300 && dotprod < 0.));
    if (!__any(alive)) return false;
    dist = num / dotprod;
    alive = alive && !(dist < 0.);
    p = o + scaled(d, dist);
    const double ax = t.ax - p.x, ay = t.ay - p.y;
    const double bx = t.bx - p.x, by = t.by - p.y;
    const double cx = t.ex - p.x, cy = t.ey - p.y;
    alive = alive && (ax * by - ay * bx > 0.);
    alive = alive && (bx * cy - by * cx > 0.);
    alive = alive && (cx * ay - cy * ax > 0.);
    return alive;
}

// ---- closest hit ---------------------------------------------------------
struct ClosestState {
    bool hit;
    double best, best_t;
    uint32_t best_pid;
};

__device__ __forceinline__ void closest_sphere(const SceneView &sc, ClosestState &c, const SphereRec &s, uint32_t pid,
                                               V3 o, V3 d) {
    double tca, d2;
    sphere_setup(s, o, d, tca, d2);
    if (d2 > s.r2) return;                                           // sphere.rs:37
    const double thc = __builtin_sqrt(s.r2 - d2);
    double t0 = tca - thc;
    const double t1 = tca + thc;
    if (t0 < 0.) t0 = t1;
    if (t0 < 0.) return;
    const V3 p = o + scaled(d, t0);
    const V3 dp = p - o;
    const double dist = dot(dp, dp);                                 // shapes.rs:128
    if (closer(sc, c.hit, dist, c.best, pid, c.best_pid)) { c.hit = true; c.best = dist; c.best_t = t0; c.best_pid = pid; }
}

__device__ __forceinline__ bool closest_hit(const SceneView &sc, V3 o, V3 d, Hit &out) {
    const double *__restrict__ S = sc.G;
    const rm_dev_header &H = sc.H;
    ClosestState c{false, 0., 0., 0u};

    uint32_t i = 0;
    for (; i + 4u <= H.n_spheres; i += 4u) {
        const SphereRec s0 = load_sphere(S, H, i), s1 = load_sphere(S, H, i + 1u);
        const SphereRec s2 = load_sphere(S, H, i + 2u), s3 = load_sphere(S, H, i + 3u);
        closest_sphere(sc, c, s0, i, o, d);
        closest_sphere(sc, c, s1, i + 1u, o, d);
        closest_sphere(sc, c, s2, i + 2u, o, d);
        closest_sphere(sc, c, s3, i + 3u, o, d);
    }
    for (; i < H.n_spheres; i++) closest_sphere(sc, c, load_sphere(S, H, i), i, o, d);

    for (uint32_t g = 0; g < H.n_polygons; g++) {
        const PolyRec pg = load_polygon(S, H, g);
        const Vert2 q[4] = {load_vertex(S, H, pg.first), load_vertex(S, H, pg.first + 1u),
                            load_vertex(S, H, pg.first + 2u), load_vertex(S, H, pg.first + 3u)};
        double dist;
        V3 p;
        if (!polygon_hit(S, H, pg, q, o, d, dist, p)) continue;
        const V3 dp = p - o;
        const double dh = dot(dp, dp);
        const uint32_t pid = H.n_spheres + g;
        if (closer(sc, c.hit, dh, c.best, pid, c.best_pid)) { c.hit = true; c.best = dh; c.best_t = dist; c.best_pid = pid; }
    }

    // obj.rs:193-210: closest triangle, strict `<`
    auto tri = [&](const TriRec &t, uint32_t k) {
        double dist;
        V3 p;
        if (!triangle_hit(t, o, d, dist, p)) return;
        const V3 dp = p - o;
        const double dh = dot(dp, dp);
        const uint32_t pid = H.n_spheres + H.n_polygons + k;
        if (closer(sc, c.hit, dh, c.best, pid, c.best_pid)) { c.hit = true; c.best = dh; c.best_t = dist; c.best_pid = pid; }
    };
    uint32_t k = 0;
    for (; k + 2u <= H.n_triangles; k += 2u) {
        const TriRec t0 = load_triangle(S, H, k), t1 = load_triangle(S, H, k + 1u);
        tri(t0, k);
        tri(t1, k + 1u);
    }
    for (; k < H.n_triangles; k++) tri(load_triangle(S, H, k), k);

    out.t = c.best_t;
    out.pid = c.best_pid;
    return c.hit;
}

// ---- any hit (shadow rays) -------------------------------------------------
// shapes.rs:92-108.  The answer does not depend on visiting order.  A lane that
// has found an occluder keeps walking with its result latched; the wave leaves
// a loop early once every active lane is occluded (one vote per batch).
__device__ __forceinline__ bool shadow_sphere(const SphereRec &s, V3 o, V3 d) {
    double tca, d2;
    sphere_setup(s, o, d, tca, d2);
    if (d2 > s.r2) return false;
    // sphere.rs:43-52: hit unless both roots are negative.  thc >= 0, so tca >= 0
    // already gives t1 = tca + thc >= 0 without the square root.
    if (!(tca < 0.)) return true;
    const double thc = __builtin_sqrt(s.r2 - d2);
    double t0 = tca - thc;
    const double t1 = tca + thc;
    if (t0 < 0.) t0 = t1;
    return !(t0 < 0.);
}

__device__ __forceinline__ bool any_hit(const SceneView &sc, V3 o, V3 d) {
    const double *__restrict__ S = sc.G;
    const rm_dev_header &H = sc.H;
    bool occ = false;

    uint32_t i = 0;
    for (; i + 4u <= H.n_spheres; i += 4u) {
        const SphereRec s0 = load_sphere(S, H, i), s1 = load_sphere(S, H, i + 1u);
        const SphereRec s2 = load_sphere(S, H, i + 2u), s3 = load_sphere(S, H, i + 3u);
        occ = occ || shadow_sphere(s0, o, d);
        occ = occ || shadow_sphere(s1, o, d);
        occ = occ || shadow_sphere(s2, o, d);
        occ = occ || shadow_sphere(s3, o, d);
        if (__all(occ)) return true;
    }
    for (; i < H.n_spheres; i++) occ = occ || shadow_sphere(load_sphere(S, H, i), o, d);
    if (__all(occ)) return true;

    for (uint32_t g = 0; g < H.n_polygons; g++) {
        const PolyRec pg = load_polygon(S, H, g);
        const Vert2 q[4] = {load_vertex(S, H, pg.first), load_vertex(S, H, pg.first + 1u),
                            load_vertex(S, H, pg.first + 2u), load_vertex(S, H, pg.first + 3u)};
        double dist;
        V3 p;
        occ = occ || polygon_hit(S, H, pg, q, o, d, dist, p);
        if (__all(occ)) return true;
    }

    uint32_t k = 0;
    for (; k + 2u <= H.n_triangles; k += 2u) {
        const TriRec t0 = load_triangle(S, H, k), t1 = load_triangle(S, H, k + 1u);
        double dist;
        V3 p;
        occ = occ || triangle_hit(t0, o, d, dist, p);
        occ = occ || triangle_hit(t1, o, d, dist, p);
        if (__all(occ)) return true;
    }
    for (; k < H.n_triangles; k++) {
        double dist;
        V3 p;
        occ = occ || triangle_hit(load_triangle(S, H, k), o, d, dist, p);
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
