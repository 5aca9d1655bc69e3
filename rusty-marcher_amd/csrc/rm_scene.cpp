// rm_scene.cpp -- host side of the C ABI: scene construction and OBJ ingest.
//
// Replaces what the reference runs on the CPU *before* Renderer::render:
// Scene::create_default (scene.rs:28-211), sphere::create (sphere.rs:13-24),
// ConvexPolygon::create (polygon.rs:16-42), Triangle::create/offset
// (triangle.rs:19-47), lights::create_light (lights.rs:10-16), obj::load
// (obj.rs:44-151) and Win::open_obj's scene recipe (main.rs:261-327).
// Output is the flat rm_scene_desc the device side uploads.
//
// Built with -ffp-contract=off: derived quantities (plane normals, centroids,
// radius^2, light colours) must be the same doubles the reference computes.
#include "rm_internal.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

thread_local std::string g_host_error;

// ---- Vec3f (geometry.rs:19-182), operation order preserved ----
inline rm_vec3 v3(double x, double y, double z) { return rm_vec3{x, y, z}; }
inline rm_vec3 operator+(rm_vec3 a, rm_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline rm_vec3 operator-(rm_vec3 a, rm_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline rm_vec3 scaled(rm_vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
inline double dot(rm_vec3 a, rm_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline rm_vec3 cross(rm_vec3 a, rm_vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline rm_vec3 normalized(rm_vec3 a) {              // geometry.rs:104-109
    double norm = std::sqrt(dot(a, a));
    return norm > 0. ? scaled(a, 1. / norm) : a;
}
inline rm_vec3 normalized_l0(rm_vec3 a) {           // geometry.rs:111-116
    double norm = std::fmax(std::fmax(a.x, a.y), a.z);
    return norm > 0. ? scaled(a, 1. / norm) : a;
}

}  // namespace

struct rm_scene {
    std::vector<rm_shape_ref> shapes;
    std::vector<rm_sphere> spheres;
    std::vector<rm_polygon> polygons;
    std::vector<rm_vec3> polygon_vertices;
    std::vector<rm_triangle> triangles;
    std::vector<rm_light> lights;
    rm_vec3 camera{0., 0., 0.};
};

void rm_set_host_error(const std::string &msg) { g_host_error = msg; }
const char *rm_get_host_error() { return g_host_error.c_str(); }

static rm_status fail(rm_status st, const std::string &msg) {
    g_host_error = msg;
    return st;
}

extern "C" {

void rm_reflectance_default(rm_reflectance *out) {
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    out->diffusion = 1.;
    out->diffuse_color = v3(1., 1., 1.);
    out->specular = 1.;
    out->specular_exponent = 30.;
    out->is_glass_like = 0;
    out->reflection = 0.95;
    out->refractive_index = 1.;
}

void rm_create_renderer(double fov, double height, double width, rm_params *out) {
    if (!out) return;
    std::memset(out, 0, sizeof *out);
    out->fov = fov;
    out->half_fov = std::tan(fov / 2.);
    out->height = height;
    out->width = width;
    out->ratio = width / height;
    out->frame_width = width > 0. ? (uint32_t)width : 0u;
    out->frame_height = height > 0. ? (uint32_t)height : 0u;
    out->max_depth = 3;
    out->patch_size = RM_PATCH_SIZE;
    out->background = v3(0.1, 0.1, 0.1);
    out->patch_row_begin = 0;
    out->patch_row_end = 0;
    out->patch_row_stride = 1;
    out->flags = RM_FLAG_NONE;
}

rm_status rm_scene_new(rm_scene **out) {
    if (!out) return fail(RM_ERR_INVALID_ARG, "rm_scene_new: out is NULL");
    *out = new rm_scene();
    return RM_OK;
}

void rm_scene_free(rm_scene *scene) { delete scene; }

rm_status rm_scene_add_sphere(rm_scene *s, rm_vec3 center, double radius, const rm_reflectance *r) {
    if (!s || !r) return fail(RM_ERR_INVALID_ARG, "rm_scene_add_sphere: NULL argument");
    rm_sphere sp;
    sp.center = center;
    sp.radius_square = radius * radius;
    sp.reflectance = *r;
    sp.reflectance._pad = 0;
    s->shapes.push_back(rm_shape_ref{RM_SHAPE_SPHERE, (uint32_t)s->spheres.size(), 1u, 0u});
    s->spheres.push_back(sp);
    return RM_OK;
}

rm_status rm_scene_add_polygon(rm_scene *s, const rm_vec3 *vertices, uint32_t n, const rm_reflectance *r) {
    if (!s || !r || !vertices) return fail(RM_ERR_INVALID_ARG, "rm_scene_add_polygon: NULL argument");
    if (n < 3) return fail(RM_ERR_INVALID_ARG, "rm_scene_add_polygon: fewer than 3 vertices (polygon.rs:18 asserts)");
    rm_polygon p;
    p.first_vertex = (uint32_t)s->polygon_vertices.size();
    p.n_vertices = n;
    rm_vec3 mean = v3(0., 0., 0.);
    for (uint32_t i = 0; i < n; i++) mean = mean + vertices[i];
    mean = scaled(mean, 1. / (double)n);
    p.plane_normal = normalized(cross(vertices[1] - vertices[0], vertices[2] - vertices[1]));
    p.plane_point = mean;
    p.reflectance = *r;
    p.reflectance._pad = 0;
    s->polygon_vertices.insert(s->polygon_vertices.end(), vertices, vertices + n);
    s->shapes.push_back(rm_shape_ref{RM_SHAPE_POLYGON, (uint32_t)s->polygons.size(), 1u, 0u});
    s->polygons.push_back(p);
    return RM_OK;
}

rm_status rm_scene_add_mesh(rm_scene *s, const double *xyz, uint32_t n_triangles, rm_vec3 offset) {
    if (!s || (!xyz && n_triangles)) return fail(RM_ERR_INVALID_ARG, "rm_scene_add_mesh: NULL argument");
    const uint32_t first = (uint32_t)s->triangles.size();
    for (uint32_t t = 0; t < n_triangles; t++) {
        const double *p = xyz + 9 * (size_t)t;
        rm_triangle tr;
        tr.vertices[0] = v3(p[0], p[1], p[2]);
        tr.vertices[1] = v3(p[3], p[4], p[5]);
        tr.vertices[2] = v3(p[6], p[7], p[8]);
        // triangle.rs:33-47: centroid and normal from the *un-offset* vertices
        tr.center = scaled((tr.vertices[0] + tr.vertices[1]) + tr.vertices[2], 1. / 3.);
        tr.normal = normalized(cross(tr.vertices[1] - tr.vertices[0], tr.vertices[2] - tr.vertices[1]));
        // obj.rs:125-138: arbitrary continuous colour ramp over the model
        rm_reflectance_default(&tr.reflectance);
        const double t_f = (double)t;
        tr.reflectance.diffuse_color = v3(1. - t_f / (double)n_triangles, t_f / (double)n_triangles, 1.);
        // triangle.rs:19-24: offset moves centre and vertices, not the normal
        tr.center = tr.center + offset;
        for (auto &v : tr.vertices) v = v + offset;
        s->triangles.push_back(tr);
    }
    s->shapes.push_back(rm_shape_ref{RM_SHAPE_MESH, first, n_triangles, 0u});
    return RM_OK;
}

rm_status rm_scene_add_light(rm_scene *s, rm_vec3 position, rm_vec3 color, double intensity) {
    if (!s) return fail(RM_ERR_INVALID_ARG, "rm_scene_add_light: NULL scene");
    s->lights.push_back(rm_light{position, normalized_l0(color), intensity});
    return RM_OK;
}

rm_status rm_scene_offset_shape(rm_scene *s, uint32_t shape_index, rm_vec3 offset) {
    if (!s) return fail(RM_ERR_INVALID_ARG, "rm_scene_offset_shape: NULL scene");
    if (shape_index >= s->shapes.size()) return fail(RM_ERR_INVALID_ARG, "rm_scene_offset_shape: index out of range");
    const rm_shape_ref &r = s->shapes[shape_index];
    if (r.kind == RM_SHAPE_POLYGON) {                       // polygon.rs:44-49
        rm_polygon &p = s->polygons[r.first];
        p.plane_point = p.plane_point + offset;
        for (uint32_t v = 0; v < p.n_vertices; v++)
            s->polygon_vertices[p.first_vertex + v] = s->polygon_vertices[p.first_vertex + v] + offset;
        return RM_OK;
    }
    if (r.kind == RM_SHAPE_MESH) {                          // obj.rs:24-29, triangle.rs:19-24
        for (uint32_t t = 0; t < r.count; t++) {
            rm_triangle &tr = s->triangles[r.first + t];
            tr.center = tr.center + offset;
            for (auto &v : tr.vertices) v = v + offset;
        }
        return RM_OK;
    }
    return fail(RM_ERR_INVALID_ARG, "rm_scene_offset_shape: spheres have no offset() in the reference");
}

rm_status rm_scene_set_camera(rm_scene *s, rm_vec3 camera) {
    if (!s) return fail(RM_ERR_INVALID_ARG, "rm_scene_set_camera: NULL scene");
    s->camera = camera;
    return RM_OK;
}

rm_status rm_scene_offset_camera(rm_scene *s, rm_vec3 offset) {
    if (!s) return fail(RM_ERR_INVALID_ARG, "rm_scene_offset_camera: NULL scene");
    s->camera = s->camera + offset;
    return RM_OK;
}

rm_status rm_scene_get_desc(const rm_scene *s, rm_scene_desc *out) {
    if (!s || !out) return fail(RM_ERR_INVALID_ARG, "rm_scene_get_desc: NULL argument");
    out->shapes = s->shapes.data();                    out->n_shapes = (uint32_t)s->shapes.size();
    out->spheres = s->spheres.data();                  out->n_spheres = (uint32_t)s->spheres.size();
    out->polygons = s->polygons.data();                out->n_polygons = (uint32_t)s->polygons.size();
    out->polygon_vertices = s->polygon_vertices.data();
    out->n_polygon_vertices = (uint32_t)s->polygon_vertices.size();
    out->triangles = s->triangles.data();              out->n_triangles = (uint32_t)s->triangles.size();
    out->lights = s->lights.data();                    out->n_lights = (uint32_t)s->lights.size();
    out->camera = s->camera;
    return RM_OK;
}

// scene.rs:28-211.  The reference mutates ONE Reflectance top to bottom, so a
// field keeps its last value until reassigned; the table below lists, per
// object in source order, only the assignments the source makes.
rm_status rm_scene_create_default(rm_scene **out) {
    if (!out) return fail(RM_ERR_INVALID_ARG, "rm_scene_create_default: out is NULL");
    rm_scene *s = new rm_scene();

    rm_reflectance cur;
    rm_reflectance_default(&cur);

    cur.diffuse_color = v3(0.8, 0., 0.);            // scene.rs:32-37
    cur.specular_exponent = 100.;
    const rm_reflectance red = cur;

    cur.diffuse_color = v3(0.6, 0., 0.7);           // :49-53
    const rm_reflectance purple = cur;

    cur.diffusion = 1.0;                            // :76-85
    cur.specular = 1.;
    cur.is_glass_like = 1;
    cur.refractive_index = 1.5;
    cur.reflection = 0.5;
    cur.diffuse_color = v3(0.3, 0.9, 0.9);
    const rm_reflectance floor_r = cur;

    cur.specular = 1.0;                             // :114-123
    cur.diffusion = 0.1;
    cur.diffuse_color = v3(0., 0., 0.2);
    cur.is_glass_like = 1;
    cur.refractive_index = 1.5;
    cur.reflection = 0.2;
    const rm_reflectance blue = cur;

    cur.diffusion = 1.;                             // :136-145
    cur.reflection = 1.;
    cur.is_glass_like = 0;
    cur.specular = 0.8;
    cur.diffuse_color = v3(0., 1., 0.);
    const rm_reflectance green = cur;

    cur.diffuse_color = v3(0.9, 0.9, 0.9);          // :158-162
    const rm_reflectance white = cur;

    const rm_vec3 tri[3] = {v3(7., -4., -8.), v3(15., 0., -9.), v3(6., 3., -8.)};          // :54-72
    const rm_vec3 quad[4] = {v3(20., -3., -50.), v3(-20., -3., -50.),                       // :87-110
                             v3(-15., -6., -3.), v3(15., -6., -3.)};

    // list order of scene.rs:201-208
    rm_scene_add_sphere(s, v3(-0.5, -1.5, -5.), 2., &blue);     // :125-133
    rm_scene_add_sphere(s, v3(6., -0.5, -18.), 3., &green);     // :147-155
    rm_scene_add_sphere(s, v3(-5., 0., -16.), 4., &red);        // :39-46
    rm_scene_add_sphere(s, v3(-10., 6., -14.), 4., &white);     // :163-171
    rm_scene_add_polygon(s, tri, 3, &purple);
    rm_scene_add_polygon(s, quad, 4, &floor_r);

    rm_scene_add_light(s, v3(0., 0., 0.), v3(1., 1., 1.), 1.);          // :174-182
    rm_scene_add_light(s, v3(20., 20., 20.), v3(1., 0.5, 0.5), 0.8);    // :184-197
    s->camera = v3(0., 0., 0.);
    *out = s;
    return RM_OK;
}

}  // extern "C"

// --------------------------------------------------------------------------
// OBJ ingest.  The reference delegates to the tobj crate (Cargo.toml:8, "*",
// 3.x API; source not under /root/reference) with LoadOptions{single_index,
// triangulate, ignore_points, ignore_lines} (obj.rs:45-50).  What obj.rs
// consumes (obj.rs:79-115) depends on these tobj behaviours, restated here:
//   * positions are parsed as f32 and widened to f64 (obj.rs:103-105);
//   * a face index i > 0 is 1-based, i < 0 is relative to the vertices read so far;
//   * "o"/"g" closes the current model only if it has faces; the last model is
//     always emitted;
//   * "usemtl" closes the current model only when the material id changes and
//     the model has faces;
//   * polygons are fan-triangulated (v0, v[i], v[i+1]); 1- and 2-vertex faces
//     are dropped (ignore_points / ignore_lines);
//   * every mtllib must load, else obj.rs:64 panics ("WOOPS").
// --------------------------------------------------------------------------
namespace {

struct ObjModel {
    std::string name;
    std::vector<double> tri_xyz;  // 9 per triangle
};

bool parse_f32(const std::string &tok, float *out) {
    if (tok.empty()) return false;
    errno = 0;
    char *end = nullptr;
    float v = std::strtof(tok.c_str(), &end);
    if (end == tok.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}

std::vector<std::string> split_ws(const std::string &line) {
    std::vector<std::string> w;
    std::istringstream is(line);
    std::string t;
    while (is >> t) w.push_back(t);
    return w;
}

std::string rest_after_keyword(const std::string &line, const std::string &kw) {
    size_t p = line.find(kw);
    std::string r = (p == std::string::npos) ? std::string() : line.substr(p + kw.size());
    size_t b = r.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) return std::string();
    size_t e = r.find_last_not_of(" \t\r\n");
    return r.substr(b, e - b + 1);
}

// Minimal MTL reader: we need the name -> id map (it drives model splitting)
// and the load/parse success that obj.rs:64 unwraps.
rm_status load_mtl(const std::string &path, std::map<std::string, size_t> &mat_map, size_t &n_materials,
                   std::string &err) {
    std::ifstream f(path);
    if (!f) {
        err = "could not open material library '" + path + "' (the reference panics: WOOPS, obj.rs:64)";
        return RM_ERR_IO;
    }
    std::string line;
    while (std::getline(f, line)) {
        auto w = split_ws(line);
        if (w.empty() || w[0] == "#") continue;
        if (w[0] == "newmtl") {
            std::string name = rest_after_keyword(line, "newmtl");
            if (name.empty()) { err = "newmtl without a name in '" + path + "'"; return RM_ERR_PARSE; }
            mat_map[name] = n_materials++;
        } else if (w[0] == "Ka" || w[0] == "Kd" || w[0] == "Ks") {
            float tmp;
            if (w.size() < 4 || !parse_f32(w[1], &tmp) || !parse_f32(w[2], &tmp) || !parse_f32(w[3], &tmp)) {
                err = "bad " + w[0] + " in '" + path + "'";
                return RM_ERR_PARSE;
            }
        } else if (w[0] == "Ns" || w[0] == "Ni" || w[0] == "d") {
            float tmp;
            if (w.size() < 2 || !parse_f32(w[1], &tmp)) { err = "bad " + w[0] + " in '" + path + "'"; return RM_ERR_PARSE; }
        }
        // everything else (maps, illum, unknown keys) carries no information we use
    }
    return RM_OK;
}

struct FaceVert { long v; };

bool parse_face_vertex(const std::string &tok, size_t n_pos, long *out) {
    // "v", "v/vt", "v//vn", "v/vt/vn": only the position index matters to obj.rs
    std::string first = tok.substr(0, tok.find('/'));
    if (first.empty()) return false;
    char *end = nullptr;
    long idx = std::strtol(first.c_str(), &end, 10);
    if (*end != '\0') return false;
    long resolved = idx < 0 ? (long)n_pos + idx : idx - 1;
    if (resolved < 0 || (size_t)resolved >= n_pos) return false;
    *out = resolved;
    return true;
}

rm_status load_obj_models(const std::string &path, std::vector<ObjModel> &models, std::string &err) {
    std::ifstream f(path);
    if (!f) {
        err = "Could not load obj from " + path;   // obj.rs:54
        return RM_ERR_IO;
    }
    std::string dir;
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) dir = path.substr(0, slash + 1);

    std::vector<float> pos;                       // f32, as tobj stores them
    std::vector<std::vector<long>> faces;         // faces of the model being read
    std::string name = "unnamed_object";
    std::map<std::string, size_t> mat_map;
    size_t n_materials = 0;
    bool have_mat = false;
    size_t mat_id = 0;

    auto flush = [&](const std::string &model_name) {
        ObjModel m;
        m.name = model_name;
        for (const auto &face : faces) {
            if (face.size() < 3) continue;        // ignore_points / ignore_lines
            for (size_t i = 1; i + 1 < face.size(); i++) {
                const long idx[3] = {face[0], face[i], face[i + 1]};
                for (long k : idx)
                    for (int c = 0; c < 3; c++) m.tri_xyz.push_back((double)pos[3 * (size_t)k + c]);
            }
        }
        models.push_back(std::move(m));
        faces.clear();
    };

    std::string line;
    size_t lineno = 0;
    while (std::getline(f, line)) {
        lineno++;
        auto w = split_ws(line);
        if (w.empty() || w[0] == "#") continue;
        const std::string &kw = w[0];
        if (kw == "v") {
            float xyz[3];
            if (w.size() < 4 || !parse_f32(w[1], &xyz[0]) || !parse_f32(w[2], &xyz[1]) || !parse_f32(w[3], &xyz[2])) {
                err = path + ":" + std::to_string(lineno) + ": bad vertex position";
                return RM_ERR_PARSE;
            }
            pos.insert(pos.end(), xyz, xyz + 3);
        } else if (kw == "f" || kw == "l") {
            std::vector<long> face;
            for (size_t i = 1; i < w.size(); i++) {
                long v;
                if (!parse_face_vertex(w[i], pos.size() / 3, &v)) {
                    err = path + ":" + std::to_string(lineno) + ": bad face index";
                    return RM_ERR_PARSE;
                }
                face.push_back(v);
            }
            if (face.empty()) {
                err = path + ":" + std::to_string(lineno) + ": empty face";
                return RM_ERR_PARSE;
            }
            faces.push_back(std::move(face));
        } else if (kw == "o" || kw == "g") {
            if (!faces.empty()) flush(name);
            name = rest_after_keyword(line, kw);
            if (name.empty()) name = "unnamed_object";
        } else if (kw == "mtllib") {
            if (w.size() < 2) { err = path + ":" + std::to_string(lineno) + ": mtllib without a file"; return RM_ERR_PARSE; }
            rm_status st = load_mtl(dir + w[1], mat_map, n_materials, err);
            if (st != RM_OK) return st;
        } else if (kw == "usemtl") {
            std::string mat_name = rest_after_keyword(line, "usemtl");
            if (mat_name.empty()) { err = path + ":" + std::to_string(lineno) + ": usemtl without a name"; return RM_ERR_PARSE; }
            auto it = mat_map.find(mat_name);
            bool new_have = it != mat_map.end();
            size_t new_id = new_have ? it->second : 0;
            bool changed = (new_have != have_mat) || (new_have && new_id != mat_id);
            if (changed && !faces.empty()) flush(name);
            have_mat = new_have;
            mat_id = new_id;
        }
        // vt, vn, s, comments glued to a token ("#f ..."), anything else: ignored
    }
    flush(name);   // the last model is emitted unconditionally
    return RM_OK;
}

}  // namespace

extern "C" {

rm_status rm_scene_load_obj(rm_scene *s, const char *path, rm_vec3 offset, uint32_t *n_models_out) {
    if (!s || !path) return fail(RM_ERR_INVALID_ARG, "rm_scene_load_obj: NULL argument");
    std::vector<ObjModel> models;
    std::string err;
    rm_status st = load_obj_models(path, models, err);
    if (st != RM_OK) return fail(st, err);
    for (const auto &m : models) {
        if (m.tri_xyz.empty())
            // obj.rs:88-92 indexes positions[0] of every model and update_bounding_box
            // indexes triangles[0]: a face-less model panics in the reference.
            return fail(RM_ERR_PARSE, "model '" + m.name + "' has no triangles (the reference panics, obj.rs:88)");
    }
    for (const auto &m : models) rm_scene_add_mesh(s, m.tri_xyz.data(), (uint32_t)(m.tri_xyz.size() / 9), offset);
    if (n_models_out) *n_models_out = (uint32_t)models.size();
    return RM_OK;
}

rm_status rm_scene_open_obj(const char *path, rm_scene **out) {
    if (!path || !out) return fail(RM_ERR_INVALID_ARG, "rm_scene_open_obj: NULL argument");
    rm_scene *s = new rm_scene();                                  // main.rs:274
    rm_status st = rm_scene_load_obj(s, path, v3(0., 0., -500.), nullptr);   // main.rs:278-286
    if (st == RM_ERR_IO && std::string(g_host_error).rfind("Could not load obj", 0) == 0) {
        // obj::load returned None (obj.rs:53-56): the scene simply has no shapes
        st = RM_OK;
    }
    if (st != RM_OK) { delete s; return st; }
    rm_scene_add_light(s, v3(0., 0., 0.), v3(1., 1., 1.), 1.);                 // main.rs:293-301
    rm_scene_add_light(s, v3(20., 20., 20.), v3(1., 0.5, 0.5), 0.8);           // main.rs:303-315
    *out = s;
    return RM_OK;
}

int rm_format_status(char *buf, size_t buflen, uint64_t ms, uint32_t w, uint32_t h) {
    // renderer.rs:111-121: fps = 1000 / ms (f64), printed `as u32` (saturating);
    // MP/s = fps * (H*W / 1e6) with two decimals; ms == 0 gives "inf".
    const double fps = 1000. / (double)ms;
    const double pix_scale = (double)((uint64_t)h * (uint64_t)w) / 1e6;
    uint32_t fps_u32 = 0;
    if (fps >= 4294967295.) fps_u32 = 4294967295u;
    else if (fps > 0.) fps_u32 = (uint32_t)fps;
    const double mps = fps * pix_scale;
    if (std::isnan(mps))
        return std::snprintf(buf, buflen, "Scene rendered in %llu ms (%u fps, NaN MP/s)", (unsigned long long)ms, fps_u32);
    if (std::isinf(mps))
        return std::snprintf(buf, buflen, "Scene rendered in %llu ms (%u fps, inf MP/s)", (unsigned long long)ms, fps_u32);
    return std::snprintf(buf, buflen, "Scene rendered in %llu ms (%u fps, %.2f MP/s)", (unsigned long long)ms, fps_u32, mps);
}

uint32_t rm_abi_version(void) { return RM_ABI_VERSION; }

}  // extern "C"
