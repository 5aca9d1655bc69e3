// rusty_marcher.hpp -- C++ host-side mirror of the reference's module surface
// (engine/src/{geometry,shapes,sphere,polygon,lights,obj,scene,framebuffer,renderer}.rs)
// over the C ABI of include/rusty_marcher_amd.h.  Header-only; link with
// librusty_marcher_amd.so.  Same names, argument order and failure behaviour as the
// Rust code: where the reference panics this throws rusty_marcher::Panic.
//
// The reference's toolchain (rustc/cargo) is absent from the build image, so this is the
// compiled-language host layer; the Rust shim a maintainer adds is in INTEGRATION.md.
#ifndef RUSTY_MARCHER_HPP
#define RUSTY_MARCHER_HPP

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rusty_marcher_amd.h"

namespace rusty_marcher {

struct Panic : std::runtime_error {
    rm_status status;
    Panic(rm_status st, const std::string &what) : std::runtime_error(what), status(st) {}
};

inline void check(rm_status st, const rm_ctx *ctx = nullptr) {
    if (st != RM_OK) throw Panic(st, rm_last_error(ctx));
}

// ---- geometry.rs ----------------------------------------------------------------
namespace geometry {
struct Vec3f {
    double x = 0., y = 0., z = 0.;
    static Vec3f zero() { return {0., 0., 0.}; }
    static Vec3f ones() { return {1., 1., 1.}; }
    Vec3f operator+(Vec3f o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3f operator-(Vec3f o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3f scaled(double s) const { return {x * s, y * s, z * s}; }
    rm_vec3 c() const { return rm_vec3{x, y, z}; }
};
}  // namespace geometry
using geometry::Vec3f;

// ---- shapes.rs --------------------------------------------------------------------
namespace shapes {
struct Reflectance {
    double diffusion, specular, specular_exponent, reflection, refractive_index;
    Vec3f diffuse_color;
    bool is_glass_like;
    static Reflectance create_default() {             // shapes.rs:49-61
        rm_reflectance r;
        rm_reflectance_default(&r);
        return from_c(r);
    }
    static Reflectance from_c(const rm_reflectance &r) {
        return {r.diffusion, r.specular, r.specular_exponent, r.reflection, r.refractive_index,
                Vec3f{r.diffuse_color.x, r.diffuse_color.y, r.diffuse_color.z}, r.is_glass_like != 0};
    }
    rm_reflectance c() const {
        return rm_reflectance{diffusion, diffuse_color.c(), specular, specular_exponent, is_glass_like ? 1 : 0, 0,
                              reflection, refractive_index};
    }
};

// `trait Shape` (shapes.rs:40-47): what a shape must be able to do here is add itself to
// the flat scene, in list order.
struct Shape {
    virtual ~Shape() = default;
    virtual void describe(rm_scene *scene, uint32_t index) const = 0;
};
}  // namespace shapes
using shapes::Reflectance;

// ---- sphere.rs ----------------------------------------------------------------------
namespace sphere {
struct Sphere : shapes::Shape {
    Vec3f center;
    double radius;
    Reflectance reflectance;
    Sphere(Vec3f c, double r, Reflectance m) : center(c), radius(r), reflectance(m) {}
    void describe(rm_scene *scene, uint32_t) const override {
        const rm_reflectance r = reflectance.c();
        check(rm_scene_add_sphere(scene, center.c(), radius, &r));
    }
};
inline std::unique_ptr<Sphere> create(Vec3f center, double radius, Reflectance r) {   // sphere.rs:13-24
    return std::unique_ptr<Sphere>(new Sphere(center, radius, r));
}
}  // namespace sphere

// ---- polygon.rs ---------------------------------------------------------------------
namespace polygon {
struct ConvexPolygon : shapes::Shape {
    std::vector<Vec3f> vertices;
    Reflectance reflectance;
    std::vector<Vec3f> offsets;
    static std::unique_ptr<ConvexPolygon> create(std::vector<Vec3f> v, Reflectance r) {   // polygon.rs:16-42
        if (v.size() <= 2) throw Panic(RM_ERR_INVALID_ARG, "assertion failed: vertices.len() > 2");
        std::unique_ptr<ConvexPolygon> p(new ConvexPolygon());
        p->vertices = std::move(v);
        p->reflectance = r;
        return p;
    }
    void offset(Vec3f off) { offsets.push_back(off); }                                   // polygon.rs:44-49
    void describe(rm_scene *scene, uint32_t index) const override {
        std::vector<rm_vec3> v;
        for (const auto &p : vertices) v.push_back(p.c());
        const rm_reflectance r = reflectance.c();
        check(rm_scene_add_polygon(scene, v.data(), (uint32_t)v.size(), &r));
        for (const auto &o : offsets) check(rm_scene_offset_shape(scene, index, o.c()));
    }

  private:
    ConvexPolygon() : reflectance(Reflectance::create_default()) {}
};
}  // namespace polygon

// ---- lights.rs ------------------------------------------------------------------------
namespace lights {
struct Light {
    Vec3f position, color;
    double intensity;
};
inline Light create_light(Vec3f position, Vec3f color, double intensity) {   // lights.rs:10-16
    return Light{position, color, intensity};                               // L-inf normalised inside the library
}
}  // namespace lights

// ---- obj.rs ---------------------------------------------------------------------------
namespace obj {
struct Obj : shapes::Shape {
    std::vector<double> tri_xyz;   // 9 per triangle, f32 positions widened (obj.rs:103-105)
    std::vector<Vec3f> offsets;
    void offset(Vec3f off) { offsets.push_back(off); }                                    // obj.rs:24-29
    void describe(rm_scene *scene, uint32_t index) const override {
        check(rm_scene_add_mesh(scene, tri_xyz.data(), (uint32_t)(tri_xyz.size() / 9), rm_vec3{0., 0., 0.}));
        for (const auto &o : offsets) check(rm_scene_offset_shape(scene, index, o.c()));
    }
};

// obj.rs:44-151: `Some(Vec<Obj>)`; `ok == false` when the file cannot be read (obj.rs:53-56).
inline std::vector<std::unique_ptr<Obj>> load(const std::string &path, bool *ok = nullptr) {
    std::vector<std::unique_ptr<Obj>> out;
    rm_scene *s = nullptr;
    check(rm_scene_new(&s));
    uint32_t n = 0;
    const rm_status st = rm_scene_load_obj(s, path.c_str(), rm_vec3{0., 0., 0.}, &n);
    if (st == RM_ERR_IO && std::string(rm_last_error(nullptr)).rfind("Could not load obj", 0) == 0) {
        std::printf("Could not load obj from %s\n", path.c_str());
        rm_scene_free(s);
        if (ok) *ok = false;
        return out;
    }
    if (st != RM_OK) { rm_scene_free(s); check(st); }
    rm_scene_desc d;
    check(rm_scene_get_desc(s, &d));
    for (uint32_t i = 0; i < d.n_shapes; i++) {
        std::unique_ptr<Obj> o(new Obj());
        for (uint32_t t = 0; t < d.shapes[i].count; t++)
            for (const rm_vec3 &v : d.triangles[d.shapes[i].first + t].vertices) {
                o->tri_xyz.push_back(v.x); o->tri_xyz.push_back(v.y); o->tri_xyz.push_back(v.z);
            }
        out.push_back(std::move(o));
    }
    rm_scene_free(s);
    if (ok) *ok = true;
    return out;
}
}  // namespace obj

// ---- scene.rs -------------------------------------------------------------------------
namespace scene {
struct Scene {
    std::vector<lights::Light> lights;
    std::vector<std::unique_ptr<shapes::Shape>> shapes;   // Vec<Box<dyn Shape + Sync>>
    Vec3f camera;

    static Scene make() { return Scene(); }                                               // Scene::new, scene.rs:16-23
    void offset_camera(Vec3f off) { camera = camera + off; }                              // scene.rs:25-27

    static Scene create_default() {                                                       // scene.rs:28-211
        Scene s;
        check(rm_scene_create_default(&s.prebuilt_));
        s.adopt();
        return s;
    }
    static Scene open_obj(const std::string &path) {                                      // main.rs:261-327
        Scene s;
        check(rm_scene_open_obj(path.c_str(), &s.prebuilt_));
        s.adopt();
        return s;
    }

    Scene() = default;
    Scene(Scene &&o) noexcept { *this = std::move(o); }
    Scene &operator=(Scene &&o) noexcept {
        lights = std::move(o.lights); shapes = std::move(o.shapes); camera = o.camera;
        std::swap(prebuilt_, o.prebuilt_);
        return *this;
    }
    ~Scene() { rm_scene_free(prebuilt_); }

    // Flat form for upload; the caller frees it unless it is the library-built scene.
    rm_scene *flatten(bool *owned) const {
        if (prebuilt_ && shapes.empty()) {                  // library-built, shapes untouched
            check(rm_scene_set_camera(prebuilt_, camera.c()));
            *owned = false;
            return prebuilt_;
        }
        rm_scene *s = nullptr;
        check(rm_scene_new(&s));
        uint32_t i = 0;
        for (const auto &sh : shapes) sh->describe(s, i++);
        for (const auto &l : lights) check(rm_scene_add_light(s, l.position.c(), l.color.c(), l.intensity));
        check(rm_scene_set_camera(s, camera.c()));
        *owned = true;
        return s;
    }

  private:
    rm_scene *prebuilt_ = nullptr;
    void adopt() {
        rm_scene_desc d;
        check(rm_scene_get_desc(prebuilt_, &d));
        camera = Vec3f{d.camera.x, d.camera.y, d.camera.z};
        for (uint32_t i = 0; i < d.n_lights; i++)
            lights.push_back({Vec3f{d.lights[i].position.x, d.lights[i].position.y, d.lights[i].position.z},
                              Vec3f{d.lights[i].color.x, d.lights[i].color.y, d.lights[i].color.z}, d.lights[i].intensity});
    }
};
}  // namespace scene

// ---- framebuffer.rs ---------------------------------------------------------------------
namespace framebuffer {
// framebuffer.rs:6-10: `buffer: Vec<Vec<geometry::Vec3f>>` -- one heap allocation per scan line.
// Vec3f is three doubles, x y z in that order (the `#[repr(C)]` the shim asks of geometry.rs:4-8).
static_assert(sizeof(Vec3f) == 3 * sizeof(double), "Vec3f must be three packed doubles");
struct FrameBuffer {
    size_t width, height;
    std::vector<std::vector<Vec3f>> buffer;   // [height] rows of [width] pixels
    // what rm_render_rows / rm_fetch_rows take: one pointer per scan line
    // (width, height and buffer are public, as in the reference: a frame whose fields disagree is refused here -- the
    // library writes width * 3 doubles into each of height rows)
    std::vector<double *> row_pointers() {
        if (buffer.size() < height) throw std::runtime_error("FrameBuffer: fewer rows than its height");
        std::vector<double *> rows(height);
        for (size_t y = 0; y < height; y++) {
            if (buffer[y].size() != width) throw std::runtime_error("FrameBuffer: a row that is not `width` pixels long");
            rows[y] = reinterpret_cast<double *>(buffer[y].data());
        }
        return rows;
    }
    static uint8_t quantize(double f) {                                   // framebuffer.rs:80-82
        const double c = f > 0. ? (f < 1. ? f : 1.) : 0.;                 // f.max(0.).min(1.): NaN -> 0
        return (uint8_t)(255. * c);
    }
    std::vector<uint8_t> to_vec() const {                                 // framebuffer.rs:40-55
        std::vector<uint8_t> out(width * height * 3);
        size_t k = 0;
        for (size_t i = 0; i < height; i++)
            for (size_t j = 0; j < width; j++) {
                out[k++] = quantize(buffer[i][j].x); out[k++] = quantize(buffer[i][j].y); out[k++] = quantize(buffer[i][j].z);
            }
        return out;
    }
    void normalize() {                                                    // framebuffer.rs:58-77
        double mx = 0., my = 0., mz = 0.;
        auto fmax = [](double a, double b) { return b > a ? b : a; };     // f64::max: NaN loses
        for (const auto &row : buffer)
            for (const Vec3f &p : row) { mx = fmax(mx, p.x); my = fmax(my, p.y); mz = fmax(mz, p.z); }
        const double max_val = fmax(fmax(mx, my), mz);
        if (max_val > 0.) {
            const double s = 1. / max_val;
            for (auto &row : buffer)
                for (Vec3f &p : row) p = p.scaled(s);
        }
    }
    void write_ppm(const std::string &filename) const {                   // framebuffer.rs:26-38
        std::ofstream f(filename, std::ios::binary);
        f << "P6\n" << width << " " << height << "\n255\n";
        const std::vector<uint8_t> rgb = to_vec();
        f.write(reinterpret_cast<const char *>(rgb.data()), (std::streamsize)rgb.size());
    }
};
inline FrameBuffer create_frame_buffer(size_t width, size_t height) {     // framebuffer.rs:12-22
    return FrameBuffer{width, height, std::vector<std::vector<Vec3f>>(height, std::vector<Vec3f>(width, Vec3f::zero()))};
}
}  // namespace framebuffer

// ---- renderer.rs --------------------------------------------------------------------------
namespace renderer {
struct Renderer {
    double fov, half_fov, height, width, ratio;
    uint32_t max_depth = 3;        // renderer.rs:262, exposed
    uint32_t flags = RM_FLAG_NONE;
    rm_timing last_timing{};

    ~Renderer() { rm_destroy(ctx_); }
    Renderer(const Renderer &) = delete;
    Renderer(Renderer &&o) noexcept : fov(o.fov), half_fov(o.half_fov), height(o.height), width(o.width), ratio(o.ratio),
                                      max_depth(o.max_depth), flags(o.flags), ctx_(o.ctx_) { o.ctx_ = nullptr; }
    Renderer(double fov_, double height_, double width_) {
        rm_params p;
        rm_create_renderer(fov_, height_, width_, &p);
        fov = p.fov; half_fov = p.half_fov; height = p.height; width = p.width; ratio = p.ratio;
    }

    // renderer.rs:36-126: synchronous, fills `frame` -- rows of rows, like the reference's -- and
    // returns the status string
    std::string render(framebuffer::FrameBuffer &frame, const scene::Scene &sc) {
        const auto t0 = std::chrono::steady_clock::now();
        const rm_params p = prepare(frame.width, frame.height, sc);
        std::vector<double *> rows = frame.row_pointers();
        check(rm_render_rows(ctx_, &p, rows.data(), &last_timing), ctx_);
        return status(t0, frame.width, frame.height);
    }
    // The same call with a device-resident FrameBuffer: the f64 frame stays on the device, what
    // comes back is `fb.to_vec()` -- the bytes the window blits (main.rs:337-346) -- into `rgb8`
    // (width * height * 3; rows below the last whole patch row keep their contents).
    std::string render_display(size_t width, size_t height, const scene::Scene &sc, std::vector<uint8_t> &rgb8) {
        const auto t0 = std::chrono::steady_clock::now();
        const rm_params p = prepare(width, height, sc);
        rgb8.resize(width * height * 3);
        check(rm_render_display(ctx_, &p, rgb8.data(), &last_timing), ctx_);
        return status(t0, width, height);
    }
    // f64 rows of the device-resident frame on demand (save_to_file: normalize + write_ppm)
    void fetch(framebuffer::FrameBuffer &frame) {
        std::vector<double *> rows = frame.row_pointers();
        // (refused with RM_ERR_INVALID_ARG when the frame on the device is of another size: a window resized since)
        check(rm_fetch_rows(context(), rows.data(), (uint32_t)frame.width, (uint32_t)frame.height, 0, 0), ctx_);
    }
    rm_ctx *context() { if (!ctx_) check(rm_init(0, &ctx_)); return ctx_; }

  private:
    rm_ctx *ctx_ = nullptr;
    rm_params prepare(size_t width_px, size_t height_px, const scene::Scene &sc) {
        if (!ctx_) check(rm_init(0, &ctx_));
        if (height_px % 32 != 0 || width_px % 32 != 0) std::printf("Dimensions mismatch\n");
        std::printf("Rendering using patches of size %d, using %zu patches overall\n", 32, (height_px / 32) * (width_px / 32));
        bool owned = false;
        rm_scene *flat = sc.flatten(&owned);
        rm_scene_desc d;
        check(rm_scene_get_desc(flat, &d));
        const rm_status up = rm_scene_upload(ctx_, &d);
        if (owned) rm_scene_free(flat);
        check(up, ctx_);
        rm_params p;
        rm_create_renderer(fov, height, width, &p);
        p.frame_width = (uint32_t)width_px;
        p.frame_height = (uint32_t)height_px;
        p.max_depth = max_depth;
        p.flags = flags;
        return p;
    }
    static std::string status(std::chrono::steady_clock::time_point t0, size_t width_px, size_t height_px) {
        const uint64_t ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(
                                std::chrono::steady_clock::now() - t0).count();
        char buf[256];
        rm_format_status(buf, sizeof buf, ms, (uint32_t)width_px, (uint32_t)height_px);
        std::printf("%s\n", buf);
        return buf;
    }
};
inline Renderer create_renderer(double fov, double height, double width) {   // renderer.rs:25-33: (fov, height, width)
    return Renderer(fov, height, width);
}

// FrameBuffer::normalize + to_vec (framebuffer.rs:40-77) run on the frame the renderer's context
// holds on the device (rm_postprocess): what save_to_file needs when the f64 frame was never
// fetched.  `frame` gives the size.
inline std::vector<uint8_t> to_vec_on_device(Renderer &r, size_t width, size_t height, bool normalize) {
    std::vector<uint8_t> out(width * height * 3);
    check(rm_postprocess(r.context(), nullptr, (uint32_t)width, (uint32_t)height, normalize ? 1 : 0, out.data(), nullptr),
          r.context());
    return out;
}
inline void write_ppm(const std::string &filename, size_t width, size_t height, const std::vector<uint8_t> &rgb) {
    std::ofstream f(filename, std::ios::binary);
    f << "P6\n" << width << " " << height << "\n255\n";
    f.write(reinterpret_cast<const char *>(rgb.data()), (std::streamsize)rgb.size());
}
}  // namespace renderer

}  // namespace rusty_marcher
#endif
