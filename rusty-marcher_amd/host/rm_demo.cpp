// rm_demo -- command-line stand-in for the reference's GTK window (engine/src/main.rs):
// "Default scene" / "Open .obj" -> render -> "Save to file" (normalize + write_ppm),
// written against the C++ host mirror (rusty_marcher.hpp).  Native harness for the C ABI.
//
//   rm_demo [--obj FILE] [--width W] [--height H] [--fov F] [--depth D] [--frames N]
//           [--camera x,y,z] [--fast-fp] [--no-normalize] [--device-frame] [--out FILE.ppm] [--dump-scene]
//
// Default: render() fills a FrameBuffer of rows of rows (framebuffer.rs:6-22) and the host
// normalises and quantises it, as save_to_file does (main.rs:353-357).  --device-frame: the f64
// frame stays on the device, render() brings back only fb.to_vec() (what the window blits) and the
// saved file comes from the device post-process.
//
// Defaults reproduce the reference's committed engine/out.ppm: 800x600, fov 1.5, depth 3.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rusty_marcher.hpp"

using namespace rusty_marcher;

static void dump_scene(const scene::Scene &sc) {
    bool owned = false;
    rm_scene *flat = sc.flatten(&owned);
    rm_scene_desc d;
    check(rm_scene_get_desc(flat, &d));
    std::printf("shapes %u spheres %u polygons %u polygon_vertices %u triangles %u lights %u camera %.17g %.17g %.17g\n",
                d.n_shapes, d.n_spheres, d.n_polygons, d.n_polygon_vertices, d.n_triangles, d.n_lights, d.camera.x,
                d.camera.y, d.camera.z);
    for (uint32_t i = 0; i < d.n_spheres; i++)
        std::printf("sphere %u c %.17g %.17g %.17g r2 %.17g glass %d exp %.17g\n", i, d.spheres[i].center.x,
                    d.spheres[i].center.y, d.spheres[i].center.z, d.spheres[i].radius_square,
                    d.spheres[i].reflectance.is_glass_like, d.spheres[i].reflectance.specular_exponent);
    for (uint32_t i = 0; i < d.n_polygons; i++)
        std::printf("polygon %u nv %u n %.17g %.17g %.17g p %.17g %.17g %.17g\n", i, d.polygons[i].n_vertices,
                    d.polygons[i].plane_normal.x, d.polygons[i].plane_normal.y, d.polygons[i].plane_normal.z,
                    d.polygons[i].plane_point.x, d.polygons[i].plane_point.y, d.polygons[i].plane_point.z);
    for (uint32_t i = 0; i < d.n_lights; i++)
        std::printf("light %u p %.17g %.17g %.17g c %.17g %.17g %.17g i %.17g\n", i, d.lights[i].position.x,
                    d.lights[i].position.y, d.lights[i].position.z, d.lights[i].color.x, d.lights[i].color.y,
                    d.lights[i].color.z, d.lights[i].intensity);
    if (owned) rm_scene_free(flat);
}

int main(int argc, char **argv) {
    std::string obj_path, out = "out.ppm";
    size_t width = 800, height = 600;
    double fov = 1.5;
    unsigned depth = 3, frames = 1;
    bool fast = false, normalize = true, dump = false, device_frame = false;
    Vec3f cam_off;
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--obj")) obj_path = next();
        else if (!std::strcmp(argv[i], "--width")) width = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--height")) height = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--fov")) fov = std::atof(next());
        else if (!std::strcmp(argv[i], "--depth")) depth = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--frames")) frames = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--camera")) { if (std::sscanf(next(), "%lf,%lf,%lf", &cam_off.x, &cam_off.y, &cam_off.z) != 3) return 2; }
        else if (!std::strcmp(argv[i], "--fast-fp")) fast = true;
        else if (!std::strcmp(argv[i], "--no-normalize")) normalize = false;
        else if (!std::strcmp(argv[i], "--device-frame")) device_frame = true;
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else if (!std::strcmp(argv[i], "--dump-scene")) dump = true;
        else { std::fprintf(stderr, "usage: rm_demo [--obj FILE] [--width W] [--height H] [--fov F] [--depth D] [--frames N] [--camera x,y,z] [--fast-fp] [--no-normalize] [--device-frame] [--out FILE.ppm] [--dump-scene]\n"); return 2; }
    }
    try {
        // main.rs:119-123 (default scene) / main.rs:261-327 (open .obj)
        scene::Scene sc = obj_path.empty() ? scene::Scene::create_default() : scene::Scene::open_obj(obj_path);
        sc.offset_camera(cam_off);                                            // main.rs:75-78
        if (dump) { dump_scene(sc); return 0; }

        framebuffer::FrameBuffer fb = framebuffer::create_frame_buffer(width, height);   // main.rs:240 uses 1600x1280
        renderer::Renderer r = renderer::create_renderer(fov, (double)fb.height, (double)fb.width);   // main.rs:367-368
        r.max_depth = depth;
        if (fast) r.flags |= RM_FLAG_FAST_FP;
        std::string msg;
        std::vector<uint8_t> display;
        for (unsigned f = 0; f < frames; f++)                                // main.rs:329-333: the whole Scene goes in every time
            msg = device_frame ? r.render_display(fb.width, fb.height, sc, display) : r.render(fb, sc);
        std::printf("kernel %.3f ms, device->host %.3f ms, call %.3f ms\n", r.last_timing.kernel_ms, r.last_timing.d2h_ms,
                    r.last_timing.total_ms);
        uint64_t up_calls = 0, up_copies = 0, bytes = 0, patches = 0, sent = 0;
        check(rm_scene_uploads(r.context(), &up_calls, &up_copies), r.context());
        std::printf("scene uploads: %llu calls, %llu copies to the device\n", (unsigned long long)up_calls,
                    (unsigned long long)up_copies);
        check(rm_hostio_stats(r.context(), &bytes, &patches, &sent, nullptr), r.context());
        std::printf("last frame: %llu bytes over the link, %llu of %llu patches sent\n", (unsigned long long)bytes,
                    (unsigned long long)sent, (unsigned long long)patches);
        // main.rs:353-357 save_to_file: fb.normalize(); fb.write_ppm("out.ppm")
        if (device_frame) {
            renderer::write_ppm(out, fb.width, fb.height, renderer::to_vec_on_device(r, fb.width, fb.height, normalize));
        } else {
            if (normalize) fb.normalize();
            fb.write_ppm(out);
        }
        std::printf("Saved rendered file %s\n", out.c_str());
    } catch (const Panic &p) {
        std::fprintf(stderr, "panic: %s (status %d)\n", p.what(), (int)p.status);
        return 101;                                                          // Rust's panic exit code
    }
    return 0;
}
