// rm_seam -- the drop-in call measured at the seam the reference really has.
//
// renderer.rs:36-126 `render(&mut FrameBuffer, &Scene)` fills framebuffer.rs:6-22
// `buffer: Vec<Vec<Vec3f>>`: one heap allocation per scan line.  This harness holds exactly
// that (rusty_marcher.hpp `FrameBuffer`: a std::vector per row) and times, per call, from
// compiled code over the C ABI:
//
//   flat           rm_render into one flat pageable array (what rounds 1-2 measured)
//   rows_of_rows   rm_render_rows into the per-row allocations (the reference's layout)
//   display_only   rm_render_display: f64 frame resident on the device, only fb.to_vec() comes
//                  back (what main.rs:337-346 blits), into pageable and into page-locked memory
//                  -- synchronous calls -- and as a render loop with frames in flight
//   fetch_rows     rm_fetch_rows of the resident frame (save_to_file's read-back, on demand)
//
// and checks that all of them hold the same frame bit for bit.  Prints ONE JSON line.
//
//   rm_seam [--scene demo|OBJFILE] [--width W] [--height H] [--depth D] [--frames N] [--fast-fp]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include "rusty_marcher.hpp"

using namespace rusty_marcher;
using clk = std::chrono::steady_clock;

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v.empty() ? 0. : v[v.size() / 2];
}

int main(int argc, char **argv) {
    std::string scene_arg = "demo";
    size_t width = 1920, height = 1080;
    unsigned depth = 5, frames = 20;
    bool fast = false;
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--scene")) scene_arg = next();
        else if (!std::strcmp(argv[i], "--width")) width = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--height")) height = std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--depth")) depth = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--frames")) frames = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--fast-fp")) fast = true;
        else { std::fprintf(stderr, "usage: rm_seam [--scene demo|OBJFILE] [--width W] [--height H] [--depth D] [--frames N] [--fast-fp]\n"); return 2; }
    }
    setenv("GPU_MAX_HW_QUEUES", "8", 0);     // frames in flight: a hardware queue per slot's stream (before HIP starts)
    try {
        scene::Scene sc = scene_arg == "demo" ? scene::Scene::create_default() : scene::Scene::open_obj(scene_arg);
        renderer::Renderer r = renderer::create_renderer(1.5, (double)height, (double)width);
        r.max_depth = depth;
        if (fast) r.flags |= RM_FLAG_FAST_FP;
        rm_ctx *ctx = r.context();
        // render() prints its status lines like the reference does: they go to /dev/null, the JSON
        // line to the real stdout
        std::fflush(stdout);
        FILE *json = fdopen(dup(1), "w");
        if (!json || !std::freopen("/dev/null", "w", stdout)) { std::fprintf(stderr, "cannot redirect stdout\n"); return 2; }

        // ---- rows of rows: the reference's FrameBuffer
        framebuffer::FrameBuffer fb = framebuffer::create_frame_buffer(width, height);
        std::vector<double> t_rows, t_rows_kernel;
        for (unsigned f = 0; f < frames + 3; f++) {
            const auto t0 = clk::now();
            r.render(fb, sc);
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            if (f >= 3) { t_rows.push_back(ms); t_rows_kernel.push_back(r.last_timing.kernel_ms); }
        }
        uint64_t bytes = 0, patches = 0, sent = 0;
        int threads = 0;
        check(rm_hostio_stats(ctx, &bytes, &patches, &sent, &threads), ctx);

        // ---- ... the way the reference calls it: a button offsets the camera by 5 on one axis, then render() (main.rs:74-78,
        // :119-170).  Twelve presses that lead back home, over and over: every call renders a view its predecessor did not.
        std::vector<double> t_move, t_move_kernel;
        {
            const Vec3f press[12] = {{5, 0, 0}, {0, 5, 0}, {0, 0, 5}, {-5, 0, 0}, {0, 0, 5}, {5, 0, 0}, {0, -5, 0}, {0, 0, -5}, {-5, 0, 0}, {0, 5, 0}, {0, 0, -5}, {0, -5, 0}};
            const unsigned n_move = ((frames + 11u) / 12u) * 12u;
            for (unsigned f = 0; f < 12u + n_move; f++) {
                sc.offset_camera(press[f % 12u]);
                const auto t0 = clk::now();
                r.render(fb, sc);
                const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
                if (f >= 12u) { t_move.push_back(ms); t_move_kernel.push_back(r.last_timing.kernel_ms); }
            }
            r.render(fb, sc);                                           // (home again: the frame the comparisons below expect)
        }

        // ---- flat pageable array through rm_render (same scene, same context)
        bool owned = false;
        rm_scene *flat_scene = sc.flatten(&owned);
        rm_scene_desc d;
        check(rm_scene_get_desc(flat_scene, &d));
        check(rm_scene_upload(ctx, &d), ctx);
        if (owned) rm_scene_free(flat_scene);
        rm_params p;
        rm_create_renderer(1.5, (double)height, (double)width, &p);
        p.max_depth = depth;
        p.flags = r.flags;
        std::vector<double> flat(width * height * 3, 0.);
        std::vector<double> t_flat;
        rm_timing tm{};
        for (unsigned f = 0; f < frames + 3; f++) {
            const auto t0 = clk::now();
            check(rm_render(ctx, &p, flat.data(), &tm), ctx);
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            if (f >= 3) t_flat.push_back(ms);
        }
        bool rows_equal_flat = true;
        for (size_t y = 0; y < height && rows_equal_flat; y++)
            rows_equal_flat = std::memcmp(fb.buffer[y].data(), flat.data() + y * width * 3, width * 3 * sizeof(double)) == 0;

        // ---- display only, pageable destination
        std::vector<uint8_t> rgb8;
        std::vector<double> t_disp, t_disp_kernel;
        for (unsigned f = 0; f < frames + 3; f++) {
            const auto t0 = clk::now();
            r.render_display(width, height, sc, rgb8);
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            if (f >= 3) { t_disp.push_back(ms); t_disp_kernel.push_back(r.last_timing.kernel_ms); }
        }
        const std::vector<uint8_t> want8 = fb.to_vec();
        const size_t rendered = (height / 32) * 32 * width * 3;
        const bool display_equal = std::memcmp(rgb8.data(), want8.data(), rendered) == 0;

        // ---- display only, page-locked destination (rm_host_alloc)
        void *pinned = nullptr;
        check(rm_host_alloc(ctx, width * height * 3, &pinned), ctx);
        std::memset(pinned, 0, width * height * 3);
        std::vector<double> t_pin;
        for (unsigned f = 0; f < frames + 3; f++) {
            const auto t0 = clk::now();
            check(rm_render_display(ctx, &p, (uint8_t *)pinned, &tm), ctx);
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            if (f >= 3) t_pin.push_back(ms);
        }
        const bool pinned_equal = std::memcmp(pinned, want8.data(), rendered) == 0;
        rm_host_free(ctx, pinned);

        // ---- display only, a render LOOP: frames in flight (rm_frame_submit_to_host, four slots, each on a
        // stream of its own): frame k+1..k+3 render while the bytes of frame k cross the link; the camera
        // moves between frames (rm_camera_update, the scene stays resident).  What a loop gets that need
        // not have frame k in hand before it starts k+1 -- rm_walk is this loop with files behind it.
        double pipelined_fps = 0.;
        bool pipelined_equal = false;
        {
            const unsigned n_slots = RM_MAX_FRAME_SLOTS, n_loop = std::max(60u, frames * 6u);
            const size_t rows32 = (height / 32) * 32, bytes8 = rows32 * width * 3;
            void *d_rgb[RM_MAX_FRAME_SLOTS] = {}, *d_gather[RM_MAX_FRAME_SLOTS] = {}, *d_disp[RM_MAX_FRAME_SLOTS] = {}, *h_disp[RM_MAX_FRAME_SLOTS] = {};
            for (unsigned k = 0; k < n_slots; k++) {
                check(rm_buffer_alloc(ctx, width * height * 3 * sizeof(double), &d_rgb[k]), ctx);
                check(rm_buffer_alloc(ctx, bytes8, &d_gather[k]), ctx);
                check(rm_buffer_alloc(ctx, bytes8, &d_disp[k]), ctx);
                check(rm_host_alloc(ctx, bytes8, &h_disp[k]), ctx);
            }
            const rm_vec3 cam0 = sc.camera.c();
            auto t0 = clk::now();
            for (unsigned f = 0; f < n_slots * 2 + n_loop; f++) {
                if (f == n_slots * 2) {                                  // warmed up: drain, start the clock
                    for (unsigned k = 0; k < n_slots; k++) check(rm_frame_wait(ctx, k), ctx);
                    t0 = clk::now();
                }
                const unsigned k = f % n_slots;
                check(rm_frame_wait(ctx, k), ctx);                         // the slot's previous frame is in h_disp[k]: consumed here
                check(rm_camera_update(ctx, rm_vec3{cam0.x + 0.01 * (f % 7), cam0.y, cam0.z}), ctx);
                check(rm_frame_submit_to_host(ctx, &p, d_rgb[k], d_gather[k], d_disp[k], h_disp[k], k), ctx);
            }
            for (unsigned k = 0; k < n_slots; k++) check(rm_frame_wait(ctx, k), ctx);
            pipelined_fps = n_loop / std::chrono::duration<double>(clk::now() - t0).count();
            // one more frame at the original camera: the same bytes as the synchronous call's
            check(rm_camera_update(ctx, cam0), ctx);
            check(rm_frame_submit_to_host(ctx, &p, d_rgb[0], d_gather[0], d_disp[0], h_disp[0], 0), ctx);
            check(rm_frame_wait(ctx, 0), ctx);
            pipelined_equal = std::memcmp(h_disp[0], want8.data(), bytes8) == 0;
            for (unsigned k = 0; k < n_slots; k++) {
                rm_buffer_free(ctx, d_rgb[k]); rm_buffer_free(ctx, d_gather[k]); rm_buffer_free(ctx, d_disp[k]); rm_host_free(ctx, h_disp[k]);
            }
        }

        // ---- the f64 rows of the resident frame, on demand
        framebuffer::FrameBuffer fb2 = framebuffer::create_frame_buffer(width, height);
        std::vector<double> t_fetch;
        for (unsigned f = 0; f < 5; f++) {
            const auto t0 = clk::now();
            r.fetch(fb2);
            t_fetch.push_back(std::chrono::duration<double, std::milli>(clk::now() - t0).count());
        }
        bool fetch_equal = true;
        for (size_t y = 0; y < height && fetch_equal; y++)
            fetch_equal = std::memcmp(fb.buffer[y].data(), fb2.buffer[y].data(), width * 3 * sizeof(double)) == 0;

        const double px = (double)width * (double)height;
        const double m_rows = median(t_rows), m_flat = median(t_flat), m_disp = median(t_disp), m_pin = median(t_pin);
        std::fprintf(json,
                     "{\"scene\": \"%s\", \"width\": %zu, \"height\": %zu, \"depth\": %u, \"frames\": %u, \"host_threads\": %d, "
                     "\"rows_of_rows\": {\"ms_per_call\": %.4f, \"mpx_per_s\": %.1f, \"kernel_ms\": %.4f, \"bytes_over_the_link\": %llu, "
                     "\"patches\": %llu, \"patches_sent\": %llu, \"identical_to_flat\": %s, "
                     "\"what\": \"Renderer::render into a FrameBuffer of per-row heap allocations (rm_render_rows), median\", "
                     "\"camera_on_the_move\": {\"ms_per_call\": %.4f, \"kernel_ms\": %.4f, \"what\": \"scene.offset_camera(one press: 5 on one axis) before every render(), median\"}}, "
                     "\"flat\": {\"ms_per_call\": %.4f, \"mpx_per_s\": %.1f, \"what\": \"rm_render into one flat pageable array, median\"}, "
                     "\"display_only\": {\"ms_per_call\": %.4f, \"frames_per_s\": %.1f, \"kernel_ms\": %.4f, \"identical_to_to_vec_of_rows\": %s, "
                     "\"into_page_locked\": {\"ms_per_call\": %.4f, \"frames_per_s\": %.1f, \"identical\": %s}, "
                     "\"render_loop_frames_in_flight\": {\"frames_per_s\": %.1f, \"frames_in_flight\": %u, \"identical\": %s, "
                     "\"what\": \"rm_frame_submit_to_host, camera moving between frames, display bytes into page-locked memory\"}, "
                     "\"what\": \"rm_render_display: f64 frame stays on the device, fb.to_vec() into host memory, synchronous, median\"}, "
                     "\"fetch_rows\": {\"ms_per_call\": %.4f, \"identical\": %s}}\n",
                     scene_arg.c_str(), width, height, depth, frames, threads, m_rows, px / m_rows / 1e3, median(t_rows_kernel),
                     (unsigned long long)bytes, (unsigned long long)patches, (unsigned long long)sent, rows_equal_flat ? "true" : "false",
                     median(t_move), median(t_move_kernel), m_flat, px / m_flat / 1e3, m_disp, 1e3 / m_disp, median(t_disp_kernel), display_equal ? "true" : "false", m_pin,
                     1e3 / m_pin, pinned_equal ? "true" : "false", pipelined_fps, (unsigned)RM_MAX_FRAME_SLOTS, pipelined_equal ? "true" : "false",
                     median(t_fetch), fetch_equal ? "true" : "false");
        std::fclose(json);
        return (rows_equal_flat && display_equal && pinned_equal && pipelined_equal && fetch_equal) ? 0 : 3;
    } catch (const Panic &p) {
        std::fprintf(stderr, "panic: %s (status %d)\n", p.what(), (int)p.status);
        return 101;
    }
}
