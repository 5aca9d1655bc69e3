// rm_walk -- a camera walk through the default scene on N GPUs, one process per GPU,
// written against the C ABI alone (no HIP, no RCCL, no torch on the host side): what the
// Rust host's interactive loop (main.rs:75-78 offset_camera -> main.rs:329-333 render ->
// main.rs:337-346 to_vec into the pixbuf) becomes with rm_camera_update + rm_frame_submit.
//
//   rm_walk --rank R --world N --id-file PATH [--run-id TEXT] [--device D] [--frames K]
//           [--step dx,dy,dz] [--width W] [--height H] [--depth D] [--fov F] [--fast-fp] [--out PREFIX]
//
// Rank 0 creates the RCCL unique id and publishes it through PATH (written to PATH.tmp,
// then renamed); the other ranks wait for the file.  The file starts with the launcher's
// --run-id (every rank gets the same one) and a file whose run id differs -- one left behind
// by an earlier run -- is ignored, so a rank that starts before rank 0 cannot join with a
// stale id; rank 0 also removes whatever is at PATH before it publishes.  Every rank renders its cyclic share
// of every frame; rank 0, the consumer, receives the display bytes of the whole frame and
// (copied to page-locked host memory as part of the frame) and writes PREFIX_%04d.ppm (the
// bytes of fb.to_vec(), not normalised -- what the UI blits).
// --world 1 needs no file: the single rank still goes through RCCL unless --no-rccl.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rusty_marcher_amd.h"

static void die(rm_ctx *ctx, const char *what, rm_status st) {
    std::fprintf(stderr, "panic: %s: status %d: %s\n", what, (int)st, rm_last_error(ctx));
    std::exit(101);                                            // the reference's failure mode is a panic
}
#define CHECK(ctx, call) do { rm_status st__ = (call); if (st__ != RM_OK) die(ctx, #call, st__); } while (0)

int main(int argc, char **argv) {
    int rank = 0, world = 1, device = -1;
    unsigned frames = 8, width = 800, height = 600, depth = 3;
    double fov = 1.5;
    rm_vec3 step{0., 0., -0.5};
    bool fast = false, use_rccl = true;
    std::string id_file, out, run_id = "rm_walk";
    for (int i = 1; i < argc; i++) {
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--rank")) rank = std::atoi(next());
        else if (!std::strcmp(argv[i], "--world")) world = std::atoi(next());
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(next());
        else if (!std::strcmp(argv[i], "--id-file")) id_file = next();
        else if (!std::strcmp(argv[i], "--run-id")) run_id = next();
        else if (!std::strcmp(argv[i], "--frames")) frames = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--width")) width = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--height")) height = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--depth")) depth = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (!std::strcmp(argv[i], "--fov")) fov = std::atof(next());
        else if (!std::strcmp(argv[i], "--step")) { if (std::sscanf(next(), "%lf,%lf,%lf", &step.x, &step.y, &step.z) != 3) return 2; }
        else if (!std::strcmp(argv[i], "--fast-fp")) fast = true;
        else if (!std::strcmp(argv[i], "--no-rccl")) use_rccl = false;
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else { std::fprintf(stderr, "usage: rm_walk --rank R --world N --id-file PATH [--run-id TEXT] [--device D] [--frames K] [--step dx,dy,dz] [--width W] [--height H] [--depth D] [--fov F] [--fast-fp] [--no-rccl] [--out PREFIX]\n"); return 2; }
    }
    if (world < 1 || rank < 0 || rank >= world) { std::fprintf(stderr, "need 0 <= rank < world\n"); return 2; }
    if (world > 1 && id_file.empty()) { std::fprintf(stderr, "--world > 1 needs --id-file\n"); return 2; }
    if (device < 0) device = rank;                              // one GPU per rank

    // each frame slot has a stream of its own; HIP's default of 4 hardware queues would make
    // slots share one and serialise (profiles/r01_slots_cost.txt).  Read when HIP starts.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    rm_ctx *ctx = nullptr;
    CHECK(nullptr, rm_init(device, &ctx));
    rm_scene *scene = nullptr;
    CHECK(ctx, rm_scene_create_default(&scene));                // scene.rs:28-211
    rm_scene_desc desc;
    CHECK(ctx, rm_scene_get_desc(scene, &desc));
    CHECK(ctx, rm_scene_upload(ctx, &desc));

    // ---- communicator: the id goes through a file
    if (world > 1 || use_rccl) {
        unsigned char id[RM_COMM_ID_BYTES];
        char tag[64] = {};                                              // fixed-size run id in front of the unique id
        std::snprintf(tag, sizeof tag, "%s", run_id.c_str());
        if (rank == 0) {
            if (!id_file.empty()) std::remove(id_file.c_str());        // nothing of an earlier run survives rank 0's start
            CHECK(ctx, rm_comm_unique_id(id));
            if (!id_file.empty()) {
                const std::string tmp = id_file + ".tmp";
                FILE *f = std::fopen(tmp.c_str(), "wb");
                if (!f || std::fwrite(tag, 1, sizeof tag, f) != sizeof tag || std::fwrite(id, 1, sizeof id, f) != sizeof id) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
                std::fclose(f);
                if (std::rename(tmp.c_str(), id_file.c_str()) != 0) { std::fprintf(stderr, "cannot publish %s\n", id_file.c_str()); return 1; }
            }
        } else {
            bool got = false;
            for (int tries = 0; tries < 600 && !got; tries++) {                 // up to a minute
                if (FILE *f = std::fopen(id_file.c_str(), "rb")) {
                    char seen[sizeof tag];
                    got = std::fread(seen, 1, sizeof seen, f) == sizeof seen && !std::memcmp(seen, tag, sizeof tag) &&
                          std::fread(id, 1, sizeof id, f) == sizeof id;             // another run's file: keep waiting
                    std::fclose(f);
                }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            if (!got) { std::fprintf(stderr, "rank %d: no unique id in %s\n", rank, id_file.c_str()); return 1; }
        }
        CHECK(ctx, rm_comm_init(ctx, id, rank, world));
    }

    rm_params p;
    rm_create_renderer(fov, (double)height, (double)width, &p);  // renderer.rs:25-33
    p.max_depth = depth;
    if (fast) p.flags |= RM_FLAG_FAST_FP;
    uint32_t rows_per_rank = 0;
    size_t chunk = 0;
    CHECK(ctx, rm_exchange_layout(&p, world, &rows_per_rank, &chunk));
    const size_t n_patch_rows = height / 32u;
    const size_t display_bytes = n_patch_rows * 32u * width * 3u;
    const size_t frame_bytes = (size_t)width * height * 3u * sizeof(double);
    if (display_bytes == 0) { std::fprintf(stderr, "frame lower than one patch row\n"); return 2; }

    const uint32_t n_slots = RM_MAX_FRAME_SLOTS;
    std::vector<void *> d_rgb(n_slots), d_gather(n_slots), d_display(n_slots, nullptr);
    for (uint32_t s = 0; s < n_slots; s++) {
        CHECK(ctx, rm_buffer_alloc(ctx, frame_bytes, &d_rgb[s]));
        CHECK(ctx, rm_buffer_alloc(ctx, chunk * (size_t)world, &d_gather[s]));
        if (rank == 0) CHECK(ctx, rm_buffer_alloc(ctx, display_bytes, &d_display[s]));
    }
    std::vector<void *> host(n_slots, nullptr);                 // page-locked: the copy of a finished frame is asynchronous
    if (rank == 0)
        for (uint32_t s = 0; s < n_slots; s++) CHECK(ctx, rm_host_alloc(ctx, display_bytes, &host[s]));
    std::vector<int> frame_in_slot(n_slots, -1);

    auto consume = [&](uint32_t s) {                            // the consumer's side of a finished slot
        CHECK(ctx, rm_frame_wait(ctx, s));
        if (rank != 0 || frame_in_slot[s] < 0) return;
        if (!out.empty()) {
            char name[512];
            std::snprintf(name, sizeof name, "%s_%04d.ppm", out.c_str(), frame_in_slot[s]);
            FILE *f = std::fopen(name, "wb");
            if (!f) { std::fprintf(stderr, "cannot write %s\n", name); std::exit(1); }
            std::fprintf(f, "P6\n%u %u\n255\n", width, (unsigned)(n_patch_rows * 32u));   // framebuffer.rs:26-38 header
            std::fwrite(host[s], 1, display_bytes, f);
            std::fclose(f);
        }
    };

    const auto t0 = std::chrono::steady_clock::now();
    rm_vec3 cam = desc.camera;
    for (unsigned k = 0; k < frames; k++) {
        const uint32_t s = k % n_slots;
        if (frame_in_slot[s] >= 0) consume(s);                  // frame k - n_slots: display it before its buffers are reused
        CHECK(ctx, rm_camera_update(ctx, cam));                 // main.rs:75-78: the camera moves, the scene stays on the device
        CHECK(ctx, rm_frame_submit_to_host(ctx, &p, d_rgb[s], d_gather[s], d_display[s], host[s], s));
        frame_in_slot[s] = (int)k;
        cam.x += step.x; cam.y += step.y; cam.z += step.z;
    }
    for (unsigned k = frames; k < frames + n_slots; k++) {      // drain in frame order
        const uint32_t s = k % n_slots;
        if (frame_in_slot[s] >= 0) { consume(s); frame_in_slot[s] = -1; }
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rank == 0)
        std::printf("%u frames on %d GPU(s) in %.2f ms (%.1f frames/s, %.2f MP/s into host memory)\n", frames, world, ms,
                    frames * 1000. / ms, (double)width * height * frames / (ms * 1e3));
    for (uint32_t s = 0; s < n_slots; s++) {
        rm_buffer_free(ctx, d_rgb[s]);
        rm_buffer_free(ctx, d_gather[s]);
        rm_buffer_free(ctx, d_display[s]);
        rm_host_free(ctx, host[s]);
    }
    rm_scene_free(scene);
    rm_destroy(ctx);
    return 0;
}
