// AddressSanitizer / UBSan exercise of the host half of the C ABI (csrc/rm_scene.cpp):
// scene builder, OBJ ingest (good and malformed files), status formatting.  CPU only.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rusty_marcher_amd.h"
#include "rm_internal.h"

// the device half (rm_device.hip) is not part of this CPU-only build: its one symbol the
// host half's callers need is the error accessor
extern "C" const char *rm_last_error(const rm_ctx *) { return rm_get_host_error(); }

int main(int argc, char **argv) {
    assert(argc >= 3);
    const std::string cornell = argv[1], tmpdir = argv[2];
    rm_scene *s = nullptr;
    assert(rm_scene_create_default(&s) == RM_OK);
    rm_scene_desc d;
    assert(rm_scene_get_desc(s, &d) == RM_OK);
    assert(d.n_shapes == 6 && d.n_lights == 2 && d.n_polygon_vertices == 7);
    rm_vec3 off = {1., 2., 3.};
    assert(rm_scene_offset_shape(s, 5, off) == RM_OK);
    assert(rm_scene_offset_shape(s, 0, off) == RM_ERR_INVALID_ARG);
    assert(rm_scene_offset_shape(s, 77, off) == RM_ERR_INVALID_ARG);
    assert(rm_scene_offset_camera(s, off) == RM_OK);
    rm_reflectance r;
    rm_reflectance_default(&r);
    std::vector<rm_vec3> poly;
    for (int i = 0; i < 9; i++) poly.push_back(rm_vec3{(double)i, (double)(i * i % 5), -10.});
    assert(rm_scene_add_polygon(s, poly.data(), (uint32_t)poly.size(), &r) == RM_OK);
    assert(rm_scene_add_polygon(s, poly.data(), 2, &r) == RM_ERR_INVALID_ARG);
    std::vector<double> tri(9 * 100);
    for (size_t i = 0; i < tri.size(); i++) tri[i] = (double)((i * 7919) % 101) - 50.;
    assert(rm_scene_add_mesh(s, tri.data(), 100, off) == RM_OK);
    assert(rm_scene_add_mesh(s, nullptr, 0, off) == RM_OK);
    assert(rm_scene_get_desc(s, &d) == RM_OK);
    assert(d.n_triangles == 100 && d.n_shapes == 9);
    rm_scene_free(s);

    rm_scene *c = nullptr;
    assert(rm_scene_open_obj(cornell.c_str(), &c) == RM_OK);
    assert(rm_scene_get_desc(c, &d) == RM_OK);
    assert(d.n_shapes == 8 && d.n_triangles == 36 && d.n_lights == 2);
    rm_scene_free(c);

    // malformed inputs: every one must come back as a status, never as a crash
    const char *bad[] = {
        "f 1 2 3\n",                                   // face before any vertex
        "v 0 0\nf 1 1 1\n",                            // short vertex
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n",        // index out of range
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -4 -1 -2\n",     // negative index out of range
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf a b c\n",        // not a number
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/ 2/x 3\n",     // odd separators
        "usemtl\nv 0 0 0\n",                           // usemtl without a name
        "mtllib\n",                                    // mtllib without a file
        "o empty\n",                                   // no faces at all
        "v 1e999 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",    // overflowing float
        "",                                            // empty file
    };
    int n_ok = 0;
    for (size_t i = 0; i < sizeof bad / sizeof bad[0]; i++) {
        const std::string path = tmpdir + "/bad" + std::to_string(i) + ".obj";
        FILE *f = std::fopen(path.c_str(), "w");
        std::fputs(bad[i], f);
        std::fclose(f);
        rm_scene *b = nullptr;
        assert(rm_scene_new(&b) == RM_OK);
        uint32_t n = 0;
        const rm_status st = rm_scene_load_obj(b, path.c_str(), off, &n);
        if (st == RM_OK) n_ok++;
        else assert(std::strlen(rm_last_error(nullptr)) > 0);
        rm_scene_free(b);
    }
    char buf[8];
    assert(rm_format_status(buf, sizeof buf, 92, 1280, 800) > 7);   // truncated, NUL-terminated
    assert(buf[7] == '\0');
    std::printf("host sanitize ok (%d of the malformed files were acceptable to the loader)\n", n_ok);
    return 0;
}
