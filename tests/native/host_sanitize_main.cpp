// AddressSanitizer / UBSan exercise of the host half of the C ABI (csrc/rm_scene.cpp):
// scene builder, OBJ ingest (good and malformed files), status formatting, and the
// hierarchy builder the upload uses (csrc/rm_bvh.hpp).  CPU only.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rusty_marcher_amd.h"
#include "rm_internal.h"
#include "rm_bvh.hpp"

// the device half (rm_device.hip) is not part of this CPU-only build: its one symbol the
// host half's callers need is the error accessor
extern "C" const char *rm_last_error(const rm_ctx *) { return rm_get_host_error(); }

// rm_build_bvh invariants the kernel's walk relies on: `order` is a permutation; every leaf
// holds 1..leaf_size consecutive primitives and every primitive sits in exactly one leaf;
// a child box contains the boxes of everything below it (with the builder's inflation).
static void walk_hierarchy(const rm_bvh &h, const std::vector<rm_aabb> &boxes, uint64_t ref, const double *box,
                           uint32_t leaf_size, std::vector<int> &seen, int depth) {
    assert(depth <= (int)h.depth && h.depth <= RM_BVH_MAX_DEPTH);     // the kernel's stack: 64 entries, one per level
    const uint32_t idx = (uint32_t)ref, cnt = (uint32_t)(ref >> 32);
    if (cnt) {                                                          // leaf
        assert(cnt <= leaf_size && (size_t)idx + cnt <= boxes.size());
        for (uint32_t k = 0; k < cnt; k++) {
            const rm_aabb &b = boxes[h.order[idx + k]];
            seen[idx + k]++;
            for (int a = 0; a < 3; a++) assert(box[a] <= b.lo[a] && b.hi[a] <= box[3 + a]);
        }
        return;
    }
    assert((size_t)(idx + 1) * RM_BVH_NODE_WORDS <= h.nodes.size());
    const double *n = &h.nodes[(size_t)idx * RM_BVH_NODE_WORDS];
    uint64_t l, r;
    std::memcpy(&l, &n[12], 8);
    std::memcpy(&r, &n[13], 8);
    for (int a = 0; a < 3; a++) {                                       // children inside the parent (root: no parent box)
        if (box) { assert(box[a] <= n[a] + 1e-6 && n[3 + a] <= box[3 + a] + 1e-6); assert(box[a] <= n[6 + a] + 1e-6 && n[9 + a] <= box[3 + a] + 1e-6); }
        assert(n[a] <= n[3 + a] && n[6 + a] <= n[9 + a]);
    }
    walk_hierarchy(h, boxes, l, n, leaf_size, seen, depth + 1);
    walk_hierarchy(h, boxes, r, n + 6, leaf_size, seen, depth + 1);
}

static void check_hierarchy(uint32_t n_prims, uint32_t leaf_size) {
    std::vector<rm_aabb> boxes(n_prims);
    uint64_t state = 0x9E3779B97F4A7C15ull * (n_prims + 1);
    auto rnd = [&]() { state = state * 6364136223846793005ull + 1442695040888963407ull; return (double)(state >> 11) / 9007199254740992.; };
    for (rm_aabb &b : boxes)
        for (int a = 0; a < 3; a++) {
            const double c = (n_prims % 2 ? 0. : 100. * rnd() - 50.), e = 2. * rnd();   // odd counts: all coincident centres
            b.lo[a] = c - e; b.hi[a] = c + e;
        }
    if (n_prims <= leaf_size) return;                                   // the upload never builds one over so few
    const rm_bvh h = rm_build_bvh(boxes, leaf_size);
    assert(h.order.size() == n_prims && h.nodes.size() % RM_BVH_NODE_WORDS == 0 && !h.nodes.empty());
    std::vector<int> perm(n_prims, 0), seen(n_prims, 0);
    for (uint32_t o : h.order) { assert(o < n_prims); perm[o]++; }
    for (int c : perm) assert(c == 1);
    walk_hierarchy(h, boxes, 0, nullptr, leaf_size, seen, 0);           // ref 0 = inner node 0, the root
    for (int c : seen) assert(c == 1);
}

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
    // Spheres whose radii double along a line: the surface-area split peels one off per level
    // (depth ~ count); the builder must switch to median splits and stay under the stack bound.
    for (uint32_t leaf : {1u, 2u, 4u}) {
        const uint32_t n = 400;
        std::vector<rm_aabb> boxes(n);
        double x = 0., r = 1e-30;
        for (uint32_t i = 0; i < n; i++) {
            x += 3. * r;
            for (int a = 0; a < 3; a++) { boxes[i].lo[a] = (a == 0 ? x : 0.) - r; boxes[i].hi[a] = (a == 0 ? x : 0.) + r; }
            r *= 2.;
        }
        const rm_bvh h = rm_build_bvh(boxes, leaf);
        assert(h.depth > RM_BVH_SAH_DEPTH && h.depth <= RM_BVH_SAH_DEPTH + 10u);
        std::vector<int> seen(n, 0);
        walk_hierarchy(h, boxes, 0, nullptr, leaf, seen, 0);
        for (int c : seen) assert(c == 1);
    }
    check_hierarchy(1, 4);
    check_hierarchy(5, 4);
    check_hierarchy(257, 4);
    check_hierarchy(1000, 2);
    check_hierarchy(64, 1);
    char buf[8];
    assert(rm_format_status(buf, sizeof buf, 92, 1280, 800) > 7);   // truncated, NUL-terminated
    assert(buf[7] == '\0');
    std::printf("host sanitize ok (%d of the malformed files were acceptable to the loader)\n", n_ok);
    return 0;
}
