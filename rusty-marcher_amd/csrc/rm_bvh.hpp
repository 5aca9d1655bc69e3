// rm_bvh.hpp -- host-side builder of the bounding-volume hierarchy the kernel walks for
// scenes with many spheres / mesh triangles (the reference computes a
// BoundingBox per shape, shapes.rs:34-86, and never consults it).
//
// The hierarchy only decides WHICH primitives a wave tests; the tests themselves, the
// closest-hit ordering and the list-order tie-break are unchanged, so results are
// identical to the brute-force walk.  Boxes are inflated by a relative margin so that
// rounding in the kernel's slab test can never cull a primitive the exact test would hit.
//
// Node (16 words): left child box min xyz, max xyz | right child box min xyz, max xyz |
// left ref | right ref | 2 pad.  A ref is a u64: low 32 bits = node index (inner) or
// first primitive (leaf), high 32 bits = 0 (inner) or the leaf's primitive count.
#ifndef RM_BVH_HPP
#define RM_BVH_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#ifndef RM_BVH4
#define RM_BVH4 0
#endif
#if RM_BVH4
// Four-wide nodes (32 words, one 256-byte fetch): the binary tree with every other level folded away.
//   [6 c .. 6 c + 5]  child c's box, min xyz | max xyz      (c = 0..3: left-left, left-right, right-left, right-right)
//   [24 .. 27]        child c's ref as two u32: index (inner: node, leaf: first primitive), count (0: inner; ~0: no such child)
//   [28]              u32: split axes -- bits 0-1 the node's own, 2-3 its left half's, 4-5 its right half's (front-to-back order)
#define RM_BVH_NODE_WORDS 32u
#else
#define RM_BVH_NODE_WORDS 16u
#endif

struct rm_aabb {
    double lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<double>::infinity(); hi[a] = -lo[a]; }
    }
    void grow(const rm_aabb &o) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); }
    }
    double centre(int a) const { return 0.5 * (lo[a] + hi[a]); }
};

struct rm_bvh {
    std::vector<double> nodes;        // RM_BVH_NODE_WORDS per node, node 0 is the root
    std::vector<uint32_t> order;      // order[k] = index (into the input boxes) of the k-th primitive after reordering
    uint32_t depth = 0;               // inner nodes on the longest root-to-leaf path = entries the kernel's walk can park
};

// The kernel's walk parks one sibling per level on a 64-entry stack per wave.  The surface-area
// split alone has no depth bound (spheres whose radii double along a line give one level per
// sphere), so below RM_BVH_SAH_DEPTH levels the builder splits at the median: from there the
// depth is at most log2(count), and the whole tree stays under RM_BVH_MAX_DEPTH for any
// primitive count a 32-bit index can address.  rm_scene_upload checks `depth` all the same.
#define RM_BVH_SAH_DEPTH 24u
#define RM_BVH_MAX_DEPTH 60u

namespace rm_bvh_detail {

inline void inflate(rm_aabb &b) {
    for (int a = 0; a < 3; a++) {
        const double m = 1e-7 * (std::fabs(b.lo[a]) + std::fabs(b.hi[a]) + (b.hi[a] - b.lo[a])) + 1e-9;
        b.lo[a] -= m;
        b.hi[a] += m;
    }
}

struct Builder {
    const std::vector<rm_aabb> &boxes;
    uint32_t leaf_size;
    rm_bvh &out;
    std::vector<int> axis_of;         // per binary inner node: the axis its halves were sorted along

    // Builds the subtree over order[first, first+count) whose root sits `depth` levels below
    // the hierarchy's root; returns its ref and box.
    uint64_t build(uint32_t first, uint32_t count, rm_aabb &box, uint32_t depth) {
        box.reset();
        for (uint32_t k = 0; k < count; k++) box.grow(boxes[out.order[first + k]]);
        if (count <= leaf_size) return ((uint64_t)count << 32) | first;           // leaf
        // surface-area heuristic: for each axis sort by centroid and sweep; cost(split) =
        // area(left) * n_left + area(right) * n_right
        auto area = [](const rm_aabb &x) {
            const double dx = x.hi[0] - x.lo[0], dy = x.hi[1] - x.lo[1], dz = x.hi[2] - x.lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        uint32_t half = count / 2;
        int axis = 0;
        double best_cost = std::numeric_limits<double>::infinity();
        std::vector<uint32_t> tmp(out.order.begin() + first, out.order.begin() + first + count), best_order;
        std::vector<double> right_area(count);
        for (int a = 0; a < 3; a++) {
            std::sort(tmp.begin(), tmp.end(), [&](uint32_t x, uint32_t y) { return boxes[x].centre(a) < boxes[y].centre(a); });
            rm_aabb acc;
            acc.reset();
            for (uint32_t k = count; k-- > 0;) { acc.grow(boxes[tmp[k]]); right_area[k] = area(acc); }
            acc.reset();
            for (uint32_t k = 1; k < count; k++) {
                acc.grow(boxes[tmp[k - 1]]);
                const double cost = area(acc) * k + right_area[k] * (count - k);
                const bool balanced = depth < RM_BVH_SAH_DEPTH || k == count / 2;   // deep down: median only
                if (balanced && cost < best_cost) { best_cost = cost; half = k; axis = a; best_order = tmp; }
            }
        }
        if (depth >= RM_BVH_SAH_DEPTH && best_order.empty()) half = count / 2;   // (costs were not finite)
        out.depth = std::max(out.depth, depth + 1u);
        if (!best_order.empty()) std::copy(best_order.begin(), best_order.end(), out.order.begin() + first);
        const uint32_t me = (uint32_t)(out.nodes.size() / 16u);
        out.nodes.resize(out.nodes.size() + 16u, 0.);
        axis_of.resize(me + 1u, 0);
        axis_of[me] = axis;
        rm_aabb lb, rb;
        const uint64_t lref = build(first, half, lb, depth + 1u);
        const uint64_t rref = build(first + half, count - half, rb, depth + 1u);
        inflate(lb);
        inflate(rb);
        double *n = &out.nodes[(size_t)me * 16u];
        for (int a = 0; a < 3; a++) { n[a] = lb.lo[a]; n[3 + a] = lb.hi[a]; n[6 + a] = rb.lo[a]; n[9 + a] = rb.hi[a]; }
        std::memcpy(&n[12], &lref, 8);
        std::memcpy(&n[13], &rref, 8);
        return me;                                                                // inner: high word 0
    }
};

}  // namespace rm_bvh_detail

// Builds a hierarchy over `boxes` with at most `leaf_size` primitives per leaf.  The root
// (node 0) is always an inner node: needs at least leaf_size + 1 primitives.
inline rm_bvh rm_build_bvh(const std::vector<rm_aabb> &boxes, uint32_t leaf_size) {
    rm_bvh out;
    out.order.resize(boxes.size());
    for (uint32_t i = 0; i < boxes.size(); i++) out.order[i] = i;
    rm_bvh_detail::Builder b{boxes, leaf_size, out, {}};
    rm_aabb root;
    b.build(0, (uint32_t)boxes.size(), root, 0);
#if RM_BVH4
    // fold every other level away: a node's children are its halves' halves (a half that is a leaf stays one child)
    const std::vector<double> bin = out.nodes;
    std::vector<double> wide;
    uint32_t depth4 = 0;
    struct Fold {
        const std::vector<double> &bin; const std::vector<int> &axis_of; std::vector<double> &wide; uint32_t &depth4;
        uint32_t fold(uint32_t node, uint32_t depth) {
            depth4 = std::max(depth4, depth + 1u);
            const uint32_t me = (uint32_t)(wide.size() / RM_BVH_NODE_WORDS);
            wide.resize(wide.size() + RM_BVH_NODE_WORDS, 0.);
            const double *n = &bin[(size_t)node * 16u];
            uint64_t ref[2];
            std::memcpy(&ref[0], &n[12], 8); std::memcpy(&ref[1], &n[13], 8);
            double box[4][6];
            uint32_t idx[4], cnt[4];
            uint32_t axes = (uint32_t)axis_of[node] & 3u;
            for (int h = 0; h < 2; h++) {
                const uint32_t hi = (uint32_t)(ref[h] >> 32), lo = (uint32_t)ref[h];
                for (int c = 0; c < 2; c++) { idx[2 * h + c] = 0u; cnt[2 * h + c] = ~0u; for (int k = 0; k < 6; k++) box[2 * h + c][k] = 0.; }
                if (hi != 0u) {                                   // a leaf: one child
                    for (int k = 0; k < 6; k++) box[2 * h][k] = n[6 * h + k];
                    idx[2 * h] = lo; cnt[2 * h] = hi;
                } else {                                          // an inner node: its two halves
                    const double *m = &bin[(size_t)lo * 16u];
                    uint64_t r2[2];
                    std::memcpy(&r2[0], &m[12], 8); std::memcpy(&r2[1], &m[13], 8);
                    axes |= ((uint32_t)axis_of[lo] & 3u) << (2 + 2 * h);
                    for (int c = 0; c < 2; c++) {
                        for (int k = 0; k < 6; k++) box[2 * h + c][k] = m[6 * c + k];
                        const uint32_t hi2 = (uint32_t)(r2[c] >> 32), lo2 = (uint32_t)r2[c];
                        if (hi2 != 0u) { idx[2 * h + c] = lo2; cnt[2 * h + c] = hi2; }
                        else { idx[2 * h + c] = fold(lo2, depth + 1u); cnt[2 * h + c] = 0u; }
                    }
                }
            }
            double *w = &wide[(size_t)me * RM_BVH_NODE_WORDS];   // (after the recursion: `wide` may have moved)
            for (int c = 0; c < 4; c++) for (int k = 0; k < 6; k++) w[6 * c + k] = box[c][k];
            uint32_t refs[8];
            for (int c = 0; c < 4; c++) { refs[2 * c] = idx[c]; refs[2 * c + 1] = cnt[c]; }
            std::memcpy(&w[24], refs, sizeof refs);
            std::memcpy(&w[28], &axes, 4);
            return me;
        }
    } f{bin, b.axis_of, wide, depth4};
    f.fold(0u, 0u);
    out.nodes.swap(wide);
    out.depth = 3u * depth4;          // entries the walk can park: three per level
#endif
    return out;
}

#endif
