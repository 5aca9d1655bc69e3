// rm_internal.h -- shared between the host (rm_scene.cpp) and device (rm_device.hip)
// halves of librusty_marcher_amd.so.  Not part of the public ABI.
#ifndef RM_INTERNAL_H
#define RM_INTERNAL_H

#include "rusty_marcher_amd.h"

#include <string>

// thread-local error text for calls that have no rm_ctx (scene builder, rm_init)
void rm_set_host_error(const std::string &msg);
const char *rm_get_host_error();

// ---- device scene blob --------------------------------------------------
// The uploaded scene is one array of 8-byte words that every workgroup copies
// into LDS.  Primitives are regrouped by kind; `pid` (device primitive id) is
// the position in [spheres | polygons | triangles].  All offsets are in
// 8-byte words from the start of the blob and are multiples of 2 so that
// 16-byte LDS reads stay aligned.
//
//   spheres   4 words each : cx cy cz r^2                       (sphere.rs:6-11)
//   polygons  8 words each : nx ny nz  ppx ppy ppz  {first_vertex,n_vertices}  0
//   pverts    2 words each : x y       (polygon.rs:54-56 reads only .z of the cross
//                                       product, i.e. only x and y of the vertices)
//   triangles 12 words each: nx ny nz  cx cy cz  v0x v0y v1x v1y v2x v2y
//   materials 10 words per pid: diffusion dcx dcy dcz specular exponent
//                               reflection refractive_index is_glass 0
//   lights    8 words each : px py pz  cx cy cz  intensity 0
//   keys      1 u32 per pid (2 per word): position in Scene.shapes order, used
//                               only to break exact distance ties (shapes.rs:130)
//   bounds    4 words per pid: centre and radius of a sphere holding every point of the
//                               primitive a ray can hit (polygons: the vertices lifted onto
//                               the plane the hit test uses), inflated by 1e-7 relative
//   planar    16 words per polygon / triangle: up to four vertices lifted onto the plane of the
//                               hit test (a triangle repeats its first), the vertex count
//                               (0: no edge test -- more than four vertices, or no finite lift)
//   groups    4 words per 64 consecutive pids: a sphere around their bounding spheres (the
//                               cull's first step)
//   bvh       16 words per node (rm_bvh.hpp), one hierarchy over the spheres and one
//                               over the triangles when there are enough of them; the
//                               primitives of a kind are then stored in leaf order
struct rm_dev_header {
    uint32_t n_spheres, n_polygons, n_triangles, n_lights;
    uint32_t off_spheres, off_polygons, off_pverts, off_triangles;
    uint32_t off_materials, off_lights, off_keys, total_words;
    uint32_t list_ordered;        // grouping by kind kept Scene.shapes order (no tie keys needed)
    uint32_t off_bvh_spheres;     // 0 = walk all spheres; else the sphere hierarchy (rm_bvh.hpp)
    uint32_t off_bvh_triangles;   // 0 = walk all triangles; else the triangle hierarchy
    uint32_t off_bounds;          // bounding sphere per pid (centre, radius: 4 words), inflated -- the bundle cull reads these
    uint32_t off_planar;          // per polygon / triangle (pid - n_spheres): lifted vertices + count, 16 words (cull_step)
    uint32_t off_groups;          // 0, or (scenes of 3 to 64 cull steps) a bounding sphere per 64 consecutive pids, 4 words each
    double shadow_rho;            // every shadow ray passes within this of its light: 1e-3 x the longest normal (renderer.rs:168-172)
};

#define RM_SPHERE_WORDS 4u
#define RM_POLYGON_WORDS 16u   /* normal, plane point, (first vertex | count), pad, x/y of the first four vertices */
#define RM_PVERT_WORDS 2u
#define RM_TRIANGLE_WORDS 12u
#define RM_MATERIAL_WORDS 10u
#define RM_LIGHT_WORDS 8u

#endif
