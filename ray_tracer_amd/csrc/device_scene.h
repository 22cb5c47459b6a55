// device_scene.h -- how a scene is laid out in HBM for the gfx950 kernels, and
// the struct the kernels receive by value.
#pragma once

#include <stdint.h>

#include "../../include/mirt.h"

namespace mirt {

// A reference to a BVH child or a stack entry.
//   bit 31      : 1 = leaf (shapes), 0 = inner node
//   inner node  : bits 0..30 = index into the DNode array
//   leaf        : bit 30 = gated: test the shapes only if the box stored beside this
//                 reference passes the slab test
//                 bit 29 = pair: two shapes, index and index + 1
//                 bits 26..28 = TraverseKind of shape `index`
//                 bits 23..25 = TraverseKind of shape `index + 1` (pair only)
//                 bits 0..22  = index into the sorted shape array
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_GATED = 0x40000000u;
constexpr uint32_t REF_PAIR = 0x20000000u;
constexpr uint32_t REF_KIND_SHIFT = 26;
constexpr uint32_t REF_KIND2_SHIFT = 23;
constexpr uint32_t REF_INDEX_MASK = 0x007FFFFFu;
constexpr uint32_t MAX_SHAPES = REF_INDEX_MASK;  // 2^23 - 1

// One inner node of the traversal tree: one 128-byte line (the L2 line size on
// gfx950) holding the boxes of BOTH children, so that one fetch serves two slab
// tests and the near child can be taken first.
//
// box[c][axis][0/1] = min/max of child c along axis.  For an inner child, or a
// gated leaf, the slab test of this box decides whether the child is entered.
// For an ungated leaf child -- a leaf carries no box in the reference and is never
// slab-tested there (src/bvh.rs:45-48) -- it is the shape's own bounding box, kept
// only to order and distance-prune, never to reject.  accel_build.h says which
// boxes these are and why the decisions equal the reference's.
struct alignas(128) DNode {
    double box[2][3][2];
    uint32_t child[2];
    uint32_t pad[6];
};
static_assert(sizeof(DNode) == 128, "DNode must be one 128-byte line");

// One shape: 96 bytes, the first 32 of which are all a sphere test needs.
struct alignas(32) DShape {
    double f[10];       // same meaning as mirt_shape.f
    uint32_t kind;      // MIRT_SHAPE_*
    uint32_t material;  // index into materials
    uint32_t cls;       // shade class of its material (SHADE_*), precomputed
    uint32_t pad;
};
static_assert(sizeof(DShape) == 96, "DShape is 96 bytes");

// How the fast traversal tests a shape.  Same numbering as MIRT_SHAPE_* except that
// TK_SPHERE also covers a moving sphere whose movement time (time_end - time_start) is a
// power of two: dividing by it equals multiplying by its (exact) reciprocal, so
//     center(time) = c0 + ((time - ts) * rcp) * (c1 - c0)
// carries the bits of Moving::center (src/hittable.rs:550-561), and a static sphere is the
// same formula with c1 - c0 = 0 (c0 + x * 0 = c0 for any finite x).  One code path for
// every sphere of the usual scenes means no divergence between the two kinds.
// Parameters of the fast traversal, ten doubles per shape (DScene::tparams):
//   TK_SPHERE          {c0x, c0y, c0z, radius, c1x - c0x, c1y - c0y, c1z - c0z, ts, 1 / (te - ts)}
//   every other kind   the shape's own parameters (mirt_shape.f)
enum TraverseKind : uint32_t {
    TK_SPHERE = 0,
    TK_XY_RECT = 1,
    TK_XZ_RECT = 2,
    TK_YZ_RECT = 3,
    TK_CUBE = 4,
    TK_MOVING_SPHERE = 5  // movement time not a power of two: keeps the division
};
constexpr int TPARAMS_PER_SHAPE = 10;

// Shade classes: paths are binned by this after each closest-hit query so that a
// wave of the shade kernel runs one material/texture kind.
enum ShadeClass : uint32_t {
    SHADE_MISS = 0,
    SHADE_LAMBERTIAN_CONSTANT = 1,
    SHADE_LAMBERTIAN_CHECKER = 2,
    SHADE_LAMBERTIAN_NOISE = 3,
    SHADE_LAMBERTIAN_IMAGE = 4,
    SHADE_METAL = 5,
    SHADE_DIELECTRIC = 6,
    SHADE_LIGHT = 7,
    SHADE_NUM_CLASSES = 8
};

// Most levels the traversal kernels can walk.  The fast kernel keeps a per-lane
// stack of this many entries in LDS; the reference-order kernel keeps as many
// frames in private memory.
constexpr int MAX_TREE_HEIGHT = 24;

// What the kernels see (passed by value as a kernel argument).
struct DScene {
    const DNode* nodes;              // inner nodes; root = nodes[root_ref] unless the root is a leaf
    const DShape* shapes;            // BVH-sorted order
    const double* tparams;           // TPARAMS_PER_SHAPE doubles per shape, see TraverseKind
    const mirt_bvh_member* members;  // the reference's own member array (post-order, root last)
    const mirt_material* materials;
    const mirt_texture* textures;
    const mirt_perlin* perlins;
    const mirt_image* images;  // .rgb holds DEVICE pointers
    uint32_t n_shapes;
    uint32_t n_members;  // 0 when the scene came without a BVH
    uint32_t n_nodes;
    uint32_t root_ref;  // REF_LEAF|... when the scene has a single shape
    // bounds of the root BVHMember::Node, [axis][min,max]: every other box lives in
    // its parent's DNode, the root has no parent.  Tested first, like bvh.rs:61 does.
    double root_box[3][2];
};

}  // namespace mirt
