// device_scene.h -- how a scene is laid out in HBM for the gfx950 kernels, and
// the struct the kernels receive by value.
#pragma once

#include <stdint.h>

#include "../../include/mirt.h"

namespace mirt {

// A reference to a child of the culling tree.
//   bit 31      : 1 = leaf (one shape), 0 = inner node
//   inner node  : bits 0..30 = index into the CNode array
//   leaf        : bits 26..28 = TraverseKind of the shape
//                 bits 23..25 = ShadeClass of its material
//                 bits 0..22  = index into the sorted shape array
constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_KIND_SHIFT = 26;
constexpr uint32_t REF_CLASS_SHIFT = 23;
constexpr uint32_t REF_INDEX_MASK = 0x007FFFFFu;
constexpr uint32_t MAX_SHAPES = REF_INDEX_MASK;  // 2^23 - 1
// A leaf reference in 16 bits, for scenes of at most REF16_MAX_SHAPES shapes: index | class << 10 |
// kind << 13.  Never zero: every shape has a class, and SHADE_MISS = 0 is none's.
constexpr uint32_t REF16_MAX_SHAPES = 1024, REF16_INDEX_MASK = 0x3FFu, REF16_CLASS_SHIFT = 10, REF16_KIND_SHIFT = 13;
__host__ __device__ constexpr uint32_t ref16_of(uint32_t ref) {
    return (ref & REF16_INDEX_MASK) | (((ref >> REF_CLASS_SHIFT) & 63u) << REF16_CLASS_SHIFT);
}

// One inner node of the culling tree (accel_build.h): the boxes of BOTH children as
// f32, laid out so that one 8-byte read fetches the same bound of the two children and the
// near / far bound of an axis is picked by address:
//   b[axis][0][c] = lower bound of child c, b[axis][1][c] = upper bound of child c.
// The boxes only ever CULL: they are the shapes' own bounding boxes, widened, and a slab
// test on them is built to pass whenever the reference's f64 arithmetic could accept a hit
// of a shape below (device_traverse.h).  What admits a hit is the gate of its shape.
struct alignas(8) CNode {
    float b[3][2][2];
    uint32_t child[2];
};
static_assert(sizeof(CNode) == 56, "CNode is 56 bytes");

// The same tree in half the bytes, for scenes whose 56-byte nodes do not fit the CU's LDS but whose
// 28-byte ones do (up to ~3 700 shapes): every bound as an IEEE half, in units of `QGrid::step` from
// `QGrid::center` per axis, rounded outward on the host (accel_build.h, build_qtree).  One 8-byte read
// fetches an axis: a dword per child with the lower bound in its low half and the upper in its high; a
// rotation by 0 or 16 bits (the ray's direction along the axis) puts the near bound low, and
// v_fma_mix_f32 takes either half as an operand, so a node visit needs no conversion instructions.
// A half resolves 2^-11 of a bound's distance from the centre: the boxes come out that much larger than
// the f32 ones, which costs shape tests and no answer.
//   QBox::b[axis][c][0 / 1] = lower / upper bound of child c (bits of a half)
//   QRefs::child[c]         = inner node: its index (< QREF_LEAF);
//                             leaf: QREF_LEAF | TraverseKind << QREF_KIND_SHIFT | index of the shape
// In memory: all QBoxes (24 bytes each, so 8-byte reads stay aligned), then all QRefs.
struct alignas(8) QBox {
    uint16_t b[3][2][2];
};
struct alignas(4) QRefs {
    uint16_t child[2];
};
static_assert(sizeof(QBox) == 24 && sizeof(QRefs) == 4, "a QNode is 24 + 4 bytes");
constexpr uint32_t QNODE_BYTES = sizeof(QBox) + sizeof(QRefs);
constexpr uint32_t QREF_LEAF = 0x8000u, QREF_KIND_SHIFT = 12, QREF_INDEX_MASK = 0xFFFu;
constexpr uint32_t QNODE_MAX_SHAPES = QREF_INDEX_MASK + 1u;
struct QGrid {
    double center[3];  // world coordinate of grid coordinate 0
    float step[3];     // world length of one grid unit: a power of two
    float half[3];     // largest magnitude of a grid coordinate of the tree
};

// The boxes of the reference's tree that decide whether a shape is tested at all: in the
// reference a leaf is reached iff AABB::hit passes for every BVHMember::Node above it
// (src/bvh.rs:61-66).  gate_of[shape] names the gate of the shape's parent node; `next`
// leads to the gate of the next ancestor that still has to be tested (NO_GATE: none).
// accel_build.h cuts the chain after the first gate when boxes nest.
constexpr uint32_t NO_GATE = 0xFFFFFFFFu;
struct alignas(8) DGate {
    double box[3][2];  // [axis][min,max], the bits of the reference's AABB
    uint32_t next;
    uint32_t pad;
};
static_assert(sizeof(DGate) == 56, "DGate is 56 bytes");

// One shape: 96 bytes, the first 32 of which are all a sphere test needs.
struct alignas(32) DShape {
    double f[10];       // same meaning as mirt_shape.f
    uint32_t kind;      // MIRT_SHAPE_*
    uint32_t material;  // index into materials
    uint32_t cls;       // shade class of its material (SHADE_*), precomputed
    uint32_t tk;        // TraverseKind
};
static_assert(sizeof(DShape) == 96, "DShape is 96 bytes");

// How the fast traversal tests a shape.  Same numbering as MIRT_SHAPE_* except that
// TK_SPHERE also covers a moving sphere whose movement time (time_end - time_start) is a
// power of two: dividing by it equals multiplying by its (exact) reciprocal, so
//     center(time) = c0 + ((time - ts) * rcp) * (c1 - c0)
// carries the bits of Moving::center (src/hittable.rs:550-561), and a static sphere is the
// same formula with c1 - c0 = 0 (c0 + x * 0 = c0 for any finite x).  One code path for
// every sphere of the usual scenes means no divergence between the two kinds.
enum TraverseKind : uint32_t {
    TK_SPHERE = 0,
    TK_XY_RECT = 1,
    TK_XZ_RECT = 2,
    TK_YZ_RECT = 3,
    TK_CUBE = 4,
    TK_MOVING_SPHERE = 5  // movement time not a power of two: keeps the division
};

// One record per shape (DScene::recs), everything the path kernel needs to test the shape
// and to shade a hit on it, so that it reads ONE place (staged in LDS when the scene fits):
//   f[0..8]    parameters of the fast traversal
//                TK_SPHERE          {c0x, c0y, c0z, radius, c1x - c0x, c1y - c0y, c1z - c0z, ts, 1 / (te - ts)}
//                every other kind   the shape's own parameters (mirt_shape.f)
//   material   index into DScene::materials | ShadeClass of the material << REC_CLASS_SHIFT
//   texture    lambertian / diffuse light whose texture is not a constant: index into
//              DScene::shade_textures; NO_TEXTURE otherwise
//   color      lambertian / diffuse light with a constant texture: its colour; metal: albedo
//   param      metal: fuzziness; dielectric: refraction index
//   sure       [axis][lo, hi]: the box of the shape's gate moved INWARD (accel_build.h); a ray that
//              provably crosses it needs no exact gate test (RayWalk::confirm)
constexpr uint32_t NO_TEXTURE = 0xFFFFFFFFu;
constexpr uint32_t REC_CLASS_SHIFT = 28, REC_MATERIAL_MASK = (1u << REC_CLASS_SHIFT) - 1u;
struct alignas(8) ShapeRec {
    double f[9];
    uint32_t material;
    uint32_t texture;
    double color[3];
    double param;
    float sure[3][2];
};
static_assert(sizeof(ShapeRec) == 136, "ShapeRec is 136 bytes");

// A scene of a handful of rects and cubes (the Cornell box) is rendered without a tree (ScanWalk,
// device_traverse.h): every ray is tested against the culling box of every ITEM, and exactly against the
// item it reaches first.  An item is a rect or ONE FACE of a cube (hittable.rs:432-445 tests a cube as
// six rects and keeps the closest, the later face on equal t -- which is what taking the faces as
// items of their own, in that order, amounts to).  Items are numbered in the order (shape, face).
//   f          {a0, a1, b0, b1, k} as rect_hit takes them
//   ref        bits 0..7 shape index | bits 8..10 TraverseKind of the rect (its orientation)
//              | bits 16.. what HitRec::side carries for a hit of this item: side | ShadeClass << 4 | TraverseKind
//              of the SHAPE << 7
constexpr uint32_t SCAN_MAX_ITEMS = 64;  // the items a ray has to test exactly are a 64-bit mask
struct alignas(8) ScanItem {
    double f[5];
    uint32_t ref;
    uint32_t pad;
};
static_assert(sizeof(ScanItem) == 48, "ScanItem is 48 bytes");
// The culling boxes of items 2 i and 2 i + 1, as a CNode holds its two children's -- but with the lower
// bounds a second time after the upper ones, so that near and far bound of an axis are read at
// base + (ray runs down the axis ? 8 : 0) and 8 bytes behind it: one address per axis and ray.
struct alignas(8) ScanBoxes {
    float b[3][3][2];  // [axis][lower, upper, lower][item]
};
static_assert(sizeof(ScanBoxes) == 72, "ScanBoxes is 72 bytes");

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
    const CNode* nodes;              // culling tree; root = nodes[root_ref] unless the root is a leaf
    const DGate* gates;              // one per BVHMember::Node of the reference's tree
    const uint32_t* gate_of;         // per shape: its first gate, or NO_GATE
    const DShape* shapes;            // BVH-sorted order
    const ShapeRec* recs;            // one per shape, see ShapeRec
    const mirt_texture* shade_textures;  // the textures ShapeRec::texture can lead to, children before parents
    uint32_t n_shade_textures;
    const mirt_bvh_member* members;  // the reference's own member array (post-order, root last)
    const mirt_material* materials;
    const mirt_texture* textures;
    const mirt_perlin* perlins;
    const mirt_image* images;  // .rgb holds DEVICE pointers
    uint32_t n_shapes;
    uint32_t n_members;  // 0 when the scene came without a BVH
    uint32_t n_nodes;
    uint32_t root_ref;  // REF_LEAF|... when the scene has a single shape
    // rays whose time lies outside [time_lo, time_hi] are not culled by boxes at all (a
    // moving sphere is boxed over its own time span only); -inf, +inf without moving shapes
    double time_lo, time_hi;
    // ... and rays that start farther than this from the origin of the coordinates (largest
    // component) are not culled either: see accel_build.h, cull_reach
    float cull_reach;
    uint32_t n_scan_items;           // 0: the scene has too many shapes to be rendered without a tree
    const ScanBoxes* scan_boxes;     // culling boxes of the items, two per entry
    const ScanItem* scan_items;
    const QBox* qnodes;              // the culling tree with bounds as halves (same numbering as `nodes`): n_nodes QBoxes, then n_nodes QRefs; or null
    QGrid qgrid;
};

}  // namespace mirt
