// accel_build.h -- host side: what the fast closest-hit query walks instead of the
// reference's tree, and why its answers are still the reference's.
//
// The reference (src/bvh.rs:37-80) returns, for a ray and (t_min, t_max), the hit with the
// smallest t -- equal t: the larger shape index -- among the shapes that
//   (a) are REACHED: AABB::hit(ray, t_min, t_max) passes for the bounds of every
//       BVHMember::Node above the shape's leaf, and
//   (b) are HIT: Shape::hit(ray, t_min, t_max) returns Some.
// Its tree is a random-axis median split (bvh.rs:146, 202), poor at keeping rays away from
// shapes, and because no query shortens t_max on the way (bvh.rs:65-66) it tests every
// reached shape.  The fast query separates the two roles the reference's boxes play:
//
//  gates   (a) is decided with the reference's own boxes and arithmetic, bit for bit, but
//          only for a shape that is about to win.  Every Node's bounds is the component-wise
//          min/max over all shapes below it (bvh.rs:185, 228-229; aabb.rs:53-71), so a
//          parent's box contains each inner child's box component by component, and the
//          slab test (aabb.rs:24-51) is monotone in the box: subtraction and multiplication
//          by a fixed 1/d round monotonically; a NaN from 0 * inf is dropped by max/min and
//          only ever removes a constraint.  Hence "child box passes => parent box passes
//          => ... => root passes", and (a) equals "the box of the leaf's OWN parent
//          passes".  build_gates() keeps one gate per Node and lets every shape point at its
//          parent's; when a caller's boxes do not nest (checked), each gate also points at
//          the gate above it and the whole chain is tested.
//
//  culling a surface-area-heuristic tree over the shapes' own bounding boxes (f32, widened)
//          decides which shapes are worth testing.  It may only ever skip a shape that
//          cannot win: its slab test is conservative (device_traverse.h says how), so a
//          shape that is hit at t <= the best t so far is always found, reached or not.
//
// The query then takes the closest hit among ALL shapes, checks (a) for that one shape, and
// in the rare case that it fails (a ray grazing a gate within rounding) walks again with (a)
// checked on every hit.  The closest hit among all shapes, if reached, is the closest among
// the reached ones, so the answer is the reference's.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "device_scene.h"

namespace mirt {

struct BuiltTree {
    std::vector<CNode> nodes;
    std::vector<DGate> gates;
    std::vector<uint32_t> gate_of;  // per shape
    uint32_t root_ref = 0;
    uint32_t height = 0;  // most stack entries a traversal can hold + 1
    double time_lo = -std::numeric_limits<double>::infinity();
    double time_hi = std::numeric_limits<double>::infinity();
    float cull_reach = std::numeric_limits<float>::infinity();
};

namespace accel {

struct Box {
    double lo[3], hi[3];
};

inline Box box_union(const Box& a, const Box& b) {
    Box r;
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = std::fmin(a.lo[k], b.lo[k]);
        r.hi[k] = std::fmax(a.hi[k], b.hi[k]);
    }
    return r;
}

inline double half_area(const Box& b) {
    const double dx = std::max(0.0, b.hi[0] - b.lo[0]);
    const double dy = std::max(0.0, b.hi[1] - b.lo[1]);
    const double dz = std::max(0.0, b.hi[2] - b.lo[2]);
    const double a = dx * dy + dy * dz + dz * dx;
    return std::isfinite(a) ? a : std::numeric_limits<double>::max();
}

// `tk` = TraverseKind << REF_KIND_SHIFT | ShadeClass << REF_CLASS_SHIFT of every shape.
inline uint32_t leaf_ref(const uint32_t* tk, uint32_t shape) { return REF_LEAF | tk[shape] | shape; }

// Bounding box of the points at which Shape::hit can report a hit (src/hittable.rs:105-137,
// 185-207, 249-271, 313-335, 432-445, 487-521), whatever the sign of a radius or the order
// of a cube's corners.  A moving sphere is boxed over its own time span [ts, te].
inline Box shape_box(const mirt_shape& s) {
    Box b;
    auto sphere = [](const double c[3], double r, Box* out) {
        const double a = std::fabs(r);
        for (int k = 0; k < 3; ++k) {
            out->lo[k] = c[k] - a;
            out->hi[k] = c[k] + a;
        }
    };
    switch (s.kind) {
        case MIRT_SHAPE_SPHERE: sphere(s.f, s.f[3], &b); break;
        case MIRT_SHAPE_MOVING_SPHERE: {
            Box b2;
            sphere(s.f, s.f[3], &b);
            sphere(s.f + 4, s.f[3], &b2);
            b = box_union(b, b2);
            break;
        }
        case MIRT_SHAPE_XY_RECT: b = Box{{s.f[0], s.f[2], s.f[4]}, {s.f[1], s.f[3], s.f[4]}}; break;
        case MIRT_SHAPE_XZ_RECT: b = Box{{s.f[0], s.f[4], s.f[2]}, {s.f[1], s.f[4], s.f[3]}}; break;
        case MIRT_SHAPE_YZ_RECT: b = Box{{s.f[4], s.f[0], s.f[2]}, {s.f[4], s.f[1], s.f[3]}}; break;
        default:  // cube: six rects on the faces of [p0, p1]
            for (int k = 0; k < 3; ++k) {
                b.lo[k] = std::fmin(s.f[k], s.f[3 + k]);
                b.hi[k] = std::fmax(s.f[k], s.f[3 + k]);
            }
            break;
    }
    // anything not finite: no culling for this shape
    for (int k = 0; k < 3; ++k) {
        if (!std::isfinite(b.lo[k])) b.lo[k] = -std::numeric_limits<double>::infinity();
        if (!std::isfinite(b.hi[k])) b.hi[k] = std::numeric_limits<double>::infinity();
    }
    return b;
}

// The widening the f32 slab test of device_traverse.h relies on: every bound moves outward by
// 2^-19 of its magnitude (that test's rounding errors proportional to the bound) plus 1e-6 of
// the box's size and a small absolute amount (hits the f64 shape tests accept lie on the
// shape only up to THEIR rounding), then to the next f32 outward.
constexpr double CULL_REL = 1.0 / 524288.0;  // 2^-19
inline float f32_below(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float f32_above(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}
inline void put_box(CNode& n, int c, const Box& b) {
    double size = 0.0;
    for (int a = 0; a < 3; ++a)
        if (std::isfinite(b.hi[a] - b.lo[a])) size = std::fmax(size, b.hi[a] - b.lo[a]);
    const double slack = 1.0e-6 * size + 1.0e-30;
    for (int a = 0; a < 3; ++a) {
        n.b[a][0][c] = f32_below(b.lo[a] - (std::fabs(b.lo[a]) * CULL_REL + slack));
        n.b[a][1][c] = f32_above(b.hi[a] + (std::fabs(b.hi[a]) * CULL_REL + slack));
    }
}

}  // namespace accel

// TraverseKind and the traversal parameters (ShapeRec::f) of every shape.
inline void traversal_shapes(const mirt_scene_desc* desc, std::vector<uint32_t>* tk, std::vector<ShapeRec>* recs) {
    tk->resize(desc->n_shapes);
    recs->resize(desc->n_shapes);
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const mirt_shape& s = desc->shapes[i];
        ShapeRec& rec = (*recs)[i];
        std::memset(&rec, 0, sizeof rec);
        double* p = rec.f;
        std::memcpy(p, s.f, sizeof rec.f);
        (*tk)[i] = s.kind;
        if (s.kind == MIRT_SHAPE_SPHERE) {
            (*tk)[i] = TK_SPHERE;
            p[4] = p[5] = p[6] = 0.0;  // c1 - c0
            p[7] = 0.0;                // ts
            p[8] = 1.0;                // 1 / (te - ts)
        } else if (s.kind == MIRT_SHAPE_MOVING_SPHERE) {
            const double movement_time = s.f[8] - s.f[7];  // hittable.rs:556
            int exp = 0;
            const bool pow2 = std::isnormal(movement_time) && std::frexp(movement_time, &exp) == 0.5 &&
                              std::isnormal(1.0 / movement_time);
            if (pow2) {
                (*tk)[i] = TK_SPHERE;
                for (int a = 0; a < 3; ++a) p[4 + a] = s.f[4 + a] - s.f[a];  // hittable.rs:558
                p[7] = s.f[7];
                p[8] = 1.0 / movement_time;  // exact
            } else {
                (*tk)[i] = TK_MOVING_SPHERE;
            }
        }
    }
}

// What shading a hit needs, folded into the shapes' records: the material's constants, and for a
// texture that is not a constant its index in a table that holds only such textures and what
// they lead to (a scene has one constant texture per plain sphere; none of them is in it).
inline void shading_records(const mirt_scene_desc* desc, std::vector<ShapeRec>* recs, std::vector<mirt_texture>* table) {
    std::vector<uint8_t> wanted(desc->n_textures, 0);
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const mirt_material& m = desc->materials[desc->shapes[i].material];
        if ((m.kind == MIRT_MAT_LAMBERTIAN || m.kind == MIRT_MAT_DIFFUSE_LIGHT) && desc->textures[m.texture].kind != MIRT_TEX_CONSTANT)
            wanted[m.texture] = 1;
    }
    // children carry smaller indices than their parent (validated): one sweep from the top
    for (uint32_t t = desc->n_textures; t-- > 0;)
        if (wanted[t] && desc->textures[t].kind == MIRT_TEX_CHECKER) wanted[desc->textures[t].even] = wanted[desc->textures[t].odd] = 1;
    std::vector<uint32_t> place(desc->n_textures, NO_TEXTURE);
    for (uint32_t t = 0; t < desc->n_textures; ++t) {
        if (!wanted[t]) continue;
        mirt_texture copy = desc->textures[t];
        if (copy.kind == MIRT_TEX_CHECKER) {
            copy.even = place[copy.even];
            copy.odd = place[copy.odd];
        }
        place[t] = (uint32_t)table->size();
        table->push_back(copy);
    }
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        ShapeRec& rec = (*recs)[i];
        const mirt_material& m = desc->materials[desc->shapes[i].material];
        rec.material = desc->shapes[i].material;
        rec.texture = NO_TEXTURE;
        rec.param = m.param;
        if (m.kind == MIRT_MAT_LAMBERTIAN || m.kind == MIRT_MAT_DIFFUSE_LIGHT) {
            const mirt_texture& tex = desc->textures[m.texture];
            if (tex.kind == MIRT_TEX_CONSTANT) std::memcpy(rec.color, tex.color, sizeof rec.color);
            else rec.texture = place[m.texture];
        } else {
            std::memcpy(rec.color, m.albedo, sizeof rec.color);
        }
    }
}

// True when every Node's box contains the boxes of its inner children component-wise.
inline bool boxes_nest(const mirt_scene_desc* desc) {
    for (uint32_t i = 0; i < desc->n_nodes; ++i) {
        const mirt_bvh_member& m = desc->nodes[i];
        if (m.is_leaf) continue;
        for (uint32_t child : {m.left, m.right}) {
            const mirt_bvh_member& ch = desc->nodes[child];
            if (ch.is_leaf) continue;
            for (int a = 0; a < 3; ++a)
                if (!(m.bmin[a] <= ch.bmin[a]) || !(m.bmax[a] >= ch.bmax[a])) return false;
        }
    }
    return true;
}

// One gate per BVHMember::Node; every shape points at its parent's.
inline void build_gates(const mirt_scene_desc* desc, BuiltTree* t) {
    const uint32_t nm = desc->n_nodes;
    const bool nest = boxes_nest(desc);
    t->gate_of.assign(desc->n_shapes, NO_GATE);
    std::vector<uint32_t> gate_index(nm, NO_GATE);
    // members come children first (post-order); walk from the root down so that a node's
    // parent gate exists when the node's own is made
    for (uint32_t i = nm; i-- > 0;) {
        const mirt_bvh_member& m = desc->nodes[i];
        if (m.is_leaf) continue;
        if (gate_index[i] == NO_GATE) {  // the root (or, in a malformed array, an orphan)
            gate_index[i] = (uint32_t)t->gates.size();
            DGate g;
            std::memset(&g, 0, sizeof g);
            g.next = NO_GATE;
            t->gates.push_back(g);
        }
        DGate& g = t->gates[gate_index[i]];
        for (int a = 0; a < 3; ++a) {
            g.box[a][0] = m.bmin[a];
            g.box[a][1] = m.bmax[a];
        }
        for (uint32_t child : {m.left, m.right}) {
            const mirt_bvh_member& ch = desc->nodes[child];
            if (ch.is_leaf) {
                t->gate_of[ch.shape] = gate_index[i];
            } else {
                gate_index[child] = (uint32_t)t->gates.size();
                DGate cg;
                std::memset(&cg, 0, sizeof cg);
                cg.next = nest ? NO_GATE : gate_index[i];
                t->gates.push_back(cg);
            }
        }
    }
}

// ShapeRec::sure of every shape: its gate's box moved inward by what the f32 test of
// RayWalk::confirm may err by (2^-19 of each bound's magnitude, as the culling boxes move
// outward), then to the next f32 inward.  A shape without a gate gets a box no ray misses; a
// shape whose gate leads on to further gates (boxes that do not nest) or whose gate is too thin
// gets one no ray is sure to cross, and the exact test decides.
inline void sure_boxes(const BuiltTree& t, std::vector<ShapeRec>* recs) {
    using namespace accel;
    const float big = std::numeric_limits<float>::max();
    for (size_t i = 0; i < recs->size(); ++i) {
        ShapeRec& rec = (*recs)[i];
        const uint32_t g = t.gate_of[i];
        for (int a = 0; a < 3; ++a) {
            if (g == NO_GATE) {
                rec.sure[a][0] = -big;
                rec.sure[a][1] = big;
                continue;
            }
            const DGate& gate = t.gates[g];
            const double lo = gate.box[a][0], hi = gate.box[a][1];
            const double size = hi - lo;
            const bool usable = gate.next == NO_GATE && std::isfinite(lo) && std::isfinite(hi) && size > 0.0 &&
                                std::fabs(lo) < 0x1.0p40 && std::fabs(hi) < 0x1.0p40;
            float l = big, h = -big;
            if (usable) {
                const double slack = 1.0e-6 * size + 1.0e-30;
                l = f32_above(lo + (std::fabs(lo) * CULL_REL + slack));
                h = f32_below(hi - (std::fabs(hi) * CULL_REL + slack));
                if (!(l < h)) {
                    l = big;
                    h = -big;
                }
            }
            rec.sure[a][0] = l;
            rec.sure[a][1] = h;
        }
    }
}

namespace accel {

struct Item {
    Box box;
    uint32_t shape;
    double centroid[3];
};

struct Builder {
    const uint32_t* tk;
    std::vector<Item>& items;
    std::vector<CNode>& out;
    uint32_t max_height;
    uint32_t height_seen = 0;

    // What a subtree looks like to its parent.
    struct Sub {
        uint32_t ref;
        Box box;
    };

    // Builds the subtree over items[first, first+count) at `depth` (root = 0).
    Sub build(uint32_t first, uint32_t count, uint32_t depth) {
        if (count == 1) {
            height_seen = std::max(height_seen, depth + 1);
            return Sub{leaf_ref(tk, items[first].shape), items[first].box};
        }
        Item* g = items.data() + first;
        // levels still available below this node
        const uint32_t budget = max_height > depth + 2 ? max_height - depth - 2 : 0;
        uint32_t need = 0;
        while ((1u << need) < count) ++need;  // ceil(log2(count))
        uint32_t split = count / 2;
        int axis = -1;
        if (need < budget) {
            // surface-area heuristic, full sweep over the three axes
            double best = std::numeric_limits<double>::infinity();
            std::vector<double> right_area(count);
            for (int a = 0; a < 3; ++a) {
                std::sort(g, g + count, [a](const Item& x, const Item& y) { return x.centroid[a] < y.centroid[a]; });
                Box acc = g[count - 1].box;
                for (uint32_t i = count - 1; i > 0; --i) {
                    acc = box_union(acc, g[i].box);
                    right_area[i] = half_area(acc);
                }
                acc = g[0].box;
                for (uint32_t i = 1; i < count; ++i) {
                    const double cost = half_area(acc) * i + right_area[i] * (count - i);
                    if (cost < best) {
                        best = cost;
                        axis = a;
                        split = i;
                    }
                    acc = box_union(acc, g[i].box);
                }
            }
        }
        if (axis < 0) {
            // out of depth budget (or degenerate): balanced split on the widest axis
            Box all = g[0].box;
            for (uint32_t i = 1; i < count; ++i) all = box_union(all, g[i].box);
            axis = 0;
            for (int a = 1; a < 3; ++a)
                if (all.hi[a] - all.lo[a] > all.hi[axis] - all.lo[axis]) axis = a;
            split = count / 2;
        }
        std::sort(g, g + count, [axis](const Item& x, const Item& y) { return x.centroid[axis] < y.centroid[axis]; });
        const uint32_t self = (uint32_t)out.size();
        out.emplace_back();
        const Sub l = build(first, split, depth + 1);
        const Sub r = build(first + split, count - split, depth + 1);
        CNode& n = out[self];
        std::memset(&n, 0, sizeof n);
        n.child[0] = l.ref;
        n.child[1] = r.ref;
        put_box(n, 0, l.box);
        put_box(n, 1, r.box);
        return Sub{self, box_union(l.box, r.box)};
    }
};

}  // namespace accel

// The items of a scene of a few rects and cubes, which is rendered without a tree (device_scene.h,
// ScanItem), and their culling boxes in pairs.  `ref_bits` as for build_fast_tree.  False: the scene
// has a sphere, or more than SCAN_MAX_ITEMS items.
inline bool build_scan_items(const mirt_scene_desc* desc, const std::vector<ShapeRec>& recs, const uint32_t* ref_bits,
                             std::vector<ScanBoxes>* pairs, std::vector<ScanItem>* items) {
    using namespace accel;
    pairs->clear();
    items->clear();
    if (desc->n_shapes > 256) return false;
    std::vector<Box> boxes;
    auto add = [&](uint32_t shape, uint32_t exact_kind, uint32_t side, const double* f, int n_f, const Box& box) {
        ScanItem it;
        std::memset(&it, 0, sizeof it);
        std::memcpy(it.f, f, sizeof(double) * n_f);
        const uint32_t tk = (ref_bits[shape] >> REF_KIND_SHIFT) & 7u, cls = (ref_bits[shape] >> REF_CLASS_SHIFT) & 7u;
        it.ref = shape | exact_kind << 8 | (side | cls << 4 | tk << 7) << 16;
        items->push_back(it);
        boxes.push_back(box);
    };
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const mirt_shape& s = desc->shapes[i];
        const uint32_t tk = (ref_bits[i] >> REF_KIND_SHIFT) & 7u;
        if (tk == TK_SPHERE || tk == TK_MOVING_SPHERE) {
            // a ray that crosses a sphere's box misses the sphere one time in four, and the item reached
            // first then settles nothing: measured slower than the tree from four spheres on
            items->clear();
            return false;
        }
        if (tk != TK_CUBE) {
            add(i, tk, 0, recs[i].f, 5, shape_box(s));
            continue;
        }
        // the six faces in the order of hittable.rs:432-445 (cube_hit, device_intersect.h)
        const double p0x = s.f[0], p0y = s.f[1], p0z = s.f[2], p1x = s.f[3], p1y = s.f[4], p1z = s.f[5];
        const double faces[6][5] = {{p0x, p1x, p0y, p1y, p0z}, {p0x, p1x, p0y, p1y, p1z}, {p0x, p1x, p0z, p1z, p1y},
                                    {p0x, p1x, p0z, p1z, p0y}, {p0y, p1y, p0z, p1z, p1x}, {p0y, p1y, p0z, p1z, p0x}};
        const uint32_t kinds[6] = {TK_XY_RECT, TK_XY_RECT, TK_XZ_RECT, TK_XZ_RECT, TK_YZ_RECT, TK_YZ_RECT};
        for (uint32_t side = 0; side < 6; ++side) {
            mirt_shape face;
            std::memset(&face, 0, sizeof face);
            face.kind = kinds[side];  // (TraverseKind and MIRT_SHAPE_* number the rects alike)
            std::memcpy(face.f, faces[side], sizeof faces[side]);
            add(i, kinds[side], side, faces[side], 5, shape_box(face));
        }
    }
    if (items->size() > SCAN_MAX_ITEMS) {
        items->clear();
        return false;
    }
    pairs->resize((items->size() + 1) / 2);
    for (size_t k = 0; k < pairs->size(); ++k) {
        CNode n;
        std::memset(&n, 0, sizeof n);
        put_box(n, 0, boxes[2 * k]);
        put_box(n, 1, boxes[std::min(2 * k + 1, boxes.size() - 1)]);  // (an odd count: the last box twice, never a candidate)
        ScanBoxes& out = (*pairs)[k];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 2; ++c) {
                out.b[a][0][c] = out.b[a][2][c] = n.b[a][0][c];
                out.b[a][1][c] = n.b[a][1][c];
            }
    }
    return true;
}

// How far from the origin of the coordinates a ray may start and still be culled.
//
// Culling assumes that a hit the f64 shape tests accept lies on its shape up to a slack far below
// the widening of the boxes.  For rects that holds at any distance (t = (k - o) / d is one
// division).  The sphere's quadratic (hittable.rs:106-119) cancels: with oc = o - c its
// discriminant b^2 - a c is a difference of two numbers near a |oc|^2, each carrying about six
// roundings, so a ray that passes the centre at distance dist is still "hit" when
// dist^2 - r^2 <= 12 eps |oc|^2, i.e. up to 6.7e-16 |oc|^2 / r outside the sphere.  The boxes give
// about 1e-6 of the larger of |o| and the box's own coordinates.  With |oc| <= 2 max(|o|, B),
// B = the largest coordinate of any box, that is enough while max(|o|, B) <= 3.7e8 r; the reach is
// set at 2^24 r for the smallest sphere radius r (a factor of 20 in hand), and to zero -- no
// culling at all, every shape is tested as the reference's linear list would -- when the scene
// itself is larger than that.  Beyond the reach the reference's answer is rounding noise; it is
// reproduced all the same.
inline float cull_reach(const mirt_scene_desc* desc) {
    double r_min = std::numeric_limits<double>::infinity(), b_max = 0.0;
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const mirt_shape& s = desc->shapes[i];
        const accel::Box b = accel::shape_box(s);
        for (int a = 0; a < 3; ++a) {
            if (std::isfinite(b.lo[a])) b_max = std::fmax(b_max, std::fabs(b.lo[a]));
            if (std::isfinite(b.hi[a])) b_max = std::fmax(b_max, std::fabs(b.hi[a]));
        }
        if (s.kind == MIRT_SHAPE_SPHERE || s.kind == MIRT_SHAPE_MOVING_SPHERE) {
            const double r = std::fabs(s.f[3]);
            if (r == r) r_min = std::fmin(r_min, r);
        }
    }
    const double reach = r_min * 16777216.0;  // inf without spheres
    if (!(b_max <= reach)) return 0.0f;
    return reach > 1.0e38 ? std::numeric_limits<float>::infinity() : (float)reach;
}

// Gates from the reference's tree, culling tree over the shapes' own boxes.
// `tk`: per shape, TraverseKind << REF_KIND_SHIFT | ShadeClass << REF_CLASS_SHIFT.
inline BuiltTree build_fast_tree(const mirt_scene_desc* desc, const uint32_t* tk, uint32_t max_height) {
    using namespace accel;
    BuiltTree t;
    build_gates(desc, &t);
    t.cull_reach = cull_reach(desc);
    std::vector<Item> items(desc->n_shapes);
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const mirt_shape& s = desc->shapes[i];
        items[i].shape = i;
        items[i].box = shape_box(s);
        for (int a = 0; a < 3; ++a) {
            const double c = 0.5 * items[i].box.lo[a] + 0.5 * items[i].box.hi[a];
            items[i].centroid[a] = std::isfinite(c) ? c : 0.0;
        }
        if (s.kind == MIRT_SHAPE_MOVING_SPHERE) {
            // Moving::center (hittable.rs:550-561) stays between its two end points for ts <= time <= te
            const double a = std::fmin(s.f[7], s.f[8]), b = std::fmax(s.f[7], s.f[8]);
            t.time_lo = std::fmax(t.time_lo, a);
            t.time_hi = std::fmin(t.time_hi, b);
        }
    }
    if (desc->n_shapes == 1) {
        t.root_ref = leaf_ref(tk, 0);
        t.height = 1;
        return t;
    }
    std::vector<CNode> built;
    Builder b{tk, items, built, max_height};
    const Builder::Sub top = b.build(0, desc->n_shapes, 0);
    t.root_ref = top.ref;  // node 0
    t.height = b.height_seen + 1;
    // breadth-first order: the first k nodes are the top of the tree, which is what a kernel
    // keeps in LDS when the whole tree does not fit
    std::vector<uint32_t> order{0}, place(built.size(), 0);
    for (size_t k = 0; k < order.size(); ++k)
        for (uint32_t child : built[order[k]].child)
            if (!(child & REF_LEAF)) {
                place[child] = (uint32_t)order.size();
                order.push_back(child);
            }
    t.nodes.resize(built.size());
    for (size_t k = 0; k < order.size(); ++k) {
        CNode n = built[order[k]];
        for (uint32_t& child : n.child)
            if (!(child & REF_LEAF)) child = place[child];
        t.nodes[k] = n;
    }
    return t;
}

// ---- the culling tree with its bounds as halves (device_scene.h, QNode) ----
struct QTree {
    std::vector<QBox> boxes;  // empty: the scene does not lend itself to the format
    std::vector<QRefs> refs;
    QGrid grid{};
    // as the device reads them: the boxes, then the references
    std::vector<unsigned char> bytes() const {
        std::vector<unsigned char> out(boxes.size() * sizeof(QBox) + refs.size() * sizeof(QRefs));
        if (!boxes.empty()) {
            std::memcpy(out.data(), boxes.data(), boxes.size() * sizeof(QBox));
            std::memcpy(out.data() + boxes.size() * sizeof(QBox), refs.data(), refs.size() * sizeof(QRefs));
        }
        return out;
    }
};

namespace accel {

// Every finite IEEE half that is zero or normal, in ascending order.  Subnormal halves are left out:
// whether an instruction flushes them is a matter of the wave's mode, and a bound must mean one thing.
struct HalfTable {
    std::vector<double> value;
    std::vector<uint16_t> bits;
    HalfTable() {
        auto decode = [](uint16_t h) {
            const int e = (h >> 10) & 31, m = h & 1023;
            const double mag = std::ldexp(1.0 + m / 1024.0, e - 15);
            return (h & 0x8000) ? -mag : mag;
        };
        for (uint32_t h = 0xFBFFu; h >= 0x8400u; --h) {  // -65504 ... -2^-14
            value.push_back(decode((uint16_t)h));
            bits.push_back((uint16_t)h);
        }
        value.push_back(0.0);
        bits.push_back(0);
        for (uint32_t h = 0x0400u; h <= 0x7BFFu; ++h) {  // 2^-14 ... 65504
            value.push_back(decode((uint16_t)h));
            bits.push_back((uint16_t)h);
        }
    }
    // the largest entry <= x / the smallest entry >= x; false when there is none
    bool below(double x, uint16_t* out) const {
        const size_t k = std::upper_bound(value.begin(), value.end(), x) - value.begin();
        if (k == 0) return false;
        *out = bits[k - 1];
        return true;
    }
    bool above(double x, uint16_t* out) const {
        const size_t k = std::lower_bound(value.begin(), value.end(), x) - value.begin();
        if (k == value.size()) return false;
        *out = bits[k];
        return true;
    }
};

}  // namespace accel

// The nodes of `t` re-expressed as QNodes.  Every bound of every node must be finite and of sane size;
// at most QNODE_MAX_SHAPES shapes and fewer than QREF_LEAF nodes.
inline QTree build_qtree(const BuiltTree& t, uint32_t n_shapes) {
    using namespace accel;
    QTree q;
    if (t.nodes.empty() || t.nodes.size() >= QREF_LEAF || n_shapes > QNODE_MAX_SHAPES) return q;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) lo[a] = std::numeric_limits<double>::infinity(), hi[a] = -lo[a];
    for (const CNode& n : t.nodes)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 2; ++c) {
                const double l = n.b[a][0][c], h = n.b[a][1][c];
                if (!std::isfinite(l) || !std::isfinite(h) || std::fabs(l) > 1.0e15 || std::fabs(h) > 1.0e15) return q;
                lo[a] = std::fmin(lo[a], l);
                hi[a] = std::fmax(hi[a], h);
            }
    for (int a = 0; a < 3; ++a) {
        // A half resolves 2^-11 of a bound's distance from the centre: the centre goes where the leaves are (the
        // median of their boxes' middles), not to the middle of the extremes, which one huge shape decides.
        std::vector<double> middles;
        for (const CNode& n : t.nodes)
            for (int c = 0; c < 2; ++c)
                if (n.child[c] & REF_LEAF) middles.push_back(0.5 * n.b[a][0][c] + 0.5 * n.b[a][1][c]);
        std::nth_element(middles.begin(), middles.begin() + middles.size() / 2, middles.end());
        q.grid.center[a] = middles.empty() ? 0.5 * lo[a] + 0.5 * hi[a] : middles[middles.size() / 2];
        const double half_extent = std::fmax(hi[a] - q.grid.center[a], q.grid.center[a] - lo[a]);
        // a power of two that brings the half extent to at most 2^14 grid units: far from both ends of a half's range
        int e = 0;
        if (half_extent > 0.0) {
            (void)std::frexp(half_extent, &e);  // half_extent = m * 2^e, 0.5 <= m < 1
            e -= 14;
        }
        if (e < -60 || e > 60) return q;
        q.grid.step[a] = (float)std::ldexp(1.0, e);
    }
    static const HalfTable table;
    q.boxes.resize(t.nodes.size());
    q.refs.resize(t.nodes.size());
    double reach[3] = {0.0, 0.0, 0.0};
    for (size_t k = 0; k < t.nodes.size(); ++k) {
        const CNode& n = t.nodes[k];
        QBox& out = q.boxes[k];
        for (int c = 0; c < 2; ++c) {
            for (int a = 0; a < 3; ++a) {
                const double step = q.grid.step[a];
                // (b - center) / step: the subtraction rounds (2^-53 of the larger term), the division is exact
                const double l = (n.b[a][0][c] - q.grid.center[a]) / step, h = (n.b[a][1][c] - q.grid.center[a]) / step;
                const double slack_l = (std::fabs(n.b[a][0][c]) + std::fabs(q.grid.center[a])) / step * 0x1.0p-50;
                const double slack_h = (std::fabs(n.b[a][1][c]) + std::fabs(q.grid.center[a])) / step * 0x1.0p-50;
                if (!table.below(l - slack_l, &out.b[a][c][0]) || !table.above(h + slack_h, &out.b[a][c][1])) {
                    q.boxes.clear();
                    q.refs.clear();
                    return q;
                }
                reach[a] = std::fmax(reach[a], std::fmax(std::fabs(l), std::fabs(h)) + 32.0);
            }
            const uint32_t ref = n.child[c];
            if (ref & REF_LEAF) q.refs[k].child[c] = (uint16_t)(QREF_LEAF | (((ref >> REF_KIND_SHIFT) & 7u) << QREF_KIND_SHIFT) | (ref & REF_INDEX_MASK));
            else q.refs[k].child[c] = (uint16_t)ref;
        }
    }
    for (int a = 0; a < 3; ++a) q.grid.half[a] = f32_above(reach[a]);
    return q;
}

}  // namespace mirt
