// accel_build.h -- host side: turns the reference's BVHMember array into the tree
// of DNodes the fast closest-hit kernel walks.
//
// Two builders:
//
//  flatten_reference()  keeps the reference's topology (src/bvh.rs:137-237): one
//      DNode per BVHMember::Node, children boxes copied into the parent.
//
//  regroup()  keeps the reference's DECISIONS but not its topology.  In the
//      reference a leaf is tested iff every ancestor's box passes AABB::hit
//      (src/bvh.rs:61-66).  Every Node's bounds is the component-wise min/max
//      over all shapes below it (bvh.rs:185, 228-229; aabb.rs:53-71), so a
//      parent's box contains each inner child's box component by component.
//      The slab test (aabb.rs:24-51) is monotone in the box: with
//      parent.min <= child.min and parent.max >= child.max, every t0 of the
//      parent is <= the child's and every t1 >= the child's (subtraction and
//      multiplication by a fixed 1/d round monotonically; a NaN from 0 * inf is
//      dropped by max/min and only ever removes a constraint, and it can appear
//      for the parent only where the child's bound is no tighter), hence
//            child box passes  =>  parent box passes  =>  ... => root passes.
//      So "all ancestors pass" equals "the nearest ancestor that is a Node
//      passes", i.e. a leaf is tested iff the box of its OWN parent passes.
//      regroup() therefore collects, for every Node with at least one leaf
//      child, a GROUP = {that node's exact box, its 1 or 2 leaf children}, and
//      builds a fresh surface-area-heuristic tree over the groups whose inner
//      boxes are exact min/max unions of group boxes.  By the same monotonicity
//      an inner box can only fail when every group below fails, so the fresh
//      tree never hides a group the reference would have reached, and the group's
//      own box test -- bit for bit the reference's -- is what admits its leaves.
//      The precondition (parent boxes contain child boxes) is checked on the
//      caller's array; if it does not hold, flatten_reference() is used instead.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "device_scene.h"

namespace mirt {

struct BuiltTree {
    std::vector<DNode> nodes;
    uint32_t root_ref = 0;
    double root_box[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    uint32_t height = 0;  // most stack entries a traversal can hold + 1
    bool regrouped = false;
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

// `tk` = TraverseKind of every shape (how the fast traversal tests it).
inline uint32_t leaf_ref(const uint32_t* tk, uint32_t shape, bool gated) {
    return REF_LEAF | (gated ? REF_GATED : 0u) | (tk[shape] << REF_KIND_SHIFT) | shape;
}

// Bounding box of a shape, used only to order and distance-prune leaf children of a
// group node (never to reject).  A moving sphere is boxed over its own time span.
inline Box advisory_box(const mirt_shape& s) {
    Box b;
    auto sphere = [](const double c[3], double r, Box* out) {
        const double a = std::fabs(r);
        for (int k = 0; k < 3; ++k) {
            out->lo[k] = c[k] - a;
            out->hi[k] = c[k] + a;
        }
    };
    const double pad = 0.0001;
    switch (s.kind) {
        case MIRT_SHAPE_SPHERE: sphere(s.f, s.f[3], &b); break;
        case MIRT_SHAPE_MOVING_SPHERE: {
            Box b2;
            sphere(s.f, s.f[3], &b);
            sphere(s.f + 4, s.f[3], &b2);
            b = box_union(b, b2);
            break;
        }
        case MIRT_SHAPE_XY_RECT:
            b = Box{{s.f[0], s.f[2], s.f[4] - pad}, {s.f[1], s.f[3], s.f[4] + pad}};
            break;
        case MIRT_SHAPE_XZ_RECT:
            b = Box{{s.f[0], s.f[4] - pad, s.f[2]}, {s.f[1], s.f[4] + pad, s.f[3]}};
            break;
        case MIRT_SHAPE_YZ_RECT:
            b = Box{{s.f[4] - pad, s.f[0], s.f[2]}, {s.f[4] + pad, s.f[1], s.f[3]}};
            break;
        default:  // cube
            for (int k = 0; k < 3; ++k) {
                b.lo[k] = std::fmin(s.f[k], s.f[3 + k]) - pad;
                b.hi[k] = std::fmax(s.f[k], s.f[3 + k]) + pad;
            }
            break;
    }
    return b;
}

inline void put_box(DNode& n, int c, const Box& b) {
    for (int a = 0; a < 3; ++a) {
        n.box[c][a][0] = b.lo[a];
        n.box[c][a][1] = b.hi[a];
    }
}

inline Box member_box(const mirt_bvh_member& m) {
    Box b;
    std::memcpy(b.lo, m.bmin, sizeof b.lo);
    std::memcpy(b.hi, m.bmax, sizeof b.hi);
    return b;
}

}  // namespace accel

// TraverseKind and the ten traversal parameters of every shape (device_scene.h).
inline void traversal_shapes(const mirt_scene_desc* desc, std::vector<uint32_t>* tk, std::vector<double>* params) {
    tk->resize(desc->n_shapes);
    params->assign((size_t)desc->n_shapes * TPARAMS_PER_SHAPE, 0.0);
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const mirt_shape& s = desc->shapes[i];
        double* p = params->data() + (size_t)i * TPARAMS_PER_SHAPE;
        std::memcpy(p, s.f, sizeof s.f);
        (*tk)[i] = s.kind;
        if (s.kind == MIRT_SHAPE_SPHERE) {
            (*tk)[i] = TK_SPHERE;
            p[4] = p[5] = p[6] = 0.0;  // c1 - c0
            p[7] = 0.0;                // ts
            p[8] = 1.0;                // 1 / (te - ts)
            p[9] = 0.0;
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
                p[9] = 0.0;
            } else {
                (*tk)[i] = TK_MOVING_SPHERE;
            }
        }
    }
}

// One DNode per BVHMember::Node, numbered breadth-first from the root.
inline BuiltTree flatten_reference(const mirt_scene_desc* desc, const uint32_t* tk) {
    using namespace accel;
    BuiltTree t;
    const uint32_t nm = desc->n_nodes;
    const mirt_bvh_member& root = desc->nodes[nm - 1];
    if (root.is_leaf) {
        t.root_ref = leaf_ref(tk, root.shape, false);
        t.height = 1;
        return t;
    }
    std::vector<uint32_t> order, node_of(nm, 0xFFFFFFFFu), level(nm, 0);
    order.push_back(nm - 1);
    node_of[nm - 1] = 0;
    level[nm - 1] = 1;
    for (size_t k = 0; k < order.size(); ++k) {
        const mirt_bvh_member& m = desc->nodes[order[k]];
        t.height = std::max(t.height, level[order[k]]);
        for (uint32_t child : {m.left, m.right})
            if (!desc->nodes[child].is_leaf) {
                node_of[child] = (uint32_t)order.size();
                level[child] = level[order[k]] + 1;
                order.push_back(child);
            }
    }
    t.nodes.resize(order.size());
    for (size_t k = 0; k < order.size(); ++k) {
        const mirt_bvh_member& m = desc->nodes[order[k]];
        DNode& n = t.nodes[k];
        std::memset(&n, 0, sizeof n);
        const uint32_t kids[2] = {m.left, m.right};
        for (int c = 0; c < 2; ++c) {
            const mirt_bvh_member& ch = desc->nodes[kids[c]];
            if (ch.is_leaf) {
                n.child[c] = leaf_ref(tk, ch.shape, false);
                put_box(n, c, advisory_box(desc->shapes[ch.shape]));
            } else {
                n.child[c] = node_of[kids[c]];
                put_box(n, c, member_box(ch));
            }
        }
    }
    t.root_ref = 0;
    const Box rb = member_box(root);
    for (int a = 0; a < 3; ++a) {
        t.root_box[a][0] = rb.lo[a];
        t.root_box[a][1] = rb.hi[a];
    }
    t.height += 1;
    return t;
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

namespace accel {

struct Group {
    Box box;             // exact bounds of the reference node the leaves hang from
    uint32_t shapes[2];  // its leaf children, left to right
    uint32_t n_shapes;
    double centroid[3];
};

struct Builder {
    const mirt_scene_desc* desc;
    const uint32_t* tk;
    std::vector<Group>& groups;
    std::vector<DNode>& out;
    uint32_t max_height;
    uint32_t height_seen = 0;

    // What a subtree looks like to its parent.
    struct Sub {
        uint32_t ref;
        Box box;
    };

    // A group becomes a gated leaf reference -- the parent's copy of the group box admits
    // it -- for one shape, or for two shapes with consecutive indices (what bvh.rs:187-188
    // always produces).  Two shapes that are not neighbours get a DNode whose children
    // are ungated leaves: the parent's copy of the group box admitted the node, nothing
    // more gates the shapes.
    Sub emit_group(const Group& g, uint32_t depth) {
        height_seen = std::max(height_seen, depth + 1);
        if (g.n_shapes == 1) return Sub{leaf_ref(tk, g.shapes[0], true), g.box};
        if (g.shapes[1] == g.shapes[0] + 1) {
            const uint32_t ref = leaf_ref(tk, g.shapes[0], true) | REF_PAIR | (tk[g.shapes[1]] << REF_KIND2_SHIFT);
            return Sub{ref, g.box};
        }
        DNode n;
        std::memset(&n, 0, sizeof n);
        for (int c = 0; c < 2; ++c) {
            n.child[c] = leaf_ref(tk, g.shapes[c], false);
            put_box(n, c, advisory_box(desc->shapes[g.shapes[c]]));
        }
        out.push_back(n);
        return Sub{(uint32_t)(out.size() - 1), g.box};
    }

    // Builds the subtree over groups[first, first+count) at `depth` (root = 0).
    Sub build(uint32_t first, uint32_t count, uint32_t depth) {
        if (count == 1) return emit_group(groups[first], depth);
        Group* g = groups.data() + first;
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
                std::sort(g, g + count, [a](const Group& x, const Group& y) { return x.centroid[a] < y.centroid[a]; });
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
        std::sort(g, g + count, [axis](const Group& x, const Group& y) { return x.centroid[axis] < y.centroid[axis]; });
        const uint32_t self = (uint32_t)out.size();
        out.emplace_back();
        const Sub l = build(first, split, depth + 1);
        const Sub r = build(first + split, count - split, depth + 1);
        DNode& n = out[self];
        std::memset(&n, 0, sizeof n);
        n.child[0] = l.ref;
        n.child[1] = r.ref;
        put_box(n, 0, l.box);
        put_box(n, 1, r.box);
        return Sub{self, box_union(l.box, r.box)};
    }
};

}  // namespace accel

// Surface-area-heuristic tree over the reference's leaf groups.  Requires boxes_nest(desc).
inline BuiltTree regroup(const mirt_scene_desc* desc, const uint32_t* tk, uint32_t max_height) {
    using namespace accel;
    BuiltTree t;
    t.regrouped = true;
    const uint32_t nm = desc->n_nodes;
    const mirt_bvh_member& root = desc->nodes[nm - 1];
    if (root.is_leaf) {
        t.root_ref = leaf_ref(tk, root.shape, false);
        t.height = 1;
        return t;
    }
    std::vector<Group> groups;
    for (uint32_t i = 0; i < nm; ++i) {
        const mirt_bvh_member& m = desc->nodes[i];
        if (m.is_leaf) continue;
        Group g;
        g.n_shapes = 0;
        for (uint32_t child : {m.left, m.right})
            if (desc->nodes[child].is_leaf) g.shapes[g.n_shapes++] = desc->nodes[child].shape;
        if (g.n_shapes == 0) continue;
        g.box = member_box(m);
        for (int a = 0; a < 3; ++a) {
            const double c = 0.5 * g.box.lo[a] + 0.5 * g.box.hi[a];
            g.centroid[a] = std::isfinite(c) ? c : 0.0;
        }
        groups.push_back(g);
    }
    Builder b{desc, tk, groups, t.nodes, max_height};
    const Builder::Sub top = b.build(0, (uint32_t)groups.size(), 0);
    t.root_ref = top.ref;
    for (int a = 0; a < 3; ++a) {
        t.root_box[a][0] = top.box.lo[a];
        t.root_box[a][1] = top.box.hi[a];
    }
    t.height = b.height_seen + 1;
    return t;
}

}  // namespace mirt
