// scene.cpp -- constructors, bounding boxes, BVH build, camera, Perlin tables
// and flattening for the host-side mirror of the reference's scene API.
// Built with -ffp-contract=off: boxes and camera vectors must carry the same
// bits the reference's unfused f64 arithmetic produces.
#include "scene.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <tuple>

namespace rt {

// ----------------------------------------------------------------- textures
Texture Texture::constant(double r, double g, double b) {
    Texture t;
    t.kind = MIRT_TEX_CONSTANT;
    t.color = Vec3(r, g, b);
    return t;
}

Texture Texture::checker_color(double r_even, double g_even, double b_even, double r_odd, double g_odd,
                               double b_odd) {
    return checker(constant(r_even, g_even, b_even), constant(r_odd, g_odd, b_odd));
}

Texture Texture::checker(const Texture& even, const Texture& odd) {
    Texture t;
    t.kind = MIRT_TEX_CHECKER;
    t.even = std::make_shared<Texture>(even);
    t.odd = std::make_shared<Texture>(odd);
    return t;
}

Texture Texture::noise(Rng& rng) { return scaled_noise(rng, 1.0); }

Texture Texture::scaled_noise(Rng& rng, double scale) {
    Texture t;
    t.kind = MIRT_TEX_NOISE;
    t.perlin = Perlin::create(rng);
    t.scale = scale;
    return t;
}

bool read_ppm_p6(const std::string& path, Image* out) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::string magic;
    in >> magic;
    if (magic != "P6") return false;
    auto next_int = [&](long* v) {
        // skip whitespace and '#' comments between header tokens
        for (;;) {
            int c = in.peek();
            if (c == '#') {
                std::string line;
                std::getline(in, line);
            } else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') {
                in.get();
            } else {
                break;
            }
        }
        return static_cast<bool>(in >> *v);
    };
    long w = 0, h = 0, maxv = 0;
    if (!next_int(&w) || !next_int(&h) || !next_int(&maxv)) return false;
    if (w <= 0 || h <= 0 || maxv != 255) return false;
    in.get();  // single whitespace after maxval
    out->width = (uint32_t)w;
    out->height = (uint32_t)h;
    out->rgb.resize((size_t)w * (size_t)h * 3);
    in.read(reinterpret_cast<char*>(out->rgb.data()), (std::streamsize)out->rgb.size());
    return in.gcount() == (std::streamsize)out->rgb.size();
}

bool read_image(const std::string& path, Image* out, std::string* why) {
    unsigned char magic[2] = {0, 0};
    {
        std::ifstream in(path, std::ios::binary);
        if (!in || !in.read(reinterpret_cast<char*>(magic), 2)) {
            if (why) *why = "cannot open the file";
            return false;
        }
    }
    if (magic[0] == 0xFF && magic[1] == 0xD8) return read_jpeg_baseline(path, out, why);
    if (magic[0] == 'P' && magic[1] == '6') {
        if (read_ppm_p6(path, out)) return true;
        if (why) *why = "malformed binary PPM";
        return false;
    }
    if (why) *why = "neither a JPEG nor a binary PPM (P6)";
    return false;
}

Texture Texture::image(const std::string& filepath) {
    auto img = std::make_shared<Image>();
    std::string why;
    if (!read_image(filepath, img.get(), &why)) {
        // texture.rs:173-177: log and fall back to the cyan error texture
        std::fprintf(stderr, "Unable to load image \"%s\": %s\n", filepath.c_str(), why.c_str());
        return constant(0.0, 1.0, 1.0);
    }
    return image(img);
}

Texture Texture::image(std::shared_ptr<Image> decoded) {
    Texture t;
    t.kind = MIRT_TEX_IMAGE;
    t.img = std::move(decoded);
    return t;
}

// ---------------------------------------------------------------- materials
Material Material::lambertian(double r, double g, double b) { return textured(Texture::constant(r, g, b)); }

Material Material::textured(const Texture& texture) {
    Material m;
    m.kind = MIRT_MAT_LAMBERTIAN;
    m.texture = texture;
    return m;
}

Material Material::metal(double e0, double e1, double e2, double fuzziness) {
    Material m;
    m.kind = MIRT_MAT_METAL;
    m.albedo = Vec3(e0, e1, e2);
    m.param = std::fmin(fuzziness, 1.0);  // Metal::new clamps, material.rs:68-71
    return m;
}

Material Material::dielectric(double ref_idx) {
    Material m;
    m.kind = MIRT_MAT_DIELECTRIC;
    m.param = ref_idx;
    return m;
}

Material Material::diffuse_light(const Texture& texture) {
    Material m;
    m.kind = MIRT_MAT_DIFFUSE_LIGHT;
    m.texture = texture;
    return m;
}

// ------------------------------------------------------------------- shapes
static Shape make_shape(uint32_t kind, std::initializer_list<double> params, const Material& m) {
    Shape s;
    s.kind = kind;
    size_t i = 0;
    for (double p : params) s.f[i++] = p;
    s.material = m;
    return s;
}

Shape Shape::sphere(double x, double y, double z, double radius, const Material& m) {
    return make_shape(MIRT_SHAPE_SPHERE, {x, y, z, radius}, m);
}
Shape Shape::xy_rect(double x0, double x1, double y0, double y1, double k, const Material& m) {
    return make_shape(MIRT_SHAPE_XY_RECT, {x0, x1, y0, y1, k}, m);
}
Shape Shape::xz_rect(double x0, double x1, double z0, double z1, double k, const Material& m) {
    return make_shape(MIRT_SHAPE_XZ_RECT, {x0, x1, z0, z1, k}, m);
}
Shape Shape::yz_rect(double y0, double y1, double z0, double z1, double k, const Material& m) {
    return make_shape(MIRT_SHAPE_YZ_RECT, {y0, y1, z0, z1, k}, m);
}
Shape Shape::cube(double x0, double y0, double z0, double x1, double y1, double z1, const Material& m) {
    return make_shape(MIRT_SHAPE_CUBE, {x0, y0, z0, x1, y1, z1}, m);
}
Shape Shape::moving_sphere(double x0, double y0, double z0, double x1, double y1, double z1, double radius,
                           const Material& m, double time_start, double time_end) {
    return make_shape(MIRT_SHAPE_MOVING_SPHERE, {x0, y0, z0, radius, x1, y1, z1, time_start, time_end}, m);
}

namespace {
struct Box {
    double lo[3], hi[3];
};
inline Box box_of(double ax, double ay, double az, double bx, double by, double bz) {
    return Box{{ax, ay, az}, {bx, by, bz}};
}
// AABB::surrounding_box, aabb.rs:53-71
inline Box box_union(const Box& a, const Box& b) {
    Box r;
    for (int c = 0; c < 3; ++c) {
        r.lo[c] = std::fmin(a.lo[c], b.lo[c]);
        r.hi[c] = std::fmax(a.hi[c], b.hi[c]);
    }
    return r;
}
// Moving::center, hittable.rs:550-561 (initial centre is always the sphere's own)
template <class S>
inline void centre_at(const S& s, double time, double c[3]) {
    const double frac = (time - s.f[7]) / (s.f[8] - s.f[7]);
    for (int a = 0; a < 3; ++a) {
        const double distance = s.f[4 + a] - s.f[a];
        c[a] = s.f[a] + frac * distance;
    }
}
inline Box sphere_box(const double c[3], double r) {
    return box_of(c[0] - r, c[1] - r, c[2] - r, c[0] + r, c[1] + r, c[2] + r);
}
// S is rt::Shape or the flat mirt_shape: both carry `kind` and `f[10]`.
template <class S>
inline Box shape_box(const S& s, double t0, double t1) {
    const double pad = 0.0001;
    switch (s.kind) {
        case MIRT_SHAPE_SPHERE: return sphere_box(s.f, s.f[3]);
        case MIRT_SHAPE_XY_RECT: return box_of(s.f[0], s.f[2], s.f[4] - pad, s.f[1], s.f[3], s.f[4] + pad);
        case MIRT_SHAPE_XZ_RECT: return box_of(s.f[0], s.f[4] - pad, s.f[2], s.f[1], s.f[4] + pad, s.f[3]);
        case MIRT_SHAPE_YZ_RECT: return box_of(s.f[4] - pad, s.f[0], s.f[2], s.f[4] + pad, s.f[1], s.f[3]);
        case MIRT_SHAPE_CUBE: {
            const double *p0 = s.f, *p1 = s.f + 3;
            // the six sides in the order Cube::new lists them (hittable.rs:405-412)
            Box b = box_of(p0[0], p0[1], p0[2] - pad, p1[0], p1[1], p0[2] + pad);
            b = box_union(b, box_of(p0[0], p0[1], p1[2] - pad, p1[0], p1[1], p1[2] + pad));
            b = box_union(b, box_of(p0[0], p1[1] - pad, p0[2], p1[0], p1[1] + pad, p1[2]));
            b = box_union(b, box_of(p0[0], p0[1] - pad, p0[2], p1[0], p0[1] + pad, p1[2]));
            b = box_union(b, box_of(p1[0] - pad, p0[1], p0[2], p1[0] + pad, p1[1], p1[2]));
            b = box_union(b, box_of(p0[0] - pad, p0[1], p0[2], p0[0] + pad, p1[1], p1[2]));
            return b;
        }
        case MIRT_SHAPE_MOVING_SPHERE: {
            double ci[3], cf[3];
            centre_at(s, t0, ci);
            centre_at(s, t1, cf);
            Box bi = sphere_box(ci, s.f[3]);
            if (ci[0] == cf[0] && ci[1] == cf[1] && ci[2] == cf[2]) return bi;
            return box_union(bi, sphere_box(cf, s.f[3]));
        }
        default: throw std::runtime_error("unknown shape kind");
    }
}
}  // namespace

void Shape::bounding_box(double time_start, double time_end, double bmin[3], double bmax[3]) const {
    Box b = shape_box(*this, time_start, time_end);
    std::memcpy(bmin, b.lo, sizeof b.lo);
    std::memcpy(bmax, b.hi, sizeof b.hi);
}

// ---------------------------------------------------------------------- BVH
namespace {
template <class S>
struct Builder {
    Rng& rng;
    std::vector<mirt_bvh_member>& out;
    double t0, t1;

    static mirt_bvh_member blank() {
        mirt_bvh_member m;
        std::memset(&m, 0, sizeof m);
        return m;
    }
    static void set_box(mirt_bvh_member& m, const Box& b) {
        std::memcpy(m.bmin, b.lo, sizeof b.lo);
        std::memcpy(m.bmax, b.hi, sizeof b.hi);
    }
    uint32_t emit(const mirt_bvh_member& m) {
        out.push_back(m);
        return (uint32_t)(out.size() - 1);
    }

    // build_tree_internal, bvh.rs:137-237.  One axis draw per call (leaf calls
    // included); the slice is re-sorted in place by the minimum corner on that
    // axis with the reference's "a - b < 0" ordering, stable for equal keys.
    mirt_bvh_member build(S* slice, uint32_t offset, uint32_t n) {
        const uint32_t axis = rng.gen_below(3);
        std::vector<std::pair<double, uint32_t>> keys(n);
        for (uint32_t i = 0; i < n; ++i) keys[i] = {shape_box(slice[i], t0, t1).lo[axis], i};
        std::stable_sort(keys.begin(), keys.end(),
                         [](const std::pair<double, uint32_t>& a, const std::pair<double, uint32_t>& b) {
                             return a.first - b.first < 0.0;
                         });
        {
            std::vector<S> sorted;
            sorted.reserve(n);
            for (uint32_t i = 0; i < n; ++i) sorted.push_back(std::move(slice[keys[i].second]));
            for (uint32_t i = 0; i < n; ++i) slice[i] = std::move(sorted[i]);
        }

        mirt_bvh_member m = blank();
        if (n == 1) {
            m.is_leaf = 1;
            m.shape = offset;
            return m;
        }
        if (n == 2) {
            mirt_bvh_member l = blank(), r = blank();
            l.is_leaf = r.is_leaf = 1;
            l.shape = offset;
            r.shape = offset + 1;
            set_box(m, box_union(shape_box(slice[0], t0, t1), shape_box(slice[1], t0, t1)));
            m.left = emit(l);
            m.right = emit(r);
            return m;
        }
        const uint32_t half = n / 2;
        m.left = emit(build(slice, offset, half));
        m.right = emit(build(slice + half, offset + half, n - half));
        Box all = shape_box(slice[0], t0, t1);
        for (uint32_t i = 1; i < n; ++i) all = box_union(all, shape_box(slice[i], t0, t1));
        set_box(m, all);
        return m;
    }
};
}  // namespace

BoundingVolumeHierarchy BoundingVolumeHierarchy::create(Rng& rng, std::vector<Shape> shapes, double time_initial,
                                                        double time_final) {
    if (shapes.empty()) throw std::runtime_error("BoundingVolumeHierarchy: empty shape list");
    BoundingVolumeHierarchy bvh;
    bvh.shapes = std::move(shapes);
    bvh.nodes.reserve(bvh.shapes.size() * 2);
    Builder<Shape> b{rng, bvh.nodes, time_initial, time_final};
    mirt_bvh_member root = b.build(bvh.shapes.data(), 0, (uint32_t)bvh.shapes.size());
    bvh.nodes.push_back(root);  // the root goes last, bvh.rs:130-131
    return bvh;
}

void bvh_build_flat(Rng& rng, mirt_shape* shapes, uint32_t n, double time_initial, double time_final,
                    std::vector<mirt_bvh_member>* nodes) {
    if (n == 0) throw std::runtime_error("bvh_build_flat: empty shape list");
    nodes->clear();
    nodes->reserve((size_t)n * 2);
    Builder<mirt_shape> b{rng, *nodes, time_initial, time_final};
    mirt_bvh_member root = b.build(shapes, 0, n);
    nodes->push_back(root);
}

void flat_shape_bbox(const mirt_shape& s, double time_start, double time_end, double bmin[3], double bmax[3]) {
    Box b = shape_box(s, time_start, time_end);
    std::memcpy(bmin, b.lo, sizeof b.lo);
    std::memcpy(bmax, b.hi, sizeof b.hi);
}

// ------------------------------------------------------------------- camera
Camera Camera::create(const Vec3& look_from, const Vec3& look_at, const Vec3& v_up, double vertical_fov,
                      double aspect, double aperture, double focus_distance, double time_start, double time_end) {
    const double pi = 3.14159265358979323846264338327950288;
    auto sub = [](const Vec3& a, const Vec3& b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); };
    auto scale = [](const Vec3& a, double k) { return Vec3(a.x * k, a.y * k, a.z * k); };
    auto unit = [&](const Vec3& a) {
        const double k = 1.0 / std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);  // vec3.rs:75-78
        return scale(a, k);
    };
    auto cross = [](const Vec3& a, const Vec3& b) {  // vec3.rs:92-98
        return Vec3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
    };
    const double theta = vertical_fov * pi / 180.0;
    const double half_height = std::tan(theta / 2.0);
    const double half_width = aspect * half_height;
    const Vec3 w = unit(sub(look_from, look_at));
    const Vec3 u = unit(cross(v_up, w));
    const Vec3 v = cross(w, u);
    const Vec3 origin = look_from;
    const Vec3 llc = sub(sub(sub(origin, scale(u, half_width * focus_distance)), scale(v, half_height * focus_distance)),
                         scale(w, focus_distance));
    const Vec3 horizontal = scale(u, 2.0 * half_width * focus_distance);
    const Vec3 vertical = scale(v, 2.0 * half_height * focus_distance);

    Camera cam;
    auto put = [](double dst[3], const Vec3& s) {
        dst[0] = s.x;
        dst[1] = s.y;
        dst[2] = s.z;
    };
    put(cam.c.origin, origin);
    put(cam.c.lower_left_corner, llc);
    put(cam.c.horizontal, horizontal);
    put(cam.c.vertical, vertical);
    put(cam.c.u, u);
    put(cam.c.v, v);
    put(cam.c.w, w);
    cam.c.lens_radius = aperture / 2.0;
    cam.c.time_start = time_start;
    cam.c.time_end = time_end;
    return cam;
}

// ------------------------------------------------------------------- perlin
std::shared_ptr<Perlin> Perlin::create(Rng& rng) {
    auto p = std::make_shared<Perlin>();
    for (int i = 0; i < 256; ++i) {
        // Vec3::random(rng, -1.0, 1.0).unit(), perlin.rs:27
        const double x = rng.gen_range(-1.0, 1.0);
        const double y = rng.gen_range(-1.0, 1.0);
        const double z = rng.gen_range(-1.0, 1.0);
        const double k = 1.0 / std::sqrt(x * x + y * y + z * z);
        p->tables.vecs[i][0] = x * k;
        p->tables.vecs[i][1] = y * k;
        p->tables.vecs[i][2] = z * k;
    }
    int32_t* perm[3] = {p->tables.perm_x, p->tables.perm_y, p->tables.perm_z};
    for (int t = 0; t < 3; ++t) {
        // generate_perm, perlin.rs:88-98: (0..256).rev() shuffled
        std::vector<int32_t> v(256);
        for (int i = 0; i < 256; ++i) v[i] = 255 - i;
        rng.shuffle(v);
        std::copy(v.begin(), v.end(), perm[t]);
    }
    return p;
}

// ------------------------------------------------------------------ flatten
namespace {
struct Flattener {
    FlatScene& fs;
    std::map<const Perlin*, uint32_t> perlin_ids;
    std::map<const Image*, uint32_t> image_ids;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, double, double, double, double>, uint32_t>
        texture_ids;
    std::map<std::tuple<uint32_t, uint32_t, double, double, double, double>, uint32_t> material_ids;

    uint32_t add_perlin(const std::shared_ptr<Perlin>& p) {
        auto it = perlin_ids.find(p.get());
        if (it != perlin_ids.end()) return it->second;
        fs.perlins.push_back(p->tables);
        return perlin_ids[p.get()] = (uint32_t)(fs.perlins.size() - 1);
    }
    uint32_t add_image(const std::shared_ptr<Image>& im) {
        auto it = image_ids.find(im.get());
        if (it != image_ids.end()) return it->second;
        fs.image_storage.push_back(im);
        mirt_image mi;
        mi.width = im->width;
        mi.height = im->height;
        mi.rgb = im->rgb.data();
        fs.images.push_back(mi);
        return image_ids[im.get()] = (uint32_t)(fs.images.size() - 1);
    }
    uint32_t add_texture(const Texture& t) {
        mirt_texture mt;
        std::memset(&mt, 0, sizeof mt);
        mt.kind = t.kind;
        switch (t.kind) {
            case MIRT_TEX_CONSTANT:
                mt.color[0] = t.color.x;
                mt.color[1] = t.color.y;
                mt.color[2] = t.color.z;
                break;
            case MIRT_TEX_CHECKER:
                mt.even = add_texture(*t.even);
                mt.odd = add_texture(*t.odd);
                break;
            case MIRT_TEX_NOISE:
                mt.perlin = add_perlin(t.perlin);
                mt.scale = t.scale;
                break;
            case MIRT_TEX_IMAGE: mt.image = add_image(t.img); break;
            default: throw std::runtime_error("unknown texture kind");
        }
        auto key = std::make_tuple(mt.kind, mt.even, mt.odd, mt.perlin, mt.image, mt.color[0], mt.color[1],
                                   mt.color[2], mt.scale);
        auto it = texture_ids.find(key);
        if (it != texture_ids.end()) return it->second;
        fs.textures.push_back(mt);
        return texture_ids[key] = (uint32_t)(fs.textures.size() - 1);
    }
    uint32_t add_material(const Material& m) {
        mirt_material mm;
        std::memset(&mm, 0, sizeof mm);
        mm.kind = m.kind;
        if (m.kind == MIRT_MAT_LAMBERTIAN || m.kind == MIRT_MAT_DIFFUSE_LIGHT) mm.texture = add_texture(m.texture);
        if (m.kind == MIRT_MAT_METAL) {
            mm.albedo[0] = m.albedo.x;
            mm.albedo[1] = m.albedo.y;
            mm.albedo[2] = m.albedo.z;
        }
        if (m.kind == MIRT_MAT_METAL || m.kind == MIRT_MAT_DIELECTRIC) mm.param = m.param;
        auto key = std::make_tuple(mm.kind, mm.texture, mm.albedo[0], mm.albedo[1], mm.albedo[2], mm.param);
        auto it = material_ids.find(key);
        if (it != material_ids.end()) return it->second;
        fs.materials.push_back(mm);
        return material_ids[key] = (uint32_t)(fs.materials.size() - 1);
    }
};
}  // namespace

std::unique_ptr<FlatScene> flatten(const BoundingVolumeHierarchy& world) {
    auto fs = std::make_unique<FlatScene>();
    Flattener fl{*fs, {}, {}, {}, {}};
    fs->shapes.reserve(world.shapes.size());
    for (const Shape& s : world.shapes) {
        mirt_shape ms;
        std::memset(&ms, 0, sizeof ms);
        ms.kind = s.kind;
        ms.material = fl.add_material(s.material);
        std::memcpy(ms.f, s.f, sizeof ms.f);
        fs->shapes.push_back(ms);
    }
    fs->nodes = world.nodes;
    mirt_scene_desc& d = fs->desc;
    std::memset(&d, 0, sizeof d);
    d.shapes = fs->shapes.data();
    d.n_shapes = (uint32_t)fs->shapes.size();
    d.materials = fs->materials.data();
    d.n_materials = (uint32_t)fs->materials.size();
    d.textures = fs->textures.data();
    d.n_textures = (uint32_t)fs->textures.size();
    d.perlins = fs->perlins.empty() ? nullptr : fs->perlins.data();
    d.n_perlins = (uint32_t)fs->perlins.size();
    d.images = fs->images.empty() ? nullptr : fs->images.data();
    d.n_images = (uint32_t)fs->images.size();
    d.nodes = fs->nodes.empty() ? nullptr : fs->nodes.data();
    d.n_nodes = (uint32_t)fs->nodes.size();
    return fs;
}

}  // namespace rt
