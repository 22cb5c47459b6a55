//! Writes a `BoundingVolumeHierarchy` (bvh.rs:11-15) out as the flat arrays of include/mirt.h, and
//! `render_world` on top of it.  Restates `flatten` of ray_tracer_amd/host/scene.cpp (which is built and
//! tested); this file itself has not been compiled (no Rust toolchain in this repository's image).
//! Needs the fields named in ffi/README.md to be `pub(crate)`.
use std::collections::HashMap;

use image::GenericImageView;

use crate::{
    bvh::{BVHMember, BoundingVolumeHierarchy},
    camera::Camera,
    ffi::*,
    hittable::{Rect, Shape},
    material::Material,
    texture::Texture,
    vec3::Vec3,
};

fn arr(v: &Vec3) -> [f64; 3] { [v[0], v[1], v[2]] }

/// FNV-1a, 64 bits.
fn fnv(bytes: &[u8]) -> u64 {
    bytes.iter().fold(0xCBF2_9CE4_8422_2325u64, |h, b| (h ^ *b as u64).wrapping_mul(0x0000_0100_0000_01B3))
}

/// What makes two textures the same texture.  `Debug` holds every field of the constant, checker and noise
/// textures, but `ImageTexture` prints the fixed word "ImageTexture" (texture.rs:100-104): an image is keyed
/// by its content -- dimensions and a hash of its RGB8 pixels -- and a checker by its children's keys, since it
/// may hold an image.
fn texture_key(t: &Texture) -> String {
    match t {
        Texture::Image(i) => {
            let (w, h) = i.image.dimensions();
            format!("image:{}x{}:{:016x}", w, h, fnv(&image_rgb8(&i.image)))
        }
        Texture::Checker(c) => format!("checker({},{})", texture_key(&c.even), texture_key(&c.odd)),
        other => format!("{:?}", other),
    }
}

/// ... and two materials the same material (a lambertian or a light may hold an image).
fn material_key(m: &Material) -> String {
    match m {
        Material::Lambertian(l) => format!("lambertian({})", texture_key(&l.albedo)),
        Material::DiffuseLight(l) => format!("light({})", texture_key(&l.texture)),
        other => format!("{:?}", other),
    }
}

/// Tightly packed RGB8, row 0 = top: what `get_pixel` hands `ImageTexture::value` (texture.rs:118-121).
fn image_rgb8(image: &image::DynamicImage) -> Vec<u8> {
    let (w, h) = image.dimensions();
    let mut rgb = Vec::with_capacity((w * h * 3) as usize);
    for y in 0..h { for x in 0..w { let p = image.get_pixel(x, y); rgb.extend_from_slice(&[p[0], p[1], p[2]]); } }
    rgb
}

/// Owns everything a `MirtSceneDesc` points into.
#[derive(Default)]
pub struct FlatScene {
    pub shapes: Vec<MirtShape>,
    pub materials: Vec<MirtMaterial>,
    pub textures: Vec<MirtTexture>,
    pub perlins: Vec<MirtPerlin>,
    pub pixels: Vec<Vec<u8>>,   // RGB8 of every image texture, row 0 = top
    pub images: Vec<MirtImage>,
    pub nodes: Vec<MirtBvhMember>,
    // the reference clones a material into every shape (hittable.rs:406-411) and a texture into every material:
    // equal values are written once (material_key / texture_key)
    material_ids: HashMap<String, u32>,
    texture_ids: HashMap<String, u32>,
}

impl FlatScene {
    fn texture(&mut self, t: &Texture) -> u32 {
        let key = texture_key(t);
        if let Some(&id) = self.texture_ids.get(&key) { return id; }
        let mut out = MirtTexture::default();
        match t {
            Texture::Constant(c) => { out.kind = TEX_CONSTANT; out.color = arr(&c.color); }
            Texture::Checker(c) => {
                // children first: libmirt wants a checker's children at smaller indices
                out.kind = TEX_CHECKER;
                out.even = self.texture(&c.even);
                out.odd = self.texture(&c.odd);
            }
            Texture::Noise(n) => {
                out.kind = TEX_NOISE;
                out.scale = n.scale;
                out.perlin = self.perlins.len() as u32;
                let mut p = MirtPerlin { vecs: [[0.0; 3]; 256], perm_x: n.noise.perm_x, perm_y: n.noise.perm_y, perm_z: n.noise.perm_z };
                for (dst, v) in p.vecs.iter_mut().zip(n.noise.random_vecs.iter()) { *dst = arr(v); }
                self.perlins.push(p);
            }
            Texture::Image(i) => {
                out.kind = TEX_IMAGE;
                out.image = self.pixels.len() as u32;
                let (w, h) = i.image.dimensions();
                self.pixels.push(image_rgb8(&i.image));
                self.images.push(MirtImage { width: w, height: h, rgb: std::ptr::null() });  // pointer set in desc()
            }
        }
        let id = self.textures.len() as u32;
        self.textures.push(out);
        self.texture_ids.insert(key, id);
        id
    }

    fn material(&mut self, m: &Material) -> u32 {
        let key = material_key(m);
        if let Some(&id) = self.material_ids.get(&key) { return id; }
        let mut out = MirtMaterial::default();
        match m {
            Material::Lambertian(l) => { out.kind = MAT_LAMBERTIAN; out.texture = self.texture(&l.albedo); }
            Material::Dielectric(d) => { out.kind = MAT_DIELECTRIC; out.param = d.ref_idx; }
            Material::Metal(x) => { out.kind = MAT_METAL; out.albedo = arr(&x.albedo); out.param = x.fuzziness; }
            Material::DiffuseLight(l) => { out.kind = MAT_DIFFUSE_LIGHT; out.texture = self.texture(&l.texture); }
        }
        let id = self.materials.len() as u32;
        self.materials.push(out);
        self.material_ids.insert(key, id);
        id
    }

    fn shape(&mut self, s: &Shape) {
        let mut out = MirtShape::default();
        match s {
            Shape::Sphere(x) => {
                out.kind = SHAPE_SPHERE;
                out.f[..3].copy_from_slice(&arr(&x.center)); out.f[3] = x.radius;
                out.material = self.material(&x.material);
            }
            Shape::XyRect(r) => { out.kind = SHAPE_XY_RECT; out.f[..5].copy_from_slice(&[r.x0, r.x1, r.y0, r.y1, r.k]); out.material = self.material(&r.material); }
            Shape::XzRect(r) => { out.kind = SHAPE_XZ_RECT; out.f[..5].copy_from_slice(&[r.x0, r.x1, r.z0, r.z1, r.k]); out.material = self.material(&r.material); }
            Shape::YzRect(r) => { out.kind = SHAPE_YZ_RECT; out.f[..5].copy_from_slice(&[r.y0, r.y1, r.z0, r.z1, r.k]); out.material = self.material(&r.material); }
            Shape::Cube(c) => {
                // Cube::new (hittable.rs:403-420) builds its six rects from (min, max) and one material
                out.kind = SHAPE_CUBE;
                out.f[..3].copy_from_slice(&arr(&c.min)); out.f[3..6].copy_from_slice(&arr(&c.max));
                // every side holds a clone of the one material: read it from sides[0], which is always Rect::Xy
                out.material = match &c.sides[0] {
                    Rect::Xy(r) => self.material(&r.material),
                    Rect::Xz(r) => self.material(&r.material),
                    Rect::Yz(r) => self.material(&r.material),
                };
            }
            Shape::MovingSphere(m) => {
                out.kind = SHAPE_MOVING_SPHERE;
                out.f[..3].copy_from_slice(&arr(&m.object.center)); out.f[3] = m.object.radius;
                out.f[4..7].copy_from_slice(&arr(&m.center_final)); out.f[7] = m.time_start; out.f[8] = m.time_end;
                out.material = self.material(&m.object.material);
            }
        }
        self.shapes.push(out);
    }

    pub fn desc(&mut self) -> MirtSceneDesc {
        for (im, px) in self.images.iter_mut().zip(self.pixels.iter()) { im.rgb = px.as_ptr(); }
        MirtSceneDesc {
            shapes: self.shapes.as_ptr(), n_shapes: self.shapes.len() as u32,
            materials: self.materials.as_ptr(), n_materials: self.materials.len() as u32,
            textures: self.textures.as_ptr(), n_textures: self.textures.len() as u32,
            perlins: self.perlins.as_ptr(), n_perlins: self.perlins.len() as u32,
            images: self.images.as_ptr(), n_images: self.images.len() as u32,
            nodes: self.nodes.as_ptr(), n_nodes: self.nodes.len() as u32,
        }
    }
}

impl BoundingVolumeHierarchy {
    /// Shapes in the order the build left them (a hit's primitive id indexes them) and the members in the
    /// reference's own order: post-order, root last (bvh.rs:33-35, 130-131).
    pub(crate) fn flatten(&self) -> FlatScene {
        let mut flat = FlatScene::default();
        for s in &self.shapes { flat.shape(s); }
        for m in &self.nodes {
            flat.nodes.push(match m {
                BVHMember::Leaf(i) => MirtBvhMember { is_leaf: 1, shape: *i as u32, ..Default::default() },
                BVHMember::Node { bounds, left, right } => MirtBvhMember {
                    is_leaf: 0,
                    left: left.expect("build_tree_internal always sets both children") as u32,
                    right: right.expect("build_tree_internal always sets both children") as u32,
                    shape: 0,
                    bmin: arr(bounds.min()),
                    bmax: arr(bounds.max()),
                },
            });
        }
        flat
    }
}

impl Camera {
    pub(crate) fn to_ffi(&self) -> MirtCamera {
        MirtCamera { origin: arr(&self.origin), lower_left_corner: arr(&self.lower_left_corner), horizontal: arr(&self.horizontal),
                     vertical: arr(&self.vertical), u: arr(&self.u), v: arr(&self.v), w: arr(&self.w),
                     lens_radius: self.lens_radius, time_start: self.time_start, time_end: self.time_end }
    }
}

/// New body for `renderer::render_world` (renderer.rs:9-62): the SAME signature, the same `Vec<u8>` (W*H*3
/// bytes, top row first).  The sampler seed is drawn from entropy, as the reference seeds every worker
/// (renderer.rs:30, 47).  With more than one device visible the frame is tile-split over all of them.
pub fn render_world<'a>(background: &Vec3, world: &'a BoundingVolumeHierarchy, camera: &Camera, screen_width: usize,
                        screen_height: usize, antialias_iterations: usize, _render_parallel: bool,
                        use_bounding_volume: bool) -> Vec<u8> {
    let mut flat = world.flatten();
    let desc = flat.desc();
    let n_devices = unsafe { mirt_device_count() }.max(1) as usize;
    let mut scenes: Vec<*mut MirtScene> = vec![std::ptr::null_mut(); n_devices];
    for (device, scene) in scenes.iter_mut().enumerate() {
        check(unsafe { mirt_scene_create(&desc, device as i32, scene) });
    }
    let mut buffer = vec![0u8; screen_width * screen_height * 3];
    let opts = MirtRenderOpts::new(use_bounding_volume);
    let multi = MirtMultiOpts { struct_size: std::mem::size_of::<MirtMultiOpts>() as u32, split: SPLIT_TILE, ..Default::default() };
    let bg = arr(background);
    let seed: u64 = rand::random();
    let handles: Vec<*const MirtScene> = scenes.iter().map(|s| *s as *const MirtScene).collect();
    check(unsafe {
        mirt_render_multi(handles.as_ptr(), handles.len() as u32, &camera.to_ffi(), bg.as_ptr(), screen_width as u32,
                          screen_height as u32, antialias_iterations as u32, 50 /* renderer.rs:83 */, seed, &opts, &multi,
                          buffer.as_mut_ptr(), std::ptr::null_mut())
    });
    for scene in scenes { unsafe { mirt_scene_destroy(scene) }; }
    buffer
}
