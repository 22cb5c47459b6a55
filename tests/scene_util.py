"""Hand-built scenes for known-answer tests: plain ctypes values of include/mirt.h,
BVH built by the oracle's restatement of BoundingVolumeHierarchy::new."""
import ctypes as C

import oracle_lib
from ray_tracer_amd import _abi as abi


def shape(kind, material, *f):
    s = abi.Shape()
    s.kind, s.material = kind, material
    for i, x in enumerate(f):
        s.f[i] = float(x)
    return s


def sphere(cx, cy, cz, r, material=0):
    return shape(abi.SHAPE_SPHERE, material, cx, cy, cz, r)


def moving_sphere(c0, c1, r, t0, t1, material=0):
    return shape(abi.SHAPE_MOVING_SPHERE, material, c0[0], c0[1], c0[2], r, c1[0], c1[1], c1[2], t0, t1)


def material(kind, texture=0, albedo=(0.0, 0.0, 0.0), param=0.0):
    m = abi.Material()
    m.kind, m.texture, m.param = kind, texture, float(param)
    for i in range(3):
        m.albedo[i] = float(albedo[i])
    return m


def constant(r, g, b):
    t = abi.Texture()
    t.kind = abi.TEX_CONSTANT
    t.color[0], t.color[1], t.color[2] = float(r), float(g), float(b)
    return t


def checker(even, odd):
    t = abi.Texture()
    t.kind, t.even, t.odd = abi.TEX_CHECKER, even, odd
    return t


def noise(perlin, scale):
    t = abi.Texture()
    t.kind, t.perlin, t.scale = abi.TEX_NOISE, perlin, float(scale)
    return t


def image_texture(image):
    t = abi.Texture()
    t.kind, t.image = abi.TEX_IMAGE, image
    return t


class Built:
    """A mirt_scene_desc and everything it points to."""

    def __init__(self, shapes, materials=None, textures=None, perlins=(), images=(), bvh=True, seed=7, t0=0.0, t1=1.0):
        materials = materials or [material(abi.MAT_LAMBERTIAN, 0)]
        textures = textures or [constant(0.5, 0.5, 0.5)]
        self.shapes = (abi.Shape * len(shapes))(*shapes)
        self.materials = (abi.Material * len(materials))(*materials)
        self.textures = (abi.Texture * len(textures))(*textures)
        self.perlins = (abi.Perlin * max(1, len(perlins)))(*perlins)
        self.image_arrays = [im for im in images]  # numpy (H, W, 3) uint8, kept alive
        self.images = (abi.Image * max(1, len(images)))()
        for i, im in enumerate(images):
            self.images[i].width, self.images[i].height = im.shape[1], im.shape[0]
            self.images[i].rgb = im.ctypes.data_as(C.POINTER(C.c_uint8))
        self.nodes = None
        n_nodes = 0
        if bvh:
            # sorts self.shapes in place, as the reference's build does
            self.nodes = (abi.BvhMember * (2 * len(shapes) - 1))()
            n_nodes = oracle_lib.lib().oracle_bvh_build(self.shapes, len(shapes), t0, t1, seed, self.nodes)
            assert n_nodes == 2 * len(shapes) - 1
        d = abi.SceneDesc()
        d.shapes, d.n_shapes = self.shapes, len(shapes)
        d.materials, d.n_materials = self.materials, len(materials)
        d.textures, d.n_textures = self.textures, len(textures)
        d.perlins, d.n_perlins = self.perlins, len(perlins)
        d.images, d.n_images = self.images, len(images)
        d.nodes, d.n_nodes = self.nodes, n_nodes
        self.desc = d


def ray(o, d, time=0.0):
    r = abi.Ray()
    for i in range(3):
        r.origin[i], r.direction[i] = float(o[i]), float(d[i])
    r.time = float(time)
    return r


def hit_detail(built, r, t_min=0.001, t_max=float("inf"), use_bvh=True):
    """-> None or dict(t,u,v,p,normal,front,material,prim,side) of the oracle's Hit."""
    out = (C.c_double * 13)()
    if not oracle_lib.lib().oracle_hit_detail(C.byref(built.desc), C.byref(r), t_min, t_max, int(use_bvh), out):
        return None
    o = list(out)
    return {"t": o[0], "u": o[1], "v": o[2], "p": o[3:6], "normal": o[6:9], "front": o[9] == 1.0,
            "material": int(o[10]), "prim": int(o[11]), "side": int(o[12])}
