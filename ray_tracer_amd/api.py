"""Python host over the C ABI: scene set-up through libmirt_host, rendering and
closest-hit queries through libmirt (HIP, gfx950).

Names follow the reference: ``gen_world`` (src/world.rs:9-26), ``Camera.new``
(src/camera.rs:21-63), ``render_world`` (src/renderer.rs:9-62),
``BoundingVolumeHierarchy.hit`` (src/bvh.rs:83-86).
"""
import ctypes as C

import numpy as np

from . import _abi as abi

# numpy views of the POD structs that cross the boundary in bulk
RAY_DTYPE = np.dtype([("origin", "<f8", 3), ("direction", "<f8", 3), ("time", "<f8")])
HIT_DTYPE = np.dtype([("prim", "<i4"), ("side", "<i4"), ("t", "<f8")])
assert RAY_DTYPE.itemsize == C.sizeof(abi.Ray) and HIT_DTYPE.itemsize == C.sizeof(abi.Hit)

SKY = (0.70, 0.80, 1.00)  # the background main.rs:80 keeps as a comment
BLACK = (0.0, 0.0, 0.0)   # the background main.rs:81 uses
MAX_DEPTH = 50            # renderer.rs:83


class MirtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"mirt status {status}: {message}")
        self.status = status


def _check(rc):
    if rc != abi.OK:
        raise MirtError(rc, abi.libmirt().mirt_last_error().decode())


def _check_host(rc):
    if rc < 0:
        raise MirtError(rc, abi.libhost().mirt_host_last_error().decode())
    return rc


def device_count():
    return abi.libmirt().mirt_device_count()


def _vec3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class Camera:
    """Camera::new (src/camera.rs:21-63); main.rs fixes v_up = (0,1,0) and focus_distance = 10."""

    def __init__(self, c_struct):
        self.c = c_struct

    @classmethod
    def new(cls, look_from, look_at, v_up=(0.0, 1.0, 0.0), vertical_fov=15.0, aspect=2.0, aperture=0.2,
            focus_distance=10.0, time_start=0.0, time_end=1.0):
        cam = abi.Camera()
        _check_host(abi.libhost().mirt_host_camera_new(_vec3(look_from), _vec3(look_at), _vec3(v_up), vertical_fov,
                                                        aspect, aperture, focus_distance, time_start, time_end,
                                                        C.byref(cam)))
        return cls(cam)


def image_struct(rgb):
    """mirt_image over an (H, W, 3) uint8 array; the array must outlive the struct."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.ndim == 3 and rgb.shape[2] == 3
    im = abi.Image(rgb.shape[1], rgb.shape[0], rgb.ctypes.data_as(C.POINTER(C.c_uint8)))
    im._keep = rgb
    return im


class World:
    """A BoundingVolumeHierarchy (src/bvh.rs:11-27) as flat arrays, built on the CPU."""

    def __init__(self, handle, keep=None):
        self._handle = handle
        self._keep = keep
        self.desc = abi.libhost().mirt_host_world_desc(handle)

    @classmethod
    def generate(cls, scene, max_objects=10, seed=0x5EED0001, time_initial=0.0, time_final=1.0, earth_rgb=None,
                 earth_path=None):
        """gen_world (src/world.rs:9-26) for `scene` in Basic, Dynamic, Checker, Perlin, Earth,
        SimpleLight, CornellBox, Textured, NextWeek."""
        handle = C.c_void_p()
        im = image_struct(earth_rgb) if earth_rgb is not None else None
        _check_host(abi.libhost().mirt_host_world_create(
            scene.encode(), max_objects, seed, time_initial, time_final, C.byref(im) if im is not None else None,
            earth_path.encode() if earth_path else None, C.byref(handle)))
        return cls(handle, keep=im)

    @property
    def n_shapes(self):
        return self.desc.contents.n_shapes

    @property
    def n_nodes(self):
        return self.desc.contents.n_nodes

    def shapes(self):
        d = self.desc.contents
        return [d.shapes[i] for i in range(d.n_shapes)]

    def close(self):
        if self._handle:
            abi.libhost().mirt_host_world_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """A scene resident on one MI355X (mirt_scene)."""

    def __init__(self, desc, device=-1):
        self._lib = abi.libmirt()
        self._handle = C.c_void_p()
        if isinstance(desc, World):
            self._keep = desc
            desc = desc.desc
        _check(self._lib.mirt_scene_create(desc, device, C.byref(self._handle)))

    @property
    def handle(self):
        return self._handle

    def trim(self):
        """Frees the per-pass buffers and the accumulation image render calls keep with the scene (mirt_scene_trim)."""
        _check(self._lib.mirt_scene_trim(self._handle))

    def close(self):
        if self._handle:
            self._lib.mirt_scene_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # Hittable::hit (src/bvh.rs:83-86 / src/hittable.rs:698-710)
    def hit(self, rays, t_min=0.001, t_max=float("inf"), use_bvh=True, traversal=abi.TRAVERSE_FAST):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        _check(self._lib.mirt_intersect(self._handle, rays.ctypes.data_as(C.POINTER(abi.Ray)), rays.shape[0], t_min,
                                        t_max, int(bool(use_bvh)), traversal,
                                        hits.ctypes.data_as(C.POINTER(abi.Hit))))
        return hits


def make_opts(use_bvh=True, traversal=abi.TRAVERSE_FAST, pixel_range=(0, 0), sample_range=(0, 0), paths_per_pass=0,
              profile=False, device=-1, tiles=(0, 1, 0), handover=(0, 0)):
    o = abi.RenderOpts()
    o.struct_size = C.sizeof(abi.RenderOpts)
    o.device = device
    o.use_bvh = int(bool(use_bvh))
    o.traversal = traversal
    o.pixel_begin, o.pixel_end = pixel_range
    o.sample_begin, o.sample_end = sample_range
    o.paths_per_pass = paths_per_pass
    o.profile = int(bool(profile))
    o.tile_size, o.tile_count, o.tile_index = tiles
    o.handover_bounce, o.handover_paths = handover
    return o


def stats_dict(s):
    return {name: getattr(s, name) for name, _ in abi.Stats._fields_}


def render_world(background, world, camera, screen_width, screen_height, antialias_iterations, render_parallel=True,
                 use_bounding_volume=True, seed=0x5EED0002, max_depth=MAX_DEPTH, traversal=abi.TRAVERSE_FAST,
                 pixel_range=(0, 0), paths_per_pass=0, profile=False, out=None, return_stats=False, tiles=(0, 1, 0),
                 handover=(0, 0)):
    """render_world (src/renderer.rs:9-62): (H, W, 3) uint8, top row first.

    `world` is a Scene (already on the device) or a World (uploaded for this call).
    `render_parallel` selects rayon vs a serial loop in the reference and has no
    meaning on the device.  `tiles` = (tile_size, tile_count, tile_index) renders only
    this device's share of a tile split; the other pixels of `out` are left untouched.
    `handover` = (handover_bounce, handover_paths) of mirt_render_opts: scheduling only, never results.
    """
    del render_parallel
    own = None
    scene = world
    if isinstance(world, World):
        own = scene = Scene(world)
    try:
        if out is None:
            out = np.zeros((screen_height, screen_width, 3), dtype=np.uint8)
        assert out.dtype == np.uint8 and out.size == screen_width * screen_height * 3 and out.flags.c_contiguous
        opts = make_opts(use_bounding_volume, traversal, pixel_range, (0, 0), paths_per_pass, profile, tiles=tiles, handover=handover)
        stats = abi.Stats()
        _check(abi.libmirt().mirt_render(scene.handle, C.byref(camera.c), _vec3(background), screen_width,
                                         screen_height, antialias_iterations, max_depth, seed, C.byref(opts),
                                         out.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(stats)))
    finally:
        if own is not None:
            own.close()
    return (out, stats_dict(stats)) if return_stats else out


def render_world_multi(background, scenes, camera, screen_width, screen_height, antialias_iterations, split="tile",
                       reduce="peer", tile_size=0, seed=0x5EED0002, max_depth=MAX_DEPTH, paths_per_pass=0, out=None,
                       return_stats=False, handover=(0, 0)):
    """render_world over several devices from this process (mirt_render_multi): `scenes` = one Scene per
    rank, each on the device it was created on (entries may repeat or share a device)."""
    n = len(scenes)
    if out is None:
        out = np.zeros((screen_height, screen_width, 3), dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size == screen_width * screen_height * 3 and out.flags.c_contiguous
    handles = (C.c_void_p * n)(*[s.handle for s in scenes])
    opts = make_opts(paths_per_pass=paths_per_pass, handover=handover)
    multi = abi.MultiOpts(C.sizeof(abi.MultiOpts), {"tile": abi.SPLIT_TILE, "sample": abi.SPLIT_SAMPLE}[split],
                          {"peer": abi.REDUCE_PEER, "rccl": abi.REDUCE_RCCL}[reduce], tile_size)
    stats = (abi.Stats * n)()
    _check(abi.libmirt().mirt_render_multi(handles, n, C.byref(camera.c), _vec3(background), screen_width, screen_height,
                                           antialias_iterations, max_depth, seed, C.byref(opts), C.byref(multi),
                                           out.ctypes.data_as(C.POINTER(C.c_uint8)), stats))
    return (out, [stats_dict(s) for s in stats]) if return_stats else out


def render_accumulate(scene, camera, background, width, height, spp, d_accum_ptr, seed=0x5EED0002, max_depth=MAX_DEPTH,
                      use_bvh=True, traversal=abi.TRAVERSE_FAST, pixel_range=(0, 0), sample_range=(0, 0),
                      paths_per_pass=0, profile=False, tiles=(0, 1, 0), handover=(0, 0), static=False):
    """Adds the radiance of a sample range into a DEVICE f64 image (address `d_accum_ptr`).
    static=True goes through mirt_render_accumulate_static, the entry point that allocates nothing."""
    opts = make_opts(use_bvh, traversal, pixel_range, sample_range, paths_per_pass, profile, tiles=tiles, handover=handover)
    stats = abi.Stats()
    entry = abi.libmirt().mirt_render_accumulate_static if static else abi.libmirt().mirt_render_accumulate
    _check(entry(scene.handle, C.byref(camera.c), _vec3(background), width, height, spp,
                                                max_depth, seed, C.byref(opts), C.c_void_p(d_accum_ptr),
                                                C.byref(stats)))
    return stats_dict(stats)


def resolve(d_accum_ptr, width, height, spp, pixel_range=(0, 0), device=-1, out=None):
    """renderer.rs:85-89 + the u8 cast, from a DEVICE f64 image into a host (H, W, 3) uint8 array."""
    if out is None:
        out = np.zeros((height, width, 3), dtype=np.uint8)
    _check(abi.libmirt().mirt_resolve(C.c_void_p(d_accum_ptr), width, height, spp, pixel_range[0], pixel_range[1],
                                      device, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def camera_rays(camera, width, height, sample=0, seed=0x5EED0002, pixel_range=(0, 0), device=-1):
    """The camera rays a render with this seed starts from (sample `sample` of each pixel)."""
    pb, pe = pixel_range
    if pe == 0:
        pe = width * height
    rays = np.zeros(pe - pb, dtype=RAY_DTYPE)
    _check(abi.libmirt().mirt_camera_rays(C.byref(camera.c), width, height, pb, pe, sample, seed, device,
                                          rays.ctypes.data_as(C.POINTER(abi.Ray))))
    return rays
