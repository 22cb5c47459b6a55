"""Two corners of the render path.

1. Rays with a zero direction component lying in a rect's plane get an order-dependent answer from
   the reference ((k - o) / d = 0 / 0, hittable.rs:189-193); mirt_intersect reproduces it through the
   reference's own recursion, the path kernel does not carry that hook.  The scene most likely to
   make such rays -- axis-aligned rects only, a pinhole camera looking straight down an axis at a
   perfect mirror -- renders the oracle's bytes, and none of the rays the oracle traces for it has a
   zero direction component: jittered camera rays and scattered rays never do.
2. Checker textures nest to any depth in the reference (texture.rs:45-54, 154-157): three levels
   on the device against the oracle."""
import numpy as np
import pytest

import scene_util as su
from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api

pytestmark = pytest.mark.gpu


def _mirror_room():
    textures = [su.constant(0.73, 0.73, 0.73), su.constant(0.65, 0.05, 0.05), su.constant(4.0, 4.0, 4.0)]
    materials = [su.material(abi.MAT_LAMBERTIAN, 0), su.material(abi.MAT_LAMBERTIAN, 1), su.material(abi.MAT_DIFFUSE_LIGHT, 2),
                 su.material(abi.MAT_METAL, albedo=(0.9, 0.9, 0.9), param=0.0)]
    shapes = [
        su.shape(abi.SHAPE_XZ_RECT, 0, -4.0, 4.0, -4.0, 8.0, 0.0),    # floor
        su.shape(abi.SHAPE_XZ_RECT, 0, -4.0, 4.0, -4.0, 8.0, 4.0),    # ceiling
        su.shape(abi.SHAPE_XZ_RECT, 2, -1.0, 1.0, 1.0, 3.0, 3.999),   # light
        su.shape(abi.SHAPE_YZ_RECT, 1, 0.0, 4.0, -4.0, 8.0, -4.0),    # red wall
        su.shape(abi.SHAPE_YZ_RECT, 0, 0.0, 4.0, -4.0, 8.0, 4.0),
        su.shape(abi.SHAPE_XY_RECT, 3, -3.0, 3.0, 0.5, 3.5, -2.0),    # the mirror, facing the camera
        su.shape(abi.SHAPE_XY_RECT, 0, -4.0, 4.0, 0.0, 4.0, -4.0),    # wall behind it
        su.shape(abi.SHAPE_XY_RECT, 0, -4.0, 4.0, 0.0, 4.0, 8.0),     # wall behind the camera
        su.shape(abi.SHAPE_CUBE, 1, -0.5, 0.0, 0.0, 0.5, 1.0, 1.0),   # an axis-aligned box in front of the mirror
    ]
    return su.Built(shapes, materials, textures, seed=3)


def test_axis_aligned_mirror_room_renders_the_oracle_bytes(oracle):
    built = _mirror_room()
    w, h, spp = 65, 49, 6  # odd sizes: a pixel centre sits exactly on the optical axis
    cam = api.Camera.new((0.0, 2.0, 6.0), (0.0, 2.0, 0.0), (0.0, 1.0, 0.0), 50.0, w / h, 0.0, 10.0, 0.0, 1.0)
    ref, _, ctr, _ = oracle.render(built.desc, cam, api.BLACK, w, h, spp)
    gpu, st = api.render_world(api.BLACK, api.Scene(built.desc), cam, w, h, spp, return_stats=True)
    assert st["segments"] == ctr["segments"]
    diff = np.abs(gpu.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 0.002 * diff.size
    assert gpu.max() > 30  # lit
    rays, hits = oracle.trace_rays(built.desc, cam, api.BLACK, w, h, np.arange(w * h, dtype=np.uint32), (0, spp))
    assert rays.shape[0] == ctr["segments"]
    mirror = np.flatnonzero([s.material == 3 for s in built.shapes])
    assert np.count_nonzero(np.isin(hits["prim"], mirror)) > 1000  # the mirror is in play
    assert np.count_nonzero(rays["direction"] == 0.0) == 0  # and no ray has a zero direction component
    # the same rays through the closest-hit entry point (which does carry the hook): the oracle's hits
    scene = api.Scene(built.desc)
    try:
        got = scene.hit(rays)
        assert np.array_equal(got["prim"], hits["prim"]) and np.array_equal(got["t"][hits["prim"] >= 0], hits["t"][hits["prim"] >= 0])
    finally:
        scene.close()


def test_three_levels_of_checker_textures(oracle):
    c = [su.constant(0.9, 0.1, 0.1), su.constant(0.1, 0.9, 0.1), su.constant(0.1, 0.1, 0.9), su.constant(0.9, 0.9, 0.9)]
    textures = c + [su.checker(0, 1),    # 4: level 1
                    su.checker(4, 2),    # 5: level 2, one child a checker
                    su.checker(3, 5),    # 6: level 3
                    su.checker(6, 4)]    # 7: both children checkers
    materials = [su.material(abi.MAT_LAMBERTIAN, 6), su.material(abi.MAT_LAMBERTIAN, 7), su.material(abi.MAT_DIFFUSE_LIGHT, 5)]
    shapes = [su.sphere(0.0, -1000.0, 0.0, 1000.0, 0), su.sphere(0.0, 1.0, 0.0, 1.0, 1), su.sphere(-2.2, 0.7, 0.5, 0.7, 0),
              su.shape(abi.SHAPE_XZ_RECT, 2, -2.0, 2.0, -2.0, 2.0, 4.0), su.shape(abi.SHAPE_CUBE, 1, 1.5, 0.0, -0.5, 2.5, 1.0, 0.5)]
    built = su.Built(shapes, materials, textures, seed=5)
    w, h, spp = 96, 54, 8
    cam = api.Camera.new((6.0, 2.5, 4.0), (0.0, 0.8, 0.0), (0.0, 1.0, 0.0), 35.0, w / h, 0.0, 10.0, 0.0, 1.0)
    ref, _, ctr, _ = oracle.render(built.desc, cam, api.SKY, w, h, spp)
    scene = api.Scene(built.desc)
    try:
        for handover in ((0, 0), (abi.HANDOVER_NEVER, 0)):  # the path kernel and the per-bounce shade kernel
            gpu, st = api.render_world(api.SKY, scene, cam, w, h, spp, return_stats=True, handover=handover)
            assert st["segments"] == ctr["segments"]
            diff = np.abs(gpu.astype(int) - ref.astype(int))
            assert diff.max() <= 1 and np.count_nonzero(diff) <= 0.002 * diff.size, (handover, diff.max(), np.count_nonzero(diff))
    finally:
        scene.close()
    # the texture really has structure at all three levels: each of the four colours shows up on the ground
    assert len(np.unique(ref.reshape(-1, 3), axis=0)) > 50
