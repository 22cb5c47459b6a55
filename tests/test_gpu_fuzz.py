"""Random worlds: every shape kind with every material and every texture kind, in combinations none
of the reference's scenes holds (a glass cube, a mirror rect, a light on a moving sphere, a noise
texture inside a checker, an image texture on a cube, fuzz 0 and fuzz > 1, refraction indices below
and above 1, hollow spheres), seen through cameras with and without an aperture and a shutter.

Per world: the image of the path kernel and of the per-bounce pipeline against the oracle's
(equal segment counts, bytes within one level), and every ray the oracle traced for it through
both tree traversals (hittable.rs:396-561, material.rs:51-243, texture.rs:45-122)."""
import ctypes as C

import numpy as np
import pytest

import scene_util as su
from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api

pytestmark = pytest.mark.gpu


def _perlin(seed):
    p = abi.Perlin()
    assert abi.libhost().mirt_host_perlin_new(seed, C.byref(p)) == 0
    return p


def _random_world(seed, n_shapes=(24, 90), planar=False):
    rng = np.random.default_rng(1000 + seed + 7919 * n_shapes[0] + (104729 if planar else 0))
    image = rng.integers(0, 256, (9, 14, 3), dtype=np.uint8)
    n_const = 6
    textures = [su.constant(*rng.uniform(0.05, 0.95, 3)) for _ in range(n_const)]
    textures.append(su.constant(*rng.uniform(2.0, 9.0, 3)))                 # 6: bright, for lights
    textures.append(su.noise(0, rng.uniform(0.5, 6.0)))                     # 7
    textures.append(su.image_texture(0))                                    # 8
    textures.append(su.checker(int(rng.integers(0, n_const)), 7))           # 9: a noise texture inside a checker
    textures.append(su.checker(8, 9))                                       # 10: image | checker
    textures.append(su.checker(int(rng.integers(0, n_const)), int(rng.integers(0, n_const))))  # 11
    materials = []
    for t in range(len(textures)):
        materials.append(su.material(abi.MAT_LAMBERTIAN, t))
    for fuzz in (0.0, 0.25, 1.0, 1.7):
        materials.append(su.material(abi.MAT_METAL, albedo=rng.uniform(0.3, 1.0, 3), param=fuzz))
    for index in (1.5, 1.0, 0.6, 2.4):
        materials.append(su.material(abi.MAT_DIELECTRIC, param=index))
    for t in (6, 9, 10):
        materials.append(su.material(abi.MAT_DIFFUSE_LIGHT, t))
    shapes = []
    n = int(rng.integers(n_shapes[0], n_shapes[1]))
    for k in range(n):
        c = rng.uniform(-6.0, 6.0, 3)
        m = int(rng.integers(0, len(materials)))
        kind = int(rng.integers(4, 7)) if planar else int(rng.integers(0, 7))
        if kind == 0:
            shapes.append(su.moving_sphere(c, c + rng.uniform(-0.8, 0.8, 3), rng.uniform(0.3, 1.4), 0.0, 1.0, m))
        elif kind == 1:
            shapes.append(su.sphere(*c, -rng.uniform(0.3, 1.2), m))        # hollow
        elif kind in (2, 3):
            shapes.append(su.sphere(*c, rng.uniform(0.2, 1.6), m))
        elif kind == 4:
            shapes.append(su.shape(abi.SHAPE_CUBE, m, *c, *(c + rng.uniform(0.4, 2.5, 3))))
        else:
            a = rng.uniform(0.8, 5.0, 2)
            shapes.append(su.shape(abi.SHAPE_XY_RECT + int(rng.integers(0, 3)), m, c[0], c[0] + a[0], c[1], c[1] + a[1], c[2]))
    if planar:
        shapes.append(su.shape(abi.SHAPE_XZ_RECT, int(rng.integers(0, len(textures))), -40.0, 40.0, -40.0, 40.0, -6.5))  # ground
    else:
        shapes.append(su.sphere(0.0, -1006.5, 0.0, 1000.0, int(rng.integers(0, len(textures)))))  # ground
    built = su.Built(shapes, materials, textures, perlins=[_perlin(77 + seed)], images=[image], seed=seed)
    eye = rng.uniform(-1.0, 1.0, 3)
    eye = eye / np.linalg.norm(eye) * rng.uniform(9.0, 16.0)
    eye[1] = abs(eye[1])
    aperture = (0.0, 0.3)[seed % 2]
    shutter = ((0.0, 1.0), (0.0, 0.0), (0.25, 0.5))[seed % 3]
    return built, eye, aperture, shutter


@pytest.mark.parametrize("seed,n_shapes,planar", [(s, (24, 90), False) for s in range(10)] + [(s, (1 + s % 5, 16), False) for s in range(8)] +
                         [(s, (1 + s % 4, 11), True) for s in range(10)] +
                         # thousands of shapes: the path kernel walks such a tree from LDS in its 28-byte form (bounds as halves)
                         [(s, (2200, 3400), False) for s in range(2)])
def test_random_worlds_render_and_intersect_like_the_oracle(seed, n_shapes, planar, oracle):
    """(planar: rects and cubes only -- such a world, when small, is rendered without a tree, cube faces as items of their own)"""
    built, eye, aperture, shutter = _random_world(seed, n_shapes, planar)
    w, h, spp = 72, 48, 4
    cam = api.Camera.new(tuple(eye), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, w / h, aperture, float(np.linalg.norm(eye)), *shutter)
    background = (api.SKY, api.BLACK)[seed % 2] if n_shapes[0] >= 24 else api.SKY
    ref, _, ctr, _ = oracle.render(built.desc, cam, background, w, h, spp)
    scene = api.Scene(built.desc)
    try:
        for handover in ((0, 0), (abi.HANDOVER_NEVER, 0), (2, 0)):  # path kernel, per-bounce pipeline, pipeline then kernel
            gpu, st = api.render_world(background, scene, cam, w, h, spp, return_stats=True, handover=handover)
            assert st["segments"] == ctr["segments"], (seed, handover)
            diff = np.abs(gpu.astype(int) - ref.astype(int))
            assert diff.max() <= 1 and np.count_nonzero(diff) <= 0.002 * diff.size, (seed, handover, diff.max())
        assert ref.max() > 20 and len(np.unique(ref)) > 50  # a picture, not a black frame
        rays, hits = oracle.trace_rays(built.desc, cam, background, w, h, np.arange(w * h, dtype=np.uint32), (0, spp))
        assert rays.shape[0] == ctr["segments"]
        for trav in (abi.TRAVERSE_FAST, abi.TRAVERSE_REFERENCE):
            got = scene.hit(rays, traversal=trav)
            assert np.array_equal(got["prim"], hits["prim"]), (seed, trav)
            assert np.array_equal(got["side"], hits["side"]), (seed, trav)
            found = hits["prim"] >= 0
            assert np.array_equal(got["t"][found], hits["t"][found]), (seed, trav)
        # the world uses what it was made to use
        used = {built.materials[built.shapes[p].material].kind for p in np.unique(hits["prim"][hits["prim"] >= 0])}
        assert len(used) >= (3 if n_shapes[0] >= 24 else 1)
    finally:
        scene.close()
