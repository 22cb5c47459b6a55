"""Worlds nobody would build (degenerate_worlds.py): radii of zero, 1e-20, 1e6 and below zero, cubes with unordered or
coinciding corners, rects of no area, repeated shapes, shutter spans that are empty, reversed or 1e-300 long, scales
from 1e-30 to 1e30, and reference boxes that were shrunk, scrambled or blown up.  Whatever the reference's arithmetic
makes of them (hittable.rs:105-137, 185-207, 432-445, 487-561; bvh.rs:37-80) is the answer: all three traversals
against the oracle on ordinary and wild rays, three (t_min, t_max) ranges, and a small render through both pipelines.
tools/world_soak.py runs the same over thousands of such worlds."""
import numpy as np
import pytest

import degenerate_worlds as dw
from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api
from test_gpu_intersect import _compare, _rays_nobody_would_send

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", list(range(21)) + [1000])
def test_degenerate_worlds(seed, oracle):
    big = seed == 1000  # one world of thousands of shapes (the tree then lives elsewhere than in LDS as a whole)
    built, scale, boxes, (t_lo, t_hi), rng = dw.degenerate_world(seed, big)
    n = 6000
    rays = dw.rays_for(built, scale, n, seed, rng, _rays_nobody_would_send)
    scene = api.Scene(built.desc)
    try:
        for t_min, t_max in ((0.001, float("inf")), (-5.0 * scale, 40.0 * scale), (0.001 * scale, 1e30)):
            ref, _ = oracle.intersect(built.desc, rays, t_min=t_min, t_max=t_max)
            ref_list, _ = oracle.intersect(built.desc, rays, t_min=t_min, t_max=t_max, use_bvh=False)
            what = f"seed {seed}, scale {scale:g}, boxes {boxes}, range ({t_min:g}, {t_max:g})"
            _compare(scene.hit(rays, t_min, t_max), ref, what + ", fast")
            _compare(scene.hit(rays, t_min, t_max, traversal=abi.TRAVERSE_REFERENCE), ref, what + ", reference-order")
            _compare(scene.hit(rays, t_min, t_max, use_bvh=False), ref_list, what + ", list")
        cam = api.Camera.new(tuple(np.array([3.0, 4.0, 13.0]) * scale), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 10.0, t_lo, t_hi)
        ref_img, _, ctr, _ = oracle.render(built.desc, cam, api.SKY, 48, 32, 2)
        for handover in ((0, 0), (abi.HANDOVER_NEVER, 0)):
            img, st = api.render_world(api.SKY, scene, cam, 48, 32, 2, return_stats=True, handover=handover)
            assert st["segments"] == ctr["segments"], (seed, handover)
            assert np.abs(img.astype(int) - ref_img.astype(int)).max() <= 1, (seed, handover)
    finally:
        scene.close()
