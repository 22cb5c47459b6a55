"""Worlds nobody would build, for tests/test_gpu_degenerate.py and tools/world_soak.py: spheres of radius zero, tiny,
huge or negative, cubes with unordered or coinciding corners, rects of no area or with their ends swapped, the same
shape several times, moving spheres whose shutter span is empty, reversed or not a power of two, everything scaled by
10^-30 .. 10^30 -- and, in some, the reference's boxes shrunk, scrambled or blown up (the answer is then what the
reference's recursion makes of THOSE boxes, bvh.rs:61-66)."""
import numpy as np

import scene_util as su
from ray_tracer_amd import _abi as abi


def degenerate_world(seed, big=False):
    """-> (Built, scale, name of what was done to the boxes, (shutter open, shutter close))"""
    rng = np.random.default_rng(4242 + seed)
    scale = 10.0 ** rng.choice([0, 0, 0, -30, -8, 5, 12, 30])
    count = int(rng.choice([900, 1500, 2500, 3600, 3800, 6000, 20000, 70000] if big else [1, 2, 3, 7, 20, 60, 400]))
    shapes = []
    for k in range(count):
        c = rng.uniform(-6.0, 6.0, 3)
        if rng.random() < 0.2:
            c = np.round(c)  # coinciding centres, faces and planes
        kind = int(rng.integers(0, 6))
        if kind == 0:
            r = rng.choice([0.0, 1e-20, 1e-3, 0.5, 1.0, 3.0, 1e6, -0.7, -1e-9])
            shapes.append(su.sphere(*(c * scale), r * scale))
        elif kind == 1:
            t_a, t_b = rng.choice([(0.0, 1.0), (0.0, 0.0), (1.0, 0.0), (0.25, 0.55), (-3.0, 5.0), (0.0, 1e-300)])
            shapes.append(su.moving_sphere(c * scale, (c + rng.uniform(-2.0, 2.0, 3)) * scale, rng.choice([0.4, 1.0, 0.0, -0.5]) * scale, t_a, t_b))
        elif kind == 2:
            e = rng.uniform(-1.0, 2.5, 3) * rng.choice([1.0, 1.0, 0.0], 3)  # unordered corners, flat cubes
            shapes.append(su.shape(abi.SHAPE_CUBE, 0, *(c * scale), *((c + e) * scale)))
        else:
            a = rng.uniform(-1.0, 5.0, 2) * rng.choice([1.0, 1.0, 0.0], 2)  # swapped ends, no area
            shapes.append(su.shape(abi.SHAPE_XY_RECT + kind - 3, 0, c[0] * scale, (c[0] + a[0]) * scale, c[1] * scale, (c[1] + a[1]) * scale, c[2] * scale))
    if rng.random() < 0.3 and shapes:
        shapes += [shapes[int(rng.integers(0, len(shapes)))] for _ in range(3)]  # the same shape several times
    t_lo, t_hi = rng.choice([(0.0, 1.0), (0.0, 0.0), (0.3, 0.4)])
    built = su.Built(shapes, seed=seed, t0=float(t_lo), t1=float(t_hi))
    boxes = ("as built", "as built", "shrunk", "as built", "scrambled", "as built", "inflated")[seed % 7]
    for i in range(built.desc.n_nodes):
        m = built.nodes[i]
        if m.is_leaf or boxes == "as built":
            continue
        for a in range(3):
            if boxes == "shrunk":  # towards a point: they still nest but cut into their shapes
                m.bmin[a] = scale + (m.bmin[a] - scale) * 0.8
                m.bmax[a] = scale + (m.bmax[a] - scale) * 0.8
            elif boxes == "scrambled":  # each bound moved at random: they no longer nest
                m.bmin[a] += rng.uniform(-1.0, 2.0) * scale
                m.bmax[a] += rng.uniform(-2.0, 1.0) * scale
            else:
                m.bmin[a] -= rng.uniform(0.0, 3.0) * scale
                m.bmax[a] += rng.uniform(0.0, 3.0) * scale
    return built, scale, boxes, (float(t_lo), float(t_hi)), rng


def rays_for(built, scale, n, seed, rng, wild_rays):
    """Half of them rays nobody would send (wild_rays(n, seed, shapes)), half in the world's own scale, aimed at it."""
    rays = wild_rays(n, 99 + seed, built.shapes)
    half = n // 2
    rays["origin"][:half] = rng.uniform(-14.0, 14.0, (half, 3)) * scale
    target = rng.uniform(-6.0, 6.0, (half, 3)) * scale
    rays["direction"][:half] = (target - rays["origin"][:half]) * (10.0 ** rng.uniform(-3, 3, (half, 1)))
    rays["time"][:half] = rng.uniform(-0.2, 1.2, half)
    return rays
