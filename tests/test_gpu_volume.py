"""Closest-hit parity at volume: at least 10^7 primary + secondary rays per benchmark scene --
every ray the oracle hands to world.hit while rendering a 640x360 frame of the scene at a few
samples per pixel -- through the fast and the reference-order traversal: hit / miss, primitive,
cube side and t bit for bit.  (The f32 culling margins of the fast traversal are the kind of
thing that fails once in 10^8 rays, not once in 10^4.)  And whole frames at the benchmark's own
sizes, one sample per pixel: the path kernel against the launch-per-bounce pipeline, byte for
byte and query for query; and frames of tens of millions of paths against the oracle's image."""
import numpy as np
import pytest

from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api, configs
from test_gpu_intersect import _compare

pytestmark = pytest.mark.gpu

W, H = 640, 360
# samples per pixel that give > 10^7 rays (2.6 / 2.1 / 6.4 / 2.4 closest-hit queries per sample)
SAMPLES = {"C1": 18, "C3": 22, "C4": 12, "C5": 19}
N_RAYS = 10_000_000


@pytest.mark.parametrize("name", ["C1", "C3", "C4", "C5"])
def test_ten_million_rays_per_scene(name, worlds, oracle):
    cfg = configs.ALL[name]
    world = worlds(name)
    cam = cfg.camera(W, H)
    pixels = np.arange(W * H, dtype=np.uint32)
    rays, ref = oracle.trace_rays(world.desc, cam, cfg.background, W, H, pixels, (0, SAMPLES[name]), cap=N_RAYS + (1 << 21))
    assert rays.shape[0] >= N_RAYS, f"{name}: only {rays.shape[0]} rays"
    assert np.count_nonzero(ref["prim"] >= 0) > N_RAYS // 4
    scene = api.Scene(world)
    try:
        for trav, label in ((abi.TRAVERSE_FAST, "fast"), (abi.TRAVERSE_REFERENCE, "reference-order")):
            gpu = scene.hit(rays, use_bvh=True, traversal=trav)
            _compare(gpu, ref, f"{name} {label}, {rays.shape[0]} rays")
    finally:
        scene.close()


@pytest.mark.parametrize("name", ["C3", "C4", "C5"])
def test_full_size_frames_path_kernel_against_pipeline(name, worlds):
    cfg = configs.ALL[name]
    w, h = cfg.width, cfg.height
    assert (w, h) == {"C3": (1920, 1080), "C4": (1024, 1024), "C5": (3840, 2160)}[name]
    world = worlds(name)
    cam = cfg.camera()
    scene = api.Scene(world)
    try:
        paths, st = api.render_world(cfg.background, scene, cam, w, h, 1, return_stats=True)
        assert st["samples"] == w * h and st["segments_paths"] == st["segments"] >= st["samples"]
        queues, st_q = api.render_world(cfg.background, scene, cam, w, h, 1, handover=(abi.HANDOVER_NEVER, 0), return_stats=True)
        assert st_q["launches_paths"] == 0 and st_q["segments"] == st["segments"]
        assert np.array_equal(paths, queues)
        assert paths.std() > 5  # an image
    finally:
        scene.close()


# frames of (nearly) the benchmark's own sizes, a few samples per pixel: 12 to 34 million paths per scene
RENDER_FRAMES = {"C2": (1200, 800, 16), "C3": (1920, 1080, 8), "C4": (1024, 1024, 32), "C5": (1920, 1080, 6)}


@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5"])
def test_tens_of_millions_of_paths_render_the_oracle_bytes(name, worlds, oracle):
    """The path kernel against the oracle with the same sampler: as many closest-hit queries, and
    the image byte for byte (renderer.rs:64-116 end to end; the oracle runs on all host cores)."""
    cfg = configs.ALL[name]
    w, h, spp = RENDER_FRAMES[name]
    world = worlds(name)
    cam = cfg.camera(w, h)
    ref, _, ctr, _ = oracle.render(world.desc, cam, cfg.background, w, h, spp)
    gpu, st = api.render_world(cfg.background, world, cam, w, h, spp, return_stats=True)
    assert st["samples"] == w * h * spp >= 12_000_000
    assert st["segments"] == ctr["segments"], f"{name}: {st['segments']} queries, the oracle made {ctr['segments']}"
    diff = np.abs(gpu.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 1e-5 * diff.size, f"{name}: {np.count_nonzero(diff)} bytes differ, by up to {diff.max()}"


@pytest.mark.parametrize("n", [1_500, 3_000, 70_000])
def test_scenes_too_large_for_lds(n, oracle):
    """1 500 shapes: the nodes fit LDS, the records do not (a third way of reading the tree).
    3 000 shapes: tree and records are read from global memory (16-bit stack entries, plain textures: a
    kernel instance no benchmark scene selects).  70 000 shapes: more than 65 535 tree nodes, so the
    traversal kernels keep 32-bit stack entries (again other instances), and the reference's own tree is 17
    levels deep.  Closest hits on 300 000 rays through both traversals, and a small render, against the oracle."""
    import scene_util as su

    rng = np.random.default_rng(77)
    centres = rng.uniform(-60.0, 60.0, (n, 3))
    sizes = rng.uniform(0.2, 1.2, n)
    shapes = []
    for k in range(n):
        c, r = centres[k], sizes[k]
        kind = k % 4
        if kind == 0:
            shapes.append(su.shape(abi.SHAPE_CUBE, k % 3, *(c - r), *(c + r)))
        elif kind == 1:
            shapes.append(su.shape(abi.SHAPE_XY_RECT + k % 3, k % 3, c[0], c[0] + 2 * r, c[1], c[1] + 2 * r, c[2]))
        elif kind == 2 and k % 8 == 2:
            shapes.append(su.moving_sphere(c, c + rng.uniform(-0.5, 0.5, 3), r, 0.0, 1.0, k % 3))
        else:
            shapes.append(su.sphere(*c, r, k % 3))
    textures = [su.constant(0.7, 0.3, 0.3), su.constant(0.9, 0.9, 0.9)]
    materials = [su.material(abi.MAT_LAMBERTIAN, 0), su.material(abi.MAT_METAL, albedo=(0.8, 0.8, 0.9), param=0.1),
                 su.material(abi.MAT_DIELECTRIC, param=1.5)]
    built = su.Built(shapes, materials, textures, seed=11)
    assert built.desc.n_nodes == 2 * n - 1 and (n < 10_000 or built.desc.n_nodes > 65535)
    m = 300_000
    rays = np.zeros(m, dtype=api.RAY_DTYPE)
    rays["origin"] = rng.uniform(-80.0, 80.0, (m, 3))
    rays["direction"] = (rng.uniform(-50.0, 50.0, (m, 3)) - rays["origin"]) * rng.uniform(0.2, 2.0, (m, 1))
    rays["time"] = rng.uniform(0.0, 1.0, m)
    ref, _ = oracle.intersect(built.desc, rays)
    assert np.count_nonzero(ref["prim"] >= 0) > m // 8
    scene = api.Scene(built.desc)
    try:
        _compare(scene.hit(rays), ref, f"{n} shapes, fast")
        _compare(scene.hit(rays, traversal=abi.TRAVERSE_REFERENCE), ref, f"{n} shapes, reference-order")
        w, h, spp = 96, 54, 4
        cam = api.Camera.new((0.0, 5.0, 110.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, w / h, 0.0, 10.0, 0.0, 1.0)
        want, _, ctr, _ = oracle.render(built.desc, cam, api.SKY, w, h, spp)
        got, st = api.render_world(api.SKY, scene, cam, w, h, spp, return_stats=True)
        assert st["segments"] == ctr["segments"]
        diff = np.abs(got.astype(int) - want.astype(int))
        assert diff.max() <= 1 and np.count_nonzero(diff) <= 0.002 * diff.size
    finally:
        scene.close()
