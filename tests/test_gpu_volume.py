"""Closest-hit parity at volume: at least 10^7 primary + secondary rays per benchmark scene --
every ray the oracle hands to world.hit while rendering a 640x360 frame of the scene at a few
samples per pixel -- through the fast and the reference-order traversal: hit / miss, primitive,
cube side and t bit for bit.  (The f32 culling margins of the fast traversal are the kind of
thing that fails once in 10^8 rays, not once in 10^4.)  And whole frames at the benchmark's own
sizes, one sample per pixel: the path kernel against the launch-per-bounce pipeline, byte for
byte and query for query."""
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
