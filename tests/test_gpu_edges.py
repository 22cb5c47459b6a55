"""The ends of the input range: frames of one pixel, one sample, one bounce; no rays at all; passes that do
not divide the frame; arguments the boundary must refuse.  The reference has no tests of its own
(.github/workflows/rust.yml runs an empty `cargo test`), so the cases follow its code: the depth budget of
renderer.rs:83 and 99-101, the pixel order of renderer.rs:74-75, Hittable::hit on an empty batch."""
import ctypes as C

import numpy as np
import pytest

from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api, configs

pytestmark = pytest.mark.gpu


def _same(gpu, ref):
    diff = np.abs(gpu.astype(int) - ref.astype(int))
    return diff.max() <= 1 and np.count_nonzero(diff) <= max(1, 0.002 * diff.size)


@pytest.mark.parametrize("frame", [(1, 1, 1), (1, 1, 37), (1, 7, 3), (7, 1, 3), (3, 2, 1), (65, 3, 2), (63, 5, 5), (129, 1, 1)])
def test_tiny_and_ragged_frames(frame, worlds, oracle):
    """Frames smaller than a wave, of one row, of one column, of one pixel; widths around the wave size."""
    w, h, spp = frame
    cfg = configs.C1
    world = worlds("C1")
    cam = cfg.camera(w, h)
    ref, _, ctr, _ = oracle.render(world.desc, cam, cfg.background, w, h, spp)
    for handover in ((0, 0), (abi.HANDOVER_NEVER, 0)):  # the path kernel, and a launch per bounce
        gpu, st = api.render_world(cfg.background, world, cam, w, h, spp, return_stats=True, handover=handover)
        assert st["samples"] == w * h * spp and st["segments"] == ctr["segments"], (frame, handover)
        assert _same(gpu, ref), (frame, handover)


@pytest.mark.parametrize("name", ["C1", "C4"])
@pytest.mark.parametrize("depth", [1, 2, 3, 7])
def test_depth_budget(name, depth, worlds, oracle):
    """color(.., depth) with a budget of `depth` closest-hit queries per path (renderer.rs:83 fixes 50): what
    would be scattered at the last of them is multiplied by color(.., 0) = 0 (renderer.rs:99-101), a light met
    there still emits."""
    cfg = configs.ALL[name]
    w, h, spp = 48, 32, 6
    world = worlds(name)
    cam = cfg.camera(w, h)
    ref, _, ctr, _ = oracle.render(world.desc, cam, cfg.background, w, h, spp, max_depth=depth)
    assert ctr["segments"] <= w * h * spp * depth
    for handover in ((0, 0), (abi.HANDOVER_NEVER, 0), (1, 0)):
        gpu, st = api.render_world(cfg.background, world, cam, w, h, spp, max_depth=depth, return_stats=True, handover=handover)
        assert st["segments"] == ctr["segments"], (name, depth, handover)
        assert _same(gpu, ref), (name, depth, handover)


def test_passes_that_do_not_divide_the_frame(worlds, oracle):
    """A pass holds whole samples of every pixel; pass sizes that leave a ragged last pass, of one sample per
    pass, and smaller than one sample of the frame (refused: a pass cannot hold less) render the same bytes."""
    cfg = configs.C1
    w, h, spp = 50, 30, 7
    world = worlds("C1")
    cam = cfg.camera(w, h)
    ref, _, ctr, _ = oracle.render(world.desc, cam, cfg.background, w, h, spp)
    scene = api.Scene(world)
    try:
        whole = api.render_world(cfg.background, scene, cam, w, h, spp)
        assert _same(whole, ref)
        for per_pass in (w * h, 2 * w * h, 3 * w * h + 17, 5 * w * h, 1 << 20):
            img, st = api.render_world(cfg.background, scene, cam, w, h, spp, paths_per_pass=per_pass, return_stats=True)
            assert np.array_equal(img, whole) and st["segments"] == ctr["segments"], per_pass
            assert st["passes"] == -(-spp // max(1, min(spp, per_pass // (w * h)))), (per_pass, st["passes"])
    finally:
        scene.close()


def test_no_rays_at_all(worlds):
    """Hittable::hit over an empty batch: nothing to do, nothing written, no error; and a null batch of a
    positive length is an error, not a fault."""
    scene = api.Scene(worlds("C1"))
    try:
        hits = scene.hit(np.zeros(0, dtype=api.RAY_DTYPE))
        assert hits.shape == (0,)
        lib = abi.libmirt()
        for traversal in (abi.TRAVERSE_FAST, abi.TRAVERSE_REFERENCE):
            assert lib.mirt_intersect(scene.handle, None, 0, 0.001, float("inf"), 1, traversal, None) == abi.OK
        assert lib.mirt_intersect(scene.handle, None, 4, 0.001, float("inf"), 1, abi.TRAVERSE_FAST, None) == abi.ERR_INVALID_ARGUMENT
    finally:
        scene.close()


def test_render_arguments_the_boundary_refuses(worlds):
    cfg = configs.C1
    scene = api.Scene(worlds("C1"))
    try:
        lib = abi.libmirt()
        cam = cfg.camera(16, 9)
        out = np.zeros((9, 16, 3), dtype=np.uint8)
        ptr = out.ctypes.data_as(C.POINTER(C.c_uint8))
        bg = api._vec3(cfg.background)

        def call(scene_handle=scene.handle, camera=C.byref(cam.c), background=bg, w=16, h=9, spp=1, depth=50, opts=None, dst=ptr):
            return lib.mirt_render(scene_handle, camera, background, w, h, spp, depth, 1, opts, dst, None)

        assert call() == abi.OK
        assert call(scene_handle=None) == abi.ERR_INVALID_ARGUMENT
        assert call(camera=None) == abi.ERR_INVALID_ARGUMENT
        assert call(background=None) == abi.ERR_INVALID_ARGUMENT
        assert call(dst=None) == abi.ERR_INVALID_ARGUMENT
        assert call(w=0) == abi.ERR_INVALID_ARGUMENT
        assert call(h=0) == abi.ERR_INVALID_ARGUMENT
        assert call(spp=0) == abi.ERR_INVALID_ARGUMENT
        assert call(depth=0) == abi.ERR_INVALID_ARGUMENT
        short = api.make_opts()
        short.struct_size = 8
        assert call(opts=C.byref(short)) == abi.ERR_INVALID_ARGUMENT
        beyond = api.make_opts(pixel_range=(0, 16 * 9 + 1))
        assert call(opts=C.byref(beyond)) == abi.ERR_INVALID_ARGUMENT
        assert lib.mirt_last_error().decode() != ""
        # the scene is still usable after every refusal
        assert call() == abi.OK
    finally:
        scene.close()


def test_a_pixel_range_of_one_pixel_and_the_last_pixel(worlds, oracle):
    """renderer.rs:74-75: idx -> (i, j) with row 0 of the output on top.  Single pixels at the corners of the
    frame come out as the oracle renders them there, and nothing else of the buffer is touched."""
    cfg = configs.C1
    w, h, spp = 40, 24, 5
    world = worlds("C1")
    cam = cfg.camera(w, h)
    ref, _, _, _ = oracle.render(world.desc, cam, cfg.background, w, h, spp)
    scene = api.Scene(world)
    try:
        for idx in (0, w - 1, w, w * h - w, w * h - 1):
            out = np.full((h, w, 3), 77, dtype=np.uint8)
            api.render_world(cfg.background, scene, cam, w, h, spp, pixel_range=(idx, idx + 1), out=out)
            got = out.reshape(-1, 3)
            want = ref.reshape(-1, 3)
            assert np.abs(got[idx].astype(int) - want[idx].astype(int)).max() <= 1, idx
            rest = np.delete(got, idx, axis=0)
            assert np.all(rest == 77), idx
    finally:
        scene.close()


def test_a_scene_keeps_its_buffers_between_calls_and_trim_frees_them(worlds):
    """mirt_render leaves its per-pass records, the f64 accumulation image and the byte buffer with the scene: the next
    call of the same size allocates nothing, mirt_scene_trim gives the memory back, and the call after that renders
    the same bytes again."""
    import torch

    cfg = configs.C1
    w, h, spp = 1024, 576, 2  # 14 MB of f64 image, 28 MB of records: large enough to show in the device's free memory
    scene = api.Scene(worlds("C1"))
    try:
        cam = cfg.camera(w, h)
        first = api.render_world(cfg.background, scene, cam, w, h, spp)
        free_kept, _ = torch.cuda.mem_get_info()
        again = api.render_world(cfg.background, scene, cam, w, h, spp)
        free_after, _ = torch.cuda.mem_get_info()
        assert np.array_equal(first, again) and free_after == free_kept
        scene.trim()
        free_trimmed, _ = torch.cuda.mem_get_info()
        assert free_trimmed >= free_kept + (w * h * (3 * 8 + 3) + w * h * spp * 24) // 2
        assert np.array_equal(api.render_world(cfg.background, scene, cam, w, h, spp), first)
        # a smaller frame after a larger one works in the larger one's buffers
        small = api.render_world(cfg.background, scene, cfg.camera(64, 36), 64, 36, spp)
        assert small.shape == (36, 64, 3)
    finally:
        scene.close()
