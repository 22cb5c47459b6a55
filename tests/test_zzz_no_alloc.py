"""mirt_render_accumulate_static, the entry point that allocates no device memory, against
mirt_render_accumulate: the same f64 sums bit for bit, on two calls in a row that share the
library's static array, the second with a smaller frame; and its limits are refused."""
import numpy as np
import pytest

from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api, configs

pytestmark = pytest.mark.gpu


def _both(scene, cam, background, w, h, spp, **kw):
    import torch

    got = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    want = torch.zeros_like(got)
    torch.cuda.synchronize()
    st_got = api.render_accumulate(scene, cam, background, w, h, spp, got.data_ptr(), static=True, **kw)
    st_want = api.render_accumulate(scene, cam, background, w, h, spp, want.data_ptr(), **kw)
    torch.cuda.synchronize()
    return got.cpu().numpy(), want.cpu().numpy(), st_got, st_want


def test_static_entry_point_matches_the_allocating_one(worlds):
    cfg = configs.C1
    world = worlds("C1")
    scene = api.Scene(world)
    try:
        # first call: more pixels than fit one pass twice over -> several passes through the array
        w, h, spp = 1200, 800, 3
        assert w * h <= abi.STATIC_PASS_PATHS < w * h * spp
        got, want, st_got, st_want = _both(scene, cfg.camera(w, h), cfg.background, w, h, spp)
        assert st_got["passes"] == spp and st_got["segments"] == st_want["segments"]
        assert want.any() and np.array_equal(got.view(np.uint64), want.view(np.uint64))
        # second call, smaller: the array still holds the first call's records and counters
        w, h, spp = 96, 54, 6
        got, want, st_got, st_want = _both(scene, cfg.camera(w, h), cfg.background, w, h, spp)
        assert st_got["passes"] == 1 and st_got["segments"] == st_want["segments"]
        assert want.any() and np.array_equal(got.view(np.uint64), want.view(np.uint64))
        # a warm call leaves the device's free memory where it was: nothing allocated, nothing freed and kept
        import torch

        cam = cfg.camera(w, h)
        accum = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
        api.render_accumulate(scene, cam, cfg.background, w, h, spp, accum.data_ptr(), static=True)
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        api.render_accumulate(scene, cam, cfg.background, w, h, spp, accum.data_ptr(), static=True)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before, (free_before, free_after)
        # a sample range and a tile split through the same entry point
        got, want, _, _ = _both(scene, cfg.camera(w, h), cfg.background, w, h, spp, sample_range=(2, 5), tiles=(w, 3, 1))
        assert want.any() and np.array_equal(got.view(np.uint64), want.view(np.uint64))
    finally:
        scene.close()


def test_static_entry_point_refuses_what_it_cannot_serve(worlds):
    import torch

    cfg = configs.C1
    scene = api.Scene(worlds("C1"))
    try:
        w, h = 1920, 1080  # more pixels than a pass of the static array holds
        assert w * h > abi.STATIC_PASS_PATHS
        accum = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
        with pytest.raises(api.MirtError) as err:
            api.render_accumulate(scene, cfg.camera(w, h), cfg.background, w, h, 1, accum.data_ptr(), static=True)
        assert err.value.status == abi.ERR_UNSUPPORTED and "pixel ranges" in str(err.value)
        # the same frame in pixel ranges is served
        half = w * h // 2
        for pr in ((0, half), (half, w * h)):
            api.render_accumulate(scene, cfg.camera(w, h), cfg.background, w, h, 1, accum.data_ptr(), static=True, pixel_range=pr)
        want = torch.zeros_like(accum)
        api.render_accumulate(scene, cfg.camera(w, h), cfg.background, w, h, 1, want.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(accum, want)
        for kw in ({"handover": (abi.HANDOVER_NEVER, 0)}, {"traversal": abi.TRAVERSE_REFERENCE}, {"use_bvh": False}, {"profile": True}):
            with pytest.raises(api.MirtError) as err:
                api.render_accumulate(scene, cfg.camera(64, 36), cfg.background, 64, 36, 1, accum.data_ptr(), static=True, **kw)
            assert err.value.status == abi.ERR_UNSUPPORTED
    finally:
        scene.close()
