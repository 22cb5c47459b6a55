"""The committed fixtures of tests/golden/ (tools/make_golden.py wrote them from the CPU oracle;
the reference has no vectors of its own and cannot run here): the oracle must still give them
(CPU), and the HIP path must give them too (GPU) -- hits bit for bit, images within one level."""
import os

import numpy as np
import pytest

import golden_cases
from ray_tracer_amd import api

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _hits(name):
    with np.load(os.path.join(GOLDEN, f"hits_{name}.npz")) as z:
        return {k: z[k] for k in ("rays", "prim", "side", "t")}


def _image(name):
    with np.load(os.path.join(GOLDEN, f"image_{name}.npz")) as z:
        return z["rgb"], int(z["segments"])


def _same_hits(got, want, what):
    assert np.array_equal(got["prim"], want["prim"]), f"{what}: primitive ids differ on {np.count_nonzero(got['prim'] != want['prim'])} rays"
    hit = want["prim"] >= 0
    assert np.array_equal(got["side"][hit], want["side"][hit]), f"{what}: cube sides differ"
    assert np.array_equal(got["t"][hit], want["t"][hit]), f"{what}: t not bit-identical"


@pytest.mark.parametrize("name", golden_cases.HIT_CASES)
def test_oracle_reproduces_golden_hits(name, oracle):
    desc, _keep, rays, kwargs = golden_cases.hit_case(name)
    want = _hits(name)
    assert np.array_equal(rays.view(np.uint8), want["rays"].view(np.uint8)), "the case's rays changed: re-run tools/make_golden.py"
    got, _ = oracle.intersect(desc, rays, **kwargs)
    _same_hits(got, want, name)


@pytest.mark.parametrize("name", golden_cases.IMAGE_CASES)
def test_oracle_reproduces_golden_images(name, oracle):
    world, cam, background, w, h, spp = golden_cases.image_case(name)
    want, segments = _image(name)
    rgb, _, ctr, _ = oracle.render(world.desc, cam, background, w, h, spp)
    assert ctr["segments"] == segments
    assert np.array_equal(rgb, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_cases.HIT_CASES)
def test_hip_reproduces_golden_hits(name):
    from ray_tracer_amd import _abi as abi

    desc, _keep, _rays, kwargs = golden_cases.hit_case(name)
    want = _hits(name)
    scene = api.Scene(desc)
    try:
        for trav in (abi.TRAVERSE_FAST, abi.TRAVERSE_REFERENCE):
            _same_hits(scene.hit(want["rays"], traversal=trav, **kwargs), want, f"{name} traversal {trav}")
    finally:
        scene.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_cases.IMAGE_CASES)
def test_hip_reproduces_golden_images(name):
    world, cam, background, w, h, spp = golden_cases.image_case(name)
    want, segments = _image(name)
    rgb, stats = api.render_world(background, world, cam, w, h, spp, return_stats=True)
    assert stats["segments"] == segments  # the same paths, bounce for bounce
    diff = np.abs(rgb.astype(int) - want.astype(int))
    # forward product of attenuations vs the reference's nested product: a channel may flip by one level, rarely
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 0.002 * diff.size, (diff.max(), np.count_nonzero(diff))
