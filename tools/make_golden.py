#!/usr/bin/env python3
"""Writes the fixtures under tests/golden/ from the CPU oracle (oracle/liboracle.so).

The reference has no tests, golden vectors or fixtures of its own and cannot be built in this
image (Rust), so these vectors do NOT come from the reference: they pin the oracle -- and, on
the GPU, the HIP path -- against their own earlier answers, next to the hand-derived known
answers of tests/test_oracle_kat.py.  Re-run after a deliberate change of the sampler or of a
scene generator:   python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_cases  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name in golden_cases.HIT_CASES:
        desc, keep, rays, kwargs = golden_cases.hit_case(name)
        hits, _ = oracle_lib.intersect(desc, rays, **kwargs)
        np.savez_compressed(os.path.join(out, f"hits_{name}.npz"), rays=rays, prim=hits["prim"], side=hits["side"], t=hits["t"])
        print(name, rays.shape[0], "rays,", int(np.count_nonzero(hits["prim"] >= 0)), "hits")
    for name in golden_cases.IMAGE_CASES:
        world, cam, background, w, h, spp = golden_cases.image_case(name)
        rgb, _, ctr, _ = oracle_lib.render(world.desc, cam, background, w, h, spp)
        np.savez_compressed(os.path.join(out, f"image_{name}.npz"), rgb=rgb, segments=np.int64(ctr["segments"]))
        print(name, rgb.shape, ctr["segments"], "segments")


if __name__ == "__main__":
    main()
