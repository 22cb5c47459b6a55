"""Micro-benchmark of the closest-hit kernels on a realistic ray set.

Takes every ray the CPU oracle traces (primary and secondary) for a spread of pixels of a
configuration, shuffles them into queue-like order, and times mirt_intersect_device for
the fast traversal (and the reference-order one as a check).  Development tool; runs on
the GPU box.

    python tools/trav_bench.py [C1..C5] [--rays N] [--stats]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="C2")
    ap.add_argument("--rays", type=int, default=4_000_000)
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--order", default="shuffled", choices=["shuffled", "path"])
    args = ap.parse_args()

    import torch

    import oracle_lib
    from ray_tracer_amd import _abi as abi
    from ray_tracer_amd import api, configs

    cfg = configs.ALL[args.config]
    w, h = cfg.width // 4, cfg.height // 4
    world = cfg.world()
    cam = cfg.camera(w, h)
    rng = np.random.default_rng(3)
    pixels = rng.permutation(w * h).astype(np.uint32)
    t0 = time.time()
    rays, ref = oracle_lib.trace_rays(world.desc, cam, cfg.background, w, h, pixels, (0, 64), cap=args.rays)
    print(f"{args.config}: {rays.shape[0]} rays traced by the oracle in {time.time() - t0:.1f} s "
          f"({np.count_nonzero(ref['prim'] >= 0) / rays.shape[0]:.2%} hit)")
    if args.order == "shuffled":
        perm = rng.permutation(rays.shape[0])
        rays, ref = rays[perm], ref[perm]
    n = rays.shape[0]
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(n, -1).copy()).cuda()
    d_hits = torch.zeros((n, 16), dtype=torch.uint8, device="cuda")
    scene = api.Scene(world)
    lib = abi.libmirt()

    def run(traversal):
        rc = lib.mirt_intersect_device(scene.handle, C.c_void_p(d_rays.data_ptr()), n, 0.001, float("inf"), 1, traversal,
                                       C.c_void_p(d_hits.data_ptr()))
        assert rc == 0, lib.mirt_last_error()

    for name, trav in (("fast", abi.TRAVERSE_FAST), ("reference-order", abi.TRAVERSE_REFERENCE)):
        run(trav)
        torch.cuda.synchronize()
        got = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
        ok = np.array_equal(got["prim"], ref["prim"]) and np.array_equal(got["t"][ref["prim"] >= 0], ref["t"][ref["prim"] >= 0])
        times = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(trav)
            times.append(time.perf_counter() - t0)
        best = min(times)
        print(f"  {name:16s} {'exact' if ok else 'MISMATCH'}  best {best * 1e3:8.3f} ms  {n / best / 1e9:6.3f} Grays/s")
    if args.stats:
        os.environ["MIRT_TRAV_STATS"] = "1"
        run(abi.TRAVERSE_FAST)
        os.environ["MIRT_TRAV_STATS"] = "0"
    scene.close()


if __name__ == "__main__":
    main()
