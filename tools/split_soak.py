"""Development tool (run on the GPU box): however a frame is cut up, the bytes are those of the frame rendered in one
go -- random frame sizes (1 .. 90 pixels a side), sample counts, pass sizes, pixel ranges, tile sizes / counts
(mirt_render_opts), sample ranges summed in f64, the entry point that allocates nothing, and mirt_render_multi
with 1 .. 8 ranks on this device in both of its splits; path kernel and launch-per-bounce pipeline.
  usage: python tools/split_soak.py [first seed] [cases]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from ray_tracer_amd import _abi as abi, api, configs

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
names = ["C1", "C4", "C3", "C5"]
worlds = {name: configs.ALL[name].world() for name in names}
scenes = {name: api.Scene(worlds[name]) for name in names}
failed = 0
t0 = time.perf_counter()


def check(ok, what):
    global failed
    if not ok:
        failed += 1
        print("FAILED", what, flush=True)


for k in range(cases):
    seed = first + k
    rng = np.random.default_rng(555 + seed)
    name = names[k % 4]
    cfg, scene = configs.ALL[name], scenes[name]
    w, h = int(rng.integers(1, 91)), int(rng.integers(1, 91))
    spp = int(rng.integers(1, 10))
    depth = int(rng.choice([1, 3, 50]))
    handover = [(0, 0), (abi.HANDOVER_NEVER, 0), (2, 0)][int(rng.integers(0, 3))]
    cam = cfg.camera(w, h)
    tag = f"seed {seed} {name} {w}x{h}x{spp} depth {depth} handover {handover}"
    full, st_full = api.render_world(cfg.background, scene, cam, w, h, spp, max_depth=depth, return_stats=True, handover=handover)
    # pass sizes
    per_pass = int(rng.integers(1, 3 * w * h * spp + 2))
    img, st = api.render_world(cfg.background, scene, cam, w, h, spp, max_depth=depth, paths_per_pass=per_pass, return_stats=True, handover=handover)
    check(np.array_equal(img, full) and st["segments"] == st_full["segments"], f"{tag}: paths_per_pass {per_pass}")
    # pixel ranges: the frame in 1 .. 5 consecutive pieces
    cuts = sorted(set([0, w * h] + [int(x) for x in rng.integers(0, w * h + 1, int(rng.integers(0, 5)))]))
    img = np.full_like(full, 91)
    for a, b in zip(cuts[:-1], cuts[1:]):
        api.render_world(cfg.background, scene, cam, w, h, spp, max_depth=depth, pixel_range=(a, b), out=img, handover=handover)
    check(np.array_equal(img, full), f"{tag}: pixel ranges {cuts}")
    # tiles of any size over 1 .. 6 ranks
    tile_size, n_ranks = int(rng.integers(1, w * h + 2)), int(rng.integers(1, 7))
    img = np.full_like(full, 92)
    for r in range(n_ranks):
        api.render_world(cfg.background, scene, cam, w, h, spp, max_depth=depth, out=img, tiles=(tile_size, n_ranks, r), handover=handover)
    check(np.array_equal(img, full), f"{tag}: tiles of {tile_size} over {n_ranks} ranks")
    # sample ranges summed in f64 (and the entry point that allocates nothing, for the path kernel from the camera ray)
    cuts = sorted(set([0, spp] + [int(x) for x in rng.integers(0, spp + 1, 2)]))
    total = torch.zeros(w * h * 3, dtype=torch.float64, device="cuda")
    static = handover == (0, 0) and bool(rng.integers(0, 2))
    for a, b in zip(cuts[:-1], cuts[1:]):
        api.render_accumulate(scene, cam, cfg.background, w, h, spp, total.data_ptr(), max_depth=depth, sample_range=(a, b), handover=handover, static=static)
    torch.cuda.synchronize()
    img = api.resolve(total.data_ptr(), w, h, spp)
    d = np.abs(img.astype(int) - full.astype(int))
    check(d.max() <= 1 and np.count_nonzero(d) <= 2, f"{tag}: sample ranges {cuts} static {static}: {np.count_nonzero(d)} bytes differ, max {d.max()}")
    # one process, several ranks (here: all on this device)
    n_ranks = int(rng.integers(1, 9))
    img = api.render_world_multi(cfg.background, [scene] * n_ranks, cam, w, h, spp, split="tile", tile_size=int(rng.integers(0, w * h + 2)), max_depth=depth, handover=handover)
    check(np.array_equal(img, full), f"{tag}: mirt_render_multi tile split over {n_ranks} ranks")
    img = api.render_world_multi(cfg.background, [scene] * n_ranks, cam, w, h, spp, split="sample", max_depth=depth, handover=handover)
    d = np.abs(img.astype(int) - full.astype(int))
    check(d.max() <= 1 and np.count_nonzero(d) <= 2, f"{tag}: mirt_render_multi sample split over {n_ranks} ranks: {np.count_nonzero(d)} bytes differ")
    if k % 50 == 49:
        print(f"{k + 1} cases, {failed} failed, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{cases} cases, {failed} checks failed")
sys.exit(1 if failed else 0)
