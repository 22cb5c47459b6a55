"""Development tool (run on the GPU box): Hittable::hit and small renders against worlds nobody would build -- spheres
of radius zero, tiny, huge or negative, cubes with unordered or coinciding corners, rects of no area or with their
ends swapped, shapes on top of each other, moving spheres whose shutter span is empty, negative or not a power of
two, everything scaled by 10^-30 .. 10^30 -- with ordinary and wild rays; all three traversals against the oracle.
A world the library refuses to create (mirt_scene_create's own checks) is counted, not an error.
  usage: python tools/world_soak.py [first seed] [worlds] [rays per world] [big]
`big`: worlds of 900 .. 20 000 shapes instead (the path kernel's other placements of the tree: nodes in LDS with the
records in global memory, 28-byte nodes, everything from global memory, 32-bit stack entries)."""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib
import scene_util as su
import test_gpu_intersect as ti
import degenerate_worlds as dw
from ray_tracer_amd import _abi as abi, api

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
worlds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8000
big = len(sys.argv) > 4
failed = refused = total = 0
t0 = time.perf_counter()
for w in range(worlds):
    seed = first + w
    built, scale, boxes, (t_lo, t_hi), rng = dw.degenerate_world(seed, big)
    shapes = built.shapes
    rays = dw.rays_for(built, scale, n, seed, rng, ti._rays_nobody_would_send)
    try:
        scene = api.Scene(built.desc)
    except api.MirtError:
        refused += 1
        continue
    try:
        for t_min, t_max in ((0.001, float("inf")), (-5.0 * scale, 40.0 * scale), (0.001 * scale, 1e30)):
            ref, _ = oracle_lib.intersect(built.desc, rays, t_min=t_min, t_max=t_max)
            ref_list, _ = oracle_lib.intersect(built.desc, rays, t_min=t_min, t_max=t_max, use_bvh=False)
            for what, got, want in (("fast", scene.hit(rays, t_min, t_max), ref),
                                    ("reference-order", scene.hit(rays, t_min, t_max, traversal=abi.TRAVERSE_REFERENCE), ref),
                                    ("list", scene.hit(rays, t_min, t_max, use_bvh=False), ref_list)):
                total += n
                try:
                    ti._compare(got, want, what)
                except AssertionError as e:
                    failed += 1
                    print(f"FAILED seed {seed} ({len(shapes)} shapes, scale {scale:g}, boxes {boxes}) range ({t_min:g}, {t_max:g}) {what}: {str(e)[:160]}", flush=True)
        # a small render of the same world through the path kernel and the launch-per-bounce pipeline
        cam = api.Camera.new(tuple(np.array([3.0, 4.0, 13.0]) * scale), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, 1.5, 0.0, 10.0, float(t_lo), float(t_hi))
        refimg, _, ctr, _ = oracle_lib.render(built.desc, cam, api.SKY, 48, 32, 2)
        for handover in ((0, 0), (abi.HANDOVER_NEVER, 0)):
            img, st = api.render_world(api.SKY, scene, cam, 48, 32, 2, return_stats=True, handover=handover)
            d = np.abs(img.astype(int) - refimg.astype(int))
            if st["segments"] != ctr["segments"] or d.max() > 1:
                failed += 1
                print(f"FAILED seed {seed} ({len(shapes)} shapes, scale {scale:g}, boxes {boxes}) render handover {handover}: segments {st['segments']} vs {ctr['segments']}, max byte difference {d.max()}", flush=True)
    finally:
        scene.close()
    if w % 20 == 19:
        print(f"{w + 1} worlds ({refused} refused), {total / 1e6:.1f} M ray queries, {failed} failed, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{worlds} worlds ({refused} refused), {total / 1e6:.1f} M ray queries, {failed} comparisons failed")
sys.exit(1 if failed else 0)
