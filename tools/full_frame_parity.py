"""Development tool (run on the GPU box): the benchmark configurations at their full frame sizes and a reduced sample
count against the CPU oracle with the same sampler -- images byte for byte (at most one level), and the number of
closest-hit queries.  Tens of millions of paths per scene; a minute or two of CPU time.
  usage: python tools/full_frame_parity.py [bench [C2 C3 C4 C5]]"""
import sys, threading, time, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib
from ray_tracer_amd import api, configs
# "bench" as the only argument: the sample counts of the bench lines themselves (C2 at its 500; the others at 128):
# 1.9 G paths, about five minutes of CPU time
counts = {"C2": 500, "C3": 128, "C4": 128, "C5": 128} if sys.argv[1:2] == ["bench"] else {"C2": 64, "C3": 16, "C4": 32, "C5": 4}
if sys.argv[2:]:
    counts = {k: v for k, v in counts.items() if k in sys.argv[2:]}  # only the configurations named


def heartbeat():  # (the oracle's render is one long call: say that the run is alive)
    while True:
        time.sleep(60)
        print("  ...", flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
for name, spp in counts.items():
    cfg = configs.ALL[name]
    w, h = cfg.width, cfg.height
    world = cfg.world()
    cam = cfg.camera(w, h)
    t0 = time.perf_counter()
    ref, _, ctr, _ = oracle_lib.render(world.desc, cam, cfg.background, w, h, spp)
    t1 = time.perf_counter()
    gpu, st = api.render_world(cfg.background, world, cam, w, h, spp, return_stats=True)
    d = np.abs(gpu.astype(int) - ref.astype(int))
    print(f"{name} {w}x{h}x{spp}: {w * h * spp / 1e6:.1f} M paths, segments {st['segments']} / {ctr['segments']}, max difference {d.max()}, "
          f"bytes differing {np.count_nonzero(d)} of {d.size} (oracle {t1 - t0:.1f} s)", flush=True)
    assert st["segments"] == ctr["segments"] and d.max() <= 1
