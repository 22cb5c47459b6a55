import sys, time
sys.path.insert(0, '.')
from ray_tracer_amd import api, configs
for name in ("C2", "C5", "C4"):
    cfg = configs.ALL[name]
    world = cfg.world()
    for k in range(3):
        t = time.perf_counter()
        s = api.Scene(world)
        dt = (time.perf_counter() - t) * 1e3
        s.close()
        print(name, world.n_shapes, "scene create", k, f"{dt:.1f} ms", flush=True)
