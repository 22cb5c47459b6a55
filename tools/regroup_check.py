"""Development tool: the kernel that regroups paths across waves (MIRT_PATHS_REGROUP=<walking waves>) against the
default one -- run once without and once with the variable and compare the lines.
  usage: python tools/regroup_check.py [small|big]"""
import hashlib, sys, time
sys.path.insert(0, '.')
from ray_tracer_amd import api, configs
big = len(sys.argv) > 1 and sys.argv[1] == "big"
frames = {"C1": (400, 225, 10), "C4": (128, 128, 16), "C3": (192, 108, 8)} if big else {"C1": (64, 36, 2)}
for name, (w, h, spp) in frames.items():
    cfg = configs.ALL[name]
    t = time.perf_counter()
    img, st = api.render_world(cfg.background, cfg.world(), cfg.camera(w, h), w, h, spp, return_stats=True)
    print(name, hashlib.sha256(img.tobytes()).hexdigest()[:16], st["segments"], round(time.perf_counter() - t, 3), flush=True)
