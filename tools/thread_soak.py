"""Development tool (run on the GPU box): render and intersect calls from several host threads at once, on one scene
shared by all of them and on a scene each -- the reference's world is `Sync` and shared by rayon's threads
(hittable.rs:11, renderer.rs:27-43); here a caller's threads may share a mirt_scene.  Every call's bytes must be
those of the same call made alone.
  usage: python tools/thread_soak.py [threads] [calls per thread]"""
import sys, threading, hashlib
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from ray_tracer_amd import _abi as abi, api, configs

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
names = ["C1", "C3", "C4", "C5"]
worlds = {name: configs.ALL[name].world() for name in names}
shared = {name: api.Scene(worlds[name]) for name in names}


def job(k):
    """The k-th call: (config, frame, samples, pipeline); its image's digest and query count."""
    rng = np.random.default_rng(k)
    name = names[k % 4]
    w, h, spp = int(rng.integers(8, 120)), int(rng.integers(8, 90)), int(rng.integers(1, 7))
    handover = [(0, 0), (abi.HANDOVER_NEVER, 0), (1, 0)][k % 3]
    return name, w, h, spp, handover, int(rng.choice([0, 5000, 1 << 16]))


def run(k, scene):
    name, w, h, spp, handover, per_pass = job(k)
    cfg = configs.ALL[name]
    img, st = api.render_world(cfg.background, scene, cfg.camera(w, h), w, h, spp, return_stats=True, handover=handover, paths_per_pass=per_pass)
    rays = api.camera_rays(cfg.camera(w, h), w, h)
    hits = scene.hit(rays)
    return hashlib.sha1(img.tobytes() + hits.tobytes()).hexdigest(), st["segments"]


total = n_threads * n_calls
alone = [run(k, shared[job(k)[0]]) for k in range(total)]
results = [None] * total
errors = []


def worker(t, own):
    try:
        scenes = {name: api.Scene(worlds[name]) for name in names} if own else shared
        for c in range(n_calls):
            k = t * n_calls + c
            results[k] = run(k, scenes[job(k)[0]])
        if own:
            for s in scenes.values():
                s.close()
    except Exception as e:  # noqa: BLE001
        errors.append(f"thread {t}: {type(e).__name__} {e}")


failed = 0
for own in (False, True):
    results = [None] * total
    threads = [threading.Thread(target=worker, args=(t, own)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    bad = sum(1 for k in range(total) if results[k] != alone[k])
    failed += bad
    print(f"{n_threads} threads x {n_calls} calls, {'a scene each' if own else 'one scene shared'}: {bad} calls differ from the same call made alone", flush=True)
for e in errors:
    print("ERROR", e)
sys.exit(1 if failed or errors else 0)
