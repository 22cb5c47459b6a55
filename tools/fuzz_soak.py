"""Development tool (run on the GPU box): the random-world parity test of tests/test_gpu_fuzz.py on seeds the test
suite does not hold.  Prints one line per failing world and a count at the end.
  usage: python tools/fuzz_soak.py [first seed] [worlds per category]"""
import sys, time, traceback
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib
import test_gpu_fuzz as fz

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cases = [(s, (24, 90), False) for s in range(first, first + count)] + [(s, (1 + s % 5, 16), False) for s in range(first, first + count)] + \
        [(s, (1 + s % 4, 11), True) for s in range(first, first + count)] + [(s, (600, 1200), False) for s in range(first, first + max(2, count // 10))] + \
        [(s, (2200, 3400), False) for s in range(first, first + max(2, count // 15))]
failed = 0
dull = 0
t0 = time.perf_counter()
for k, (seed, n_shapes, planar) in enumerate(cases):
    try:
        fz.test_random_worlds_render_and_intersect_like_the_oracle(seed, n_shapes, planar, oracle_lib)
    except Exception as e:  # noqa: BLE001
        where = traceback.extract_tb(e.__traceback__)[-1]
        if "a picture, not a black frame" in (where.line or "") or "len(used)" in (where.line or ""):
            dull += 1  # the world itself is dull (nothing lit, one material): said by the test's own sanity lines, not a difference
            continue
        failed += 1
        print(f"FAILED seed {seed} shapes {n_shapes} planar {planar}: {type(e).__name__} {str(e)[:300]} at line {where.lineno}: {where.line}", flush=True)
    if k % 20 == 19:
        print(f"{k + 1} worlds, {failed} failed, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{len(cases)} worlds, {dull} too dull to count, {failed} failed")
sys.exit(1 if failed else 0)
