"""Development tool (run on the GPU box): mirt_scene_create on scene descriptions with one structural damage each --
indices past the end of their tables, kinds that do not exist, BVH members that point at themselves, at each other,
past the end, leaves out of order, a member count that is not 2n - 1, images without pixels, permutation entries
beyond 255 -- must refuse them (a description it accepts is rendered by kernels that trust it).
  usage: python tools/desc_soak.py [first seed] [cases]"""
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_fuzz as fz
from ray_tracer_amd import _abi as abi, api

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
accepted = []
refused = 0
for k in range(cases):
    seed = first + k
    rng = np.random.default_rng(9000 + seed)
    built, _, _, _ = fz._random_world(seed % 50, (3, 30))
    d = built.desc
    ns, nm, nt, nn = d.n_shapes, d.n_materials, d.n_textures, d.n_nodes
    big = int(rng.choice([0, 1, 7, 1000, 0x7FFFFF00, 0xFFFFFF00]))  # (sums below stay within 32 bits)
    damage = int(rng.integers(0, 17))
    what = None
    if damage == 0:
        i = int(rng.integers(0, ns)); built.shapes[i].material = nm + big; what = f"shape {i}: material {nm + big} of {nm}"
    elif damage == 1:
        i = int(rng.integers(0, ns)); built.shapes[i].kind = 6 + big; what = f"shape {i}: kind {6 + big}"
    elif damage == 2:
        textured = [i for i in range(nm) if built.materials[i].kind in (abi.MAT_LAMBERTIAN, abi.MAT_DIFFUSE_LIGHT)]  # (metal and glass have none)
        i = int(rng.choice(textured)); built.materials[i].texture = nt + big; what = f"material {i}: texture {nt + big} of {nt}"
    elif damage == 3:
        i = int(rng.integers(0, nm)); built.materials[i].kind = 4 + big; what = f"material {i}: kind {4 + big}"
    elif damage == 4:
        i = int(rng.integers(0, nt)); built.textures[i].kind = 4 + big; what = f"texture {i}: kind {4 + big}"
    elif damage == 5:
        i = 9; built.textures[i].even = i + min(big, 1000) if rng.random() < 0.5 else nt + big; what = f"checker {i}: child {built.textures[i].even}"
    elif damage == 6:
        i = 7; built.textures[i].perlin = 1 + big; what = f"noise texture: perlin table {1 + big} of 1"
    elif damage == 7:
        i = 8; built.textures[i].image = 1 + big; what = f"image texture: image {1 + big} of 1"
    elif damage == 8:
        built.images[0].width = 0 if rng.random() < 0.5 else built.images[0].width; built.images[0].height = 0; what = "image without pixels"
    elif damage == 9:
        built.perlins[0].perm_x[int(rng.integers(0, 256))] = int(rng.choice([256, -1, 70000])); what = "permutation entry outside 0..255"
    elif damage == 10:
        d.n_nodes = int(rng.choice([nn - 1, nn + 1, 0 if False else nn - 2, 1])); what = f"{d.n_nodes} BVH members for {ns} shapes"
    elif damage == 11:
        inner = [i for i in range(nn) if not built.nodes[i].is_leaf]
        i = int(rng.choice(inner)); built.nodes[i].left = i; what = f"member {i} is its own child"
    elif damage == 12:
        inner = [i for i in range(nn) if not built.nodes[i].is_leaf]
        i = int(rng.choice(inner)); built.nodes[i].right = nn + big; what = f"member {i}: child {nn + big} of {nn}"
    elif damage == 13:
        leaves = [i for i in range(nn) if built.nodes[i].is_leaf]
        i = int(rng.choice(leaves)); built.nodes[i].shape = ns + big; what = f"leaf {i}: shape {ns + big} of {ns}"
    elif damage == 14:
        leaves = [i for i in range(nn) if built.nodes[i].is_leaf]
        if len(leaves) < 2:
            continue
        a, b = rng.choice(leaves, 2, replace=False); built.nodes[int(a)].shape, built.nodes[int(b)].shape = built.nodes[int(b)].shape, built.nodes[int(a)].shape
        what = f"leaves {a} and {b} swapped"
    elif damage == 15:
        inner = [i for i in range(nn) if not built.nodes[i].is_leaf]
        i = int(rng.choice(inner)); built.nodes[i].left = built.nodes[i].right; what = f"member {i}: both children the same"
    else:
        inner = [i for i in range(nn) if not built.nodes[i].is_leaf]
        i = int(rng.choice(inner)); built.nodes[i].left = nn - 1; what = f"member {i}: the root as its child"
    try:
        scene = api.Scene(built.desc)
        scene.close()
        accepted.append(f"seed {seed}: {what}")
    except api.MirtError:
        refused += 1
print(f"{cases} damaged descriptions: {refused} refused, {len(accepted)} accepted")
for a in accepted[:40]:
    print("ACCEPTED", a)
sys.exit(1 if accepted else 0)
