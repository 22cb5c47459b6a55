"""The C ABI boundary: both shared libraries load, export every function the headers
declare (and bind with the signatures of ray_tracer_amd/_abi.py), the ctypes structs have
the sizes the C compiler gives them, and without a GPU the product refuses to work
instead of falling back to anything.  CPU only: no compute call is made."""
import ctypes as C
import os
import re
import subprocess

import pytest

from ray_tracer_amd import _abi as abi
from ray_tracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mirt_(?:host_)?[a-z_0-9]+)\s*\(", text)))


def test_libmirt_exports_every_declared_function():
    names = declared_functions("mirt.h")
    assert "mirt_render" in names and "mirt_intersect" in names and len(names) >= 12
    assert sorted(abi.MIRT_SYMBOLS) == names  # the Python table and the header list the same functions
    lib = abi.libmirt()
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.mirt_abi_version() >= 1


def test_libmirt_host_exports_every_declared_function():
    names = declared_functions("mirt_host.h")
    assert sorted(abi.HOST_SYMBOLS) == names
    lib = abi.libhost()
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_layouts_match_the_c_compiler(tmp_path):
    structs = {"mirt_shape": abi.Shape, "mirt_material": abi.Material, "mirt_texture": abi.Texture,
               "mirt_perlin": abi.Perlin, "mirt_image": abi.Image, "mirt_bvh_member": abi.BvhMember,
               "mirt_scene_desc": abi.SceneDesc, "mirt_camera": abi.Camera, "mirt_ray": abi.Ray, "mirt_hit": abi.Hit,
               "mirt_render_opts": abi.RenderOpts, "mirt_stats": abi.Stats, "mirt_multi_opts": abi.MultiOpts}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "mirt.h"\nint main(void) {\n' +
                   "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in structs) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)  # also proves the header is plain C
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    sizes = dict(line.split() for line in out.splitlines())
    for n, s in structs.items():
        assert int(sizes[n]) == C.sizeof(s), n


def test_the_rng_header_is_plain_c_and_matches_splitmix64(tmp_path):
    # include/mirt_rng.h is shared by the oracle and the device code; its mixer is splitmix64's
    # finaliser, whose first outputs for state 0 are published constants
    src = tmp_path / "rng.c"
    src.write_text('#include <stdio.h>\n#include "mirt_rng.h"\nint main(void) {\n'
                   '  mirt_seq s = mirt_seq_new(0);\n'
                   '  printf("%llu\\n", (unsigned long long)mirt_mix64(0x9E3779B97F4A7C15ull));\n'
                   '  double d = mirt_rng_double(mirt_rng_stream(1, 2, 3), 0, 0);\n'
                   '  printf("%d\\n", d >= 0.0 && d < 1.0);\n  (void)s; return 0;\n}\n')
    exe = tmp_path / "rng"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == 0xE220A8397B1DCDAF  # splitmix64: first output from seed 0
    assert out[1] == "1"


@pytest.mark.skipif(api.device_count() > 0, reason="checks the behaviour WITHOUT a GPU")
def test_without_a_gpu_the_library_refuses_loudly(worlds):
    assert abi.libmirt().mirt_device_count() == 0
    world = worlds("Basic")
    with pytest.raises(api.MirtError) as e:
        api.Scene(world)
    assert e.value.status in (abi.ERR_NO_DEVICE, abi.ERR_HIP)
    assert "device" in str(e.value).lower()


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    monkeypatch.setattr(abi, "_libmirt", None)
    monkeypatch.setattr(abi, "LIBMIRT_PATH", "/nonexistent/libmirt.so")
    with pytest.raises(RuntimeError):
        abi.libmirt()
