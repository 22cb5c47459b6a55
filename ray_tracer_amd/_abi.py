"""ctypes view of the C ABI declared in include/mirt.h and include/mirt_host.h.

Loads the two in-tree shared libraries and declares every entry point.  There is
no fallback of any kind: if ``csrc/libmirt.so`` (the HIP kernels) is missing or
cannot be loaded, importing the device API raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIRT_LIBMIRT: development builds of the same library (e.g. csrc/libmirt_stats.so with clock statistics)
LIBMIRT_PATH = os.environ.get("MIRT_LIBMIRT") or os.path.join(_HERE, "csrc", "libmirt.so")
LIBHOST_PATH = os.path.join(_HERE, "host", "libmirt_host.so")

c_double3 = C.c_double * 3


class Shape(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("material", C.c_uint32), ("f", C.c_double * 10)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("albedo", c_double3), ("param", C.c_double)]


class Texture(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("even", C.c_uint32),
        ("odd", C.c_uint32),
        ("perlin", C.c_uint32),
        ("image", C.c_uint32),
        ("_pad", C.c_uint32),
        ("color", c_double3),
        ("scale", C.c_double),
    ]


class Perlin(C.Structure):
    _fields_ = [
        ("vecs", (C.c_double * 3) * 256),
        ("perm_x", C.c_int32 * 256),
        ("perm_y", C.c_int32 * 256),
        ("perm_z", C.c_int32 * 256),
    ]


class Image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_uint8))]


class BvhMember(C.Structure):
    _fields_ = [
        ("is_leaf", C.c_uint32),
        ("left", C.c_uint32),
        ("right", C.c_uint32),
        ("shape", C.c_uint32),
        ("bmin", c_double3),
        ("bmax", c_double3),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("shapes", C.POINTER(Shape)),
        ("n_shapes", C.c_uint32),
        ("materials", C.POINTER(Material)),
        ("n_materials", C.c_uint32),
        ("textures", C.POINTER(Texture)),
        ("n_textures", C.c_uint32),
        ("perlins", C.POINTER(Perlin)),
        ("n_perlins", C.c_uint32),
        ("images", C.POINTER(Image)),
        ("n_images", C.c_uint32),
        ("nodes", C.POINTER(BvhMember)),
        ("n_nodes", C.c_uint32),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("origin", c_double3),
        ("lower_left_corner", c_double3),
        ("horizontal", c_double3),
        ("vertical", c_double3),
        ("u", c_double3),
        ("v", c_double3),
        ("w", c_double3),
        ("lens_radius", C.c_double),
        ("time_start", C.c_double),
        ("time_end", C.c_double),
    ]


class Ray(C.Structure):
    _fields_ = [("origin", c_double3), ("direction", c_double3), ("time", C.c_double)]


class Hit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("side", C.c_int32), ("t", C.c_double)]


class RenderOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("use_bvh", C.c_int32),
        ("traversal", C.c_int32),
        ("pixel_begin", C.c_uint32),
        ("pixel_end", C.c_uint32),
        ("sample_begin", C.c_uint32),
        ("sample_end", C.c_uint32),
        ("paths_per_pass", C.c_uint32),
        ("profile", C.c_int32),
        ("tile_size", C.c_uint32),
        ("tile_count", C.c_uint32),
        ("tile_index", C.c_uint32),
        ("handover_bounce", C.c_int32),
        ("handover_paths", C.c_uint32),
    ]


class MultiOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("split", C.c_int32), ("reduce", C.c_int32), ("tile_size", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [
        ("samples", C.c_uint64),
        ("segments", C.c_uint64),
        ("passes", C.c_uint64),
        ("launches_extend", C.c_uint64),
        ("launches_shade", C.c_uint64),
        ("ms_total", C.c_double),
        ("ms_raygen", C.c_double),
        ("ms_extend", C.c_double),
        ("ms_bin", C.c_double),
        ("ms_shade", C.c_double),
        ("ms_accumulate", C.c_double),
        ("ms_resolve", C.c_double),
        ("ms_paths", C.c_double),
        ("launches_paths", C.c_uint64),
        ("segments_paths", C.c_uint64),
        ("handover_bounce", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


# enum values of include/mirt.h
SHAPE_SPHERE, SHAPE_XY_RECT, SHAPE_XZ_RECT, SHAPE_YZ_RECT, SHAPE_CUBE, SHAPE_MOVING_SPHERE = range(6)
MAT_LAMBERTIAN, MAT_DIELECTRIC, MAT_METAL, MAT_DIFFUSE_LIGHT = range(4)
TEX_CONSTANT, TEX_CHECKER, TEX_NOISE, TEX_IMAGE = range(4)
TRAVERSE_REFERENCE, TRAVERSE_FAST = 0, 1
HANDOVER_NEVER, HANDOVER_PROBE = -1, -2
SPLIT_TILE, SPLIT_SAMPLE = 0, 1
REDUCE_PEER, REDUCE_RCCL = 0, 1
STATIC_PASS_PATHS = 1 << 20  # MIRT_STATIC_PASS_PATHS
OK, ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_OUT_OF_MEMORY = 0, -1, -2, -3, -4, -5

# name -> (restype, argtypes) for every symbol the headers declare
MIRT_SYMBOLS = {
    "mirt_abi_version": (C.c_uint32, []),
    "mirt_build_id": (C.c_char_p, []),
    "mirt_multi_trim": (C.c_int, []),
    "mirt_last_error": (C.c_char_p, []),
    "mirt_device_count": (C.c_int, []),
    "mirt_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "mirt_scene_destroy": (None, [C.c_void_p]),
    "mirt_scene_trim": (C.c_int, [C.c_void_p]),
    "mirt_intersect": (
        C.c_int,
        [C.c_void_p, C.POINTER(Ray), C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(Hit)],
    ),
    "mirt_intersect_device": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p],
    ),
    "mirt_render": (
        C.c_int,
        [
            C.c_void_p,
            C.POINTER(Camera),
            C.POINTER(C.c_double),
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint64,
            C.POINTER(RenderOpts),
            C.POINTER(C.c_uint8),
            C.POINTER(Stats),
        ],
    ),
    "mirt_render_accumulate": (
        C.c_int,
        [
            C.c_void_p,
            C.POINTER(Camera),
            C.POINTER(C.c_double),
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint64,
            C.POINTER(RenderOpts),
            C.c_void_p,
            C.POINTER(Stats),
        ],
    ),
    "mirt_render_accumulate_static": (
        C.c_int,
        [
            C.c_void_p,
            C.POINTER(Camera),
            C.POINTER(C.c_double),
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint64,
            C.POINTER(RenderOpts),
            C.c_void_p,
            C.POINTER(Stats),
        ],
    ),
    "mirt_render_multi": (
        C.c_int,
        [
            C.POINTER(C.c_void_p),
            C.c_uint32,
            C.POINTER(Camera),
            C.POINTER(C.c_double),
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint32,
            C.c_uint64,
            C.POINTER(RenderOpts),
            C.POINTER(MultiOpts),
            C.POINTER(C.c_uint8),
            C.POINTER(Stats),
        ],
    ),
    "mirt_accum_alloc": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "mirt_accum_free": (None, [C.c_void_p, C.c_int]),
    "mirt_accum_read": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]),
    "mirt_accum_write": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]),
    "mirt_resolve": (
        C.c_int,
        [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint8)],
    ),
    "mirt_resolve_tiles": (
        C.c_int,
        [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
         C.c_int, C.POINTER(C.c_uint8)],
    ),
    "mirt_camera_rays": (
        C.c_int,
        [C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(Ray)],
    ),
}

HOST_SYMBOLS = {
    "mirt_host_last_error": (C.c_char_p, []),
    "mirt_host_world_create": (
        C.c_int,
        [C.c_char_p, C.c_uint64, C.c_uint64, C.c_double, C.c_double, C.POINTER(Image), C.c_char_p, C.POINTER(C.c_void_p)],
    ),
    "mirt_host_world_desc": (C.POINTER(SceneDesc), [C.c_void_p]),
    "mirt_host_world_destroy": (None, [C.c_void_p]),
    "mirt_host_camera_new": (
        C.c_int,
        [
            C.POINTER(C.c_double),
            C.POINTER(C.c_double),
            C.POINTER(C.c_double),
            C.c_double,
            C.c_double,
            C.c_double,
            C.c_double,
            C.c_double,
            C.c_double,
            C.POINTER(Camera),
        ],
    ),
    "mirt_host_bvh_build": (
        C.c_int,
        [C.POINTER(Shape), C.c_uint32, C.c_double, C.c_double, C.c_uint64, C.POINTER(BvhMember)],
    ),
    "mirt_host_shape_bbox": (
        C.c_int,
        [C.POINTER(Shape), C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    ),
    "mirt_host_perlin_new": (C.c_int, [C.c_uint64, C.POINTER(Perlin)]),
    "mirt_host_ppm_create": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]),
    "mirt_host_image_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.c_size_t]),
    "mirt_host_ppm_create_binary": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8)]),
}


def _bind(lib, table):
    for name, (restype, argtypes) in table.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_libmirt = None
_libhost = None


def libmirt():
    """The HIP library.  Raises if it has not been built: there is no CPU path."""
    global _libmirt
    if _libmirt is None:
        if not os.path.exists(LIBMIRT_PATH):
            raise RuntimeError(
                f"{LIBMIRT_PATH} is missing: build it with `make -C ray_tracer_amd` "
                "(or __graft_entry__.build()); the render path has no fallback"
            )
        _libmirt = _bind(C.CDLL(LIBMIRT_PATH, mode=C.RTLD_GLOBAL), MIRT_SYMBOLS)
    return _libmirt


def libhost():
    """The CPU-only host library (scene generation, BVH build, camera)."""
    global _libhost
    if _libhost is None:
        if not os.path.exists(LIBHOST_PATH):
            raise RuntimeError(f"{LIBHOST_PATH} is missing: build it with `make -C ray_tracer_amd`")
        _libhost = _bind(C.CDLL(LIBHOST_PATH), HOST_SYMBOLS)
    return _libhost
