/*
 * mirt.h -- C ABI of libmirt.so: the render hot path of mbStavola/ray_tracer
 * (per-pixel spp x bounce loop, BVH traversal, primitive intersection,
 * material scatter, texture evaluation) as hand-written HIP for gfx950.
 *
 * The reference has no FFI layer.  The seam this library is bound at is the
 * one call main.rs makes into the hot path,
 *     render_world(&background, &world, &camera, W, H, spp,
 *                  render_parallel, use_bounding_volume) -> Vec<u8>
 * (reference src/renderer.rs:9-62, called from src/main.rs:83-92), plus the
 * trait method the intersection parity check targets,
 *     Hittable::hit(&self, &Ray, t_min, t_max) -> Option<Hit>
 * (reference src/hittable.rs:11-14, implemented by BoundingVolumeHierarchy
 * at src/bvh.rs:83-91 and by Vec<Shape> at src/hittable.rs:667-710).
 *
 * Everything that crosses the boundary is plain-old-data: the caller flattens
 * its scene graph (Shape / Material / Texture / Perlin / BVHMember values)
 * into the arrays below; the library copies them to the device.  No C++ or
 * torch types appear in any signature.
 *
 * Conventions: every function returns MIRT_OK (0) or a negative mirt_status;
 * nothing throws or aborts across the boundary; the caller owns every input
 * array and every output buffer; the library owns the device memory behind a
 * mirt_scene.  mirt_last_error() returns a thread-local message for the last
 * failing call on the calling thread.
 */
#ifndef MIRT_H
#define MIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum mirt_status {
    MIRT_OK = 0,
    MIRT_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad index, bad enum, zero size */
    MIRT_ERR_NO_DEVICE = -2,        /* no HIP device / HIP runtime failure at init */
    MIRT_ERR_HIP = -3,              /* a HIP call failed; see mirt_last_error() */
    MIRT_ERR_UNSUPPORTED = -4,      /* scene exceeds a documented limit */
    MIRT_ERR_OUT_OF_MEMORY = -5
} mirt_status;

/* ------------------------------------------------------------- scene (POD) */

/* Shape kinds, in the order of `enum Shape` (src/hittable.rs:563-571). */
enum {
    MIRT_SHAPE_SPHERE = 0,        /* f = {cx, cy, cz, radius}                      hittable.rs:87-92  */
    MIRT_SHAPE_XY_RECT = 1,       /* f = {x0, x1, y0, y1, k}                       hittable.rs:153-161 */
    MIRT_SHAPE_XZ_RECT = 2,       /* f = {x0, x1, z0, z1, k}                       hittable.rs:217-225 */
    MIRT_SHAPE_YZ_RECT = 3,       /* f = {y0, y1, z0, z1, k}                       hittable.rs:281-289 */
    MIRT_SHAPE_CUBE = 4,          /* f = {p0x, p0y, p0z, p1x, p1y, p1z}            hittable.rs:396-420 */
    MIRT_SHAPE_MOVING_SPHERE = 5  /* f = {c0x, c0y, c0z, radius, c1x, c1y, c1z,
                                          time_start, time_end}                    hittable.rs:468-485 */
};

typedef struct mirt_shape {
    uint32_t kind;     /* MIRT_SHAPE_* */
    uint32_t material; /* index into mirt_scene_desc.materials */
    double f[10];      /* kind-specific parameters, see above; unused tail = 0 */
} mirt_shape;

/* Material kinds, in the order of `enum Material` (src/material.rs:154-160). */
enum {
    MIRT_MAT_LAMBERTIAN = 0,   /* texture = albedo texture                  material.rs:40-59   */
    MIRT_MAT_DIELECTRIC = 1,   /* param = ref_idx                           material.rs:93-131  */
    MIRT_MAT_METAL = 2,        /* albedo = rgb, param = fuzziness (<= 1)    material.rs:61-91   */
    MIRT_MAT_DIFFUSE_LIGHT = 3 /* texture = emission texture                material.rs:133-152 */
};

typedef struct mirt_material {
    uint32_t kind;    /* MIRT_MAT_* */
    uint32_t texture; /* index into textures (lambertian, diffuse light) */
    double albedo[3]; /* metal only */
    double param;     /* metal: fuzziness; dielectric: ref_idx */
} mirt_material;

/* Texture kinds, in the order of `enum Texture` (src/texture.rs:124-130). */
enum {
    MIRT_TEX_CONSTANT = 0, /* color                                   texture.rs:13-28   */
    MIRT_TEX_CHECKER = 1,  /* even, odd = indices of child textures   texture.rs:30-54   */
    MIRT_TEX_NOISE = 2,    /* perlin = index into perlins, scale      texture.rs:56-87   */
    MIRT_TEX_IMAGE = 3     /* image = index into images               texture.rs:89-122  */
};

typedef struct mirt_texture {
    uint32_t kind;   /* MIRT_TEX_* */
    uint32_t even;   /* checker: child texture used when the sine product is >= 0 */
    uint32_t odd;    /* checker: child texture used when the sine product is < 0 */
    uint32_t perlin; /* noise */
    uint32_t image;  /* image */
    uint32_t _pad;
    double color[3]; /* constant */
    double scale;    /* noise */
} mirt_texture;

/* One `Perlin` value (src/perlin.rs:10-17). */
typedef struct mirt_perlin {
    double vecs[256][3];
    int32_t perm_x[256];
    int32_t perm_y[256];
    int32_t perm_z[256];
} mirt_perlin;

/* A decoded image: tightly packed RGB8, row 0 = top (what image::open +
 * get_pixel give the reference, src/texture.rs:106-122,169-181). */
typedef struct mirt_image {
    uint32_t width;
    uint32_t height;
    const uint8_t* rgb;
} mirt_image;

/* One `BVHMember` (src/bvh.rs:93-101) in the reference's own order:
 * post-order, children before parents, root LAST (src/bvh.rs:33-35,130-131).
 * A leaf carries no box.  `shape` indexes the BVH-sorted shape array: the leaves hold shapes
 * 0, 1, 2, ... from left to right, as the reference's build leaves them (mirt_scene_create
 * refuses any other order: ties between equal t go by it, bvh.rs:70-74). */
typedef struct mirt_bvh_member {
    uint32_t is_leaf;
    uint32_t left;   /* node: index of left member */
    uint32_t right;  /* node: index of right member */
    uint32_t shape;  /* leaf: index into shapes */
    double bmin[3];  /* node: bounds.min */
    double bmax[3];  /* node: bounds.max */
} mirt_bvh_member;

typedef struct mirt_scene_desc {
    const mirt_shape* shapes; /* in BVH-sorted order (BoundingVolumeHierarchy.shapes, bvh.rs:12-15) */
    uint32_t n_shapes;
    const mirt_material* materials;
    uint32_t n_materials;
    const mirt_texture* textures;
    uint32_t n_textures;
    const mirt_perlin* perlins;
    uint32_t n_perlins;
    const mirt_image* images;
    uint32_t n_images;
    const mirt_bvh_member* nodes; /* may be NULL: then only use_bvh = 0 is served */
    uint32_t n_nodes;             /* 2 * n_shapes - 1 when present */
} mirt_scene_desc;

/* The ten derived fields of `Camera` (src/camera.rs:7-18), as Camera::new
 * (src/camera.rs:21-63) leaves them. */
typedef struct mirt_camera {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
    double u[3];
    double v[3];
    double w[3];
    double lens_radius;
    double time_start;
    double time_end;
} mirt_camera;

/* `Ray` (src/ray.rs:4-8). */
typedef struct mirt_ray {
    double origin[3];
    double direction[3];
    double time;
} mirt_ray;

/* What the parity check reads out of an Option<Hit>: prim = index of the hit
 * shape in the BVH-sorted shape array (-1 = None), side = which of a cube's
 * six rects was hit (0 for other shapes; order of Cube::new,
 * hittable.rs:405-412), t = Hit::t(). */
typedef struct mirt_hit {
    int32_t prim;
    int32_t side;
    double t;
} mirt_hit;

typedef struct mirt_scene mirt_scene; /* opaque; immutable after create */

/* ------------------------------------------------------------------ options */

/* How a closest-hit query walks the BVH.  All three return the reference's
 * answer (BoundingVolumeHierarchy::indexed_hit, src/bvh.rs:37-80). */
enum {
    /* Every member whose ancestors' boxes pass AABB::hit(ray, t_min, t_max) is
     * visited, both children with the same t_max, combined exactly as
     * bvh.rs:68-79 does (left wins only if left.t < right.t). */
    MIRT_TRAVERSE_REFERENCE = 0,
    /* The same answer by another route: the closest hit among all shapes is found
     * through a culling tree of the library's own (near child first, subtrees
     * beyond the best t skipped with a safety margin); it counts only if
     * AABB::hit passes for the boxes above its shape, tested in the reference's
     * arithmetic, else the query is repeated with that test on every hit; ties
     * resolved as bvh.rs:70-74 would. */
    MIRT_TRAVERSE_FAST = 1
};

/* Values of mirt_render_opts.handover_bounce below zero. */
enum {
    MIRT_HANDOVER_NEVER = -1,
    MIRT_HANDOVER_PROBE = -2
};

typedef struct mirt_render_opts {
    uint32_t struct_size;    /* = sizeof(mirt_render_opts) */
    int32_t device;          /* HIP device ordinal; -1 = current device */
    int32_t use_bvh;         /* use_bounding_volume (renderer.rs:17,21-25) */
    int32_t traversal;       /* MIRT_TRAVERSE_*; ignored when use_bvh = 0 */
    uint32_t pixel_begin;    /* render pixels idx in [pixel_begin, pixel_end) of 0..W*H */
    uint32_t pixel_end;      /* 0 = W*H */
    uint32_t sample_begin;   /* render samples s in [sample_begin, sample_end) of 0..spp */
    uint32_t sample_end;     /* 0 = spp */
    uint32_t paths_per_pass; /* target number of paths per pass; 0 = default (2^29 when the path kernel
                              * runs from the camera ray, 24 B of device memory per path; 2^26 otherwise,
                              * 173 B per path; either way at most a quarter of the free device memory) */
    int32_t profile;         /* 1 = time each kernel class with HIP events (fills mirt_stats.ms_*) */
    /* Tile split across devices: of the pixel range, cut into tiles of tile_size
     * consecutive pixels, only tiles t with t % tile_count == tile_index are
     * rendered (and written to out_rgb).  tile_count <= 1 renders the whole range. */
    uint32_t tile_size;
    uint32_t tile_count;
    uint32_t tile_index;
    /* Scheduling only, never results.  Two ways to run a path's bounces exist: one lane
     * follows the path from its ray at bounce N to its end inside one kernel ("path kernel"),
     * or every bounce is three launches (closest hit, sort by material, shade) over the
     * queue of paths still alive.  handover_bounce = N chooses where the first takes over:
     *   0 (default)           from the camera ray: no launches per bounce at all (fastest)
     *   N > 0                 launches per bounce for bounces < N, path kernel from bounce N
     *   MIRT_HANDOVER_NEVER   launches per bounce throughout
     *   MIRT_HANDOVER_PROBE   like N, N = the first bounce that at most handover_paths paths
     *                         reach (read back during the first pass of the call)
     * The path kernel uses the fast traversal; with MIRT_TRAVERSE_REFERENCE or use_bvh = 0
     * the setting is ignored and every bounce is launched. */
    int32_t handover_bounce;
    uint32_t handover_paths; /* threshold of MIRT_HANDOVER_PROBE; 0 = half the paths of a pass */
} mirt_render_opts;

typedef struct mirt_stats {
    uint64_t samples;       /* camera paths started */
    uint64_t segments;      /* closest-hit queries issued (primary + secondary rays) */
    uint64_t passes;
    uint64_t launches_extend;
    uint64_t launches_shade;
    double ms_total;        /* wall time of the call, host clock */
    double ms_raygen;       /* sum of kernel durations by class, HIP events; 0 unless profile = 1 */
    double ms_extend;
    double ms_bin;
    double ms_shade;
    double ms_accumulate;
    double ms_resolve;
    double ms_paths;           /* the path kernel */
    uint64_t launches_paths;
    uint64_t segments_paths;   /* of `segments`, those issued by the path kernel */
    uint32_t handover_bounce;  /* bounce at which the last pass went over to the path kernel (max_depth: never) */
    uint32_t reserved;
} mirt_stats;

/* --------------------------------------------------------------- functions */

/* Version of this ABI (bumped on any incompatible change). */
uint32_t mirt_abi_version(void);

/* Identifies the sources this library was built from: the first 16 hex digits of the SHA-256 of its kernel
 * and header files (ray_tracer_amd/Makefile computes it).  No counterpart in the reference; measurement
 * records (profiles/, bench.py) carry it so that counters are never read beside a different build's timings. */
const char* mirt_build_id(void);

/* Thread-local text for the last error on this thread ("" if none). */
const char* mirt_last_error(void);

/* Number of HIP devices visible (0 when there is none; never fails). */
int mirt_device_count(void);

/* Copy a flattened scene to device `device` (-1 = current).  Validates every
 * index.  Replaces what BoundingVolumeHierarchy owns in the reference
 * (src/bvh.rs:11-27): the sorted shapes and the member array. */
int mirt_scene_create(const mirt_scene_desc* desc, int device, mirt_scene** out);
void mirt_scene_destroy(mirt_scene* scene);

/* Frees the per-pass buffers a render call left with the scene for the next one (by default up to
 * 13 GB, see mirt_render_opts.paths_per_pass) and mirt_render's accumulation image (27 bytes per
 * pixel of the frame); the next render call allocates them again.  For
 * processes that hold several scenes or need the device memory for something else between frames. */
int mirt_scene_trim(mirt_scene* scene);

/* Closest hit of each of `n` rays: Hittable::hit(&ray, t_min, t_max) on the
 * BVH (use_bvh != 0; src/bvh.rs:83-86) or on the linear shape list
 * (use_bvh = 0; src/hittable.rs:698-710).  rays/hits are HOST arrays. */
int mirt_intersect(const mirt_scene* scene, const mirt_ray* rays, size_t n,
                   double t_min, double t_max, int use_bvh, int traversal,
                   mirt_hit* hits);

/* Same, on DEVICE arrays (no copies; asynchronous work is finished on return). */
int mirt_intersect_device(const mirt_scene* scene, const mirt_ray* d_rays, size_t n,
                          double t_min, double t_max, int use_bvh, int traversal,
                          mirt_hit* d_hits);

/* render_world (src/renderer.rs:9-62): fills out_rgb (HOST, width*height*3,
 * RGB, row-major, top row first) for the pixel range in `opts` (all pixels by
 * default).  max_depth is the reference's hard-coded 50 (renderer.rs:83);
 * seed keys the counter-based sampler (the reference seeds from entropy,
 * renderer.rs:30,47, so its streams are not reproducible).  opts and stats
 * may be NULL. */
int mirt_render(const mirt_scene* scene, const mirt_camera* camera,
                const double background[3], uint32_t width, uint32_t height,
                uint32_t spp, uint32_t max_depth, uint64_t seed,
                const mirt_render_opts* opts, uint8_t* out_rgb, mirt_stats* stats);

/* The sample loop of `render` (src/renderer.rs:77-84) without its tail: adds
 * the radiance of samples [sample_begin, sample_end) of pixels
 * [pixel_begin, pixel_end) into d_accum (DEVICE, width*height*3 doubles,
 * pixel-major RGB; the caller zeroes it).  Lets several devices split one
 * image by samples and sum their buffers before mirt_resolve. */
int mirt_render_accumulate(const mirt_scene* scene, const mirt_camera* camera,
                           const double background[3], uint32_t width, uint32_t height,
                           uint32_t spp, uint32_t max_depth, uint64_t seed,
                           const mirt_render_opts* opts, double* d_accum, mirt_stats* stats);

/* An accumulation image for mirt_render_accumulate / mirt_resolve on `device` (-1 = current) for
 * hosts that do not link the HIP runtime themselves (the reference's host is Rust): allocate (zeroed),
 * free, copy to and from HOST memory (width*height*3 doubles, pixel-major RGB).  With these a host
 * renders a frame in instalments -- samples [a, b) now, [b, c) later, even in another process after
 * writing the doubles to a file -- and gets the sums of the one-call render bit for bit: samples are
 * added to a pixel in sample order whatever the instalments (renderer.rs:77-84). */
int mirt_accum_alloc(uint32_t width, uint32_t height, int device, double** d_accum);
void mirt_accum_free(double* d_accum, int device);
int mirt_accum_read(const double* d_accum, uint32_t width, uint32_t height, int device, double* host);
int mirt_accum_write(double* d_accum, uint32_t width, uint32_t height, int device, const double* host);

/* mirt_render_accumulate for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, processes whose memory is budgeted up front).  No counterpart in the
 * reference.  It allocates nothing; it is NOT fit for HIP graph capture: it works on the null stream,
 * synchronises it, copies a few words back to the host and queries the kernel's occupancy.
 * Same arguments, same sums in d_accum bit for bit; it calls no hipMalloc / hipFree: the radiance
 * records of a pass live in an array of MIRT_STATIC_PASS_PATHS records (+16 KB of counters) that every
 * device the library runs on holds from load until the process ends.  Limits: the pixels selected by
 * opts (range and tile split) must number at most MIRT_STATIC_PASS_PATHS (split larger frames into
 * pixel ranges), use_bvh = 1 with MIRT_TRAVERSE_FAST, handover_bounce = 0, profile = 0; anything else
 * returns MIRT_ERR_UNSUPPORTED.  Passes are at most MIRT_STATIC_PASS_PATHS paths, so a large frame
 * takes more launches than mirt_render_accumulate gives it.  All calls on one device share the one
 * array: calls from one process are serialised per device by the library; nothing keeps the contents
 * between calls. */
#define MIRT_STATIC_PASS_PATHS (1u << 20)
int mirt_render_accumulate_static(const mirt_scene* scene, const mirt_camera* camera,
                                  const double background[3], uint32_t width, uint32_t height,
                                  uint32_t spp, uint32_t max_depth, uint64_t seed,
                                  const mirt_render_opts* opts, double* d_accum, mirt_stats* stats);

/* The tail of `render` + the byte cast of render_world (src/renderer.rs:85-89,
 * 42): pixel / spp, sqrt per channel, * 255.99, saturating truncation to u8
 * (NaN -> 0).  d_accum is DEVICE memory, out_rgb HOST memory; pixels
 * [pixel_begin, pixel_end) (pixel_end = 0 means width*height) are written at
 * their own offsets in out_rgb. */
int mirt_resolve(const double* d_accum, uint32_t width, uint32_t height, uint32_t spp,
                 uint32_t pixel_begin, uint32_t pixel_end, int device, uint8_t* out_rgb);

/* Same for the tile subset (tile_size, tile_count, tile_index) of the range, as in
 * mirt_render_opts: only the selected pixels of out_rgb are written. */
int mirt_resolve_tiles(const double* d_accum, uint32_t width, uint32_t height, uint32_t spp,
                       uint32_t pixel_begin, uint32_t pixel_end, uint32_t tile_size, uint32_t tile_count,
                       uint32_t tile_index, int device, uint8_t* out_rgb);

/* ---- one frame over several devices from one process.  The reference spreads the pixels of a
 * frame over the threads of one process (rayon, src/renderer.rs:27-43); mirt_render_multi spreads
 * them, or the samples, over devices: one host thread drives each entry of `scenes` on that
 * scene's device. */
enum {
    /* Rank r renders tiles r, r + n, r + 2n, ... of the pixel order (tile = tile_size consecutive
     * pixels, default one image row) and copies them into out_rgb; devices exchange nothing.  The
     * bytes are those of mirt_render on one device. */
    MIRT_SPLIT_TILE = 0,
    /* Rank r renders every pixel for samples [r * spp / n, (r + 1) * spp / n) into an f64 image on
     * its device; the images are summed slice-wise (each rank sums and resolves 1/n of the pixels)
     * and the bytes gathered into out_rgb.  f64 sums in another order than on one device: a byte
     * may differ from mirt_render's by one level. */
    MIRT_SPLIT_SAMPLE = 1
};
enum {
    /* slices fetched with peer copies (peer access is queried and enabled per pair of devices; a pair
     * without it is served through host memory) and added in rank order: the result does not depend
     * on timing; scenes may share a device */
    MIRT_REDUCE_PEER = 0,
    /* ncclReduceScatter (f64, sum) of RCCL, loaded from librccl.so on first use; every scene must be
     * on a device of its own */
    MIRT_REDUCE_RCCL = 1
};

typedef struct mirt_multi_opts {
    uint32_t struct_size; /* = sizeof(mirt_multi_opts) */
    int32_t split;        /* MIRT_SPLIT_* */
    int32_t reduce;       /* MIRT_REDUCE_*; sample split only */
    uint32_t tile_size;   /* tile split: pixels per tile; 0 = width (one row) */
} mirt_multi_opts;

/* render_world (src/renderer.rs:9-62) on the devices of scenes[0 .. n_scenes): every entry is the
 * SAME scene description created on the device that is to take part (mirt_scene_create(desc,
 * device)); an entry may repeat, or several entries may share a device (they then take turns on
 * it).  At most 16 entries.  opts: traversal, pass size, handover as for mirt_render; its pixel,
 * sample and tile fields must be zero.  multi may be NULL (tile split).  stats: n_scenes entries,
 * one per rank, or NULL.  out_rgb as for mirt_render. */
int mirt_render_multi(const mirt_scene* const* scenes, uint32_t n_scenes, const mirt_camera* camera,
                      const double background[3], uint32_t width, uint32_t height, uint32_t spp,
                      uint32_t max_depth, uint64_t seed, const mirt_render_opts* opts,
                      const mirt_multi_opts* multi, uint8_t* out_rgb, mirt_stats* stats);

/* Frees the device memory mirt_render_multi's sample split keeps between calls (per rank: its f64 image,
 * a staging area and the bytes of its slice).  No counterpart in the reference. */
int mirt_multi_trim(void);

/* Camera::ray (src/camera.rs:65-88) for sample `sample` of pixel `pixel`
 * (idx of renderer.rs:74-75) for n consecutive (pixel, sample) pairs starting
 * at pixel_begin/sample: the rays a render with the same seed starts from.
 * rays is a HOST array of (pixel_end - pixel_begin) entries. */
int mirt_camera_rays(const mirt_camera* camera, uint32_t width, uint32_t height,
                     uint32_t pixel_begin, uint32_t pixel_end, uint32_t sample,
                     uint64_t seed, int device, mirt_ray* rays);

#ifdef __cplusplus
}
#endif
#endif /* MIRT_H */
