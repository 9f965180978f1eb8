/* rt_abi.h — C ABI of librt_mi355x.so, the MI355X (gfx950) path-tracing backend that replaces the BODY of
 * the reference's `Camera::render` (reference: src/camera/mod.rs:54-117) and everything it calls.
 *
 * The reference (arnfaldur/raytracer, Rust) has no FFI/plugin API of its own; the narrowest seam is the
 * method `Camera::render(&self, world, sender)` (src/camera/mod.rs:54-58), reached through
 * `Scene::render` (src/scene.rs:30-32).  A Rust `extern "C"` block binding exactly these entry points,
 * plus the `describe()` flattening the host needs, is shown in INTEGRATION.md.  Each entry point below cites
 * the reference interface it replaces.
 *
 * Conventions: plain C, POD structs, little-endian, no torch / C++ types.  Every function returns an
 * rt_status (0 = OK, <0 = error) unless stated otherwise and never unwinds, throws or aborts across the
 * boundary.  `rt_last_error(ctx)` returns a UTF-8 message owned by the ctx.  A ctx is bound to ONE GPU and
 * is NOT thread-safe (one ctx per calling thread; the reference calls render from one dedicated thread,
 * src/main.rs:50-58).  Multi-GPU: either the rt_multi_* entry points below (one call, library-owned contexts and
 * host threads, frames summed over xGMI inside the library), or one process + one ctx per GPU, each rendering a
 * disjoint residue class of the per-pixel sample indices with the caller summing the partial frames (RCCL reduce:
 * what bench.py does).
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2 /* 2: rt_stats grew samples_nan / samples_clamped / lib_alloc_calls; rt_render_batch_device */
#define RT_NOALLOC_MAX_PIXELS (1u << 24) /* largest frame rt_render_device_noalloc accepts (width * height); no buffer imposes it
                                          * any more (the fixed scratch is 1 KiB of control words): kept for interface stability */

typedef enum rt_status {
  RT_OK = 0,
  RT_E_INVALID_ARG = -1, /* reference: the expect()/panic! sites of builder.rs:58-72 */
  RT_E_HIP = -2,         /* a hip* call failed (message carries hipGetErrorString) */
  RT_E_OOM = -3,
  RT_E_UNSUPPORTED = -4,
  RT_E_INTERNAL = -5
} rt_status;

typedef struct rt_ctx rt_ctx;     /* one GPU: device id, stream, work buffers */
typedef struct rt_scene rt_scene; /* device-resident flattened world (prims, BVH, materials, textures) */

/* ---- world description (host side flattening of the Hittable tree) ------------------------------------
 * Replaces: the `Box<dyn Hittable>` world handed to Camera::render (src/camera/mod.rs:56) — Sphere /
 * MovingSphere (src/hittable/geometry.rs:12-17,84-87) inside HittableList / BVHNode
 * (src/hittable/containers.rs:13-16,48-52).  Containers are NOT described: closest-hit semantics make
 * the result independent of BVH topology (containers.rs:29-40,177-191), so the library builds its own
 * acceleration structure from the flat primitive list. */
typedef enum rt_prim_type {
  RT_PRIM_SPHERE = 0,        /* geometry.rs:12-81: a=center, radius (negative allowed: flips the normal) */
  RT_PRIM_MOVING_SPHERE = 1, /* geometry.rs:84-114: a=center at time 0, b=destination at time 1 */
  RT_PRIM_QUAD = 2,          /* EXT (not in reference, SURVEY aX1): a=Q, b=u, c=v */
  RT_PRIM_MEDIUM_SPHERE = 3, /* EXT (aX3): constant medium bounded by sphere a/radius, `density` */
  RT_PRIM_MEDIUM_BOX = 4     /* EXT (aX3): constant medium bounded by the box a + s*b + t*c + w*d, s,t,w in [0,1] */
} rt_prim_type;

typedef struct rt_prim {
  uint32_t type;        /* rt_prim_type */
  uint32_t material_id; /* index into rt_scene_desc.materials */
  int32_t object_id;    /* caller's id, reported by rt_intersect (insertion order of world.add) */
  uint32_t reserved;
  double a[3];
  double b[3];
  double c[3];
  double d[3];
  double radius;
  double density;
} rt_prim;

typedef enum rt_material_kind {
  RT_MAT_LAMBERTIAN = 0, /* materials.rs:14-16,28-43: albedo = texture `tex` */
  RT_MAT_METAL = 1,      /* materials.rs:59-80: albedo rgb, fuzz */
  RT_MAT_DIELECTRIC = 2, /* materials.rs:89-129: ior */
  RT_MAT_DIFFUSE_LIGHT = 3, /* EXT (aX4): emits texture `tex`, never scatters */
  RT_MAT_ISOTROPIC = 4      /* EXT (aX3): uniform phase function, albedo = texture `tex` */
} rt_material_kind;

typedef struct rt_material {
  uint32_t kind; /* rt_material_kind */
  int32_t tex;   /* texture index (LAMBERTIAN, DIFFUSE_LIGHT, ISOTROPIC) or -1 */
  double albedo[3];
  double fuzz;
  double ior;
} rt_material;

typedef enum rt_texture_kind {
  RT_TEX_SOLID = 0,   /* texture.rs:20-36 */
  RT_TEX_CHECKER = 1, /* texture.rs:39-66: inv_scale, odd / even = child texture indices */
  RT_TEX_IMAGE = 2,   /* texture.rs:69-93: RGBA8 texels at texel_offset (bytes) in the blob, width x height */
  RT_TEX_NOISE = 3    /* texture.rs:96-154: inv_scale; grey Hermite lattice noise (noise_at variant) */
} rt_texture_kind;

typedef struct rt_texture {
  uint32_t kind; /* rt_texture_kind */
  int32_t odd;   /* CHECKER: texture used when (x+y+z) rem_euclid 2 == 0 (texture.rs:60-61) */
  int32_t even;
  uint32_t width, height; /* IMAGE */
  uint32_t reserved;
  uint64_t texel_offset; /* IMAGE: byte offset into rt_scene_desc.texels, RGBA8 row-major */
  double rgb[3];         /* SOLID */
  double inv_scale;      /* CHECKER, NOISE */
} rt_texture;

typedef struct rt_scene_desc {
  const rt_prim* prims;
  uint64_t n_prims;
  const rt_material* materials;
  uint64_t n_materials;
  const rt_texture* textures;
  uint64_t n_textures;
  const uint8_t* texels; /* may be NULL when no IMAGE texture */
  uint64_t n_texel_bytes;
} rt_scene_desc;

/* ---- camera block -------------------------------------------------------------------------------------
 * Replaces: the private fields of `Camera` (src/camera/mod.rs:26-51) as derived by
 * CameraBuilder::build (src/camera/builder.rs:84-107). */
typedef enum rt_sampler_kind {
  RT_SAMPLER_UNIFORM = 0, /* PixelSampler::Uniform(n): n x n fixed grid, no jitter (mod.rs:150-163) */
  RT_SAMPLER_RANDOM = 1   /* PixelSampler::Random(n): n jittered samples (mod.rs:164-172) */
} rt_sampler_kind;

typedef struct rt_camera {
  uint32_t width, height;
  uint32_t sampler_kind; /* rt_sampler_kind */
  uint32_t sampler_n;    /* grid side (UNIFORM, spp = n*n) or sample count (RANDOM, spp = n) */
  uint32_t max_depth;    /* `depth` (mod.rs:30): rays at depth >= max_depth return black (mod.rs:195-197) */
  uint32_t background_mode; /* 0 = sky gradient of mod.rs:205-207 (reference), 1 = solid `background` (EXT aX4) */
  double center[3];
  double pixel00[3];
  double du[3], dv[3];               /* pixel_delta_u / pixel_delta_v */
  double defocus_u[3], defocus_v[3]; /* defocus_disk_u / defocus_disk_v */
  double defocus_angle;              /* <= 0: pinhole (mod.rs:178) */
  double background[3];
} rt_camera;

/* ---- parity probe records ----------------------------------------------------------------------------- */
typedef struct rt_ray {
  double o[3], d[3], time; /* src/ray.rs:7-11 */
} rt_ray;
typedef struct rt_hit { /* src/hittable/mod.rs:31-39 (material replaced by object_id) */
  int32_t object_id;    /* -1 = miss */
  int32_t front_face;
  double t;
  double p[3];
  double n[3];
  double u, v;
} rt_hit;

typedef struct rt_stats {
  uint64_t paths;          /* samples rendered by the last rt_render* call */
  uint64_t rays;           /* rays traced (all bounces) */
  uint32_t passes;         /* sample batches the render was split into */
  uint32_t bounces_run;    /* bounce iterations launched */
  float ms_total;          /* hipEvent time of the whole render on the library's stream */
  float ms_generate, ms_intersect, ms_shade, ms_accumulate; /* per stage (only when profiling is enabled) */
  uint64_t intersect_launches;
  uint64_t probe_rays, probe_node_visits, probe_prim_tests; /* rt_intersect traversal counters since rt_init */
  float dbg[16]; /* "fused_stats" diagnostic build: lane-work / wave-trip tallies (nodes, leaves, shade, regen) */
  /* Samples whose radiance did not reach the frame as computed (fused mode; the reference has no counterpart: its f64
   * sums carry NaN / inf into the pixel, color.rs:68-73 then quantises NaN to 0): a NaN or negative-zero-or-less radiance
   * channel adds nothing, a channel above 2^20 is added as 2^20.  Both are zero for every frame of the shipped scenes
   * (tests/test_gpu_parity.py); a caller that sees them non-zero knows that the frame dropped energy. */
  uint64_t samples_nan;     /* radiance channels that were NaN (added as 0) */
  uint64_t samples_clamped; /* radiance channels above 2^20 (added as 2^20) */
  uint64_t dbg_counts[12];  /* "fused_stats" build, exact: [0] child boxes tested (node steps x 2; x 4 on the four-wide tree),
                             * [1..5] primitive tests by type (static sphere, moving sphere, quad, medium, merged axis-aligned
                             * box; tree leaves + primitives kept outside the tree), [6..8] node-loop trips a lane sat out
                             * waiting at a leaf / finished / without a path */
  uint64_t lib_alloc_calls; /* hipMalloc / hipFree / stream / event creations this library has made in this process since
                             * it was loaded (all contexts): unchanged across a call <=> that call allocated nothing */
} rt_stats;

/* ---- entry points --------------------------------------------------------------------------------------*/

/* Create a context on HIP device `device_id`.  No reference counterpart (the reference is CPU-only; the
 * closest analogue is thread::available_parallelism() at src/camera/mod.rs:63). */
int rt_init(int device_id, rt_ctx** out);
int rt_shutdown(rt_ctx* ctx);
/* Message of the last failing call on this ctx ("" if none).  rt_last_error(NULL) returns the message of the
 * last failing rt_init. */
const char* rt_last_error(rt_ctx* ctx);
int rt_abi_version(void);

/* Upload a flattened world: validates it, builds the acceleration structure (SAH BVH2, host) and copies
 * everything into HBM.  The library copies; the caller keeps ownership of every pointer in `desc`.
 * Replaces: HittableList::into_bvh / BVHNode::from_vec (src/hittable/containers.rs:23-25,54-152). */
int rt_scene_upload(rt_ctx* ctx, const rt_scene_desc* desc, rt_scene** out);
int rt_scene_free(rt_ctx* ctx, rt_scene* scene);

/* Closest hit of `n` host rays over t in (1e-6, inf), exclusive (src/camera/mod.rs:198).
 * Replaces: `world.hit(ray, &(0.000001..f64::INFINITY))` = Hittable::hit (src/hittable/mod.rs:27) through
 * BVHNode::hit / HittableList::hit / Sphere::calculate_hit / MovingSphere::hit.  Parity probe. */
int rt_intersect(rt_ctx* ctx, const rt_scene* scene, const rt_ray* rays, uint64_t n, rt_hit* out_hits);

/* Render samples {s = sample_begin + k*sample_stride < sample_end} of every pixel and write the
 * UN-NORMALISED linear RGB sums (row-major, W*H*3 f32) to `out_sum`.  For the UNIFORM sampler s enumerates the
 * grid cells (yi, xi) = (s / n, s % n); for RANDOM s = 0..n-1.  `seed` keys the counter-based RNG: Philox4x32-10 is
 * evaluated once per path with counter = (pixel, sample index, 0, 0) and key = seed; two of its words feed the lens
 * sample and two seed the path's own xoroshiro64** stream for all later draws.  A path therefore depends only on
 * (seed, pixel, sample index), so the sum over any partition of the sample indices equals the single-call result up
 * to f32 summation order.
 * Replaces: the body of Camera::render / render_rect / sample_pixel / sample_point / ray_color
 * (src/camera/mod.rs:54-214).  rt_render takes a HOST output pointer (blocking; includes the D2H copy);
 * rt_render_device takes a DEVICE pointer and a hipStream_t (NULL = the ctx's stream) and is asynchronous
 * with respect to the host.
 * Frame completeness: the persistent kernel's scheduler has a watchdog (a wave that finds no work for ~10 s gives up and
 * raises a flag instead of hanging the GPU; the tiles it skipped keep whatever `d_out_sum` held).  The blocking entry
 * points (rt_render, rt_multi_render, rt_multi_render_device) check the flag and return RT_E_INTERNAL for such a frame.
 * Callers of the asynchronous ones (rt_render_device, rt_render_device_noalloc, rt_render_batch_device) MUST call
 * rt_get_stats after synchronising to learn whether the frame is complete: it returns RT_E_INTERNAL when the flag is set.
 * Streams: work of one ctx is ordered through the stream it is issued on; calls on a ctx must not overlap in time on
 * different streams (tile-order feedback from the previous frame is only used when that frame ran on the same stream). */
int rt_render(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
              uint32_t sample_end, uint32_t sample_stride, float* out_sum);
int rt_render_device(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
                     uint32_t sample_end, uint32_t sample_stride, float* d_out_sum, void* hip_stream);

/* rt_render_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arguments, same result bit for bit, but
 * every scratch buffer of the render (the camera block of the frame in flight, the ray counter, the tile counter the
 * persistent workgroups claim their tiles from, the scheduler's watchdog flag) lives in a fixed file-scope device
 * array inside the library instead of memory the library allocates: the call performs no hipMalloc / hipFree and
 * creates no stream or event.  Limits: fused mode only (option "mode" = 1, the default); frames of at most
 * RT_NOALLOC_MAX_PIXELS = 2^24 pixels (covers 4K; the array is 1 KiB per GPU, held from library load until the
 * process ends — partial sums of a tile live in LDS, so nothing frame-sized is needed); larger frames return
 * RT_E_UNSUPPORTED.  Every ctx on a GPU shares that GPU's array: calls that overlap in time on one GPU race (the
 * library clears what must start at zero, stream-ordered, at the start of each call). */
int rt_render_device_noalloc(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
                             uint32_t sample_end, uint32_t sample_stride, float* d_out_sum, void* hip_stream);

/* Several frames of ONE scene through ONE camera in a single persistent launch (animation frames / progressive passes /
 * the per-GPU slices of successive frames of a multi-GPU job): frame i renders the sample indices {sample_begins[i] + k *
 * sample_stride < sample_end} with key seeds[i] into d_out_sums[i] (device pointers, W*H*3 f32 each, may not overlap).
 * Every frame must select the same non-zero number of sample indices (else RT_E_INVALID_ARG: split the batch).  Each
 * frame is bit-identical to the rt_render_device frame of the same arguments; what changes is the time — the tile list is
 * the frames' lists back to back, so the next frame's tiles are handed out while the deepest paths of the previous frame
 * still bounce, and the per-frame drain (~0.4 ms at any spp) is paid once per batch.  Fused mode only; asynchronous on
 * `hip_stream` (NULL = the ctx's stream) like rt_render_device; allocates (grow-only) the frame table.
 * Replaces: successive calls of Camera::render (src/camera/mod.rs:54-58) on one world — no counterpart as one call. */
int rt_render_batch_device(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint32_t n_frames, const uint64_t* seeds,
                           const uint32_t* sample_begins, uint32_t sample_end, uint32_t sample_stride, float* const* d_out_sums,
                           void* hip_stream);

/* mean = sum / spp, gamma correction powf(1/gamma) per channel, optional 8-bit quantisation
 * `(256.0 * x) as u8` (saturating, NaN -> 0).  Either output may be NULL.  Host pointers.
 * Replaces: `accumulator / n` (mod.rs:162,171), Color::gamma_corrected (src/color.rs:58-61, gamma 2.2 at
 * mod.rs:134) and Color::into_u8 (src/color.rs:68-73).  `out_rgb` is what the reference's tile messages
 * carry (gamma-corrected f64, row-major). */
int rt_resolve(rt_ctx* ctx, const float* sum, uint32_t width, uint32_t height, uint32_t spp, double gamma,
               double* out_rgb, uint8_t* out_rgb8);
/* Same on device pointers (f32 sum in, f64 / u8 out), asynchronous on `hip_stream` (NULL = ctx stream). */
int rt_resolve_device(rt_ctx* ctx, const float* d_sum, uint32_t width, uint32_t height, uint32_t spp, double gamma,
                      double* d_out_rgb, uint8_t* d_out_rgb8, void* hip_stream);

/* Counters / timings of the last render on this ctx.  Replaces: the wall-clock print of src/main.rs:51-57. */
int rt_get_stats(rt_ctx* ctx, rt_stats* out);

/* Tunables.  "mode" (1 = fused persistent kernel, default; 0 = wavefront kernels), "max_paths" (wavefront: paths
 * in flight per pass), "profile" (wavefront: 1 = per-stage hipEvent timing, serialises stages), "fused_ku" (samples
 * per work unit, 0 = auto), "fused_lds" (1 = keep the scene in LDS when it fits), "fused_resume" (traversal
 * schedule: 0 bounce-synchronous, 1 resumable, 2 = by scene size, default) with "fused_kb" / "fused_kn" /
 * "fused_kr" (hand-back, node-loop exit and regeneration thresholds, 1..64; defaults 16 / 8 / by residency: 16 for scenes kept in LDS, 12 for scenes read through the caches), "fused_wide" / "fused_half" (scenes
 * that do not fit LDS: 1 = four-wide tree / its binary16 form with the top levels cached in LDS, defaults; 0 = the
 * binary tree / f32 planes, kept as cross-checks), "fused_tail4" / "fused_tail2" (height, in per mille of the frame's rows, of the bands of 4x4 / 2x2 tiles at the
 * end of the tile list that keep the frame's tail short; 50 / 30; never changes the frame), "fused_slots" (tiles a workgroup keeps in flight,
 * 0 = auto), "fused_partner_min" (partner workgroups from this many Mi samples per frame), "merge_boxes" (1 = default, read by
 * rt_scene_upload: six consecutive quads of one material that are the faces of an axis-aligned box enter the tree as ONE
 * primitive; a hit still reports the face's quad and the same t), "dbg_watchdog_polls" (diagnostic: polls before a wave with
 * nothing to do gives up, 0 = 2^24), "global_prims" (1 = default, read by rt_scene_upload: spheres and spherical media whose box is a quarter of the scene's are kept out of
 * the tree and tested first by every ray, and scenes of at most 8 primitives get no tree at all; hits unchanged, frames unchanged unless the primitive is a medium, whose
 * free-path draw then comes first in the ray's random stream), "fused_kr_sync" (regeneration threshold of the bounce-synchronous schedule, 1..64, default 8), "fused_feedback" (1 = default: a frame of
 * the same scene and camera as the last one hands out first the tiles whose paths went deepest then; frames unchanged), "fused_partner" (1 = default: sphere scenes resident in LDS get a second, 256-thread
 * workgroup per CU of the HBM-resident instance; the frame is unchanged), "probe_tree" (which traversal rt_intersect runs: 0 the binary tree in HBM, plain loop, default;
 * 1 four-wide f32; 2 four-wide binary16; 3 and 4 the PRODUCTION forms under the render loop's own schedule — resumable
 * traversals, 16-bit sentinel stack — 3 = the scene copied into LDS as the render kernels keep it (slab nodes), 4 = the
 * four-wide binary16 tree with its top cached in LDS (partner workgroups / HBM-resident instances); with 3 and 4 rt_hit.n
 * and rt_hit.front_face are the render path's own f32 shading normal and front test; "dbg_max_cached" (test hook: at most that many four-wide nodes are kept in LDS, the rest is read from memory; 0 = as many as fit; frames unchanged), "uv_tables" (read at scene upload, default 1: image textures get a table of their texel boundaries, against which sphere hits check an f32 guess of the texel instead of evaluating the f64 acos / atan2 of get_sphere_uv; a failed check evaluates the formulas; 0 = the formulas always; frames identical), "bvh_optimise" (read at scene upload: rounds of the reinsertion pass over the binned-SAH tree, 0..16, default 3; the tree's depth never grows and hits do not depend on it); "probe_count" = 1 makes 3 and 4 run the
 * instrumented C++ traversal that fills rt_stats.probe_node_visits / probe_prim_tests, 0 = default: the render kernels' own
 * instantiation, for sphere scenes in LDS the hand-scheduled asm traversal), "fused_stats" (1 = instrumented kernel, fills rt_stats.dbg).  Thresholds, unit size and tile order never change the frame, and
 * the kernel variants give bit-identical frames (FP contraction is fixed per source expression in all device code).  Unknown keys and out-of-range values
 * return RT_E_INVALID_ARG.  No reference counterpart (constants are edited in main.rs:28-43). */
int rt_set_option(rt_ctx* ctx, const char* key, int64_t value);

/* Host-side reference implementation of the counter-based generator used on the device (Philox4x32-10),
 * exported so tests can check it against the published Random123 known-answer vectors. */
void rt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Host-side rounding used when a scene's four-wide tree is encoded with binary16 planes: the binary16 bit pattern
 * of x rounded towards +infinity (up != 0) or -infinity.  Exported so tests can check that encoded boxes can only
 * grow.  No reference counterpart. */
uint32_t rt_half_round_bits(float x, int up);

/* Synchronise the ctx's stream (for callers of the *_device entry points). */
int rt_synchronize(rt_ctx* ctx);

/* ---- several GPUs of one node behind one call -----------------------------------------------------------
 * Replaces: the same blocking seam, Camera::render (src/camera/mod.rs:54-58) reached through Scene::render
 * (src/scene.rs:30-32), for a caller with N GPUs (no counterpart in the reference, which is one process on CPU
 * threads; its pixel is a plain sum over independent samples, mod.rs:151-172, which is what makes the split exact).
 * An rt_multi owns one rt_ctx and one persistent host thread per listed device.  rt_multi_render renders, on device i
 * of n, the sample indices {sample_begin + (i + k n) * sample_stride < sample_end} of every pixel, then sums the n
 * partial f32 frames on the FIRST device of the list with one kernel that reads the other devices' frames in place
 * over xGMI (peer access; seven dedicated links feed the root concurrently), and returns the un-normalised sums like
 * rt_render.  The result equals the single-device frame up to f32 summation order.  A device may be listed more than
 * once (several contexts on one GPU: how the one-GPU tests drive this path).  Not thread-safe: one caller at a time. */
typedef struct rt_multi rt_multi;
typedef struct rt_multi_stats {
  uint32_t n_devices, reserved;
  uint64_t paths, rays;          /* over all devices, last render */
  float ms_render[16];           /* hipEvent time of each device's render */
  float ms_reduce, ms_copy;      /* the summing kernel on the root device / the copy of the result to the caller */
} rt_multi_stats;
int rt_multi_init(const int* device_ids, int n_devices, rt_multi** out); /* 1..16 devices; the first is the root */
int rt_multi_shutdown(rt_multi* m);
const char* rt_multi_last_error(rt_multi* m); /* NULL: message of the last failing rt_multi_init */
int rt_multi_device_count(rt_multi* m);
/* rt_scene_upload on every device (replaces the previous scene); the caller keeps ownership of `desc`. */
int rt_multi_scene_upload(rt_multi* m, const rt_scene_desc* desc);
/* HOST output pointer (W*H*3 f32), blocking. */
int rt_multi_render(rt_multi* m, const rt_camera* cam, uint64_t seed, uint32_t sample_begin, uint32_t sample_end,
                    uint32_t sample_stride, float* out_sum);
/* Output pointer in the ROOT device's memory, blocking (the sum stays on the GPU: for rt_resolve_device on it). */
int rt_multi_render_device(rt_multi* m, const rt_camera* cam, uint64_t seed, uint32_t sample_begin, uint32_t sample_end,
                           uint32_t sample_stride, float* d_out_sum_root);
int rt_multi_set_option(rt_multi* m, const char* key, int64_t value); /* rt_set_option on every device */
int rt_multi_get_stats(rt_multi* m, rt_multi_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
