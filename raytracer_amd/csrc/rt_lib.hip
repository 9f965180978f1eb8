// rt_lib.hip — librt_mi355x.so: HIP kernels (gfx950) + the C ABI of include/rt_abi.h.
//
// Wavefront pipeline replacing the body of Camera::render (reference src/camera/mod.rs:54-214):
//   k_generate   sample_pixel + sample_point            (mod.rs:143-186, 211-214)
//   k_intersect  world.hit: BVH traversal, LDS stack    (containers.rs:177-191, aabb.rs:48-71, geometry.rs:36-71)
//   k_shade      ray_color_inner body: scatter / sky    (mod.rs:195-207, materials.rs, texture.rs)
//                + wave64 ballot/prefix compaction of surviving rays into the next queue
//   k_accumulate accumulator += sample                  (mod.rs:159,169)
//   k_resolve    / n, gamma, into_u8                    (mod.rs:162,171,134; color.rs:58-73)
// All queues live in HBM as structure-of-arrays; every path of a pass is in flight at once.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <map>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_internal.h"

std::atomic<uint64_t> rti::g_alloc_calls{0};

using namespace rtd;

// ================================================================================================ kernels

struct PassParams {
  uint32_t n_paths;  // paths in this pass = n_pix_pass * S
  uint32_t pix0;     // first pixel of the pass
  uint32_t S;        // samples per pixel in this pass
  uint32_t s_first;  // sample index of k = 0
  uint32_t s_stride;
  uint32_t k0, k1;  // Philox key (seed)
};

__device__ __forceinline__ GlobalAcc global_acc(const Scene& sc) { return GlobalAcc{sc.nodes, sc.prims}; }
__device__ __forceinline__ SphereAcc sphere_acc(const Scene& sc) {
  return SphereAcc{reinterpret_cast<const float4*>(sc.nodes), sc.spheres};
}

// ------------------------------------------------------------------------------------------------ wavefront mode
__global__ __launch_bounds__(RT_BLOCK) void k_generate(Camera cam, Queue q, float* __restrict__ Lr,
                                                       float* __restrict__ Lg, float* __restrict__ Lb, PassParams pp,
                                                       uint32_t* __restrict__ counts) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < pp.n_paths; i += stride) {
    uint32_t pl = i / pp.S, k = i - pl * pp.S;
    uint32_t pixel = pp.pix0 + pl, s = pp.s_first + k * pp.s_stride;
    d3 o, d;
    double time;
    PathRng rng;
    generate_ray(cam, pixel / cam.width, pixel % cam.width, s, pp.k0, pp.k1, o, d, time, rng);
    q.r0[i] = rng.s0; q.r1[i] = rng.s1;
    q.ox[i] = o.x; q.oy[i] = o.y; q.oz[i] = o.z;
    q.dx[i] = d.x; q.dy[i] = d.y; q.dz[i] = d.z;
    q.time[i] = time;
    q.tr[i] = 1.f; q.tg[i] = 1.f; q.tb[i] = 1.f;
    q.id[i] = i;
    Lr[i] = 0.f; Lg[i] = 0.f; Lb[i] = 0.f;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = pp.n_paths;
}

#define RT_N_CLASSES 6
template <bool SPHERES>
__device__ __forceinline__ uint32_t shade_class(const Scene& sc, int32_t prim) {
  if (prim < 0) return 0u;
  const uint32_t mat = SPHERES ? sc.spheres[prim].mat : sc.prims[prim].mat;
  return 1u + min(sc.mats[mat].kind, 4u);
}
template <bool SPHERES>
__global__ __launch_bounds__(RT_BLOCK) void k_intersect(Scene sc, uint32_t features, Queue q, Hits h,
                                                        const uint32_t* __restrict__ count, uint32_t* __restrict__ bins) {
  uint32_t cls_total[RT_N_CLASSES] = {0, 0, 0, 0, 0, 0};  // material binning: this wave's rays per shading class
  __shared__ uint32_t stack_rows[(RT_STACK_DEPTH + 1) * RT_BLOCK];  // row 0: guard row below the columns (trav_pop)
  uint32_t* const stack = stack_rows + RT_BLOCK;
  const uint32_t n = *count;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    d3 o = {q.ox[i], q.oy[i], q.oz[i]}, d = {q.dx[i], q.dy[i], q.dz[i]};
    HitInfo hit;
    PathRng rng = {0, 0};
    const bool media = !SPHERES && (features & RT_F_MEDIUM);  // constant media draw from the path's stream inside hit()
    if (media) { rng.s0 = q.r0[i]; rng.s1 = q.r1[i]; }
    if (SPHERES)
      hit = traverse<SphereAcc, uint32_t, false>(sphere_acc(sc), sc.root, features, o, d, q.time[i], RT_TMIN,
                                                 stack + threadIdx.x, RT_BLOCK, nullptr, rng, global_prims(sc));
    else
      hit = traverse<GlobalAcc, uint32_t, false>(global_acc(sc), sc.root, features, o, d, q.time[i], RT_TMIN,
                                                 stack + threadIdx.x, RT_BLOCK, nullptr, rng, global_prims(sc));
    if (media) { q.r0[i] = rng.s0; q.r1[i] = rng.s1; }
    h.prim[i] = hit.prim;
    h.t[i] = hit.t;
    if (bins) {  // (lanes of a wave may run different numbers of trips: the ballots count whoever is here)
      const uint32_t cls = shade_class<SPHERES>(sc, hit.prim);
#pragma unroll
      for (uint32_t c = 0; c < RT_N_CLASSES; c++) cls_total[c] += (uint32_t)__popcll(__ballot(cls == c));
    }
  }
  // the wave's lowest lane ran every trip the wave ran: its totals are the wave's
  if (bins && (threadIdx.x & 63u) == 0u) {
#pragma unroll
    for (uint32_t c = 0; c < RT_N_CLASSES; c++)
      if (cls_total[c]) atomicAdd(&bins[c], cls_total[c]);
  }
}

// wave64 compaction: ballot -> exclusive prefix via mbcnt -> one atomic per wave
__device__ __forceinline__ uint32_t wave_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t* counter) {
  unsigned long long mask = __ballot(pred);
  uint32_t lane_prefix = wave_prefix(mask);
  uint32_t total = (uint32_t)__popcll(mask);
  uint32_t base = 0;
  if (total) {
    int leader = __ffsll((long long)mask) - 1;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(counter, total);
    base = __shfl(base, leader, 64);
  }
  return base + lane_prefix;
}

// ---- material binning (north_star: "wave ballot / prefix primitives for ray compaction and material binning").
// Between intersect and shade the surviving rays are sorted by what their hit will execute — class 0 = miss (sky /
// background), 1 + material kind otherwise (Lambertian, metal, dielectric, light, isotropic: materials.rs:28-129) — so
// that a wave of k_shade runs ONE material's code with all its lanes instead of up to six in turn.  Counting sort with
// wave64 primitives: per class a ballot, a popcount and one atomic per wave (in k_intersect), an exclusive prefix over
// the six totals (every wave of k_bin_scatter, from the totals), then ballot + mbcnt prefix + one atomic per wave and class to place every ray's
// index (k_bin_scatter).  k_shade then walks the permutation.  Option "wf_bin".  The class totals are counted by
// k_intersect itself (it has the hit in registers), so binning costs one extra launch per bounce.
template <bool SPHERES>
__global__ __launch_bounds__(RT_BLOCK) void k_bin_scatter(Scene sc, Hits h, const uint32_t* __restrict__ count, uint32_t* __restrict__ bins,
                                                          uint32_t* __restrict__ perm) {
  const uint32_t n = *count;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  // Every wave owns a contiguous range of rays.  Pass 1 counts the range's rays per class, ONE atomic per class then
  // reserves the wave's slots in the class's region (a few thousand atomics per bounce in all: appending 64 rays at a
  // time, every wave hammering the same six cursors, took 0.46 ns per ray — 99 ms per C2 frame), pass 2 places the rays.
  const uint32_t per = ((n + n_waves - 1) / n_waves + 63u) & ~63u;
  const uint32_t r0 = min(n, wave * per), r1 = min(n, r0 + per);
  uint32_t cnt[RT_N_CLASSES] = {0, 0, 0, 0, 0, 0};
  for (uint32_t base = r0; base < r1; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t cls = i < r1 ? shade_class<SPHERES>(sc, h.prim[i]) : RT_N_CLASSES;
#pragma unroll
    for (uint32_t c = 0; c < RT_N_CLASSES; c++) cnt[c] += (uint32_t)__popcll(__ballot(cls == c));
  }
  uint32_t at[RT_N_CLASSES];  // next slot of this wave in each class's region
  {
    uint32_t off = 0;  // first slot of the class: exclusive prefix of the totals k_intersect counted
#pragma unroll
    for (uint32_t c = 0; c < RT_N_CLASSES; c++) {
      uint32_t got = 0;
      if (lane == 0 && cnt[c]) got = atomicAdd(&bins[8 + c], cnt[c]);
      at[c] = off + (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
      off += bins[c];
    }
  }
  for (uint32_t base = r0; base < r1; base += 64) {
    const uint32_t i = base + lane;
    const uint32_t cls = i < r1 ? shade_class<SPHERES>(sc, h.prim[i]) : RT_N_CLASSES;
#pragma unroll
    for (uint32_t c = 0; c < RT_N_CLASSES; c++) {
      const unsigned long long m = __ballot(cls == c);
      if (cls == c) perm[at[c] + wave_prefix(m)] = i;
      at[c] += (uint32_t)__popcll(m);
    }
  }
}

template <uint32_t F, bool SPHERES>
__global__ __launch_bounds__(RT_BLOCK) void k_shade(Scene sc, Camera cam, Queue qi, Hits h, Queue qo,
                                                    const uint32_t* __restrict__ count_in,
                                                    uint32_t* __restrict__ count_out, float* __restrict__ Lr,
                                                    float* __restrict__ Lg, float* __restrict__ Lb, PassParams pp,
                                                    uint32_t bounce, uint32_t last_bounce, const uint32_t* __restrict__ perm) {
  const uint32_t n = *count_in;
  const uint32_t stride = gridDim.x * blockDim.x;
  // wave-uniform trip count: every lane of a wave reaches the ballot
  for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < n; base += stride) {
    const uint32_t slot_in = base + (threadIdx.x & 63u);
    // with material binning the wave walks the permutation: neighbouring lanes shade the same material class
    const uint32_t i = (perm && slot_in < n) ? perm[slot_in] : slot_in;
    bool alive = false;
    ShadeOut so;
    uint32_t id = 0;
    double time = 0;
    PathRng rng = {0, 0};
    if (slot_in < n) {
      d3 o = {qi.ox[i], qi.oy[i], qi.oz[i]}, d = {qi.dx[i], qi.dy[i], qi.dz[i]};
      time = qi.time[i];
      f3 thr = {qi.tr[i], qi.tg[i], qi.tb[i]};
      id = qi.id[i];
      HitInfo hit;
      hit.prim = h.prim[i];
      hit.t = h.t[i];
      rng.s0 = qi.r0[i]; rng.s1 = qi.r1[i];
      if (SPHERES) so = shade<F>(sc, sphere_acc(sc), cam, o, d, time, thr, hit, rng);
      else so = shade<F>(sc, global_acc(sc), cam, o, d, time, thr, hit, rng);
      if (!so.alive) {
        Lr[id] = so.radiance.x; Lg[id] = so.radiance.y; Lb[id] = so.radiance.z;
      }
      // a ray scattered at the last allowed depth returns black (mod.rs:195-197): drop it, L stays 0
      alive = so.alive && !last_bounce;
    }
    uint32_t slot = wave_append(alive, count_out);
    if (alive) {
      qo.ox[slot] = so.o.x; qo.oy[slot] = so.o.y; qo.oz[slot] = so.o.z;
      qo.dx[slot] = so.d.x; qo.dy[slot] = so.d.y; qo.dz[slot] = so.d.z;
      qo.time[slot] = time;
      qo.tr[slot] = so.thr.x; qo.tg[slot] = so.thr.y; qo.tb[slot] = so.thr.z;
      qo.id[slot] = id;
      qo.r0[slot] = rng.s0; qo.r1[slot] = rng.s1;
    }
  }
}

// fb[pixel] += sum_k L[pl*S + k] (one wave per pixel: coalesced reads, DPP/shuffle reduction);
// also tallies the rays traced (sum of the per-bounce queue counts)
__global__ __launch_bounds__(RT_BLOCK) void k_accumulate(const float* __restrict__ Lr, const float* __restrict__ Lg,
                                                         const float* __restrict__ Lb, float* __restrict__ fb,
                                                         PassParams pp, uint32_t n_pix_pass,
                                                         const uint32_t* __restrict__ counts, uint32_t n_counts,
                                                         unsigned long long* __restrict__ ray_total) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t pl = wave; pl < n_pix_pass; pl += n_waves) {
    float r = 0.f, g = 0.f, b = 0.f;
    const size_t base = (size_t)pl * pp.S;
    for (uint32_t k = lane; k < pp.S; k += 64) {
      r += Lr[base + k]; g += Lg[base + k]; b += Lb[base + k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      r += __shfl_down(r, off, 64); g += __shfl_down(g, off, 64); b += __shfl_down(b, off, 64);
    }
    if (lane == 0) {
      const size_t p = (size_t)(pp.pix0 + pl) * 3;
      fb[p] += r; fb[p + 1] += g; fb[p + 2] += b;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long t = 0;
    for (uint32_t i = 0; i < n_counts; i++) t += counts[i];
    *ray_total += t;
  }
}

// ------------------------------------------------------------------------------------------------ fused mode
// One persistent kernel.  A WORKGROUP owns whole 8x8-pixel tiles: it claims a tile from a global counter, renders ALL of
// the tile's sample indices itself and writes the tile's 64 x 3 sums to the frame exactly once, with plain stores — no
// frame-wide accumulator, no global atomics, nothing to clear or convert afterwards.  Inside the workgroup the tile's
// samples are handed out to the waves in chunks (64 pixels x Ku sample indices) from a cursor in LDS, radiance is summed
// in the tile's LDS slot as 2^-40 fixed point (integer adds are order independent => bit-reproducible frames, whatever
// wave traces a sample), and the lane that finishes the tile's last path has its wave write the slot out.  A workgroup
// keeps up to RT_MAX_SLOTS tiles in flight (one being handed out, the others waiting for their longest paths), so
// no lane ever idles through a tile's drain.  Per lane: path regeneration from the wave's current chunk (wave64 ballot
// + prefix).  No path state ever touches HBM.
#define RT_MAX_SLOTS 64
#define RT_TILE_LEVELS 3
struct BatchFrame {  // one frame of a batch (device table, read with scalar loads)
  uint32_t k0, k1;   // Philox key (seed)
  uint32_t s_begin;  // first sample index
  uint32_t pad;
  float* out;        // W*H*3 sums
};
struct FusedParams {
  uint32_t K, s_begin, s_stride, Ku;
  // Tile list.  The frame is cut into horizontal bands, each tiled with squares of its own side (8, 4 or 2 pixels):
  // band i holds tiles [lvl_begin[i], lvl_begin[i + 1]) of the list, starts at image row lvl_y0[i] and has lvl_tiles_x[i]
  // tiles of side 1 << lvl_shift[i] per row.  Large tiles come first; the small ones at the end of the list keep the
  // frame's tail short (a workgroup's last tile is then a few hundred paths, not thousands).  The main kernel takes
  // tiles from the front of the list, the partner workgroups (from_back) from its end: one 64-bit counter, low word =
  // tiles taken at the front, high word = tiles taken at the back, a claim is valid while front + back < n_tiles.
  uint32_t n_tiles, from_back;
  uint32_t n_slots;  // tile slots per workgroup (tiles in flight), <= RT_MAX_SLOTS
  uint32_t lvl_begin[RT_TILE_LEVELS + 1], lvl_y0[RT_TILE_LEVELS], lvl_tiles_x[RT_TILE_LEVELS], lvl_shift[RT_TILE_LEVELS];
  uint32_t k0, k1;
  uint32_t stack_depth;
  uint32_t kb, kn;  // RESUME variant: hand-back / node-loop exit thresholds (traverse_resume)
  uint32_t noise_lds;  // the LDS-resident instances copy the noise jump table into LDS (scene has a noise texture)
  uint32_t kr;      // regenerate only when at least kr lanes are free (or none carries a path)
  uint32_t n_cached;  // four-wide nodes each workgroup copies into LDS (KIND_*_WIDE_HALF)
  // Tile-order feedback (render_fused): `depth_out[tile]` receives the depth class of the tile's deepest path (0: < 6
  // bounces, 1: < 12, 2: < 24, 3: deeper); `order`, when not NULL, is the list position -> tile permutation built from
  // the classes the previous frame of the same scene and camera recorded (deepest first).
  const uint32_t* order;
  unsigned char* depth_out;
  // Batches (rt_render_batch_device): the tile list is n_frames copies of the frame's list, list position T = frame *
  // tiles_per_frame + tile; `frames` holds what differs between the frames (key, first sample index, output).  n_tiles is
  // the whole list's length.  Paths of frame k + 1 are handed out while the deep paths of frame k drain.
  const BatchFrame* frames;  // never NULL: a single frame is a table of one (FrameScratch::frame1), so the kernel has one path
  uint32_t n_frames, tiles_per_frame;
  // Fixed-point format of the tile sums (to_fix / flush_finished): a sample channel is clamped to 2^20 and scaled by
  // 2^fix_bits, fix_bits = min(40, 43 - ceil(log2 K)), so that the K samples of a pixel cannot overflow the u64 slot
  // whatever they are (K <= 8: 2^-40 resolution; C2's 100: 2^-36; 10 000: 2^-29; the output is f32).
  float fix_scale;
  double fix_inv;
  uint32_t poll_limit;  // watchdog: polls of the "nothing to hand out" loop before a wave gives up (2^24; option dbg_watchdog_polls)
  unsigned long long* timeline;  // instrumented build: [0] earliest wave start, [1] latest wave end, [2] sum of wave ends, [3] sum of the times waves found the tile list used up (wall_clock64 ticks, low 40 bits in the sums)
};
#define RT_TILE_MASK 0x03FFFFFFu  // WgSched::tile: low 26 bits = tile, bits 26..28 = depth-class flags of its paths so far
// Workgroup scheduler state (LDS).  cursor: low word = next sample index (0..K) of the OPEN tile to hand out, high word =
// (tile << 6 | slot) of that tile, RT_CUR_NONE before the first tile, RT_CUR_OPENING while a wave opens the next tile,
// RT_CUR_EXHAUSTED when the frame has no tiles left.
struct WgSched {
  unsigned long long cursor;
  uint32_t pad[2];
  uint32_t rem[RT_MAX_SLOTS];    // paths of the slot's tile that have not finished yet
  uint32_t tile[RT_MAX_SLOTS];   // the slot's tile
  uint32_t state[RT_MAX_SLOTS];  // 0 = free (accumulators zero), 1 = in use
};
#define RT_SCHED_BYTES sizeof(WgSched)
static_assert(sizeof(WgSched) % 16 == 0, "scheduler block keeps the LDS regions 16-byte aligned");
#define RT_CUR_NONE 0xFFFFFFFEu
#define RT_CUR_OPENING 0xFFFFFFFDu
#define RT_CUR_EXHAUSTED 0xFFFFFFFFu
#define RT_FIX_CLAMP 1048576.0f  // 2^20: largest radiance channel a sample can add

// workgroup size of the HBM-resident instances that carry the whole feature set (noise / image / medium code).  768 threads
// (3 waves per SIMD, 168 VGPRs) until the node loop became an asm statement: compiled, the 128-VGPR form spilled inside the
// traversal loop.  Now the loop's registers are named, the spills that remain sit in the per-round and texture code, and
// four waves win: final scene 41.7 -> 41.1 ms.
#ifndef RT_HEAVY_BLOCK
#define RT_HEAVY_BLOCK 1024
#endif
#define RT_STATS_HEAVY_BLOCK 768  // the instrumented build of the same instance (see render_fused)
enum { KIND_GENERIC = 0, KIND_SPHERES_GLOBAL = 1, KIND_SPHERES_LDS = 2, KIND_GENERIC_LDS = 3, KIND_GENERIC_WIDE = 4, KIND_SPHERES_WIDE = 5,
       KIND_GENERIC_WIDE_HALF = 6, KIND_SPHERES_WIDE_HALF = 7 };
constexpr bool kind_caches_top(int k) { return k == KIND_GENERIC_WIDE_HALF || k == KIND_SPHERES_WIDE_HALF; }
constexpr bool kind_in_lds(int k) { return k == KIND_SPHERES_LDS || k == KIND_GENERIC_LDS; }

// tile index -> top-left pixel and log2 of the tile's side (all wave-uniform)
// The kernel's FusedParams as the scheduling code reads it: through a pointer into the kernel-argument segment (constant
// address space, scalar loads) made opaque where it is used, NOT through the by-value argument — fields read that way are
// loaded where the (rare) tile / chunk bookkeeping runs instead of being held in scalar registers across the whole render
// loop, where they were spilled to VGPR lanes (v_writelane / v_readlane) and those VGPRs in turn to scratch.
typedef const __attribute__((address_space(4))) FusedParams* ColdParams;
// argument list of k_fused / k_fused_lean, to locate FusedParams in the kernel-argument segment
struct FusedKernelArgs {
  Scene sc;
  const Camera* camp;
  FusedParams fp;
  unsigned long long* tile_counter;
  float* out_sum;
  unsigned long long* ray_total;
  float* dbg;
};
__device__ __forceinline__ ColdParams cold_params() {
  const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  ColdParams p = (ColdParams)(ka + offsetof(FusedKernelArgs, fp));
  asm volatile("" : "+s"(p));
  return p;
}
template <class P>
__device__ __forceinline__ void tile_geom(P fp, uint32_t t, uint32_t& x0, uint32_t& y0, uint32_t& shift) {
  uint32_t lv = 0;
#pragma unroll
  for (uint32_t i = 1; i < RT_TILE_LEVELS; i++) lv = t >= fp->lvl_begin[i] ? i : lv;
  const uint32_t local = t - fp->lvl_begin[lv];
  shift = fp->lvl_shift[lv];
  const uint32_t ty = local / fp->lvl_tiles_x[lv], tx = local - ty * fp->lvl_tiles_x[lv];
  x0 = tx << shift;
  y0 = fp->lvl_y0[lv] + (ty << shift);
}
// list position -> (frame, tile inside the frame); one frame: the position is the tile
template <class P>
__device__ __forceinline__ void tile_split(P fp, uint32_t T, uint32_t& frame, uint32_t& tile) {
  // (T is wave-uniform; the readfirstlane tells the compiler so — the values end up in scalar registers and addresses)
  T = (uint32_t)__builtin_amdgcn_readfirstlane((int)T);
  frame = 0; tile = T;
  if (fp->n_frames > 1u) { frame = T / fp->tiles_per_frame; tile = T - frame * fp->tiles_per_frame; }
}
// sample indices per chunk for a tile of side 1 << shift: a chunk is about Ku x 64 paths whatever the tile's size
template <class P>
__device__ __forceinline__ uint32_t chunk_samples(P fp, uint32_t shift) { return fp->Ku << (6u - 2u * shift); }

// The scheduler block's LDS address, made opaque at the point of use: the cold scheduling code then forms its per-lane
// addresses where it runs instead of holding them in VGPRs across the whole render loop (where they get spilled).
typedef __attribute__((address_space(3))) WgSched LdsSched;
__device__ __forceinline__ LdsSched* sched_here(WgSched* s) {
  uint32_t a = (uint32_t)(uintptr_t)(LdsSched*)s;
  asm volatile("" : "+s"(a));
  return (LdsSched*)(uintptr_t)a;
}
// cursor store (release).  The words are made opaque right here so that the compiler builds the 64-bit value in VGPRs at
// the (rare) store instead of keeping it in registers across the whole render loop.
__device__ __forceinline__ void cursor_store(LdsSched* sched, uint32_t hi, uint32_t lo) {
  asm volatile("" : "+s"(hi), "+s"(lo));
  __hip_atomic_store(&sched->cursor, ((unsigned long long)hi << 32) | lo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Radiance channel -> fixed point.  The two things that make a sample reach the frame differently from what was
// computed are counted (rt_stats.samples_nan / samples_clamped; `anomalies` = the render's ray counter block: [1] NaN
// channels, [2] clamped channels): rare by construction, so each is a device atomic on the spot.
// v_med3_f32 does the whole clamp: med3(v, 0, 2^20) is 0 for negatives and — a NaN operand makes med3 return the
// minimum of the others — for NaN; the counting is a branch that is practically never taken.
__device__ __forceinline__ unsigned long long to_fix(float v, float scale, unsigned long long* anomalies) {
  const float c = __builtin_amdgcn_fmed3f(v, 0.f, RT_FIX_CLAMP);
  if (!(v <= RT_FIX_CLAMP)) atomicAdd(anomalies + (v != v ? 1 : 2), 1ull);
  return (unsigned long long)(c * scale);
}

// camera block and (single frames) the one-entry frame table travel as kernel arguments into their device slots:
// stream-ordered, and no host buffer has to outlive an asynchronous copy
__global__ void k_store_camera(Camera cam, Camera* __restrict__ dst, BatchFrame frame, BatchFrame* __restrict__ frame_dst) {
  if (threadIdx.x == 0) {
    *dst = cam;
    if (frame_dst) *frame_dst = frame;
  }
}

// Tile-order feedback: list position -> tile, deepest depth class first, raster order inside a class (a stable counting
// sort by one workgroup; the tile list of a 4K frame is ~130 000 entries).  cls[] is what the previous frame of this
// view recorded (FusedParams::depth_out).  Why: a path that bounces 50 times takes ~0.25 ms however idle the GPU is
// (50 dependent traversals of one lane); if the tile that holds it is handed out last, the whole frame waits for it.
__global__ __launch_bounds__(1024) void k_order_tiles(const unsigned char* __restrict__ cls, uint32_t* __restrict__ order, uint32_t n) {
  __shared__ uint32_t wave_tot[4][16];
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint32_t per = (n + 1023u) / 1024u, i0 = min(n, t * per), i1 = min(n, i0 + per);
  uint32_t c[4] = {0, 0, 0, 0};
  for (uint32_t i = i0; i < i1; i++) c[3u - min(3u, (uint32_t)cls[i])]++;  // bucket 0 = deepest
  // exclusive prefix of every bucket's per-thread counts: inclusive scan inside the wave, then the 16 wave totals
  uint32_t inc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t v = c[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t u = __shfl_up(v, d, 64);
      if (lane >= (uint32_t)d) v += u;
    }
    inc[k] = v;
    if (lane == 63u) wave_tot[k][wave] = v;
  }
  __syncthreads();
  uint32_t at[4], base = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 16u; w++) { const uint32_t v = wave_tot[k][w]; total += v; if (w < wave) before += v; }
    at[k] = base + before + inc[k] - c[k];
    base += total;
  }
  for (uint32_t i = i0; i < i1; i++) order[at[3u - min(3u, (uint32_t)cls[i])]++] = i;
}

template <int KIND> struct AccOf;
template <> struct AccOf<KIND_SPHERES_LDS> {
  using type = SlabLdsAcc;
  static __device__ __forceinline__ type make(const Scene&, const unsigned char* ln, const SphereRec* ls, LdsTablePtr nt) {
    return type{ln, ls, nt};
  }
};
template <> struct AccOf<KIND_GENERIC_LDS> {
  using type = GenericLdsAcc;
  static __device__ __forceinline__ type make(const Scene&, const unsigned char* ln, const SphereRec* lp, LdsTablePtr nt) {
    return type{ln, reinterpret_cast<const Prim*>(lp), nt};
  }
};
template <> struct AccOf<KIND_SPHERES_GLOBAL> {
  using type = SphereAcc;
  static __device__ __forceinline__ type make(const Scene& sc, const unsigned char*, const SphereRec*, LdsTablePtr) { return sphere_acc(sc); }
};
template <> struct AccOf<KIND_GENERIC> {
  using type = GlobalAcc;
  static __device__ __forceinline__ type make(const Scene& sc, const unsigned char*, const SphereRec*, LdsTablePtr) { return global_acc(sc); }
};
template <> struct AccOf<KIND_GENERIC_WIDE> {
  using type = WideAcc<GlobalAcc>;
  static __device__ __forceinline__ type make(const Scene& sc, const unsigned char*, const SphereRec*, LdsTablePtr) {
    return type{global_acc(sc), sc.nodes4};
  }
};
template <> struct AccOf<KIND_GENERIC_WIDE_HALF> {
  using type = WideHalfAcc<GlobalAcc>;
  static __device__ __forceinline__ type make(const Scene& sc, const unsigned char* ln, const SphereRec*, LdsTablePtr) {
    return type{global_acc(sc), sc.nodes4h, reinterpret_cast<const uint4*>(ln), 0};  // n_cached: set by k_fused
  }
};
template <> struct AccOf<KIND_SPHERES_WIDE_HALF> {
  using type = WideHalfAcc<SphereAcc>;
  static __device__ __forceinline__ type make(const Scene& sc, const unsigned char* ln, const SphereRec*, LdsTablePtr) {
    return type{sphere_acc(sc), sc.nodes4h, reinterpret_cast<const uint4*>(ln), 0};
  }
};
template <> struct AccOf<KIND_SPHERES_WIDE> {
  using type = WideAcc<SphereAcc>;
  static __device__ __forceinline__ type make(const Scene& sc, const unsigned char*, const SphereRec*, LdsTablePtr) {
    return type{sphere_acc(sc), sc.nodes4};
  }
};

// Register budget: 4 waves/SIMD (128 VGPRs) — what a 1024-thread workgroup needs anyway; the 256-thread instances that
// carry the noise / image / medium code (HBM-resident scenes: C5) run better at 3 waves/SIMD with 168 VGPRs and
// 272 B of scratch than at 4 with 432 B (26.4 vs 32.6 ms on C5 at 4 spp; 2 waves / 243 VGPRs / no scratch: 30.9).
// The lean spheres-in-LDS instance and its 256-thread partner share a CU (4 + 1 waves per SIMD): the partner is held to
// the 5-waves budget (96 VGPRs), the main instance to 104 (fused_max_vgprs).
constexpr int fused_waves_per_simd(uint32_t F, int KIND, int BLOCK) {
  if (BLOCK == 768 || (BLOCK == 256 && (F & RT_F_TEX_NOISE))) return 3;  // (RT_HEAVY_BLOCK 1024: 4 waves, 128 VGPRs)
  if (F == RT_F_TEX_CHECKER && KIND == KIND_SPHERES_WIDE_HALF && BLOCK == 256) return 5;
  return 4;
}
// The per-workgroup copy of the scene in LDS (instances that keep it there), or of the top of the four-wide tree
// (instances that cache it): shared by the render kernels and by the probe kernels that walk the same structures.
// Returns the bytes used; the caller synchronises the workgroup before the first traversal.
template <uint32_t F, int KIND, int BLOCK>
__device__ __forceinline__ size_t lds_scene_setup(const Scene& sc, uint32_t noise_lds, uint32_t n_cached, unsigned char* smem,
                                                  SphereRec*& l_sph, LdsTablePtr& l_noise) {
  size_t off = 0;
  unsigned char* l_nodes = smem;
  if (kind_in_lds(KIND)) {
    off += RT_SLAB_REGION_BYTES(sc.n_nodes);
    l_sph = reinterpret_cast<SphereRec*>(smem + off);
    const int rec16 = KIND != KIND_GENERIC_LDS ? 4 : 8;  // record size in float4s: SphereRec 64 B / Prim 128 B
    off += (size_t)sc.n_prims * rec16 * 16;
    for (int i = threadIdx.x; i < sc.n_nodes; i += BLOCK) slab_node_store(l_nodes, i, sc.nodes[i]);
    const float4* gs = KIND != KIND_GENERIC_LDS ? reinterpret_cast<const float4*>(sc.spheres)
                                                : reinterpret_cast<const float4*>(sc.prims);
    float4* ls = reinterpret_cast<float4*>(l_sph);
    for (int i = threadIdx.x; i < sc.n_prims * rec16; i += BLOCK) ls[i] = gs[i];
    if ((F & RT_F_TEX_NOISE) && noise_lds) {  // 64 KB noise jump table: 40 random 16-B lookups per noise sample
      uint4* lt = reinterpret_cast<uint4*>(smem + off);
      l_noise = (LdsTablePtr)(reinterpret_cast<u32x4*>(lt));
      off += 16 * 256 * 16;
      for (int i = threadIdx.x; i < 16 * 256; i += BLOCK) lt[i] = sc.noise_table[i];
    }
  }
  if (kind_caches_top(KIND)) {
    off += (size_t)n_cached * 64;
    uint4* lc = reinterpret_cast<uint4*>(smem);
    for (uint32_t i = threadIdx.x; i < n_cached * 4u; i += BLOCK) lc[i] = sc.nodes4h[i];
  }
  return off;
}

// BATCH = false: the single-frame form of every production instance — no frame index, key and first sample index straight
// from the kernel arguments: compiled in, the batch bookkeeping measured 1.4-2.2 % on a C2 frame and 2 % on C5.
template <uint32_t F, int KIND, int BLOCK, class StackT, bool STATS, bool RESUME, bool BATCH = true>
__device__ __forceinline__ void fused_body(const Scene& sc, const Camera* __restrict__ camp, const FusedParams& fp, unsigned long long* __restrict__ tile_counter,
                                           float* __restrict__ out_sum, unsigned long long* __restrict__ ray_total,
                                           float* __restrict__ dbg) {
  // The camera block (26 doubles) stays in memory: read with scalar loads where a ray is generated instead of
  // occupying ~60 SGPRs for the whole kernel (they spilled to VGPR lanes: 68 v_readlane per regeneration).
  struct { uint32_t width, height, max_depth; } cam = {camp->width, camp->height, camp->max_depth};
  TravCounters tc = {0, 0, 0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0, 0, 0, 0, 0};
  float st_regen = 0.f, st_wregen = 0.f, st_shade = 0.f, st_wshade = 0.f, st_tshade = 0.f, st_tregen = 0.f, st_ttotal = 0.f;
  uint64_t st_t0 = 0, st_tk = 0;
  float st_twait = 0.f;                  // cycles in the "nothing to hand out" polling loop
  float st_texh = 0.f;                   // wave age when it first found the tile list used up
  uint64_t st_abs_exh = 0;               // ... and the time itself
  uint64_t st_wall0 = 0;
  if (STATS) { st_tk = now_cycles(); st_wall0 = wall_clock64(); }
  extern __shared__ __align__(16) unsigned char smem[];
  // LDS layout: [nodes][spheres] (KIND_SPHERES_LDS only) [scheduler][tile slots][stack]
  const uint32_t S = fp.n_slots;  // tile slots of the workgroup
  unsigned char* l_nodes = smem;
  SphereRec* l_sph = nullptr;
  LdsTablePtr l_noise = nullptr;
  size_t off = lds_scene_setup<F, KIND, BLOCK>(sc, fp.noise_lds, fp.n_cached, smem, l_sph, l_noise);
  WgSched* sched = reinterpret_cast<WgSched*>(smem + off);
  off += sizeof(WgSched);
  unsigned long long* l_acc = reinterpret_cast<unsigned long long*>(smem + off);  // [slot][pixel][channel]
  off += (size_t)S * 192 * 8;
  StackT* stack = reinterpret_cast<StackT*>(smem + off) + threadIdx.x;
  for (uint32_t i = threadIdx.x; i < S * 192u; i += BLOCK) l_acc[i] = 0;
  if (threadIdx.x < RT_MAX_SLOTS) { sched->rem[threadIdx.x] = 0; sched->tile[threadIdx.x] = 0; sched->state[threadIdx.x] = threadIdx.x < S ? 0u : 1u; }
  if (threadIdx.x == 0) sched->cursor = ((unsigned long long)RT_CUR_NONE << 32) | fp.K;  // nothing to hand out yet
  __syncthreads();
  typename AccOf<KIND>::type acc_init = AccOf<KIND>::make(sc, l_nodes, l_sph, l_noise);
  if constexpr (kind_caches_top(KIND)) acc_init.n_cached = (int32_t)fp.n_cached;
  const typename AccOf<KIND>::type acc = acc_init;

  const uint32_t lane = threadIdx.x & 63u;
  uint32_t my_rays = 0;
  // ---- wave state.  The wave works on one chunk (a tile slot x Ku sample indices) at a time; when the chunk's sample pool
  // is exhausted it claims the next one AT ONCE — of the same tile or, when that is handed out completely, of the next
  // tile the workgroup opens.  Paths in flight keep accumulating into their own tile's slot, so lanes never idle through
  // a drain — only through the last one of the workgroup.
  uint32_t x0 = 0, y0 = 0, c_shift = 3, k_lo = 0, total = 0, next = 0, c_slot = 0;  // wave-uniform: current chunk and its pool cursor
  uint32_t c_frame = 0;                      // wave-uniform: frame (of a batch) the current chunk belongs to
  bool exhausted = false;                    // wave-uniform: the frame has no more chunks
  bool opening = false;                      // wave-uniform: this wave has to open the workgroup's next tile
  bool active = false;
  d3 o = {0, 0, 0}, d = {0, 0, 0};
  double time = 0;
  f3 thr = {1.f, 1.f, 1.f};
  // one register for two small values: low 12 bits = lpix (slot * 64 + pixel index inside the tile), the rest = depth
  // (saturating at 2^20 - 1: a max_depth beyond that is never reached by a path anyway)
  uint32_t dl = 0;
#define RT_LPIX (dl & 4095u)
#define RT_DEPTH (dl >> 12)
  PathRng rng = {0, 0};
  // RESUME: the lane's traversal progress (node, stack, best hit) survives shade / regenerate rounds of its
  // neighbours; the per-ray constants are recomputed each round (unconditionally, so that they are not live across
  // the shading code)
  constexpr int32_t kDone = TravDone<StackT>::v;
  const int32_t stack_addr = lds_addr(stack);  // the lane's stack column (LDS byte address of its first row)
  int32_t t_cur = kDone, t_sp = stack_addr, t_top = kDone, t_best_prim = -1;
  double t_best_t = 0;
  bool fresh_lane = false;  // this lane holds a ray whose traversal has not started
  bool fresh = false;       // wave-uniform: some lane received a new ray since the last traversal call
  // shade one finished traversal; returns true when the path continues with a new ray in (o, d)
  auto bounce = [&](HitInfo hit, int32_t& flush_slot) -> bool {
    if (STATS) { st_shade += 1.f; st_wshade += wave_share(); st_t0 = now_cycles(); }
    ShadeOut so = shade<F>(sc, acc, *camp, o, d, time, thr, hit, rng);
    if (STATS) st_tshade += (float)(now_cycles() - st_t0) * wave_share();
    bool alive = so.alive;
    if (!alive) {
      const unsigned long long r = to_fix(so.radiance.x, fp.fix_scale, ray_total), g = to_fix(so.radiance.y, fp.fix_scale, ray_total),
                               b = to_fix(so.radiance.z, fp.fix_scale, ray_total);
      if (r) atomicAdd(&l_acc[RT_LPIX * 3 + 0], r);
      if (g) atomicAdd(&l_acc[RT_LPIX * 3 + 1], g);
      if (b) atomicAdd(&l_acc[RT_LPIX * 3 + 2], b);
    } else {
      if (RT_DEPTH < 0xFFFFFu) dl += 4096u;
      if (RT_DEPTH >= cam.max_depth) alive = false;  // mod.rs:195-197: black
    }
    if (!alive) {
      // deep paths mark their tile (tile-order feedback: such tiles go first in the next frame of this view)
      const uint32_t depth = RT_DEPTH, lpix = RT_LPIX;
      if (depth >= 6u) __hip_atomic_fetch_or(&sched->tile[lpix >> 6], depth >= 24u ? (7u << 26) : depth >= 12u ? (3u << 26) : (1u << 26),
                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // one path fewer in flight for its tile.  acq_rel at workgroup scope: the adds above are in LDS before the count
      // drops, and the lane that takes it to zero sees every other wave's adds.
      const uint32_t before = __hip_atomic_fetch_sub(&sched->rem[lpix >> 6], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (before == 1u) flush_slot = (int32_t)(lpix >> 6);
      return false;
    }
    o = so.o; d = so.d; thr = so.thr;
    return true;
  };
  // tiles whose last path one of this wave's lanes just finished: write the slot to the frame, free it
  auto flush_finished = [&](int32_t flush_slot) {
    unsigned long long fm = __ballot(flush_slot >= 0);
    while (fm) {
      const int l = __ffsll((long long)fm) - 1;
      fm &= fm - 1;
      const uint32_t s = (uint32_t)__builtin_amdgcn_readlane(flush_slot, l);
      const uint32_t tile_word = (uint32_t)__builtin_amdgcn_readfirstlane((int)sched->tile[s]);
      uint32_t frame = 0, tile = tile_word & RT_TILE_MASK;
      const ColdParams cfp = cold_params();
      if (BATCH) tile_split(cfp, tile_word & RT_TILE_MASK, frame, tile);
      unsigned char* const depth_out = cfp->depth_out;
      if (depth_out && lane == 0) depth_out[tile] = (unsigned char)__popc(tile_word >> 26);
      uint32_t tx0, ty0, tsh;
      tile_geom(cfp, tile, tx0, ty0, tsh);
      float* const frame_out = BATCH ? cfp->frames[(uint32_t)__builtin_amdgcn_readfirstlane((int)frame)].out : out_sum;  // (uniform address: a scalar load)
      const double fix_inv = cfp->fix_inv;
      const uint32_t px = tx0 + (lane & ((1u << tsh) - 1u)), py = ty0 + (lane >> tsh);
      unsigned long long* a = l_acc + s * 192u + lane * 3u;
      if (lane < (1u << (2u * tsh)) && px < cam.width && py < cam.height) {
        float* q = frame_out + ((size_t)py * cam.width + px) * 3;
        q[0] = (float)((double)a[0] * fix_inv);  // 2^-fix_bits
        q[1] = (float)((double)a[1] * fix_inv);
        q[2] = (float)((double)a[2] * fix_inv);
      }
      a[0] = 0; a[1] = 0; a[2] = 0;
      if (lane == 0) __hip_atomic_store(&sched->state[s], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  while (true) {
    // ---- next chunk when the pool is empty
    if (next >= total && !exhausted) {
      LdsSched* const sp = sched_here(sched);
      const ColdParams cfp = cold_params();  // tile / chunk bookkeeping reads the parameters where it runs
      total = 0; next = 0;
      if (opening) {
        // open the workgroup's next tile in a free slot (none free: every slot still waits for paths in flight — try
        // again after the next round of work)
        const uint32_t st = lane < S ? __hip_atomic_load(&sp->state[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 1u;
        const unsigned long long free_mask = __ballot(st == 0u);
        if (free_mask) {
          const uint32_t s = (uint32_t)__ffsll((long long)free_mask) - 1u;
          // one 64-bit counter for both ends of the tile list (see FusedParams)
          unsigned long long tc64 = 0;
          if (lane == 0) {
            uint32_t back = cfp->from_back;
            asm volatile("" : "+s"(back));  // the increment is built here, not kept in VGPRs across the render loop
            tc64 = atomicAdd(tile_counter, (unsigned long long)(back ^ 1u) | ((unsigned long long)back << 32));
          }
          const uint32_t t_front = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tc64);
          const uint32_t t_back = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tc64 >> 32));
          const bool none_left = (unsigned long long)t_front + t_back >= cfp->n_tiles;
          uint32_t t = cfp->from_back ? cfp->n_tiles - 1u - t_back : t_front;
          uint32_t t_in = t;  // tile inside its frame
          if (!none_left) {
            if (BATCH) tile_split(cfp, t, c_frame, t_in);
            if (cfp->order) t_in = cfp->order[t_in];  // (uniform address: a scalar load)
            if (BATCH) t = c_frame * cfp->tiles_per_frame + t_in;
            else t = t_in;
          }
          opening = false;
          if (none_left) {
            exhausted = true;
            if (STATS && st_texh == 0.f) { st_texh = (float)(now_cycles() - st_tk) * (1.0f / 64.0f); st_abs_exh = wall_clock64(); }
            if (lane == 0) cursor_store(sp, RT_CUR_EXHAUSTED, cfp->K);
          } else {
            tile_geom(cfp, t_in, x0, y0, c_shift);
            const uint32_t side = 1u << c_shift;
            const uint32_t n_valid = min(side, cam.width - x0) * min(side, cam.height - y0);
            const uint32_t ks = chunk_samples(cfp, c_shift);
            if (lane == 0) {
              sp->tile[s] = t;
              sp->rem[s] = n_valid * cfp->K;
              sp->state[s] = 1u;
              // the opener keeps the tile's first chunk; release: the slot's words are in LDS before the cursor moves
              cursor_store(sp, (t << 6) | s, ks);
            }
            c_slot = s; k_lo = 0;
            total = min(ks, cfp->K) << (2u * c_shift);
          }
        }
      } else {
        // a chunk of the open tile: look first (no add once the tile is handed out), then claim with one LDS atomic.
        // The first wave to find the tile handed out completely (or no tile opened yet) takes the duty of opening the
        // next one: it swaps RT_CUR_OPENING into the cursor — the value it saw is stable, nobody adds to a cursor at or
        // beyond K except adds already under way, and those make the swap fail harmlessly (next round).
        unsigned long long c = 0;
        uint32_t won = 0, inc = 0;
        if (lane == 0) {
          c = __hip_atomic_load(&sp->cursor, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
          const uint32_t chi = (uint32_t)(c >> 32);
          if ((uint32_t)c < cfp->K) {
            // the chunk is as long as what this wave adds (sized for the tile it saw; another tile may have been opened
            // in between: then the chunk is merely longer or shorter than that tile's usual ones)
            uint32_t gx, gy, gs, gf = 0, gt = chi >> 6;
            if (BATCH) tile_split(cfp, chi >> 6, gf, gt);
            tile_geom(cfp, gt, gx, gy, gs);
            inc = chunk_samples(cfp, gs);
            c = __hip_atomic_fetch_add(&sp->cursor, (unsigned long long)inc, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else if (chi != RT_CUR_EXHAUSTED && chi != RT_CUR_OPENING) {
            unsigned long long expect = c;
            uint32_t want_hi = RT_CUR_OPENING, want_lo = cfp->K;
            asm volatile("" : "+s"(want_hi), "+s"(want_lo));  // built here, not hoisted out of the render loop
            won = __hip_atomic_compare_exchange_strong(&sp->cursor, &expect, ((unsigned long long)want_hi << 32) | want_lo,
                                                       __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
          }
        }
        const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)c);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(c >> 32));
        won = (uint32_t)__builtin_amdgcn_readfirstlane((int)won);
        inc = (uint32_t)__builtin_amdgcn_readfirstlane((int)inc);
        if (k < cfp->K) {
          c_slot = hi & 63u;
          uint32_t t_in = hi >> 6;
          if (BATCH) tile_split(cfp, hi >> 6, c_frame, t_in);
          tile_geom(cfp, t_in, x0, y0, c_shift);
          k_lo = k;
          total = min(inc, cfp->K - k) << (2u * c_shift);
        } else if (hi == RT_CUR_EXHAUSTED) {
          exhausted = true;
          if (STATS && st_texh == 0.f) { st_texh = (float)(now_cycles() - st_tk) * (1.0f / 64.0f); st_abs_exh = wall_clock64(); }
        } else if (won) {
          opening = true;
          continue;
        }
      }
    }
    // ---- regenerate: lanes without a path take the next samples of the pool (wave64 ballot + prefix)
    unsigned long long need = __ballot(!active);
    if (STATS) st_t0 = now_cycles();
    const uint32_t n_free = (uint32_t)__popcll(need);
    if (n_free && (n_free >= fp.kr || n_free == 64u) && next < total) {
      uint32_t j = next + wave_prefix(need);
      if (!active && j < total) {
        const uint32_t pit = j & ((1u << (2u * c_shift)) - 1u), ks = j >> (2u * c_shift);
        const uint32_t px = x0 + (pit & ((1u << c_shift) - 1u)), py = y0 + (pit >> c_shift);
        if (px < cam.width && py < cam.height) {
          if (STATS) { st_regen += 1.f; st_wregen += wave_share(); }
          // key and first sample index of the chunk's frame (a batch: from the frame table, scalar loads made here)
          uint32_t f_k0 = fp.k0, f_k1 = fp.k1, f_s0 = fp.s_begin;
          if (BATCH) {
            const BatchFrame* bf = fp.frames + (uint32_t)__builtin_amdgcn_readfirstlane((int)c_frame);
            f_k0 = bf->k0; f_k1 = bf->k1; f_s0 = bf->s_begin;
          }
          uint32_t s = f_s0 + (k_lo + ks) * fp.s_stride;
          const Camera* cp = camp;
          asm volatile("" : "+s"(cp));  // keep the camera loads here (not hoisted into long-lived SGPRs)
          generate_ray(*cp, py, px, s, f_k0, f_k1, o, d, time, rng);
          thr = {1.f, 1.f, 1.f};
          dl = c_slot * 64u + pit;  // depth 0
          active = true;
          fresh_lane = true;
        }
      }
      next = min(total, next + (uint32_t)__popcll(need));
      fresh = true;
    }
    if (STATS) st_tregen += (float)(now_cycles() - st_t0) * wave_share();
    if (__ballot(active) == 0ull) {
      if (exhausted) break;
      if (next >= total) {
        // Nothing to hand out right now: another wave is opening the next tile, or (this wave being the opener) every
        // slot still waits for paths in flight.  Poll until that changes.  Safety valve: after ~10 s of polling the
        // wave gives up and flags the frame as failed (tile_counter[1], reported by rt_get_stats) rather than hang the
        // GPU on a scheduling bug.
        LdsSched* const sp = sched_here(sched);
        bool go = false;
        uint32_t polls = 0;
        if (STATS) st_t0 = now_cycles();
        do {
          __builtin_amdgcn_s_sleep(8);
          if (opening) {
            const uint32_t st = lane < S ? __hip_atomic_load(&sp->state[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) : 1u;
            go = __ballot(st == 0u) != 0ull;
          } else {
            uint32_t chi = 0;
            if (lane == 0) chi = (uint32_t)(__hip_atomic_load(&sp->cursor, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32);
            go = (uint32_t)__builtin_amdgcn_readfirstlane((int)chi) != RT_CUR_OPENING;
          }
        } while (!go && ++polls < cold_params()->poll_limit);
        if (STATS) { st_twait += (float)(now_cycles() - st_t0) * (1.0f / 64.0f); }
        if (!go) {
          if (lane == 0) atomicOr(reinterpret_cast<uint32_t*>(tile_counter + 1), 1u);
          break;
        }
      }
      continue;
    }
    if (RESUME) {
      TravState ts;
      trav_ray_constants(ts, o, d);  // all lanes (garbage in lanes without a path is never used)
      const bool starting = fresh_lane;
      if (fresh_lane) { t_cur = sc.tree_empty ? kDone : acc.root(sc.root); t_sp = stack_addr; t_top = kDone; t_best_prim = -1; t_best_t = __builtin_inf(); my_rays++; }
      fresh_lane = false;
      ts.cur = t_cur; ts.sp = t_sp; ts.top = t_top; ts.best_prim = t_best_prim; ts.best_t = t_best_t;
      ts.tmax_w = widen_tmax(t_best_t);
      if (sc.n_global && starting) global_prims_test(acc, global_prims(sc), ts, F, o, d, time, RT_TMIN, rng);  // the ground: all starting lanes at once
      // ---- traverse until fewer than kb lanes are unfinished (stragglers keep their progress and go on next round
      //      beside the re-launched lanes); when nothing new was launched, run to completion
      traverse_resume<typename AccOf<KIND>::type, StackT, STATS, BLOCK>(acc, ts, active, F, o, d, time, RT_TMIN, stack, BLOCK,
                                                                 &tc, rng, fresh ? fp.kb : 1u, fp.kn);
      t_cur = ts.cur; t_sp = ts.sp; t_top = ts.top; t_best_prim = ts.best_prim; t_best_t = ts.best_t;
      const bool ready = active && t_cur == kDone;
      fresh = __ballot(ready) != 0ull;
      int32_t flush_slot = -1;  // this lane finished the last path of that slot's tile
      if (ready) {
        HitInfo hit;
        hit.t = t_best_t;
        hit.prim = t_best_prim;
        if (bounce(hit, flush_slot)) fresh_lane = true;
        else active = false;
      }
      flush_finished(flush_slot);
    } else {
      // ---- one bounce for every lane that carries a path: closest hit, then shade
      int32_t flush_slot = -1;
      if (active) {
        my_rays++;
        HitInfo hit = traverse<typename AccOf<KIND>::type, StackT, STATS>(acc, acc.root(sc.root), F, o, d, time, RT_TMIN,
                                                                          stack, BLOCK, &tc, rng, global_prims(sc));
        if (!bounce(hit, flush_slot)) active = false;
      }
      flush_finished(flush_slot);
    }
  }
  // rays traced (stats): one atomic per wave
  unsigned long long wave_rays = my_rays;
  for (int off2 = 32; off2 > 0; off2 >>= 1) wave_rays += __shfl_down(wave_rays, off2, 64);
  if (lane == 0 && wave_rays) atomicAdd(ray_total, wave_rays);
  if (STATS && dbg) {  // diagnostic build only: lane-work vs wave-trip tallies per phase
    st_ttotal = (float)(now_cycles() - st_tk) * (1.0f / 64.0f);
    float v[13] = {(float)tc.nodes, tc.wnodes, (float)tc.prims, tc.wprims, st_shade, st_wshade, st_regen, st_wregen,
                   tc.t_nodes, tc.t_prims, st_tshade, st_tregen, st_ttotal};
    for (int k = 0; k < 13; k++) {
      float x = v[k];
      for (int off2 = 32; off2 > 0; off2 >>= 1) x += __shfl_down(x, off2, 64);
      if (lane == 0) atomicAdd(&dbg[k], x);
    }
    // longest wave lifetime (positive floats order like their bit patterns): against the mean, the frame's tail
    if (lane == 0) atomicMax(reinterpret_cast<int*>(&dbg[13]), __float_as_int(st_ttotal * 64.0f));
    if (fp.timeline) {  // node-loop trips sat out, by reason
      // [4..6] node-loop trips sat out by reason, [7] child boxes tested (node steps x 2, or x 4 on the four-wide tree),
      // [8..11] leaf tests by primitive type (exact counts)
      constexpr unsigned long long kBoxes = AccOf<KIND>::type::kWide ? 4ull : 2ull;
      unsigned long long v8[9] = {tc.idle_leaf, tc.idle_done, tc.idle_free, tc.nodes * kBoxes, tc.t_sph, tc.t_mov, tc.t_quad, tc.t_med, tc.t_box};
      for (int k = 0; k < 9; k++) {
        unsigned long long x = v8[k];
        for (int off2 = 32; off2 > 0; off2 >>= 1) x += __shfl_down(x, off2, 64);
        if (lane == 0 && x) atomicAdd(&fp.timeline[4 + k], x);
      }
    }
    if (fp.timeline && lane == 0) {
      const unsigned long long t_end = wall_clock64();
      atomicMin(&fp.timeline[0], (unsigned long long)st_wall0);
      atomicMax(&fp.timeline[1], t_end);
      atomicAdd(&fp.timeline[2], t_end & 0xFFFFFFFFFFull);
      atomicAdd(&fp.timeline[3], (unsigned long long)(st_abs_exh ? st_abs_exh : t_end) & 0xFFFFFFFFFFull);
    }
    {
      float w[2] = {st_twait, st_texh};
      for (int k = 0; k < 2; k++) {
        float x = w[k];
        for (int off2 = 32; off2 > 0; off2 >>= 1) x += __shfl_down(x, off2, 64);
        if (lane == 0) atomicAdd(&dbg[14 + k], x);
      }
    }
  }
}

template <uint32_t F, int KIND, int BLOCK, class StackT, bool STATS = false, bool RESUME = false, bool BATCH = true>
__global__ __launch_bounds__(BLOCK, fused_waves_per_simd(F, KIND, BLOCK)) void k_fused(Scene sc, const Camera* __restrict__ camp, FusedParams fp,
                                                                                       unsigned long long* __restrict__ tile_counter, float* __restrict__ out_sum,
                                                                                       unsigned long long* __restrict__ ray_total,
                                                                                       float* __restrict__ dbg = nullptr) {
  fused_body<F, KIND, BLOCK, StackT, STATS, RESUME, BATCH>(sc, camp, fp, tile_counter, out_sum, ray_total, dbg);
}
// The lean spheres-in-LDS instance (C1 / C2) with its own register ceiling: 4 x 104 VGPRs + the partner's 96 = 512, the
// SIMD's register file (the attribute takes no template-dependent value, hence a kernel of its own).
// (this compiler doubles the attribute's value on gfx950's unified register file: 52 caps the kernel at 104 VGPRs)
#ifndef RT_LEAN_VGPR_ATTR
#define RT_LEAN_VGPR_ATTR 52
#endif
template <bool STATS, bool RESUME, bool BATCH>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(RT_LEAN_VGPR_ATTR))) void k_fused_lean(Scene sc, const Camera* __restrict__ camp, FusedParams fp,
                                                                                            unsigned long long* __restrict__ tile_counter, float* __restrict__ out_sum,
                                                                                            unsigned long long* __restrict__ ray_total,
                                                                                            float* __restrict__ dbg = nullptr) {
  fused_body<RT_F_TEX_CHECKER, KIND_SPHERES_LDS, 1024, int16_t, STATS, RESUME, BATCH>(sc, camp, fp, tile_counter, out_sum, ray_total, dbg);
}
typedef void (*FusedKernel)(Scene, const Camera*, FusedParams, unsigned long long*, float*, unsigned long long*, float*);
template <uint32_t F, int KIND, int BLOCK, class StackT, bool STATS, bool RESUME>
static FusedKernel fused_kernel(bool batch) {
  if constexpr (F == RT_F_TEX_CHECKER && KIND == KIND_SPHERES_LDS && BLOCK == 1024 && sizeof(StackT) == 2)
    return batch ? k_fused_lean<STATS, RESUME, true> : k_fused_lean<STATS, RESUME, false>;
  else if constexpr (STATS) return k_fused<F, KIND, BLOCK, StackT, STATS, RESUME, true>;  // (instrumented builds: the one form)
  else return batch ? k_fused<F, KIND, BLOCK, StackT, STATS, RESUME, true> : k_fused<F, KIND, BLOCK, StackT, STATS, RESUME, false>;
}

__global__ __launch_bounds__(RT_BLOCK) void k_resolve(const float* __restrict__ sum, uint32_t n_values, double inv_spp,
                                                      double inv_gamma, double* __restrict__ out_rgb,
                                                      uint8_t* __restrict__ out_u8) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride) {
    double v = pow((double)sum[i] * inv_spp, inv_gamma);  // color.rs:58-61
    if (out_rgb) out_rgb[i] = v;
    if (out_u8) {  // color.rs:68-73: (256.0 * x) as u8 — saturating, NaN -> 0
      double x = 256.0 * v;
      uint8_t q = 0;
      if (x == x) q = x <= 0.0 ? 0 : (x >= 255.0 ? 255 : (uint8_t)x);
      out_u8[i] = q;
    }
  }
}

// parity probe: closest hit + full hit record for caller-supplied f64 rays (+ traversal counters).
// TREE: which form of the acceleration structure is walked — 0 binary (64-B nodes), 1 four-wide with f32 planes,
// 2 four-wide with binary16 planes (read from HBM: no LDS cache here) — so that every traversal the render kernels
// use can be held against the oracle's hits ray by ray.
template <int TREE> struct ProbeAcc;
template <> struct ProbeAcc<0> {
  using type = GlobalAcc;
  static __device__ __forceinline__ type make(const Scene& sc) { return global_acc(sc); }
};
template <> struct ProbeAcc<1> {
  using type = WideAcc<GlobalAcc>;
  static __device__ __forceinline__ type make(const Scene& sc) { return type{global_acc(sc), sc.nodes4}; }
};
template <> struct ProbeAcc<2> {
  using type = WideHalfAcc<GlobalAcc>;
  static __device__ __forceinline__ type make(const Scene& sc) { return type{global_acc(sc), sc.nodes4h, nullptr, 0}; }
};
template <int TREE>
__global__ __launch_bounds__(RT_BLOCK) void k_probe(Scene sc, uint32_t features, const rt_ray* __restrict__ rays,
                                                    uint32_t n, rt_hit* __restrict__ out,
                                                    unsigned long long* __restrict__ trav_counters) {
  __shared__ uint32_t stack_rows[(RT_STACK_DEPTH + 1) * RT_BLOCK];  // row 0: guard row below the columns (trav_pop)
  uint32_t* const stack = stack_rows + RT_BLOCK;
  const uint32_t stride = gridDim.x * blockDim.x;
  TravCounters tc = {0, 0, 0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    rt_ray r = rays[i];
    d3 o = {r.o[0], r.o[1], r.o[2]}, d = {r.d[0], r.d[1], r.d[2]};
    const typename ProbeAcc<TREE>::type acc = ProbeAcc<TREE>::make(sc);
    PathRng rng = {i * 2654435761u + 1u, 0x9E3779B9u};  // only constant media draw (their probe hits are random)
    HitInfo hit = traverse<typename ProbeAcc<TREE>::type, uint32_t, true>(acc, acc.root(sc.root), features, o, d, r.time,
                                                                          RT_TMIN, stack + threadIdx.x, RT_BLOCK, &tc, rng, global_prims(sc));
    rt_hit ho;
    memset(&ho, 0, sizeof(ho));
    ho.object_id = -1;
    if (hit.prim >= 0) {
      HitRec h = make_hit(acc, hit.prim, o, d, r.time, hit.t, true);
      ho.object_id = sc.prims[hit.prim].obj;
      ho.front_face = h.front ? 1 : 0;
      ho.t = hit.t;
      ho.p[0] = h.p.x; ho.p[1] = h.p.y; ho.p[2] = h.p.z;
      ho.n[0] = h.n.x; ho.n[1] = h.n.y; ho.n[2] = h.n.z;
      ho.u = h.u; ho.v = h.v;
    }
    out[i] = ho;
  }
  if (trav_counters) {
    atomicAdd(&trav_counters[0], (unsigned long long)tc.nodes);
    atomicAdd(&trav_counters[1], (unsigned long long)tc.prims);
  }
}

// Parity probe through the PRODUCTION traversal: caller rays walk exactly what the render kernels walk — the same LDS
// copy of the scene (KIND_*_LDS: 112-byte slab nodes addressed ref << 4, records in LDS) or the four-wide binary16 tree
// with its top cached in LDS (KIND_*_WIDE_HALF: the partner / HBM-resident form), the same 16-bit sentinel stack, and
// traverse_resume under the render loop's schedule: a wave keeps a pool of rays, lanes take the next ray when at least
// `kr` are free, a traversal round hands back when fewer than `kb` lanes are unfinished (stragglers resume next round
// beside the re-launched lanes), the node loop exits below `kn` walkers.  The hit record carries the render path's own
// f32 shading normal and front flag (shading_frame) instead of make_hit's f64 ones; u, v as the texture code gets them.
template <int KIND, int BLOCK, bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_probe_fused(Scene sc, uint32_t features, const rt_ray* __restrict__ rays, uint32_t n,
                                                       rt_hit* __restrict__ out, uint32_t n_cached, uint32_t kb, uint32_t kn,
                                                       uint32_t kr, unsigned long long* __restrict__ trav_counters) {
  typedef int16_t StackT;
  constexpr int32_t kDone = TravDone<StackT>::v;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* l_nodes = smem;
  SphereRec* l_sph = nullptr;
  LdsTablePtr l_noise = nullptr;
  size_t off = lds_scene_setup<RT_F_TEX_CHECKER, KIND, BLOCK>(sc, 0u, n_cached, smem, l_sph, l_noise);
  off += (size_t)BLOCK * sizeof(StackT);  // guard row below the stack columns (trav_pop)
  StackT* stack = reinterpret_cast<StackT*>(smem + off) + threadIdx.x;
  __syncthreads();
  typename AccOf<KIND>::type acc_init = AccOf<KIND>::make(sc, l_nodes, l_sph, l_noise);
  if constexpr (kind_caches_top(KIND)) acc_init.n_cached = (int32_t)n_cached;
  const typename AccOf<KIND>::type acc = acc_init;
  const uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6, n_waves = (gridDim.x * BLOCK) >> 6;
  // this wave's pool: a contiguous range of the caller's rays
  const uint32_t per = (n + n_waves - 1) / n_waves;
  uint32_t next = min(n, wave * per);
  const uint32_t total = min(n, next + per);
  TravCounters tc = {0, 0, 0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0, 0, 0, 0, 0};
  bool active = false, fresh_lane = false, fresh = false;
  uint32_t ray_index = 0;
  d3 o = {0, 0, 0}, d = {0, 0, 0};
  double time = 0;
  PathRng rng = {0, 0};
  const int32_t stack_addr = lds_addr(stack);  // the lane's stack column (LDS byte address of its first row)
  int32_t t_cur = kDone, t_sp = stack_addr, t_top = kDone, t_best_prim = -1;
  double t_best_t = 0;
  while (true) {
    const unsigned long long need = __ballot(!active);
    const uint32_t n_free = (uint32_t)__popcll(need);
    if (n_free && (n_free >= kr || n_free == 64u) && next < total) {
      const uint32_t j = next + wave_prefix(need);
      if (!active && j < total) {
        const rt_ray r = rays[j];
        o = {r.o[0], r.o[1], r.o[2]}; d = {r.d[0], r.d[1], r.d[2]};
        time = r.time;
        ray_index = j;
        rng = {j * 2654435761u + 1u, 0x9E3779B9u};  // only constant media draw (their probe hits are random)
        active = true;
        fresh_lane = true;
      }
      next = min(total, next + n_free);
      fresh = true;
    }
    if (__ballot(active) == 0ull) break;  // (the pool is empty too: a wave without rays regenerates all 64 lanes)
    TravState ts;
    trav_ray_constants(ts, o, d);
    const bool starting = fresh_lane;
    if (fresh_lane) { t_cur = sc.tree_empty ? kDone : acc.root(sc.root); t_sp = stack_addr; t_top = kDone; t_best_prim = -1; t_best_t = __builtin_inf(); }
    fresh_lane = false;
    ts.cur = t_cur; ts.sp = t_sp; ts.top = t_top; ts.best_prim = t_best_prim; ts.best_t = t_best_t;
    ts.tmax_w = widen_tmax(t_best_t);
    if (sc.n_global && starting) global_prims_test(acc, global_prims(sc), ts, features, o, d, time, RT_TMIN, rng);
    // COUNT = false is the render kernels' instantiation exactly (for sphere scenes in LDS: the asm traversal)
    traverse_resume<typename AccOf<KIND>::type, StackT, COUNT, BLOCK>(acc, ts, active, features, o, d, time, RT_TMIN, stack, BLOCK, &tc,
                                                                      rng, fresh ? kb : 1u, kn);
    t_cur = ts.cur; t_sp = ts.sp; t_top = ts.top; t_best_prim = ts.best_prim; t_best_t = ts.best_t;
    const bool ready = active && t_cur == kDone;
    fresh = __ballot(ready) != 0ull;
    if (ready) {
      rt_hit ho;
      memset(&ho, 0, sizeof(ho));
      ho.object_id = -1;
      if (t_best_prim >= 0) {
        d3 p;
        f3 nf;
        bool front;
        shading_frame(acc, t_best_prim, o, d, time, t_best_t, p, nf, front);
        const HitRec h = make_hit(acc, t_best_prim, o, d, time, t_best_t, true);
        ho.object_id = sc.prims[t_best_prim].obj;
        ho.front_face = front ? 1 : 0;
        ho.t = t_best_t;
        ho.p[0] = p.x; ho.p[1] = p.y; ho.p[2] = p.z;
        ho.n[0] = (double)nf.x; ho.n[1] = (double)nf.y; ho.n[2] = (double)nf.z;
        ho.u = h.u; ho.v = h.v;
      }
      out[ray_index] = ho;
      active = false;
    }
  }
  if (COUNT && trav_counters) {
    atomicAdd(&trav_counters[0], (unsigned long long)tc.nodes);
    atomicAdd(&trav_counters[1], (unsigned long long)tc.prims);
  }
}

// ================================================================================================ host side

// ---- fixed scratch of rt_render_device_noalloc (callers that cannot allocate: HIP graph capture, serving loops).
// One copy per GPU, held until the process ends: 1 KiB of control words — the camera block of the render in flight, the
// ray counter, the two-ended tile counter and the scheduler's watchdog flag.  (Round 1 also kept a W*H*3 u64 frame
// accumulator here, 24 MiB; since the workgroups own their tiles the partial sums live in LDS and no frame-sized
// scratch exists.)  Host code never passes the array's name to a kernel: rt_init asks for its address on the ctx's
// device (hipGetSymbolAddress) and keeps it.
#define RT_NOALLOC_BYTES ((size_t)1024)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];
static_assert(sizeof(Camera) <= 512, "camera block must fit the scratch's control area");

struct rt_scene {
  BvhNode* d_nodes = nullptr;
  float4* d_nodes4 = nullptr;  // four-wide form of the tree (HBM-resident fused instances)
  uint4* d_nodes4h = nullptr;  // ... with binary16 planes
  int32_t* d_global = nullptr; // Scene::global_list
  bool half_ok = false;        // every finite plane is inside the binary16 range
  int wide_stack = 0;          // stack entries a traversal of it needs
  Prim* d_prims = nullptr;
  Material* d_mats = nullptr;
  Texture* d_texs = nullptr;
  uint32_t* d_texels = nullptr;
  SphereRec* d_spheres = nullptr;
  Scene view{};
  int bvh_depth = 0;
  uint64_t serial = 0;       // unique per upload (tile-order feedback recognises "the same scene" by it)
  uint32_t features = 0;     // RT_F_* of what the scene contains
  uint32_t global_types[5] = {0, 0, 0, 0, 0};  // primitives on the global list by type (static sphere, moving sphere, quad, medium, merged box)
  bool spheres_only = false;
};

struct WorkBuffers {
  size_t cap = 0;
  void* slab = nullptr;  // one allocation holding both queues, hits and L
  Queue q[2]{};
  Hits hits{};
  float *Lr = nullptr, *Lg = nullptr, *Lb = nullptr;
  uint32_t* perm = nullptr;  // ray indices sorted by material class (option wf_bin)
};

// Device memory a fused render needs beyond its inputs and outputs.
struct FrameScratch {
  unsigned long long* unit_counter = nullptr;  // [0] two-ended tile counter of the persistent kernel, [1] its watchdog flag
  Camera* cam = nullptr;                    // camera block of the render in flight (read with scalar loads)
  BatchFrame* frame1 = nullptr;             // frame table of a single-frame render (key, first sample index, output)
  unsigned long long* ray_total = nullptr;  // 4 words: [0] rays traced, [1] NaN radiance channels, [2] clamped ones (stats)
};

struct rt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int64_t max_paths = 128ll << 20;
  int profile = 0;
  int mode = 1;              // 0 = wavefront kernels, 1 = fused persistent kernel
  int fused_ku = 0;          // samples per work unit (0 = auto)
  int fused_lds = 1;         // keep the scene in LDS when it fits
  int fused_stats = 0;       // diagnostic: instrumented fused kernel (lean LDS variant only)
  int fused_kr = 0;  // regeneration threshold of the resumable schedule; 0 = by residency: 16 for scenes in LDS, 12 read through the caches
  std::map<std::pair<const void*, size_t>, int> fused_occupancy;  // workgroups per CU of (kernel instance, LDS bytes)
  std::map<const void*, size_t> fused_lds_opt_in;                 // dynamic LDS ceiling set per kernel instance
  int fused_partner_min = 32;  // partner workgroups only for frames of at least this many Mi samples (C2 split 1 / 2 / 4 / 8 ways:
                               // 9.62 / 5.25 / 3.01 / 1.76 ms with them, 10.13 / 5.29 / 2.96 / 1.67 without)
  int fused_slots = 0;  // tile slots per workgroup (0 = auto)
  int fused_tail4 = 50, fused_tail2 = 30;  // share of the frame's rows (per mille) tiled 4x4 / 2x2 at the end of the tile list
  float* cur_out = nullptr;   // output frame of the render being issued
  bool cur_partner = false;   // ... and whether partner workgroups join it
  int dbg_main_cap = 0;       // diagnostic: cap on the main kernel's workgroups (0 = none)
  int fused_partner = 1;  // C1/C2-class scenes: a 256-thread workgroup per CU beside the 1024-thread one (fifth wave per SIMD)
  hipStream_t partner_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int fused_wide = 1;  // HBM-resident fused instances walk the four-wide tree
  int probe_tree = 0;  // rt_intersect walks: 0 the binary tree, 1 the four-wide tree (f32 planes), 2 its binary16 form
  int fused_half = 1;  // ... in its binary16 form, top levels cached in LDS, when the scene allows
  int fused_resume = 2, fused_kb = 16, fused_kn = 8;  // 0 bounce-synchronous, 1 resumable traversals, 2 by scene size  // RESUME variant of k_fused + its thresholds    // diagnostic: workgroup size of the LDS-resident variant (1024 / 512 / 256)
  float* d_dbg = nullptr;
  WorkBuffers wb;
  // per-render device scratch: `own` is allocated by the library (rt_render_device grows it on demand), `fixed` lives in
  // the file-scope __device__ array `scratch` (rt_render_device_noalloc); `fs` is the set of the render being issued
  FrameScratch own, fixed;
  const FrameScratch* fs = &own;
  const unsigned long long* last_ray_total = nullptr;  // ray counter of the last render (rt_get_stats)
  const uint32_t* last_watchdog = nullptr;             // watchdog flag of the last fused render
  unsigned long long* d_trav = nullptr;  // probe counters: node visits, prim tests
  int dbg_max_cached = 0;  // test hook: cap of the four-wide nodes kept in LDS (0 = as many as fit), so that small scenes reach the memory path
  int bvh_optimise = 3;  // rounds of the reinsertion pass after the binned-SAH build (read at scene upload; 0 = the built tree)
  int uv_tables = 1;  // read at scene upload: image textures get the table that lets sphere hits find their texel without f64 acos / atan2 (0 = the formula)
  int probe_count = 0;  // probe_tree 3 / 4: count node visits / primitive tests (instrumented traversal) instead of running the render kernels' own
  unsigned long long probe_rays = 0;
  uint32_t* d_counts = nullptr;
  uint32_t* d_bins = nullptr;  // wavefront material binning: class totals [0..5], class cursors [8..13]
  // tile-order feedback (render_fused / k_order_tiles): depth classes the last fused frame recorded per tile, the list
  // order built from them, and what that frame looked at
  unsigned long long* d_timeline = nullptr;  // instrumented build (FusedParams::timeline)
  int fused_kr_sync = 8;  // regeneration threshold of the bounce-synchronous schedule (small scenes)
  int merge_boxes = 1;   // scene upload: six quads that form an axis-aligned box become one RT_PRIM_AABOX in the tree
  int global_prims = 1;  // scene upload: keep huge spheres out of the tree and test them first (Scene::n_global)
  int fused_feedback = 1;
  hipStream_t fb_stream = nullptr;  // stream of the frame that recorded d_tile_class (feedback is only ordered through it)
  uint32_t last_global_types[5] = {0, 0, 0, 0, 0};  // global primitives of the last instrumented render's scene, by type
  int dbg_watchdog_polls = 0;       // diagnostic: polls before a wave with nothing to do gives up (0 = 2^24)
  unsigned char* d_tile_class = nullptr;
  uint32_t* d_tile_order = nullptr;
  size_t tile_cap = 0;
  bool fb_valid = false;
  uint64_t fb_key = 0;
  int wf_bin = 0;              // wavefront mode: sort rays by material class before shading.  Off by default: measured on
                               // C2 (100 spp in one pass, 216 M rays) it takes 9.2 ms off k_shade (36.7 -> 27.5 ms)
                               // and costs 6.3 ms of sorting + 3.1 ms of class counting in k_intersect: 57.6 vs 57.4 ms;
                               // at 16 spp, where the late bounces are launch-bound, 10.5 -> 14.2 ms
  float ms_bin = 0.f;          // ... and the time that takes (option profile)
  size_t counts_cap = 0;
  uint4* d_noise_table = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_a = nullptr, ev_b = nullptr;
  bool timing_pending = false;
  rt_stats stats{};
  int n_cu = 256;
  // Device staging of the entry points that take HOST pointers (rt_render: the frame; rt_resolve: sums, f64 and u8
  // outputs; rt_intersect: rays and hits): grow-only, kept until rt_shutdown, so that a caller that renders pass after
  // pass (Camera::render_progressive) does not pay a hipMalloc + hipFree pair per call.
  BatchFrame* d_batch = nullptr;  // frame table of rt_render_batch_device (grow-only)
  size_t batch_cap = 0;
  const BatchFrame* cur_frames = nullptr;  // batch being issued (render_fused), or NULL
  uint32_t cur_n_frames = 1;
  void* stage[3] = {nullptr, nullptr, nullptr};
  size_t stage_cap[3] = {0, 0, 0};
};

static int ensure_stage(rt_ctx* ctx, int slot, size_t bytes) {
  if (ctx->stage_cap[slot] >= bytes) return RT_OK;
  if (ctx->stage[slot]) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)rti::dev_free(ctx->stage[slot]);
    ctx->stage[slot] = nullptr;
    ctx->stage_cap[slot] = 0;
  }
  hipError_t e = rti::dev_malloc(&ctx->stage[slot], bytes);
  if (e != hipSuccess) {
    ctx->err = std::string("device staging buffer: ") + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP;
  }
  ctx->stage_cap[slot] = bytes;
  return RT_OK;
}

static std::string g_init_err;

#define HIP_TRY(ctx, call)                                                                                   \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) {                                                                                  \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                        \
      return e_ == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP;                                                \
    }                                                                                                        \
  } while (0)

static int fail(rt_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  else g_init_err = msg;
  return code;
}

// xoroshiro128+ short jump (reference src/random.rs:18-28,37-57) — used only to tabulate the jump's GF(2)
// matrix for the noise texture's lattice hash (src/hittable/texture.rs:148-154).
static void xoro_short_jump(uint64_t s[2]) {
  static const uint64_t J[2] = {0xdf900294d8f554a5ULL, 0x170865df4b3201fcULL};
  uint64_t s0 = 0, s1 = 0;
  for (int w = 0; w < 2; w++)
    for (int b = 0; b < 64; b++) {
      if (J[w] & (1ULL << b)) { s0 ^= s[0]; s1 ^= s[1]; }
      uint64_t a = s[0], c = s[1] ^ s[0];
      s[0] = ((a << 24) | (a >> 40)) ^ c ^ (c << 16);
      s[1] = (c << 37) | (c >> 27);
    }
  s[0] = s0;
  s[1] = s1;
}

// binary16 bits of x rounded towards +inf (up) or -inf: the nearest value, stepped once if it lies on the wrong side
static uint16_t half_round(float x, bool up) {
  _Float16 h = (_Float16)x;
  uint16_t b;
  memcpy(&b, &h, 2);
  const float back = (float)h;
  if (std::isnan(x)) return b;
  if (up ? back < x : back > x) {
    const bool neg = b & 0x8000u;
    if ((b & 0x7FFFu) == 0) b = up ? 0x0001u : 0x8001u;  // from +-0 to the smallest denormal of the wanted sign
    else if (up != neg) b++;                            // away from zero
    else b--;                                          // towards zero
  }
  return b;
}

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_last_error(rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_init_err.c_str(); }

uint32_t rt_half_round_bits(float x, int up) { return half_round(x, up != 0); }

void rt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

int rt_init(int device_id, rt_ctx** out) {
  if (!out) return fail(nullptr, RT_E_INVALID_ARG, "rt_init: out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, RT_E_HIP, std::string("rt_init: no HIP device available: ") + hipGetErrorString(e));
  if (device_id < 0 || device_id >= n) return fail(nullptr, RT_E_INVALID_ARG, "rt_init: device_id out of range");
  rt_ctx* ctx = new (std::nothrow) rt_ctx();
  if (!ctx) return fail(nullptr, RT_E_OOM, "rt_init: out of host memory");
  ctx->device = device_id;
  auto bail = [&](const char* what, hipError_t err) {
    g_init_err = std::string("rt_init: ") + what + ": " + hipGetErrorString(err);
    delete ctx;
    return (int)RT_E_HIP;
  };
  if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = rti::stream_create(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = rti::event_create(&ctx->ev_start)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = rti::event_create(&ctx->ev_stop)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = rti::event_create(&ctx->ev_a)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = rti::event_create(&ctx->ev_b)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = rti::dev_malloc(&ctx->own.ray_total, 4 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = rti::dev_malloc(&ctx->own.unit_counter, 2 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = rti::dev_malloc(&ctx->own.cam, sizeof(Camera))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = rti::dev_malloc(&ctx->own.frame1, sizeof(BatchFrame))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemset(ctx->own.ray_total, 0, 4 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
  ctx->last_ray_total = ctx->own.ray_total;
  {
    // rt_render_device_noalloc: this device's copy of the file-scope array `scratch`.  The array's name is only a
    // host-side handle; the address the kernels get is the one hipGetSymbolAddress returns with the device current.
    void* base = nullptr;
    if ((e = hipGetSymbolAddress(&base, HIP_SYMBOL(scratch))) != hipSuccess || !base) return bail("hipGetSymbolAddress(scratch)", e);
    unsigned char* b = static_cast<unsigned char*>(base);
    unsigned char* ctl = b;
    ctx->fixed.cam = reinterpret_cast<Camera*>(ctl);
    ctx->fixed.frame1 = reinterpret_cast<BatchFrame*>(ctl + 640);
    ctx->fixed.ray_total = reinterpret_cast<unsigned long long*>(ctl + 512);
    ctx->fixed.unit_counter = reinterpret_cast<unsigned long long*>(ctl + 768);
  }
  // the partner workgroups' stream and fork / join events (launch_partner): created here so that no render call has to
  if ((e = rti::stream_create(&ctx->partner_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = rti::event_create(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = rti::event_create(&ctx->ev_join, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = rti::dev_malloc(&ctx->d_trav, 2 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = rti::dev_malloc(&ctx->d_bins, 16 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = rti::dev_malloc(&ctx->d_dbg, 16 * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = rti::dev_malloc(&ctx->d_timeline, 16 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemset(ctx->d_dbg, 0, 16 * sizeof(float))) != hipSuccess) return bail("hipMemset", e);
  if ((e = hipMemset(ctx->d_trav, 0, 2 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
  // noise jump table: entry [byte position i][value v] = short_jump(state with only that byte set)
  std::vector<uint32_t> tab(16 * 256 * 4);
  for (int i = 0; i < 16; i++)
    for (int v = 0; v < 256; v++) {
      uint64_t s[2] = {0, 0};
      s[i / 8] = (uint64_t)v << (8 * (i % 8));
      xoro_short_jump(s);
      uint32_t* t = &tab[(size_t)(i * 256 + v) * 4];
      t[0] = (uint32_t)s[0]; t[1] = (uint32_t)(s[0] >> 32); t[2] = (uint32_t)s[1]; t[3] = (uint32_t)(s[1] >> 32);
    }
  // Byte positions 4..7 hold the sign extension of a lattice coordinate (texture.rs:149: `x as u64`): all 0x00 or all
  // 0xff, never anything else.  Their four 0xff entries are folded into ONE (position 4's; the other three become zero),
  // so the sign costs the device one row fetch instead of four (noise_corners_inline) and the row-by-row form
  // (noise_jump_part) still gets the same XOR.
  for (int k = 0; k < 4; k++) {
    uint32_t& folded = tab[(size_t)(4 * 256 + 255) * 4 + k];
    for (int i = 5; i < 8; i++) {
      uint32_t& t = tab[(size_t)(i * 256 + 255) * 4 + k];
      folded ^= t;
      t = 0;
    }
  }
  if ((e = rti::dev_malloc(&ctx->d_noise_table, tab.size() * 4)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemcpy(ctx->d_noise_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)
    return bail("hipMemcpy", e);
  *out = ctx;
  return RT_OK;
}

static void free_work(rt_ctx* ctx) {
  if (ctx->wb.slab) (void)rti::dev_free(ctx->wb.slab);
  ctx->wb = WorkBuffers();
}

int rt_shutdown(rt_ctx* ctx) {
  if (!ctx) return RT_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_work(ctx);
  if (ctx->d_counts) (void)rti::dev_free(ctx->d_counts);
  if (ctx->d_tile_class) (void)rti::dev_free(ctx->d_tile_class);
  if (ctx->d_tile_order) (void)rti::dev_free(ctx->d_tile_order);
  if (ctx->own.ray_total) (void)rti::dev_free(ctx->own.ray_total);
  if (ctx->own.unit_counter) (void)rti::dev_free(ctx->own.unit_counter);
  if (ctx->own.cam) (void)rti::dev_free(ctx->own.cam);
  if (ctx->own.frame1) (void)rti::dev_free(ctx->own.frame1);
  if (ctx->d_trav) (void)rti::dev_free(ctx->d_trav);
  if (ctx->d_bins) (void)rti::dev_free(ctx->d_bins);
  if (ctx->d_dbg) (void)rti::dev_free(ctx->d_dbg);
  if (ctx->d_timeline) (void)rti::dev_free(ctx->d_timeline);
  if (ctx->d_noise_table) (void)rti::dev_free(ctx->d_noise_table);
  for (int i = 0; i < 3; i++)
    if (ctx->stage[i]) (void)rti::dev_free(ctx->stage[i]);
  if (ctx->d_batch) (void)rti::dev_free(ctx->d_batch);
  if (ctx->ev_start) (void)rti::event_destroy(ctx->ev_start);
  if (ctx->ev_stop) (void)rti::event_destroy(ctx->ev_stop);
  if (ctx->ev_a) (void)rti::event_destroy(ctx->ev_a);
  if (ctx->ev_b) (void)rti::event_destroy(ctx->ev_b);
  if (ctx->ev_fork) (void)rti::event_destroy(ctx->ev_fork);
  if (ctx->ev_join) (void)rti::event_destroy(ctx->ev_join);
  if (ctx->partner_stream) (void)rti::stream_destroy(ctx->partner_stream);
  if (ctx->stream) (void)rti::stream_destroy(ctx->stream);
  delete ctx;
  return RT_OK;
}

int rt_synchronize(rt_ctx* ctx) {
  if (!ctx) return RT_E_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RT_OK;
}

int rt_set_option(rt_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return RT_E_INVALID_ARG;
  std::string k(key);
  if (k == "max_paths") {
    if (value < 64 || value > (int64_t)0xFFFFFFFFll - 4096) return fail(ctx, RT_E_INVALID_ARG, "max_paths out of range");
    ctx->max_paths = value;
  } else if (k == "profile") {
    ctx->profile = value ? 1 : 0;
  } else if (k == "mode") {
    if (value != 0 && value != 1) return fail(ctx, RT_E_INVALID_ARG, "mode must be 0 (wavefront) or 1 (fused)");
    ctx->mode = (int)value;
  } else if (k == "fused_ku") {
    if (value < 0 || value > 65535) return fail(ctx, RT_E_INVALID_ARG, "fused_ku out of range");
    ctx->fused_ku = (int)value;
  } else if (k == "fused_lds") {
    ctx->fused_lds = value ? 1 : 0;
  } else if (k == "probe_tree") {
    if (value < 0 || value > 4) return fail(ctx, RT_E_INVALID_ARG, "probe_tree must be in 0..4");
    ctx->probe_tree = (int)value;
  } else if (k == "dbg_max_cached") {
    if (value < 0 || value > 65535) return fail(ctx, RT_E_INVALID_ARG, "dbg_max_cached out of range");
    ctx->dbg_max_cached = (int)value;
  } else if (k == "bvh_optimise") {
    if (value < 0 || value > 16) return fail(ctx, RT_E_INVALID_ARG, "bvh_optimise must be in 0..16");
    ctx->bvh_optimise = (int)value;
  } else if (k == "uv_tables") {
    ctx->uv_tables = value ? 1 : 0;
  } else if (k == "probe_count") {
    ctx->probe_count = value ? 1 : 0;
  } else if (k == "fused_tail4" || k == "fused_tail2") {
    if (value < 0 || value > 400) return fail(ctx, RT_E_INVALID_ARG, k + " must be in 0..400 (per mille of the frame's rows)");
    (k == "fused_tail4" ? ctx->fused_tail4 : ctx->fused_tail2) = (int)value;
  } else if (k == "wf_bin") {
    ctx->wf_bin = value ? 1 : 0;
  } else if (k == "fused_slots") {
    if (value < 0 || value > RT_MAX_SLOTS) return fail(ctx, RT_E_INVALID_ARG, "fused_slots must be in 0.." + std::to_string(RT_MAX_SLOTS));
    ctx->fused_slots = (int)value;
  } else if (k == "dbg_main_cap") {
    ctx->dbg_main_cap = (int)std::max<int64_t>(0, std::min<int64_t>(value, 65536));
  } else if (k == "dbg_watchdog_polls") {
    if (value < 0 || value > (1 << 24)) return fail(ctx, RT_E_INVALID_ARG, "dbg_watchdog_polls must be in 0..2^24");
    ctx->dbg_watchdog_polls = (int)value;
  } else if (k == "fused_kr_sync") {
    if (value < 1 || value > 64) return fail(ctx, RT_E_INVALID_ARG, "fused_kr_sync must be in 1..64");
    ctx->fused_kr_sync = (int)value;
  } else if (k == "merge_boxes") {
    ctx->merge_boxes = value ? 1 : 0;  // (applies to scenes uploaded afterwards)
  } else if (k == "global_prims") {
    ctx->global_prims = value ? 1 : 0;  // (applies to scenes uploaded afterwards)
  } else if (k == "fused_feedback") {
    ctx->fused_feedback = value ? 1 : 0;
    ctx->fb_valid = false;
  } else if (k == "fused_partner_min") {
    if (value < 0 || value > 65536) return fail(ctx, RT_E_INVALID_ARG, "fused_partner_min out of range");
    ctx->fused_partner_min = (int)value;
  } else if (k == "fused_partner") {
    ctx->fused_partner = value ? 1 : 0;
  } else if (k == "fused_half") {
    ctx->fused_half = value ? 1 : 0;
  } else if (k == "fused_wide") {
    ctx->fused_wide = value ? 1 : 0;
  } else if (k == "fused_stats") {
    ctx->fused_stats = value ? 1 : 0;
  } else if (k == "fused_resume") {
    if (value < 0 || value > 2) return fail(ctx, RT_E_INVALID_ARG, "fused_resume must be 0, 1 or 2");
    ctx->fused_resume = (int)value;
  } else if (k == "fused_kb" || k == "fused_kn" || k == "fused_kr") {
    if (value < 1 || value > 64) return fail(ctx, RT_E_INVALID_ARG, k + " must be in 1..64");
    (k == "fused_kb" ? ctx->fused_kb : k == "fused_kn" ? ctx->fused_kn : ctx->fused_kr) = (int)value;
  } else {
    return fail(ctx, RT_E_INVALID_ARG, "rt_set_option: unknown key " + k);
  }
  return RT_OK;
}

// ------------------------------------------------------------------------------------------------ scene upload
static bool finite3(const double* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

static int validate_texture_graph(rt_ctx* ctx, const rt_scene_desc* d) {
  for (uint64_t i = 0; i < d->n_textures; i++) {
    const rt_texture& t = d->textures[i];
    if (t.kind > RT_TEX_NOISE) return fail(ctx, RT_E_INVALID_ARG, "texture " + std::to_string(i) + ": unknown kind");
    if (t.kind == RT_TEX_CHECKER) {
      if (t.odd < 0 || (uint64_t)t.odd >= d->n_textures || t.even < 0 || (uint64_t)t.even >= d->n_textures)
        return fail(ctx, RT_E_INVALID_ARG, "texture " + std::to_string(i) + ": checker child out of range");
      // children must come earlier in the table (acyclic, bounded nesting)
      if ((uint64_t)t.odd >= i || (uint64_t)t.even >= i)
        return fail(ctx, RT_E_INVALID_ARG, "texture " + std::to_string(i) + ": checker children must precede it");
    }
    if (t.kind == RT_TEX_IMAGE) {
      // checked arithmetic: width * height * 4 and offset + bytes must not wrap, and the device indexes texels with 32 bits
      const uint64_t texels = (uint64_t)t.width * t.height;  // < 2^64: both factors are 32-bit
      const uint64_t avail = d->n_texel_bytes / 4;
      if ((t.texel_offset & 3) || texels > avail || t.texel_offset / 4 > avail - texels || (texels && !d->texels) ||
          t.texel_offset / 4 + texels > 0xFFFFFFFFull)
        return fail(ctx, RT_E_INVALID_ARG, "texture " + std::to_string(i) + ": texel range outside the blob");
    }
  }
  // nesting depth of checkers <= 7 (device loop bound)
  std::vector<int> depth(d->n_textures, 0);
  for (uint64_t i = 0; i < d->n_textures; i++)
    if (d->textures[i].kind == RT_TEX_CHECKER) {
      depth[i] = 1 + std::max(depth[(size_t)d->textures[i].odd], depth[(size_t)d->textures[i].even]);
      if (depth[i] > 7) return fail(ctx, RT_E_UNSUPPORTED, "checker textures nested deeper than 7");
    }
  return RT_OK;
}

int rt_scene_free(rt_ctx* ctx, rt_scene* s) {
  if (!s) return RT_OK;
  if (ctx) (void)hipSetDevice(ctx->device);
  if (s->d_nodes) (void)rti::dev_free(s->d_nodes);
  if (s->d_nodes4) (void)rti::dev_free(s->d_nodes4);
  if (s->d_nodes4h) (void)rti::dev_free(s->d_nodes4h);
  if (s->d_global) (void)rti::dev_free(s->d_global);
  if (s->d_prims) (void)rti::dev_free(s->d_prims);
  if (s->d_mats) (void)rti::dev_free(s->d_mats);
  if (s->d_texs) (void)rti::dev_free(s->d_texs);
  if (s->d_texels) (void)rti::dev_free(s->d_texels);
  if (s->d_spheres) (void)rti::dev_free(s->d_spheres);
  delete s;
  return RT_OK;
}

static int scene_upload_impl(rt_ctx* ctx, const rt_scene_desc* d, rt_scene* sc) {
  if (d->n_prims == 0) return fail(ctx, RT_E_INVALID_ARG, "scene has no primitives");
  if (d->n_prims > (1u << 30)) return fail(ctx, RT_E_UNSUPPORTED, "too many primitives");
  if (!d->prims || !d->materials || d->n_materials == 0) return fail(ctx, RT_E_INVALID_ARG, "prims/materials missing");
  if (d->n_textures && !d->textures) return fail(ctx, RT_E_INVALID_ARG, "textures missing");
  int rc = validate_texture_graph(ctx, d);
  if (rc) return rc;
  for (uint64_t i = 0; i < d->n_materials; i++) {
    const rt_material& m = d->materials[i];
    if (m.kind > RT_MAT_ISOTROPIC) return fail(ctx, RT_E_INVALID_ARG, "material " + std::to_string(i) + ": unknown kind");
    bool needs_tex = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
    if (needs_tex && (m.tex < 0 || (uint64_t)m.tex >= d->n_textures))
      return fail(ctx, RT_E_INVALID_ARG, "material " + std::to_string(i) + ": texture index out of range");
  }
  std::vector<Prim> prims(d->n_prims);
  std::vector<rtb::Box> boxes(d->n_prims);
  sc->features = 0;
  sc->spheres_only = true;
  for (uint64_t i = 0; i < d->n_textures; i++) {
    if (d->textures[i].kind == RT_TEX_CHECKER) sc->features |= RT_F_TEX_CHECKER;
    if (d->textures[i].kind == RT_TEX_IMAGE) sc->features |= RT_F_TEX_IMAGE;
    if (d->textures[i].kind == RT_TEX_NOISE) sc->features |= RT_F_TEX_NOISE;
  }
  for (uint64_t i = 0; i < d->n_prims; i++) {
    const rt_prim& p = d->prims[i];
    Prim& q = prims[i];
    memset(&q, 0, sizeof(q));
    if (p.material_id >= d->n_materials)
      return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": material_id out of range");
    q.type = p.type;
    q.mat = p.material_id;
    q.obj = p.object_id;
    rtb::Box& b = boxes[i];
    if (p.type == RT_PRIM_SPHERE || p.type == RT_PRIM_MOVING_SPHERE) {
      const double* c1 = p.type == RT_PRIM_MOVING_SPHERE ? p.b : p.a;
      if (!finite3(p.a) || !finite3(c1) || !std::isfinite(p.radius) || p.radius == 0.0)
        return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": non-finite sphere or zero radius");
      for (int a = 0; a < 3; a++) {
        q.f[a] = p.a[a];
        q.f[3 + a] = c1[a];
        // geometry.rs:20-28,90-100: box of center -/+ radius (ordered: radius may be negative), union over
        // both end points for a moving sphere
        double r = std::fabs(p.radius);
        b.lo[a] = std::fmin(p.a[a], c1[a]) - r;
        b.hi[a] = std::fmax(p.a[a], c1[a]) + r;
      }
      q.f[6] = p.radius;
    } else if (p.type == RT_PRIM_QUAD) {
      sc->features |= RT_F_QUAD;
      sc->spheres_only = false;
      if (!finite3(p.a) || !finite3(p.b) || !finite3(p.c))
        return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": non-finite quad");
      const double *Q = p.a, *u = p.b, *v = p.c;
      double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
      double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (!(len > 0)) return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": degenerate quad");
      // w = n_raw / (n_raw . n_raw); the device evaluates alpha = (p - Q).(v x w), beta = (p - Q).(w x u) (quad_hit)
      const double w[3] = {n[0] / (len * len), n[1] / (len * len), n[2] / (len * len)};
      const double vxw[3] = {v[1] * w[2] - v[2] * w[1], v[2] * w[0] - v[0] * w[2], v[0] * w[1] - v[1] * w[0]};
      const double wxu[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
      for (int a = 0; a < 3; a++) {
        q.f[a] = Q[a]; q.f[3 + a] = vxw[a]; q.f[6 + a] = wxu[a]; q.f[9 + a] = n[a] / len;
        double c[4] = {Q[a], Q[a] + u[a], Q[a] + v[a], Q[a] + u[a] + v[a]};
        b.lo[a] = std::fmin(std::fmin(c[0], c[1]), std::fmin(c[2], c[3]));
        b.hi[a] = std::fmax(std::fmax(c[0], c[1]), std::fmax(c[2], c[3]));
        if (b.hi[a] - b.lo[a] < 0.0001) { b.lo[a] -= 0.00005; b.hi[a] += 0.00005; }
      }
      q.f[12] = q.f[9] * Q[0] + q.f[10] * Q[1] + q.f[11] * Q[2];
      q.f[13] = 1.0 / len;
    } else if (p.type == RT_PRIM_MEDIUM_SPHERE || p.type == RT_PRIM_MEDIUM_BOX) {
      sc->features |= RT_F_MEDIUM;
      sc->spheres_only = false;
      if (!(p.density > 0) || !std::isfinite(p.density))
        return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": medium density must be > 0");
      if (d->materials[p.material_id].kind != RT_MAT_ISOTROPIC)
        return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": a medium needs an ISOTROPIC material");
      q.f[12] = -1.0 / p.density;
      if (p.type == RT_PRIM_MEDIUM_SPHERE) {
        if (!finite3(p.a) || !std::isfinite(p.radius) || p.radius == 0.0)
          return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": bad medium sphere");
        for (int a = 0; a < 3; a++) {
          q.f[a] = p.a[a];
          b.lo[a] = p.a[a] - std::fabs(p.radius);
          b.hi[a] = p.a[a] + std::fabs(p.radius);
        }
        q.f[6] = p.radius;
        q.f[3] = q.f[12];  // (inside the first 64 bytes of the record: prim_test's one-round-trip fetch)
      } else {
        if (!finite3(p.a) || !finite3(p.b) || !finite3(p.c) || !finite3(p.d))
          return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": non-finite medium box");
        const double* e[3] = {p.b, p.c, p.d};
        for (int k = 0; k < 3; k++)
          if (!(e[k][0] * e[k][0] + e[k][1] * e[k][1] + e[k][2] * e[k][2] > 0))
            return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": degenerate medium box edge");
        for (int a = 0; a < 3; a++) {
          q.f[a] = p.a[a];
          // device record: edges pre-divided by their squared length, so that dot(x, f[3+3k..]) is the box coordinate
          // of x along edge k in [0, 1] without a division per test
          for (int k = 0; k < 3; k++)
            q.f[3 + 3 * k + a] = e[k][a] / (e[k][0] * e[k][0] + e[k][1] * e[k][1] + e[k][2] * e[k][2]);
          double lo = p.a[a], hi = p.a[a];
          for (int k = 0; k < 3; k++) { lo += std::fmin(0.0, e[k][a]); hi += std::fmax(0.0, e[k][a]); }
          b.lo[a] = lo; b.hi[a] = hi;
        }
      }
    } else {
      return fail(ctx, RT_E_INVALID_ARG, "prim " + std::to_string(i) + ": unknown primitive type");
    }
  }
  // Merged boxes (option merge_boxes): six consecutive quads of one material that are the six faces of an axis-aligned
  // box — what the host's box() helper emits (raytracer.cpp; the book's box()) when no rotation is baked in — are tested
  // as ONE primitive (RT_PRIM_AABOX, aabox_hit): one slab test and one division instead of up to six plane tests, one
  // leaf instead of six in the tree (final scene: 3 409 -> 1 409 tree primitives).  The quads stay in the table — a hit
  // reports the face's quad, so object ids, normals, uv and materials are untouched — and simply leave the tree.
  const size_t n_orig = prims.size();
  std::vector<char> in_group(n_orig, 0);
  if (ctx->merge_boxes) {
    for (size_t i = 0; i + 6 <= n_orig;) {
      bool ok = true;
      int32_t face[6] = {-1, -1, -1, -1, -1, -1};
      double lo[3], hi[3];
      for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
      for (size_t k = i; k < i + 6 && ok; k++) {
        ok = d->prims[k].type == RT_PRIM_QUAD && d->prims[k].material_id == d->prims[i].material_id;
        if (ok)
          for (int a = 0; a < 3; a++) {
            const double c0 = d->prims[k].a[a], c1 = c0 + d->prims[k].b[a], c2 = c0 + d->prims[k].c[a], c3 = c1 + d->prims[k].c[a];
            lo[a] = std::fmin(lo[a], std::fmin(std::fmin(c0, c1), std::fmin(c2, c3)));
            hi[a] = std::fmax(hi[a], std::fmax(std::fmax(c0, c1), std::fmax(c2, c3)));
          }
      }
      for (int a = 0; a < 3 && ok; a++) ok = hi[a] > lo[a];
      for (size_t k = i; k < i + 6 && ok; k++) {
        // u and v along two different axes, the quad lies in a face plane of [lo, hi] and spans that face exactly
        const rt_prim& q = d->prims[k];
        int au = -1, av = -1;
        for (int a = 0; a < 3; a++) {
          if (q.b[a] != 0.0) { if (au >= 0) ok = false; au = a; }
          if (q.c[a] != 0.0) { if (av >= 0) ok = false; av = a; }
        }
        if (!ok || au < 0 || av < 0 || au == av) { ok = false; break; }
        const int an = 3 - au - av;  // the axis of the normal
        const double e[2][2] = {{std::fmin(q.a[au], q.a[au] + q.b[au]), std::fmax(q.a[au], q.a[au] + q.b[au])},
                                {std::fmin(q.a[av], q.a[av] + q.c[av]), std::fmax(q.a[av], q.a[av] + q.c[av])}};
        if (e[0][0] != lo[au] || e[0][1] != hi[au] || e[1][0] != lo[av] || e[1][1] != hi[av]) { ok = false; break; }
        const int side = q.a[an] == lo[an] ? 0 : q.a[an] == hi[an] ? 1 : -1;
        if (side < 0 || face[2 * an + side] >= 0) { ok = false; break; }
        face[2 * an + side] = (int32_t)k;
      }
      if (!ok) { i++; continue; }
      Prim b;
      memset(&b, 0, sizeof(b));
      b.type = RT_PRIM_AABOX;
      b.mat = d->prims[i].material_id;
      b.obj = d->prims[i].object_id;
      for (int a = 0; a < 3; a++) { b.f[a] = lo[a]; b.f[3 + a] = hi[a]; }
      memcpy(&b.f[6], face, sizeof(face));
      rtb::Box bb;
      for (int a = 0; a < 3; a++) { bb.lo[a] = lo[a]; bb.hi[a] = hi[a]; }
      prims.push_back(b);
      boxes.push_back(bb);
      for (size_t k = i; k < i + 6; k++) in_group[k] = 1;
      sc->features |= RT_F_BOX;
      i += 6;
    }
  }
  in_group.resize(prims.size(), 0);
  // what rays are tested against: every primitive that is not a face of a merged box
  std::vector<int32_t> cand;
  for (size_t i = 0; i < prims.size(); i++)
    if (!in_group[i]) cand.push_back((int32_t)i);
  // Global primitives (Scene::n_global): (moving) spheres whose box is at least a quarter of the scene's, at most two, in
  // scenes with a tree to speak of.  They stay in the primitive tables; the tree is built over the others.
  std::vector<int32_t> in_tree;
  int32_t globals[RT_MAX_GLOBAL] = {0, 0, 0, 0, 0, 0, 0, 0};
  int n_global = 0;
  // tiny scenes (the Cornell boxes: 8 primitives): no tree, every ray tests every primitive, uniformly across the wave
  const bool tiny = ctx->global_prims && !cand.empty() && cand.size() <= RT_MAX_GLOBAL;
  if (tiny) {
    for (int32_t i : cand) globals[n_global++] = i;
  } else if (ctx->global_prims && cand.size() >= 32) {
    rtb::Box all = boxes[(size_t)cand[0]];
    for (int32_t i : cand)
      for (int a = 0; a < 3; a++) { all.lo[a] = std::min(all.lo[a], boxes[(size_t)i].lo[a]); all.hi[a] = std::max(all.hi[a], boxes[(size_t)i].hi[a]); }
    auto area = [](const rtb::Box& b) {
      const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
      return x * y + y * z + z * x;
    };
    const double whole = area(all);
    for (size_t k = 0; k < cand.size() && n_global < 2; k++) {
      const size_t i = (size_t)cand[k];
      if ((prims[i].type <= RT_PRIM_MOVING_SPHERE || prims[i].type == RT_PRIM_MEDIUM_SPHERE) && area(boxes[i]) >= 0.25 * whole)
        globals[n_global++] = (int32_t)i;
    }
  }
  std::vector<rtb::Box> tree_boxes;
  for (int32_t i : cand)
    if (tiny || !((n_global > 0 && i == globals[0]) || (n_global > 1 && i == globals[1]))) {
      in_tree.push_back(i);
      tree_boxes.push_back(boxes[(size_t)i]);
    }
  // (a tiny scene keeps its tree in memory — probe forms and sizes are unchanged — but no render or probe walks it)
  rtb::BuildResult bvh = rtb::build_bvh(tree_boxes, RT_STACK_DEPTH, ctx->bvh_optimise);
  {  // leaf references back to positions in the primitive tables
    auto fix = [&in_tree](int32_t& ref) { if (ref < 0 && ref != rtb::EMPTY_CHILD) ref = ~in_tree[(size_t)~ref]; };
    for (rtb::FlatNode& nd : bvh.nodes) { fix(nd.left); fix(nd.right); }
    fix(bvh.root);
  }
  sc->view.n_global = n_global;
  for (int g = 0; g < n_global; g++) {
    const uint32_t ty = prims[(size_t)globals[g]].type;
    sc->global_types[ty <= 2u ? ty : ty == RT_PRIM_AABOX ? 4u : 3u]++;
  }
  sc->view.tree_empty = tiny ? 1 : 0;
  HIP_TRY(ctx, rti::dev_malloc(&sc->d_global, RT_MAX_GLOBAL * sizeof(int32_t)));
  HIP_TRY(ctx, hipMemcpy(sc->d_global, globals, RT_MAX_GLOBAL * sizeof(int32_t), hipMemcpyHostToDevice));
  sc->view.global_list = sc->d_global;
  sc->bvh_depth = bvh.depth;
  if (getenv("RT_VERBOSE"))
    fprintf(stderr, "[rt] scene: %zu prims, %zu nodes, BVH depth %d\n", boxes.size(), bvh.nodes.size(), bvh.depth);
  if (bvh.depth >= RT_STACK_DEPTH) return fail(ctx, RT_E_INTERNAL, "BVH deeper than the traversal stack");

  std::vector<Material> mats(d->n_materials);
  for (uint64_t i = 0; i < d->n_materials; i++) {
    const rt_material& m = d->materials[i];
    Material& o = mats[i];
    memset(&o, 0, sizeof(o));
    o.kind = m.kind;
    // METAL and DIELECTRIC take no texture: whatever the caller left in `tex` (a zero-initialised struct says 0) is ignored
    const bool takes_tex = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
    o.tex = takes_tex ? m.tex : -1;
    for (int a = 0; a < 3; a++) o.albedo[a] = (float)m.albedo[a];
    o.fuzz = (float)m.fuzz;
    o.ior = (float)m.ior;
    // inline solid-colour textures so the common case needs no texture fetch
    if (o.tex >= 0 && d->textures[(size_t)m.tex].kind == RT_TEX_SOLID) {
      for (int a = 0; a < 3; a++) o.albedo[a] = (float)d->textures[(size_t)m.tex].rgb[a];
      o.tex = -1;
    }
  }
  std::vector<Texture> texs(std::max<uint64_t>(d->n_textures, 1));
  memset(texs.data(), 0, texs.size() * sizeof(Texture));
  for (uint64_t i = 0; i < d->n_textures; i++) {
    const rt_texture& t = d->textures[i];
    Texture& o = texs[i];
    o.kind = t.kind;
    o.odd = t.odd;
    o.even = t.even;
    o.width = t.width;
    o.height = t.height;
    o.texel_off = (uint32_t)(t.texel_offset / 4);
    for (int a = 0; a < 3; a++) o.rgb[a] = (float)t.rgb[a];
    o.inv_scale = t.inv_scale;
    if (t.kind == RT_TEX_CHECKER) {  // inline solid children (children precede their checker: validated above)
      if (d->textures[(size_t)t.odd].kind == RT_TEX_SOLID) {
        for (int a = 0; a < 3; a++) o.rgb[a] = (float)d->textures[(size_t)t.odd].rgb[a];
        o.odd = -1;
      }
      if (d->textures[(size_t)t.even].kind == RT_TEX_SOLID) {
        for (int a = 0; a < 3; a++) o.rgb2[a] = (float)d->textures[(size_t)t.even].rgb[a];
        o.even = -1;
      }
    }
  }
  // Compact the device texture table: solid colours were inlined into their users above, so nothing refers to
  // them by index any more (book_cover: 320 textures -> 1).  The table is small enough for LDS afterwards.
  {
    std::vector<int32_t> remap(texs.size(), -1);
    std::vector<Texture> kept;
    for (uint64_t i = 0; i < d->n_textures; i++) {
      if (texs[i].kind == RT_TEX_SOLID) continue;
      Texture t = texs[i];
      if (t.kind == RT_TEX_CHECKER) {  // children precede their checker, so they are already remapped
        if (t.odd >= 0) t.odd = remap[(size_t)t.odd];
        if (t.even >= 0) t.even = remap[(size_t)t.even];
      }
      remap[i] = (int32_t)kept.size();
      kept.push_back(t);
    }
    for (Material& m : mats)
      if (m.tex >= 0) m.tex = remap[(size_t)m.tex];
    if (kept.empty()) kept.emplace_back(Texture{});
    texs.swap(kept);
  }
  static_assert(sizeof(rtb::FlatNode) == sizeof(BvhNode), "node layout mismatch");
  size_t n_nodes = std::max<size_t>(bvh.nodes.size(), 1);
  HIP_TRY(ctx, rti::dev_malloc(&sc->d_nodes, n_nodes * sizeof(BvhNode)));
  if (!bvh.nodes.empty())
    HIP_TRY(ctx, hipMemcpy(sc->d_nodes, bvh.nodes.data(), bvh.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
  {
    static_assert(sizeof(rtb::WideNode) == 128, "wide node layout mismatch");
    rtb::WideResult wide = rtb::collapse_bvh4(bvh);
    sc->wide_stack = wide.stack_need;
    sc->view.n_nodes4 = (int32_t)wide.nodes.size();
    // the device never tests for unused slots: their inverted boxes cannot be entered, and their links repeat slot 0's
    for (rtb::WideNode& w : wide.nodes)
      for (int k = 1; k < 4; k++)
        if (w.child[k] == rtb::EMPTY_CHILD) w.child[k] = w.child[0];
    if (getenv("RT_VERBOSE")) fprintf(stderr, "[rt] four-wide tree: %zu nodes, stack %d\n", wide.nodes.size(), wide.stack_need);
    {
      struct HalfNode { uint16_t p[3][8]; int32_t child[4]; };
      static_assert(sizeof(HalfNode) == 64, "half node layout mismatch");
      std::vector<HalfNode> hn(std::max<size_t>(wide.nodes.size(), 1));
      memset(hn.data(), 0, hn.size() * sizeof(HalfNode));
      sc->half_ok = !wide.nodes.empty();
      for (size_t i = 0; i < wide.nodes.size(); i++)
        for (int k = 0; k < 4; k++) {
          for (int a = 0; a < 3; a++) {
            // beyond the binary16 range a plane would round to infinity: still conservative, but it culls nothing
            const float lo = wide.nodes[i].lo[a][k], hi = wide.nodes[i].hi[a][k];
            if (lo <= hi && (std::fabs(lo) > 60000.f || std::fabs(hi) > 60000.f)) sc->half_ok = false;
            hn[i].p[a][k] = half_round(wide.nodes[i].lo[a][k], false);
            hn[i].p[a][4 + k] = half_round(wide.nodes[i].hi[a][k], true);
          }
          hn[i].child[k] = wide.nodes[i].child[k];
        }
      HIP_TRY(ctx, rti::dev_malloc(&sc->d_nodes4h, hn.size() * sizeof(HalfNode)));
      HIP_TRY(ctx, hipMemcpy(sc->d_nodes4h, hn.data(), hn.size() * sizeof(HalfNode), hipMemcpyHostToDevice));
      sc->view.nodes4h = sc->d_nodes4h;
    }
    HIP_TRY(ctx, rti::dev_malloc(&sc->d_nodes4, std::max<size_t>(wide.nodes.size(), 1) * sizeof(rtb::WideNode)));
    if (!wide.nodes.empty())
      HIP_TRY(ctx, hipMemcpy(sc->d_nodes4, wide.nodes.data(), wide.nodes.size() * sizeof(rtb::WideNode), hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, rti::dev_malloc(&sc->d_prims, prims.size() * sizeof(Prim)));
  HIP_TRY(ctx, hipMemcpy(sc->d_prims, prims.data(), prims.size() * sizeof(Prim), hipMemcpyHostToDevice));
  HIP_TRY(ctx, rti::dev_malloc(&sc->d_mats, mats.size() * sizeof(Material)));
  HIP_TRY(ctx, hipMemcpy(sc->d_mats, mats.data(), mats.size() * sizeof(Material), hipMemcpyHostToDevice));
  if (d->n_texel_bytes) {
    // Behind the texels: per image size, the table that sphere_texel_checked (rt_device.h) verifies its f32 guess of the
    // texel against — W + 1 unit vectors (cos b_i, sin b_i), b_i = 2 pi i / W - pi (the longitudes where the texel column
    // changes), then H + 1 values cos(pi j / H) (the heights where the row changes); computed in long double.
    std::vector<double> tables;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> table_of;  // (W, H) -> offset in 32-bit words from the texel base
    const uint64_t texel_words = ((d->n_texel_bytes + 15) & ~15ull) / 4;
    for (Texture& t : texs) {
      if (t.kind != RT_TEX_IMAGE || !t.width || !t.height || !ctx->uv_tables) continue;
      if ((uint64_t)t.width + t.height > (1u << 20)) continue;  // (absurd sizes: the f64 formulas serve)
      const auto key = std::make_pair(t.width, t.height);
      auto it = table_of.find(key);
      if (it == table_of.end()) {
        const uint64_t off = texel_words + tables.size() * 2;
        if (off > 0xFFFFFFFFull) continue;
        // (cos, sin) of (pi / 2) * m / n with the quarter turns taken out in integers: multiples of pi / 2 come out as
        // exactly (+-1, 0) / (0, +-1), so a normal exactly ON an axis meets an exact boundary (inclusive below, like floor)
        auto unit = [](int64_t m, int64_t n, double& c, double& sn) {
          const long double half_pi = 1.57079632679489661923132169163975144L;
          int64_t q = (2 * m + n) / (2 * n);
          if ((2 * m + n) % (2 * n) < 0) q--;  // floor division
          const int64_t r = m - q * n;           // |r| <= n / 2
          const long double a = half_pi * (long double)r / (long double)n;
          const long double cr = cosl(a), sr = sinl(a);
          switch (((q % 4) + 4) % 4) {
            case 0: c = (double)cr; sn = (double)sr; break;
            case 1: c = (double)-sr; sn = (double)cr; break;
            case 2: c = (double)-cr; sn = (double)-sr; break;
            default: c = (double)sr; sn = (double)-cr; break;
          }
        };
        for (uint32_t i = 0; i <= t.width; i++) {  // b_i = 2 pi i / W - pi = (pi / 2) * (4 i - 2 W) / W
          double c, sn;
          unit(4 * (int64_t)i - 2 * (int64_t)t.width, (int64_t)t.width, c, sn);
          tables.push_back(c);
          tables.push_back(sn + 0.0);  // (-0 -> +0: the seam's two entries are the same direction, exactly (-1, 0))
        }
        for (uint32_t j = 0; j <= t.height; j++) {  // cos(pi j / H) = cos((pi / 2) * 2 j / H)
          double c, sn;
          unit(2 * (int64_t)j, (int64_t)t.height, c, sn);
          tables.push_back(c + 0.0);
        }
        if (tables.size() & 1) tables.push_back(0.0);  // the next table starts on 16 bytes
        it = table_of.emplace(key, (uint32_t)off).first;
      }
      t.uvtab_off = it->second;
    }
    HIP_TRY(ctx, rti::dev_malloc(&sc->d_texels, texel_words * 4 + tables.size() * sizeof(double) + 16));
    HIP_TRY(ctx, hipMemcpy(sc->d_texels, d->texels, d->n_texel_bytes, hipMemcpyHostToDevice));
    if (!tables.empty())
      HIP_TRY(ctx, hipMemcpy(sc->d_texels + texel_words, tables.data(), tables.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, rti::dev_malloc(&sc->d_texs, texs.size() * sizeof(Texture)));
  HIP_TRY(ctx, hipMemcpy(sc->d_texs, texs.data(), texs.size() * sizeof(Texture), hipMemcpyHostToDevice));
  if (sc->spheres_only) {
    std::vector<SphereRec> recs(prims.size());
    for (size_t i = 0; i < prims.size(); i++) {
      SphereRec& r = recs[i];
      memset(&r, 0, sizeof(r));
      for (int a = 0; a < 3; a++) { r.c0[a] = prims[i].f[a]; r.c1[a] = prims[i].f[3 + a]; }
      r.r = prims[i].f[6];
      r.type = prims[i].type;
      r.mat = prims[i].mat;
    }
    HIP_TRY(ctx, rti::dev_malloc(&sc->d_spheres, recs.size() * sizeof(SphereRec)));
    HIP_TRY(ctx, hipMemcpy(sc->d_spheres, recs.data(), recs.size() * sizeof(SphereRec), hipMemcpyHostToDevice));
  }
  sc->view.spheres = sc->d_spheres;
  sc->view.bvh_depth = bvh.depth;
  sc->view.nodes = sc->d_nodes;
  sc->view.nodes4 = sc->d_nodes4;
  sc->view.prims = sc->d_prims;
  sc->view.mats = sc->d_mats;
  sc->view.texs = sc->d_texs;
  sc->view.texels = sc->d_texels;
  sc->view.noise_table = ctx->d_noise_table;
  sc->view.n_nodes = (int32_t)bvh.nodes.size();
  sc->view.n_prims = (int32_t)prims.size();
  sc->view.n_mats = (int32_t)mats.size();
  sc->view.n_texs = (int32_t)texs.size();
  sc->view.root = bvh.root;
  return RT_OK;
}

int rt_scene_upload(rt_ctx* ctx, const rt_scene_desc* desc, rt_scene** out) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!desc || !out) return fail(ctx, RT_E_INVALID_ARG, "rt_scene_upload: NULL argument");
  *out = nullptr;
  try {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rt_scene* sc = new rt_scene();
    static std::atomic<uint64_t> next_serial{1};
    sc->serial = next_serial.fetch_add(1);
    int rc = scene_upload_impl(ctx, desc, sc);
    if (rc != RT_OK) {
      rt_scene_free(ctx, sc);
      return rc;
    }
    *out = sc;
    return RT_OK;
  } catch (const std::bad_alloc&) {
    return fail(ctx, RT_E_OOM, "rt_scene_upload: out of host memory");
  } catch (...) {
    return fail(ctx, RT_E_INTERNAL, "rt_scene_upload: unexpected exception");
  }
}

// ------------------------------------------------------------------------------------------------ probe
}  // extern "C" (templates need C++ linkage)
// probe_tree 3: the LDS-resident form the render kernels use for scenes that fit (spheres: SlabLdsAcc, general scenes:
// GenericLdsAcc), 1024 threads; probe_tree 4: the four-wide binary16 tree with its top cached in LDS, 256 threads (the
// partner workgroups' form; the HBM-resident render instances walk the same nodes).
template <int KIND, int BLOCK>
static int launch_probe_kind(rt_ctx* ctx, const rt_scene* scene, const rt_ray* d_rays, uint32_t m, rt_hit* d_hits, size_t smem,
                             uint32_t n_cached) {
  auto kern = ctx->probe_count ? k_probe_fused<KIND, BLOCK, true> : k_probe_fused<KIND, BLOCK, false>;
  size_t& opted = ctx->fused_lds_opt_in[reinterpret_cast<const void*>(kern)];
  if (smem > opted) {
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    opted = smem;
  }
  const uint32_t blocks = std::max<uint32_t>(1u, std::min<uint32_t>((m + BLOCK - 1) / BLOCK, (uint32_t)ctx->n_cu));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), smem, ctx->stream, scene->view, scene->features, d_rays, m, d_hits, n_cached,
                     (uint32_t)ctx->fused_kb, (uint32_t)ctx->fused_kn, (uint32_t)(ctx->fused_kr ? ctx->fused_kr : kind_in_lds(KIND) ? 16 : 12), ctx->d_trav);
  HIP_TRY(ctx, hipGetLastError());
  return RT_OK;
}
static int launch_probe_fused(rt_ctx* ctx, const rt_scene* scene, const rt_ray* d_rays, uint32_t m, rt_hit* d_hits) {
  const bool ids16 = scene->view.n_nodes4 < 32768 && scene->view.n_prims < 32000 &&
                     (int64_t)scene->view.n_nodes * (RT_SLAB_NODE_BYTES / 16) < 32768;
  if (!ids16) return fail(ctx, RT_E_UNSUPPORTED, "rt_intersect: probe_tree 3 / 4 need 16-bit node and primitive references");
  if (ctx->probe_tree == 3) {
    const uint32_t sdepth = (uint32_t)std::max(1, scene->bvh_depth + 1);
    const size_t rec = scene->spheres_only ? 64 : 128;
    const size_t smem = RT_SLAB_REGION_BYTES(scene->view.n_nodes) + (size_t)scene->view.n_prims * rec + (size_t)(sdepth + 1) * 1024 * 2;
    if (smem > 160 * 1024) return fail(ctx, RT_E_UNSUPPORTED, "rt_intersect: probe_tree 3: the scene does not fit in LDS");
    if (scene->spheres_only) return launch_probe_kind<KIND_SPHERES_LDS, 1024>(ctx, scene, d_rays, m, d_hits, smem, 0);
    return launch_probe_kind<KIND_GENERIC_LDS, 1024>(ctx, scene, d_rays, m, d_hits, smem, 0);
  }
  const uint32_t sd = (uint32_t)scene->wide_stack + 2u;
  const size_t fixed = (size_t)(sd + 1) * 256 * 2;
  // cache about half of the tree (at least its top node): lanes then fetch from LDS and from HBM/L2 in the same walk
  const uint32_t n_cached = (uint32_t)std::min<size_t>(std::max<size_t>(1, (size_t)scene->view.n_nodes4 / 2), (96 * 1024 - fixed) / 64);
  const size_t smem = fixed + (size_t)n_cached * 64;
  if (scene->spheres_only) return launch_probe_kind<KIND_SPHERES_WIDE_HALF, 256>(ctx, scene, d_rays, m, d_hits, smem, n_cached);
  return launch_probe_kind<KIND_GENERIC_WIDE_HALF, 256>(ctx, scene, d_rays, m, d_hits, smem, n_cached);
}
extern "C" {
int rt_intersect(rt_ctx* ctx, const rt_scene* scene, const rt_ray* rays, uint64_t n, rt_hit* out_hits) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!scene || (n && (!rays || !out_hits))) return fail(ctx, RT_E_INVALID_ARG, "rt_intersect: NULL argument");
  if (n == 0) return RT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t CH = 1u << 20;  // rays per staged chunk (the staging buffers are kept: 56 + 88 B per ray)
  if ((ctx->probe_tree == 1 || ctx->probe_tree == 2) && scene->wide_stack + 2 > RT_STACK_DEPTH)
    return fail(ctx, RT_E_UNSUPPORTED, "rt_intersect: the four-wide tree of this scene needs a deeper stack than the probe has");
  if ((ctx->probe_tree == 2 || ctx->probe_tree == 4) && !scene->half_ok)
    return fail(ctx, RT_E_UNSUPPORTED, "rt_intersect: scene coordinates exceed the binary16 range");
  uint64_t chunk = n < CH ? n : CH;
  int rc = ensure_stage(ctx, 1, chunk * sizeof(rt_ray));
  if (rc == RT_OK) rc = ensure_stage(ctx, 2, chunk * sizeof(rt_hit));
  if (rc != RT_OK) return rc;
  rt_ray* d_rays = static_cast<rt_ray*>(ctx->stage[1]);
  rt_hit* d_hits = static_cast<rt_hit*>(ctx->stage[2]);
  hipError_t e = hipSuccess;
  for (uint64_t off = 0; off < n && rc == RT_OK; off += chunk) {
    uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n - off);
    uint32_t blocks = std::min<uint32_t>((m + RT_BLOCK - 1) / RT_BLOCK, (uint32_t)ctx->n_cu * 8);
    if ((e = hipMemcpyAsync(d_rays, rays + off, (size_t)m * sizeof(rt_ray), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) { rc = RT_E_HIP; break; }
    if (ctx->probe_tree >= 3) {
      rc = launch_probe_fused(ctx, scene, d_rays, m, d_hits);
      if (rc != RT_OK) { e = hipSuccess; break; }
    } else {
      auto kern = ctx->probe_tree == 2 ? k_probe<2> : ctx->probe_tree == 1 ? k_probe<1> : k_probe<0>;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(RT_BLOCK), 0, ctx->stream, scene->view, scene->features, d_rays, m,
                         d_hits, ctx->d_trav);
    }
    ctx->probe_rays += m;
    if ((e = hipGetLastError()) != hipSuccess) { rc = RT_E_HIP; break; }
    if ((e = hipMemcpyAsync(out_hits + off, d_hits, (size_t)m * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) { rc = RT_E_HIP; break; }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { rc = RT_E_HIP; break; }
  }
  if (rc != RT_OK && e != hipSuccess) ctx->err = std::string("rt_intersect: ") + hipGetErrorString(e);
  return rc;
}

// ------------------------------------------------------------------------------------------------ render
static int ensure_work(rt_ctx* ctx, size_t cap, uint32_t depth) {
  if (ctx->wb.cap < cap) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    free_work(ctx);
    // per path: 2 queues x (7 f64 + 3 f32 + 3 u32) + hits (i32 + f64) + L (3 f32)
    const size_t capa = (cap + 63) & ~(size_t)63;
    const size_t per_q = capa * (7 * 8 + 3 * 4 + 3 * 4);
    const size_t total = 2 * per_q + capa * (4 + 8) + capa * 12 + capa * 4;
    HIP_TRY(ctx, rti::dev_malloc(&ctx->wb.slab, total));
    char* p = (char*)ctx->wb.slab;
    auto take = [&](size_t bytes) { char* r = p; p += bytes; return r; };
    ctx->wb.hits.t = (double*)take(capa * 8);
    for (int k = 0; k < 2; k++) {
      Queue& q = ctx->wb.q[k];
      q.ox = (double*)take(capa * 8); q.oy = (double*)take(capa * 8); q.oz = (double*)take(capa * 8);
      q.dx = (double*)take(capa * 8); q.dy = (double*)take(capa * 8); q.dz = (double*)take(capa * 8);
      q.time = (double*)take(capa * 8);
    }
    for (int k = 0; k < 2; k++) {
      Queue& q = ctx->wb.q[k];
      q.tr = (float*)take(capa * 4); q.tg = (float*)take(capa * 4); q.tb = (float*)take(capa * 4);
      q.id = (uint32_t*)take(capa * 4);
      q.r0 = (uint32_t*)take(capa * 4); q.r1 = (uint32_t*)take(capa * 4);
    }
    ctx->wb.hits.prim = (int32_t*)take(capa * 4);
    ctx->wb.Lr = (float*)take(capa * 4); ctx->wb.Lg = (float*)take(capa * 4); ctx->wb.Lb = (float*)take(capa * 4);
    ctx->wb.perm = (uint32_t*)take(capa * 4);
    ctx->wb.cap = capa;
  }
  if (ctx->counts_cap < (size_t)depth + 2) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_counts) (void)rti::dev_free(ctx->d_counts);
    ctx->d_counts = nullptr;
    ctx->counts_cap = 0;
    HIP_TRY(ctx, rti::dev_malloc(&ctx->d_counts, ((size_t)depth + 2) * 4));
    ctx->counts_cap = (size_t)depth + 2;
  }
  return RT_OK;
}

static int to_device_camera(rt_ctx* ctx, const rt_camera* c, Camera& o) {
  if (c->width == 0 || c->height == 0) return fail(ctx, RT_E_INVALID_ARG, "camera: zero image size");
  if ((uint64_t)c->width * c->height > 0x3FFFFFFFull) return fail(ctx, RT_E_UNSUPPORTED, "camera: image too large");
  if (c->sampler_kind > RT_SAMPLER_RANDOM) return fail(ctx, RT_E_INVALID_ARG, "camera: unknown sampler kind");
  if (c->sampler_n == 0) return fail(ctx, RT_E_INVALID_ARG, "camera: sampler_n must be >= 1");
  if (c->sampler_kind == RT_SAMPLER_UNIFORM && c->sampler_n > 65535) return fail(ctx, RT_E_INVALID_ARG, "camera: grid side too large");
  if (c->background_mode > 1) return fail(ctx, RT_E_INVALID_ARG, "camera: unknown background mode");
  for (int a = 0; a < 3; a++) {
    o.center[a] = c->center[a]; o.pixel00[a] = c->pixel00[a]; o.du[a] = c->du[a]; o.dv[a] = c->dv[a];
    o.dfu[a] = c->defocus_u[a]; o.dfv[a] = c->defocus_v[a];
    o.background[a] = (float)c->background[a];
  }
  if (!finite3(c->center) || !finite3(c->pixel00) || !finite3(c->du) || !finite3(c->dv))
    return fail(ctx, RT_E_INVALID_ARG, "camera: non-finite frame vectors");
  o.defocus_angle = c->defocus_angle;
  o.grid_interval = 1.0 / (double)c->sampler_n;
  o.grid_offset = o.grid_interval / 2.0 + 0.5;
  o.width = c->width; o.height = c->height; o.sampler_kind = c->sampler_kind; o.sampler_n = c->sampler_n;
  o.max_depth = c->max_depth; o.background_mode = c->background_mode;
  return RT_OK;
}

struct StageTimer {
  rt_ctx* ctx;
  hipStream_t st;
  explicit StageTimer(rt_ctx* c, hipStream_t s) : ctx(c), st(s) {}
  void begin() { if (ctx->profile) (void)hipEventRecord(ctx->ev_a, st); }
  void end(float& acc) {
    if (!ctx->profile) return;
    (void)hipEventRecord(ctx->ev_b, st);
    (void)hipEventSynchronize(ctx->ev_b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev_a, ctx->ev_b);
    acc += ms;
  }
};

}  // extern "C" (templates below need C++ linkage)

// ---- wavefront mode: one kernel per stage per bounce, queues in HBM
template <uint32_t F, bool SPH>
static void launch_bounce(hipStream_t st, uint32_t blocks, const rt_scene* scene, const Camera& dc, rt_ctx* ctx, Queue& qi,
                          Queue& qo, const PassParams& pp, uint32_t b, uint32_t depth, StageTimer& tm) {
  if (ctx->wf_bin) (void)hipMemsetAsync(ctx->d_bins, 0, 16 * sizeof(uint32_t), st);
  tm.begin();
  hipLaunchKernelGGL((k_intersect<SPH>), dim3(blocks), dim3(RT_BLOCK), 0, st, scene->view, scene->features, qi,
                     ctx->wb.hits, ctx->d_counts + b, ctx->wf_bin ? ctx->d_bins : (uint32_t*)nullptr);
  tm.end(ctx->stats.ms_intersect);
  const uint32_t* perm = nullptr;
  if (ctx->wf_bin) {  // sort the rays by material class (counting sort on wave64 ballots / prefixes)
    tm.begin();
    hipLaunchKernelGGL((k_bin_scatter<SPH>), dim3(blocks), dim3(RT_BLOCK), 0, st, scene->view, ctx->wb.hits, ctx->d_counts + b, ctx->d_bins,
                       ctx->wb.perm);
    tm.end(ctx->ms_bin);  // per-stage timing (option profile): reported as rt_stats.dbg[11]
    perm = ctx->wb.perm;
  }
  tm.begin();
  hipLaunchKernelGGL((k_shade<F, SPH>), dim3(blocks), dim3(RT_BLOCK), 0, st, scene->view, dc, qi, ctx->wb.hits, qo,
                     ctx->d_counts + b, ctx->d_counts + b + 1, ctx->wb.Lr, ctx->wb.Lg, ctx->wb.Lb, pp, b,
                     (uint32_t)(b + 1 == depth), perm);
  tm.end(ctx->stats.ms_shade);
}

static int render_wavefront(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, const Camera& dc, uint64_t seed,
                            uint32_t sample_begin, uint32_t sample_stride, uint32_t K, float* d_out_sum, hipStream_t st) {
  const uint32_t n_pix = cam->width * cam->height;
  // pass plan: all pixels x S samples when that fits max_paths, else pixel chunks x 1 sample
  const uint64_t cap = (uint64_t)ctx->max_paths;
  uint32_t pix_chunk = n_pix, S = K;
  if ((uint64_t)n_pix * K > cap) {
    if (n_pix <= cap) S = (uint32_t)std::max<uint64_t>(1, cap / n_pix);
    else { S = 1; pix_chunk = (uint32_t)cap; }
  }
  int rc = ensure_work(ctx, (size_t)pix_chunk * S, cam->max_depth);
  if (rc) return rc;
  const uint32_t depth = cam->max_depth;
  const uint32_t max_blocks = (uint32_t)ctx->n_cu * 8;
  const bool lean = (scene->features & ~RT_F_TEX_CHECKER) == 0;
  StageTimer tm(ctx, st);
  for (uint32_t pix0 = 0; pix0 < n_pix; pix0 += pix_chunk) {
    const uint32_t npp = std::min(pix_chunk, n_pix - pix0);
    for (uint32_t kk = 0; kk < K; kk += S) {
      const uint32_t Sp = std::min(S, K - kk);
      PassParams pp;
      pp.n_paths = npp * Sp; pp.pix0 = pix0; pp.S = Sp;
      pp.s_first = sample_begin + kk * sample_stride; pp.s_stride = sample_stride;
      pp.k0 = (uint32_t)seed; pp.k1 = (uint32_t)(seed >> 32);
      const uint32_t blocks = std::min<uint32_t>((pp.n_paths + RT_BLOCK - 1) / RT_BLOCK, max_blocks);
      HIP_TRY(ctx, hipMemsetAsync(ctx->d_counts, 0, ((size_t)depth + 2) * 4, st));
      tm.begin();
      hipLaunchKernelGGL(k_generate, dim3(blocks), dim3(RT_BLOCK), 0, st, dc, ctx->wb.q[0], ctx->wb.Lr, ctx->wb.Lg,
                         ctx->wb.Lb, pp, ctx->d_counts);
      tm.end(ctx->stats.ms_generate);
      for (uint32_t b = 0; b < depth; b++) {
        Queue& qi = ctx->wb.q[b & 1];
        Queue& qo = ctx->wb.q[(b + 1) & 1];
        if (scene->spheres_only && lean) launch_bounce<RT_F_TEX_CHECKER, true>(st, blocks, scene, dc, ctx, qi, qo, pp, b, depth, tm);
        else if (scene->spheres_only) launch_bounce<RT_F_ALL, true>(st, blocks, scene, dc, ctx, qi, qo, pp, b, depth, tm);
        else launch_bounce<RT_F_ALL, false>(st, blocks, scene, dc, ctx, qi, qo, pp, b, depth, tm);
        ctx->stats.bounces_run++;
        ctx->stats.intersect_launches++;
      }
      tm.begin();
      const uint32_t ablocks = std::min<uint32_t>((npp + 3) / 4, max_blocks);
      hipLaunchKernelGGL(k_accumulate, dim3(ablocks), dim3(RT_BLOCK), 0, st, ctx->wb.Lr, ctx->wb.Lg, ctx->wb.Lb,
                         d_out_sum, pp, npp, ctx->d_counts, depth, ctx->fs->ray_total);
      tm.end(ctx->stats.ms_accumulate);
      ctx->stats.passes++;
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  return RT_OK;
}

// ---- fused mode: one persistent kernel, no path state in HBM
// Partner of the LDS-resident spheres instance.  That kernel is one 1024-thread workgroup per CU: four waves per SIMD
// at 96 VGPRs, 134 KB of LDS at C2 — a fifth wave per SIMD has registers (512 - 4 x 96 = 128) but no room for a
// second copy of the scene.  The partner is a 256-thread workgroup per CU of the instance that needs no copy: the
// four-wide binary16 tree in whatever LDS is left (all or most of its 15 KB at C2), sphere records through L1/L2.
// It runs on a second stream and takes its tiles from the same counter.
// LDS is allocated in granules of 1280 bytes on gfx950 (160 KiB / 128): two workgroups share a CU only if their ROUNDED
// sizes fit together (measured: 134 432 + 28 368 B = 162 800 B did not co-reside, 134 224 + 28 592 B did).
static size_t lds_round(size_t bytes) { return (bytes + 1279) / 1280 * 1280; }
// Tile slots per workgroup = tiles it can have in flight (1.5 KB of LDS each).  A slot is held until its tile's LONGEST
// path ends, long after the tile's samples are handed out, so a workgroup needs several times the slots its lanes can
// keep busy — more when a tile holds few paths (small K) or paths run long (C5 at 16 spp, 12 waves: 12 slots 69.4 ms,
// 16: 64.2, 24: 60.5, 32: 59.4, 48: 60.6 — each slot costs the LDS node cache 24 nodes).  `want` is what the caller
// would like, `room` the LDS bytes it can spend; option fused_slots overrides (0 = auto).
static uint32_t fused_slots(const rt_ctx* ctx, uint32_t want, size_t room = (size_t)1 << 30) {
  if (ctx->fused_slots > 0) want = (uint32_t)ctx->fused_slots;
  const uint32_t fit = (uint32_t)std::min<size_t>(room / (192 * 8), RT_MAX_SLOTS);
  return std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(want, RT_MAX_SLOTS), fit));
}
// Samples per chunk (what a wave claims from its workgroup's open tile at a time).  With slots to spare (at least two
// per wave) a wave takes a whole tile, up to 16 sample indices (1024 paths) at a time: claims are rare and the waves
// do not wait for each other's tiles (C5, 32 slots: 1 -> 59.4 ms, 4 -> 56.9, 8 -> 55.9, 16 -> 55.7).  With one slot
// per wave (the lean LDS-resident instance beside its partner) the waves share tiles: about one chunk per wave and
// tile, at most 8 sample indices (C2, 100 spp: 2 -> 9.67 ms, 3 -> 9.75, 6 -> 9.53, 8 -> 9.62, 12 -> 9.63, 16 -> 9.84,
// 25 -> 10.97; 13 spp: 1-2 best).  Option fused_ku overrides.  Never changes the frame.
static uint32_t fused_chunk(const rt_ctx* ctx, uint32_t K, uint32_t waves, uint32_t slots, uint64_t tiles_per_wave) {
  if (ctx->fused_ku > 0) return (uint32_t)std::min<int64_t>(ctx->fused_ku, std::max<uint32_t>(K, 1u));
  // (whole tiles only where every wave gets dozens of them: with a handful per wave the frame ends unbalanced — C2 at
  // 13 spp, 3.5 tiles per wave: 2.79 ms against 1.64)
  if (slots >= 2 * waves && tiles_per_wave >= 32) return std::max<uint32_t>(1u, std::min<uint32_t>(16u, K));
  return std::max<uint32_t>(1u, std::min<uint32_t>(8u, K / waves));
}
static bool partner_possible(const rt_ctx* ctx, const rt_scene* scene, size_t lds_main) {
  if (!ctx->fused_partner || !scene->half_ok || scene->view.n_nodes4 >= 32768 || scene->view.n_prims >= 32768) return false;
  const size_t fixed = RT_SCHED_BYTES + (size_t)fused_slots(ctx, 4) * 192 * 8 + (size_t)(scene->wide_stack + 2) * 256 * 2;
  return lds_round(lds_main) + lds_round(fixed + 16 * 64) <= 160 * 1024;
}
static int launch_partner(rt_ctx* ctx, const rt_scene* scene, const FusedParams& fp_main, size_t lds_main, hipStream_t st) {
  const FusedKernel kern = ctx->cur_frames ? k_fused<RT_F_TEX_CHECKER, KIND_SPHERES_WIDE_HALF, 256, int16_t, false, true, true>
                                           : k_fused<RT_F_TEX_CHECKER, KIND_SPHERES_WIDE_HALF, 256, int16_t, false, true, false>;
  FusedParams fp = fp_main;  // same frame, same tile counter — taken from the other end
  fp.from_back = 1;
  fp.stack_depth = (uint32_t)scene->wide_stack + 2u;
  fp.n_slots = fused_slots(ctx, 4);
  fp.Ku = fused_chunk(ctx, fp.K, 4, fp.n_slots, 0);
  const size_t fixed = RT_SCHED_BYTES + (size_t)fp.n_slots * 192 * 8 + (size_t)fp.stack_depth * 256 * 2;
  const size_t room = 160 * 1024 - lds_round(lds_main);  // what the main workgroup leaves on its CU (a multiple of the granule)
  fp.n_cached = (uint32_t)std::min<size_t>((size_t)scene->view.n_nodes4, (room - fixed) / 64);
  const size_t smem = fixed + (size_t)fp.n_cached * 64;
  size_t& opted = ctx->fused_lds_opt_in[reinterpret_cast<const void*>(kern)];
  if (smem > opted) {
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    opted = smem;
  }
  if (getenv("RT_VERBOSE")) fprintf(stderr, "[rt] partner: LDS %zu B (main %zu B), %u cached nodes of %d, stack %u\n", smem, lds_main, fp.n_cached, scene->view.n_nodes4, fp.stack_depth);
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->partner_stream, ctx->ev_fork, 0));
  hipLaunchKernelGGL(kern, dim3((uint32_t)ctx->n_cu), dim3(256), smem, ctx->partner_stream, scene->view, ctx->fs->cam, fp,
                     ctx->fs->unit_counter, ctx->cur_out, ctx->fs->ray_total, (float*)nullptr);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->partner_stream));
  HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
  return RT_OK;
}
template <uint32_t F, int KIND, int BLOCK, class StackT, bool STATS = false, bool RESUME = false>
static int launch_fused(rt_ctx* ctx, const rt_scene* scene, const Camera& dc, const FusedParams& fp0, size_t smem,
                        hipStream_t st) {
  const FusedKernel kern = fused_kernel<F, KIND, BLOCK, StackT, STATS, RESUME>(ctx->cur_frames != nullptr);
  // the LDS opt-in and the occupancy query cost tens of microseconds of host time: once per (instance, LDS size)
  const std::pair<const void*, size_t> key(reinterpret_cast<const void*>(kern), smem);
  auto cached = ctx->fused_occupancy.find(key);
  if (cached == ctx->fused_occupancy.end()) {
    size_t& opted = ctx->fused_lds_opt_in[key.first];  // the attribute is a ceiling: only ever raise it
    if (smem > opted) {
      HIP_TRY(ctx, hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      opted = smem;
    }
    int q = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, BLOCK, smem));
    cached = ctx->fused_occupancy.emplace(key, q).first;
  }
  const int per_cu = cached->second;
  if (per_cu < 1) return fail(ctx, RT_E_INTERNAL, "fused kernel does not fit on a CU");
  FusedParams fp = fp0;
  fp.Ku = fused_chunk(ctx, fp.K, BLOCK / 64, fp.n_slots,  // (n_slots: set by render_fused, which sizes the LDS)
                      fp.n_tiles / ((uint64_t)ctx->n_cu * (uint64_t)per_cu * (BLOCK / 64)));
  uint32_t blocks = std::min<uint32_t>((uint32_t)ctx->n_cu * (uint32_t)per_cu, fp.n_tiles);
  if (ctx->dbg_main_cap > 0) blocks = std::min<uint32_t>(blocks, (uint32_t)ctx->dbg_main_cap);  // diagnostic
  // the camera travels as a kernel argument (captured at launch) into its device slot: stream-ordered, and no host
  // buffer has to outlive an asynchronous copy
  {
    BatchFrame one;
    one.k0 = fp.k0; one.k1 = fp.k1; one.s_begin = fp.s_begin; one.pad = 0; one.out = ctx->cur_out;
    hipLaunchKernelGGL(k_store_camera, dim3(1), dim3(64), 0, st, dc, ctx->fs->cam, one, ctx->cur_frames ? (BatchFrame*)nullptr : ctx->fs->frame1);
  }
  constexpr bool kHasPartner = F == RT_F_TEX_CHECKER && KIND == KIND_SPHERES_LDS && BLOCK == 1024 && RESUME && !STATS;
  const bool partner = kHasPartner && per_cu == 1 && ctx->cur_partner;  // decided by render_fused (it shapes the tile list)
  if (partner) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));  // counter reset, camera stored
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), smem, st, scene->view, ctx->fs->cam, fp, ctx->fs->unit_counter, ctx->cur_out,
                     ctx->fs->ray_total, ctx->d_dbg);
  HIP_TRY(ctx, hipGetLastError());
  if (partner) return launch_partner(ctx, scene, fp, smem, st);
  return RT_OK;
}

// Resumable traversals pay when rays of one wave need very different numbers of BVH steps — i.e. when there is a
// tree to speak of.  On one- or two-object scenes (C3) every traversal is a couple of steps and the plain
// bounce-synchronous loop is 15-20 % faster; from a few dozen primitives on (C1/C2, C5) the resumable form wins.
static bool use_resume(const rt_ctx* ctx, const rt_scene* scene) {
  return ctx->fused_resume == 1 || (ctx->fused_resume == 2 && scene->view.n_prims >= 32);
}
// production instances: the traversal schedule (RESUME or bounce-synchronous) is a run-time choice
template <uint32_t F, int KIND, int BLOCK, class StackT>
static int launch_variant(rt_ctx* ctx, const rt_scene* scene, const Camera& dc, const FusedParams& fp, size_t smem,
                          hipStream_t st) {
  if (use_resume(ctx, scene)) return launch_fused<F, KIND, BLOCK, StackT, false, true>(ctx, scene, dc, fp, smem, st);
  return launch_fused<F, KIND, BLOCK, StackT, false, false>(ctx, scene, dc, fp, smem, st);
}

// ... or the instrumented build of the same instance (option fused_stats: bench.py's own-tree operation counts)
template <uint32_t F, int KIND, int BLOCK, class StackT>
static int launch_variant_or_stats(rt_ctx* ctx, const rt_scene* scene, const Camera& dc, const FusedParams& fp, size_t smem,
                                   hipStream_t st) {
  if (!ctx->fused_stats) return launch_variant<F, KIND, BLOCK, StackT>(ctx, scene, dc, fp, smem, st);
  if (use_resume(ctx, scene)) return launch_fused<F, KIND, BLOCK, StackT, true, true>(ctx, scene, dc, fp, smem, st);
  return launch_fused<F, KIND, BLOCK, StackT, true, false>(ctx, scene, dc, fp, smem, st);
}

static int render_fused(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, const Camera& dc, uint64_t seed,
                        uint32_t sample_begin, uint32_t sample_stride, uint32_t K, float* d_out_sum, hipStream_t st) {
  HIP_TRY(ctx, hipMemsetAsync(ctx->fs->unit_counter, 0, 2 * sizeof(unsigned long long), st));  // [0] tile counter, [1] watchdog flag
  ctx->cur_out = d_out_sum;
  ctx->last_watchdog = reinterpret_cast<const uint32_t*>(ctx->fs->unit_counter + 1);
  FusedParams fp;
  memset(&fp, 0, sizeof(fp));
  fp.K = K; fp.s_begin = sample_begin; fp.s_stride = sample_stride;
  fp.k0 = (uint32_t)seed; fp.k1 = (uint32_t)(seed >> 32);
  const uint32_t n_frames = ctx->cur_frames ? ctx->cur_n_frames : 1u;  // a batch: the frame table replaces seed / sample_begin / output
  fp.frames = ctx->cur_frames ? ctx->cur_frames : ctx->fs->frame1;
  fp.n_frames = n_frames;
  fp.poll_limit = ctx->dbg_watchdog_polls > 0 ? (uint32_t)ctx->dbg_watchdog_polls : (1u << 24);
  {
    int lg = 0;  // ceil(log2 K)
    while ((1ull << lg) < (unsigned long long)K) lg++;
    const int bits = std::min(40, 43 - lg);  // K * 2^20 * 2^bits <= 2^63
    fp.fix_scale = std::ldexp(1.0f, bits);
    fp.fix_inv = std::ldexp(1.0, -bits);
  }
  const uint32_t sdepth = (uint32_t)std::max(1, scene->bvh_depth + 1);
  fp.stack_depth = sdepth;
  fp.kb = (uint32_t)ctx->fused_kb; fp.kn = (uint32_t)ctx->fused_kn; fp.kr = use_resume(ctx, scene) ? (uint32_t)ctx->fused_kr : (uint32_t)ctx->fused_kr_sync;
  const bool lean = (scene->features & ~RT_F_TEX_CHECKER) == 0;
  // texture kinds present (kernels are specialised on them); merged boxes count as "more than the Cornell instance has"
  const uint32_t tex_only = scene->features & (RT_F_TEX_ANY | RT_F_BOX);
  const bool small_ids = (int64_t)scene->view.n_nodes * (RT_SLAB_NODE_BYTES / 16) < 32768 && scene->view.n_prims < 32000;  // int16 stack entries
  int rc;
  // LDS budget of the 1024-thread spheres-in-LDS variant: scene + accumulators + u16 stack
  const size_t lds_noise = (scene->features & RT_F_TEX_NOISE) ? (size_t)16 * 256 * 16 : 0;
  fp.noise_lds = lds_noise ? 1u : 0u;
  const size_t lds_tables = lds_noise;  // material / texture records stay in HBM/L2: LDS copies bought nothing measurable
  const size_t lds_scene = RT_SLAB_REGION_BYTES(scene->view.n_nodes) + (size_t)scene->view.n_prims * 64 + lds_tables;
  // LDS of the 1024-thread LDS-resident instances without their tile slots, and with the 16 (one per wave) they need
  const size_t lds_base_1024 = lds_scene + RT_SCHED_BYTES + (size_t)sdepth * 1024 * 2;
  const size_t lds_base_generic_1024 = RT_SLAB_REGION_BYTES(scene->view.n_nodes) + (size_t)scene->view.n_prims * 128 + lds_tables +
                                       RT_SCHED_BYTES + (size_t)sdepth * 1024 * 2;
  size_t lds_1024 = lds_base_1024 + 16 * 192 * 8;
  size_t lds_generic_1024 = lds_base_generic_1024 + 16 * 192 * 8;
  // partner workgroups (launch_partner) join the lean spheres-in-LDS instance on frames that last long enough for a
  // second launch and its join to pay (option fused_partner_min, Mi samples)
  ctx->cur_partner = ctx->fused_partner && lean && !ctx->fused_stats && scene->spheres_only && ctx->fused_lds && small_ids &&
                     lds_1024 <= 160 * 1024 && use_resume(ctx, scene) &&
                     (uint64_t)cam->width * cam->height * K * n_frames >= ((uint64_t)ctx->fused_partner_min << 20) &&
                     (uint64_t)cam->width * cam->height >= 64ull * 4 * (uint64_t)ctx->n_cu && partner_possible(ctx, scene, lds_1024);
  // tile slots: one per wave beside partner workgroups (they get the rest of the CU's LDS), two per wave otherwise
  fp.n_slots = 16;
  if (!ctx->cur_partner) {
    if (scene->spheres_only && lds_1024 <= 160 * 1024) {
      fp.n_slots = std::max<uint32_t>(fused_slots(ctx, 32, 160 * 1024 - lds_base_1024), ctx->fused_slots > 0 ? 1u : 16u);
      lds_1024 = lds_base_1024 + (size_t)fp.n_slots * 192 * 8;
    } else if (!scene->spheres_only && lds_generic_1024 <= 160 * 1024) {
      fp.n_slots = std::max<uint32_t>(fused_slots(ctx, 32, 160 * 1024 - lds_base_generic_1024), ctx->fused_slots > 0 ? 1u : 16u);
      lds_generic_1024 = lds_base_generic_1024 + (size_t)fp.n_slots * 192 * 8;
    }
  }
  {
    // Tile list (FusedParams): 8x8 tiles for most of the frame, then two short bands of small tiles.  A small tile must
    // still be worth opening (>= ~200 paths): 4x4 from 12 samples per pixel on, 2x2 from 48.  Band heights in tile rows
    // of 8 pixels: options fused_tail4 / fused_tail2 (per mille of the frame's rows; defaults 50 / 30).
    const uint32_t rows8 = (cam->height + 7) / 8;
    // (with partner workgroups the 4x4 band is what they render — about a tenth of the frame at C2 — hence 3x as tall)
    const uint32_t pm4 = (uint32_t)ctx->fused_tail4 * (ctx->cur_partner ? 3u : 1u);
    uint32_t r4 = K >= 12 ? (rows8 * pm4 + 500) / 1000 : 0;
    uint32_t r2 = K >= 48 ? (rows8 * (uint32_t)ctx->fused_tail2 + 500) / 1000 : 0;
    if (r4 + r2 >= rows8) { r4 = 0; r2 = 0; }
    // list order: with partner workgroups (they take tiles from the END of the list, four waves to a tile) the 4x4 band
    // is last and the 2x2 band in the middle, where the two kinds of workgroup meet; otherwise sizes simply decrease
    const bool back_is_4 = ctx->cur_partner;
    const uint32_t shifts[3] = {3u, back_is_4 ? 1u : 2u, back_is_4 ? 2u : 1u};
    const uint32_t band_rows8[3] = {rows8 - r4 - r2, back_is_4 ? r2 : r4, back_is_4 ? r4 : r2};
    uint32_t y = 0, t = 0;
    for (int i = 0; i < RT_TILE_LEVELS; i++) {
      const uint32_t y_end = std::min<uint32_t>(cam->height, y + band_rows8[i] * 8u);
      const uint32_t side = 1u << shifts[i];
      fp.lvl_begin[i] = t; fp.lvl_y0[i] = y; fp.lvl_shift[i] = shifts[i];
      fp.lvl_tiles_x[i] = (cam->width + side - 1) / side;
      t += fp.lvl_tiles_x[i] * ((y_end - y + side - 1) / side);
      y = y_end;
    }
    fp.lvl_begin[RT_TILE_LEVELS] = t;
    fp.tiles_per_frame = t;
    // (a workgroup's LDS cursor carries list position << 6 | slot in 32 bits)
    if ((uint64_t)t * n_frames >= (1ull << 26)) return fail(ctx, RT_E_UNSUPPORTED, "rt_render: more than 2^26 tiles (frame too large, or too many frames in the batch)");
    fp.n_tiles = t * n_frames;
    fp.from_back = 0;
  }
  // Tile-order feedback: when the last fused frame of this context showed the same scene through the same camera with
  // the same tile list, its tiles with deep paths are handed out first (and the partner workgroups, which take from
  // the end of the list, get the shallow ones).  Frames are unchanged: a tile's sums do not depend on when it is rendered.
  // Not under rt_render_device_noalloc (the lists are allocated memory).
  fp.order = nullptr;
  fp.depth_out = nullptr;
  // (LDS-resident instances only: where the scene is read through the caches — C5 — handing tiles out in image order
  // keeps neighbouring rays on neighbouring nodes and texels, and the reordered frame measured 1 % slower)
  const bool in_lds = ctx->fused_lds && small_ids &&
                      (scene->spheres_only ? lds_1024 <= 160 * 1024 : lds_generic_1024 <= 160 * 1024);
  // (final scene, kr x kb swept after the traversal went into asm: 12 / 16 -> 38.1 ms, 16 / 16 -> 38.4, 8 / 16 -> 38.2;
  // LDS-resident C2 is flat from 16 to 24 and loses below)
  if (use_resume(ctx, scene) && ctx->fused_kr == 0) fp.kr = in_lds ? 16u : 12u;
  if (ctx->fused_feedback && in_lds && ctx->fs == &ctx->own) {
    if (fp.tiles_per_frame > ctx->tile_cap) {
      HIP_TRY(ctx, hipStreamSynchronize(st));
      if (ctx->d_tile_class) (void)rti::dev_free(ctx->d_tile_class);
      if (ctx->d_tile_order) (void)rti::dev_free(ctx->d_tile_order);
      ctx->d_tile_class = nullptr; ctx->d_tile_order = nullptr; ctx->tile_cap = 0; ctx->fb_valid = false;
      HIP_TRY(ctx, rti::dev_malloc(&ctx->d_tile_class, fp.tiles_per_frame));
      HIP_TRY(ctx, rti::dev_malloc(&ctx->d_tile_order, (size_t)fp.tiles_per_frame * sizeof(uint32_t)));
      ctx->tile_cap = fp.tiles_per_frame;
    }
    uint64_t key = 0xCBF29CE484222325ull;
    auto mix = [&key](const void* p, size_t n) {
      const unsigned char* b = static_cast<const unsigned char*>(p);
      for (size_t i = 0; i < n; i++) key = (key ^ b[i]) * 0x100000001B3ull;
    };
    mix(&scene->serial, sizeof(scene->serial));
    mix(cam, sizeof(*cam));
    mix(&fp.tiles_per_frame, sizeof(fp.tiles_per_frame));
    mix(fp.lvl_begin, sizeof(fp.lvl_begin)); mix(fp.lvl_y0, sizeof(fp.lvl_y0)); mix(fp.lvl_shift, sizeof(fp.lvl_shift));
    // The classes are written by the previous frame's kernels and read twice by k_order_tiles: that is ordered only
    // through the stream, so a frame issued on another stream than its predecessor starts from image order.
    if (ctx->fb_stream != st) ctx->fb_valid = false;
    ctx->fb_stream = st;
    if (ctx->fb_valid && key == ctx->fb_key) {
      hipLaunchKernelGGL(k_order_tiles, dim3(1), dim3(1024), 0, st, ctx->d_tile_class, ctx->d_tile_order, fp.tiles_per_frame);
      fp.order = ctx->d_tile_order;
    }
    if (fp.order && getenv("RT_VERBOSE")) {  // diagnostic: what the previous frame recorded
      std::vector<unsigned char> h(fp.tiles_per_frame);
      HIP_TRY(ctx, hipStreamSynchronize(st));
      HIP_TRY(ctx, hipMemcpy(h.data(), ctx->d_tile_class, fp.tiles_per_frame, hipMemcpyDeviceToHost));
      size_t c[4] = {0, 0, 0, 0};
      for (unsigned char v : h) c[std::min<int>(3, v)]++;
      fprintf(stderr, "[rt] tile order from depth classes: %zu shallow, %zu (>=6), %zu (>=12), %zu (>=24) of %u tiles\n", c[0], c[1], c[2], c[3], fp.tiles_per_frame);
    }
    fp.depth_out = ctx->d_tile_class;
    ctx->fb_key = key;
    ctx->fb_valid = true;
  }
  StageTimer tm(ctx, st);
  tm.begin();
  fp.timeline = nullptr;
  if (ctx->fused_stats) {
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_dbg, 0, 16 * sizeof(float), st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_timeline, 0xFF, sizeof(unsigned long long), st));     // [0]: minimum
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_timeline + 1, 0, 15 * sizeof(unsigned long long), st));
    for (int k = 0; k < 5; k++) ctx->last_global_types[k] = scene->global_types[k];
    fp.timeline = ctx->d_timeline;
  }
  if (scene->spheres_only && ctx->fused_lds && small_ids && lds_1024 <= 160 * 1024) {
    if (lean && ctx->fused_stats) {
      if (use_resume(ctx, scene)) rc = launch_fused<RT_F_TEX_CHECKER, KIND_SPHERES_LDS, 1024, int16_t, true, true>(ctx, scene, dc, fp, lds_1024, st);
      else rc = launch_fused<RT_F_TEX_CHECKER, KIND_SPHERES_LDS, 1024, int16_t, true, false>(ctx, scene, dc, fp, lds_1024, st);
    } else if (lean) rc = launch_variant<RT_F_TEX_CHECKER, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
    else if (tex_only == RT_F_TEX_IMAGE) rc = launch_variant_or_stats<RT_F_TEX_IMAGE, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
    else if (!(tex_only & RT_F_TEX_IMAGE)) rc = launch_variant_or_stats<RT_F_TEX_CHECKER | RT_F_TEX_NOISE, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
    else rc = launch_variant<RT_F_TEX_ANY, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
  } else if (!scene->spheres_only && ctx->fused_lds && small_ids && lds_generic_1024 <= 160 * 1024) {
    if (tex_only == 0 && ctx->fused_stats) {  // instrumented build of the Cornell instance (both schedules)
      if (use_resume(ctx, scene)) rc = launch_fused<RT_F_QUAD | RT_F_MEDIUM, KIND_GENERIC_LDS, 1024, int16_t, true, true>(ctx, scene, dc, fp, lds_generic_1024, st);
      else rc = launch_fused<RT_F_QUAD | RT_F_MEDIUM, KIND_GENERIC_LDS, 1024, int16_t, true, false>(ctx, scene, dc, fp, lds_generic_1024, st);
    } else if (tex_only == 0) rc = launch_variant<RT_F_QUAD | RT_F_MEDIUM, KIND_GENERIC_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_generic_1024, st);
    else rc = launch_variant<RT_F_ALL, KIND_GENERIC_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_generic_1024, st);
  } else {
    // HBM-resident scene: four-wide tree by default (fused_wide = 0: the binary tree, kept as a cross-check)
    const bool wide = ctx->fused_wide != 0;
    const uint32_t sd = wide ? (uint32_t)scene->wide_stack + 2u : sdepth;
    fp.stack_depth = sd;
    fp.n_slots = fused_slots(ctx, 8);
    const size_t smem = RT_SCHED_BYTES + (size_t)fp.n_slots * 192 * 8 + (size_t)sd * 256 * 4;
    if (smem > 160 * 1024) return fail(ctx, RT_E_INTERNAL, "traversal stack does not fit in LDS");
#define RT_GLOBAL_DISPATCH(KS, KG)                                                                                              \
    if (scene->spheres_only && lean) rc = launch_variant<RT_F_TEX_CHECKER, KS, 256, int32_t>(ctx, scene, dc, fp, smem, st);       \
    else if (scene->spheres_only) rc = launch_variant<RT_F_TEX_ANY, KS, 256, int32_t>(ctx, scene, dc, fp, smem, st);              \
    else if (tex_only == 0) rc = launch_variant<RT_F_QUAD | RT_F_MEDIUM, KG, 256, int32_t>(ctx, scene, dc, fp, smem, st);        \
    else if (ctx->fused_stats) {                                                                                                \
      if (use_resume(ctx, scene)) rc = launch_fused<RT_F_ALL, KG, 256, int32_t, true, true>(ctx, scene, dc, fp, smem, st);      \
      else rc = launch_fused<RT_F_ALL, KG, 256, int32_t, true, false>(ctx, scene, dc, fp, smem, st);                            \
    } else rc = launch_variant<RT_F_ALL, KG, 256, int32_t>(ctx, scene, dc, fp, smem, st);
    // Preferred form: binary16 planes, one workgroup per CU with the top of the tree (or all of it) in LDS — when the
    // scene's coordinates fit binary16 and its references the 16-bit stack.  768 threads (3 waves/SIMD, 168 VGPRs)
    // for the instances with the noise code, 1024 otherwise.
    const bool heavy = !(scene->spheres_only && lean);  // instances compiled with RT_F_TEX_NOISE
    // (the INSTRUMENTED general instance keeps 768 threads / 168 VGPRs: with its counters on top, the 128-VGPR form spills
    // 165 VGPRs and 98 SGPRs, and its counts — never the frame — came back different from run to run)
    const bool stats_general = ctx->fused_stats && !scene->spheres_only;
    const uint32_t cblock = heavy ? (stats_general ? (uint32_t)RT_STATS_HEAVY_BLOCK : (uint32_t)RT_HEAVY_BLOCK) : 1024u;
    // (slots before cache: 2 per wave + 8, if the stacks leave that much)
    const size_t cbase = RT_SCHED_BYTES + (size_t)sd * cblock * 2;
    const uint32_t cslots = cbase + 16 * 64 < 160 * 1024 ? fused_slots(ctx, 2 * (cblock / 64) + 8, 160 * 1024 - cbase - 16 * 64) : cblock / 64;
    const size_t cfixed = cbase + (size_t)cslots * 192 * 8;
    const bool ids16 = scene->view.n_nodes4 < 32768 && scene->view.n_prims < 32768;
    if (wide && ctx->fused_half && scene->half_ok && ids16 && cfixed + 16 * 64 <= 160 * 1024) {
      fp.n_slots = cslots;
      fp.n_cached = (uint32_t)std::min<size_t>((size_t)scene->view.n_nodes4, (160 * 1024 - cfixed) / 64);
      if (ctx->dbg_max_cached > 0) fp.n_cached = std::min<uint32_t>(fp.n_cached, (uint32_t)ctx->dbg_max_cached);  // test hook
      const size_t csmem = cfixed + (size_t)fp.n_cached * 64;
      if (scene->spheres_only && lean) rc = launch_variant<RT_F_TEX_CHECKER, KIND_SPHERES_WIDE_HALF, 1024, int16_t>(ctx, scene, dc, fp, csmem, st);
      else if (scene->spheres_only) rc = launch_variant<RT_F_TEX_ANY, KIND_SPHERES_WIDE_HALF, RT_HEAVY_BLOCK, int16_t>(ctx, scene, dc, fp, csmem, st);
      else if (ctx->fused_stats) {
        if (use_resume(ctx, scene)) rc = launch_fused<RT_F_ALL, KIND_GENERIC_WIDE_HALF, RT_STATS_HEAVY_BLOCK, int16_t, true, true>(ctx, scene, dc, fp, csmem, st);
        else rc = launch_fused<RT_F_ALL, KIND_GENERIC_WIDE_HALF, RT_STATS_HEAVY_BLOCK, int16_t, true, false>(ctx, scene, dc, fp, csmem, st);
      } else if (scene->features & RT_F_BOX) rc = launch_variant<RT_F_ALL, KIND_GENERIC_WIDE_HALF, RT_HEAVY_BLOCK, int16_t>(ctx, scene, dc, fp, csmem, st);
      else rc = launch_variant<RT_F_ALL_NOBOX, KIND_GENERIC_WIDE_HALF, RT_HEAVY_BLOCK, int16_t>(ctx, scene, dc, fp, csmem, st);
    } else if (wide) { RT_GLOBAL_DISPATCH(KIND_SPHERES_WIDE, KIND_GENERIC_WIDE) }
    else { RT_GLOBAL_DISPATCH(KIND_SPHERES_GLOBAL, KIND_GENERIC) }
#undef RT_GLOBAL_DISPATCH
  }
  if (rc) return rc;
  tm.end(ctx->stats.ms_intersect);
  ctx->stats.passes = 1;
  return RT_OK;
}

extern "C" {

static int render_device_impl(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
                              uint32_t sample_end, uint32_t sample_stride, float* d_out_sum, void* hip_stream, bool no_alloc) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!scene || !cam || !d_out_sum) return fail(ctx, RT_E_INVALID_ARG, "rt_render: NULL argument");
  if (sample_stride == 0) return fail(ctx, RT_E_INVALID_ARG, "rt_render: sample_stride must be >= 1");
  Camera dc;
  int rc = to_device_camera(ctx, cam, dc);
  if (rc) return rc;
  const uint64_t spp_total = cam->sampler_kind == RT_SAMPLER_UNIFORM ? (uint64_t)cam->sampler_n * cam->sampler_n : cam->sampler_n;
  if (sample_end > spp_total) return fail(ctx, RT_E_INVALID_ARG, "rt_render: sample_end exceeds the sampler's sample count");
  const uint32_t n_pix = cam->width * cam->height;
  if (no_alloc) {
    if (ctx->mode != 1)
      return fail(ctx, RT_E_UNSUPPORTED, "rt_render_device_noalloc: only the fused mode (option mode = 1) runs without allocating");
    if (n_pix > RT_NOALLOC_MAX_PIXELS)
      return fail(ctx, RT_E_UNSUPPORTED, "rt_render_device_noalloc: frame of " + std::to_string(n_pix) + " pixels exceeds the stated maximum (" +
                                             std::to_string(RT_NOALLOC_MAX_PIXELS) + " pixels); use rt_render_device");
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  // (64-bit: a stride close to 2^32 must not wrap the rounding-up)
  const uint64_t K64 = sample_begin < sample_end ? ((uint64_t)(sample_end - sample_begin) + sample_stride - 1) / sample_stride : 0;
  if (K64 > 65535ull * 64ull) return fail(ctx, RT_E_UNSUPPORTED, "rt_render: too many samples per pixel in one call");
  const uint32_t K = (uint32_t)K64;
  ctx->fs = no_alloc ? &ctx->fixed : &ctx->own;
  ctx->stats = rt_stats();
  ctx->ms_bin = 0.f;
  HIP_TRY(ctx, hipEventRecord(ctx->ev_start, st));
  HIP_TRY(ctx, hipMemsetAsync(ctx->fs->ray_total, 0, 4 * sizeof(unsigned long long), st));
  ctx->last_ray_total = ctx->fs->ray_total;
  ctx->last_watchdog = nullptr;
  if (K > 0 && cam->max_depth > 0) {
    ctx->stats.paths = (uint64_t)n_pix * K;
    if (ctx->mode == 0) {
      HIP_TRY(ctx, hipMemsetAsync(d_out_sum, 0, (size_t)n_pix * 3 * sizeof(float), st));
      rc = render_wavefront(ctx, scene, cam, dc, seed, sample_begin, sample_stride, K, d_out_sum, st);
    } else {
      rc = render_fused(ctx, scene, cam, dc, seed, sample_begin, sample_stride, K, d_out_sum, st);
    }
    if (rc) return rc;
  } else {
    HIP_TRY(ctx, hipMemsetAsync(d_out_sum, 0, (size_t)n_pix * 3 * sizeof(float), st));
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, st));
  ctx->timing_pending = true;
  return RT_OK;
}

int rt_render_device(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
                     uint32_t sample_end, uint32_t sample_stride, float* d_out_sum, void* hip_stream) {
  return render_device_impl(ctx, scene, cam, seed, sample_begin, sample_end, sample_stride, d_out_sum, hip_stream, false);
}

// Same render with every scratch buffer (the camera block of the frame in flight, the two-ended tile counter, the
// watchdog flag, the ray / anomaly counters: 1 KiB of control words) in the file-scope device array `scratch`: no
// hipMalloc / hipFree / stream or event creation on this path (rt_stats.lib_alloc_calls does not move across it).
// RT_NOALLOC_MAX_PIXELS is a stated limit kept for interface stability; no buffer imposes it.
int rt_render_device_noalloc(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
                             uint32_t sample_end, uint32_t sample_stride, float* d_out_sum, void* hip_stream) {
  return render_device_impl(ctx, scene, cam, seed, sample_begin, sample_end, sample_stride, d_out_sum, hip_stream, true);
}

// Several frames of one scene and camera in ONE persistent launch: the tile list is the frames' lists back to back, so
// the workgroups hand out frame k + 1 while the deepest paths of frame k still bounce — the ~0.4 ms drain that every
// separately launched frame pays is paid once per batch.  Every frame is bit-identical to what
// rt_render_device gives for the same (seed, sample_begin, sample_end, sample_stride).
int rt_render_batch_device(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint32_t n_frames, const uint64_t* seeds,
                           const uint32_t* sample_begins, uint32_t sample_end, uint32_t sample_stride, float* const* d_out_sums,
                           void* hip_stream) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!scene || !cam || !seeds || !sample_begins || !d_out_sums || n_frames == 0)
    return fail(ctx, RT_E_INVALID_ARG, "rt_render_batch_device: NULL argument or empty batch");
  if (n_frames > 65536) return fail(ctx, RT_E_UNSUPPORTED, "rt_render_batch_device: more than 65536 frames in one batch");
  if (sample_stride == 0) return fail(ctx, RT_E_INVALID_ARG, "rt_render: sample_stride must be >= 1");
  if (ctx->mode != 1) return fail(ctx, RT_E_UNSUPPORTED, "rt_render_batch_device: only the fused mode (option mode = 1) renders batches");
  Camera dc;
  int rc = to_device_camera(ctx, cam, dc);
  if (rc) return rc;
  const uint64_t spp_total = cam->sampler_kind == RT_SAMPLER_UNIFORM ? (uint64_t)cam->sampler_n * cam->sampler_n : cam->sampler_n;
  if (sample_end > spp_total) return fail(ctx, RT_E_INVALID_ARG, "rt_render: sample_end exceeds the sampler's sample count");
  // every frame renders the same NUMBER of sample indices (the kernel hands out tiles of K samples per pixel)
  uint64_t K64 = 0;
  std::vector<BatchFrame> table(n_frames);
  for (uint32_t i = 0; i < n_frames; i++) {
    if (!d_out_sums[i]) return fail(ctx, RT_E_INVALID_ARG, "rt_render_batch_device: NULL output pointer");
    const uint64_t k = sample_begins[i] < sample_end ? ((uint64_t)(sample_end - sample_begins[i]) + sample_stride - 1) / sample_stride : 0;
    if (i == 0) K64 = k;
    if (k != K64 || k == 0)
      return fail(ctx, RT_E_INVALID_ARG, "rt_render_batch_device: every frame of a batch must select the same, non-zero number of sample "
                                         "indices (split the batch)");
    table[i].k0 = (uint32_t)seeds[i]; table[i].k1 = (uint32_t)(seeds[i] >> 32);
    table[i].s_begin = sample_begins[i]; table[i].pad = 0; table[i].out = d_out_sums[i];
  }
  if (K64 > 65535ull * 64ull) return fail(ctx, RT_E_UNSUPPORTED, "rt_render: too many samples per pixel in one call");
  const uint32_t K = (uint32_t)K64;
  const uint32_t n_pix = cam->width * cam->height;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  ctx->fs = &ctx->own;
  ctx->stats = rt_stats();
  ctx->ms_bin = 0.f;
  if (ctx->batch_cap < n_frames) {
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ctx->d_batch) (void)rti::dev_free(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HIP_TRY(ctx, rti::dev_malloc(&ctx->d_batch, (size_t)n_frames * sizeof(BatchFrame)));
    ctx->batch_cap = n_frames;
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_start, st));
  // (pageable source: the runtime stages the table before the call returns, so `table` may go out of scope)
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_batch, table.data(), (size_t)n_frames * sizeof(BatchFrame), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(ctx->fs->ray_total, 0, 4 * sizeof(unsigned long long), st));
  ctx->last_ray_total = ctx->fs->ray_total;
  ctx->last_watchdog = nullptr;
  if (cam->max_depth > 0) {
    ctx->stats.paths = (uint64_t)n_pix * K * n_frames;
    ctx->cur_frames = ctx->d_batch;
    ctx->cur_n_frames = n_frames;
    rc = render_fused(ctx, scene, cam, dc, seeds[0], sample_begins[0], sample_stride, K, d_out_sums[0], st);
    ctx->cur_frames = nullptr;
    ctx->cur_n_frames = 1;
    if (rc) return rc;
  } else {
    for (uint32_t i = 0; i < n_frames; i++) HIP_TRY(ctx, hipMemsetAsync(d_out_sums[i], 0, (size_t)n_pix * 3 * sizeof(float), st));
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, st));
  ctx->timing_pending = true;
  return RT_OK;
}

int rt_render(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
              uint32_t sample_end, uint32_t sample_stride, float* out_sum) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!scene || !cam || !out_sum) return fail(ctx, RT_E_INVALID_ARG, "rt_render: NULL argument");
  if (cam->width == 0 || cam->height == 0) return fail(ctx, RT_E_INVALID_ARG, "camera: zero image size");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)cam->width * cam->height * 3 * sizeof(float);
  int rc = ensure_stage(ctx, 0, bytes);
  if (rc != RT_OK) return rc;
  float* d_out = static_cast<float*>(ctx->stage[0]);
  rc = rt_render_device(ctx, scene, cam, seed, sample_begin, sample_end, sample_stride, d_out, nullptr);
  if (rc == RT_OK) {
    hipError_t e = hipMemcpyAsync(out_sum, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, RT_E_HIP, std::string("rt_render: ") + hipGetErrorString(e));
    else rc = rt_internal_frame_status(ctx);  // a frame cut short by the scheduler's watchdog is an error, not a frame
  }
  return rc;
}

}  // extern "C"
int rt_internal_frame_status(rt_ctx* ctx) {
  if (!ctx->last_watchdog) return RT_OK;  // wavefront mode, or nothing rendered
  uint32_t flag = 0;
  HIP_TRY(ctx, hipMemcpy(&flag, ctx->last_watchdog, sizeof(flag), hipMemcpyDeviceToHost));
  if (flag) return fail(ctx, RT_E_INTERNAL, "fused kernel: a wave found no work for seconds and gave up (scheduler watchdog); the frame is incomplete");
  return RT_OK;
}
extern "C" {

int rt_get_stats(rt_ctx* ctx, rt_stats* out) {
  if (!ctx || !out) return RT_E_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->timing_pending) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    ctx->stats.ms_total = ms;
    unsigned long long rays[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(rays, ctx->last_ray_total, sizeof(rays), hipMemcpyDeviceToHost));
    ctx->stats.rays = rays[0];
    ctx->stats.samples_nan = rays[1];
    ctx->stats.samples_clamped = rays[2];
    HIP_TRY(ctx, hipMemcpy(ctx->stats.dbg, ctx->d_dbg, 16 * sizeof(float), hipMemcpyDeviceToHost));
    if (ctx->fused_stats) {  // exact counters of the instrumented build (rt_stats.dbg_counts)
      unsigned long long tl[16];
      HIP_TRY(ctx, hipMemcpy(tl, ctx->d_timeline, sizeof(tl), hipMemcpyDeviceToHost));
      ctx->stats.dbg_counts[0] = tl[7];
      // leaf tests by type + the global primitives every ray tests before it enters the tree
      for (int k = 0; k < 5; k++) ctx->stats.dbg_counts[1 + k] = tl[8 + k] + (uint64_t)ctx->last_global_types[k] * rays[0];
      for (int k = 0; k < 3; k++) ctx->stats.dbg_counts[6 + k] = tl[4 + k];
    }
    if (ctx->fused_stats && getenv("RT_VERBOSE")) {
      unsigned long long tl[8];
      HIP_TRY(ctx, hipMemcpy(tl, ctx->d_timeline, sizeof(tl), hipMemcpyDeviceToHost));
      const double span = (double)(tl[1] - tl[0]);
      const double n_waves = ctx->stats.dbg[13] > 0 ? (double)ctx->stats.dbg[12] : 0;  // (unused: counts come from the caller)
      (void)n_waves;
      fprintf(stderr, "[rt] node-loop trips sat out (lane-trips): at a leaf %llu, finished %llu, no path %llu; walked %.0f\n", tl[4], tl[5], tl[6], (double)ctx->stats.dbg[0]);
      fprintf(stderr, "[rt] timeline (100 MHz ticks): span %.0f, sum_end %llu, sum_exhaust %llu, start %llu\n", span, tl[2], tl[3], tl[0] & 0xFFFFFFFFFFull);
    }
    ctx->stats.dbg[11] = ctx->mode == 0 ? ctx->ms_bin : ctx->stats.dbg[11];
    if (ctx->last_watchdog) {
      // (instrumented build: dbg[14] / dbg[15] = wave cycles spent polling for work / visits of that loop, from the kernel)
      // tiles the main workgroups took from the front of the list / the partner workgroups from its end (diagnostic)
      uint32_t taken[2] = {0, 0};
      HIP_TRY(ctx, hipMemcpy(taken, ctx->last_watchdog - 2, sizeof(taken), hipMemcpyDeviceToHost));
      if (!ctx->fused_stats) {
        ctx->stats.dbg[14] = (float)taken[0];
        ctx->stats.dbg[15] = (float)taken[1];
      }
      const int frc = rt_internal_frame_status(ctx);
      if (frc != RT_OK) return frc;
    }
    ctx->timing_pending = false;
  }
  {  // probe counters: rt_intersect is synchronous, so they are current whether or not a render is pending
    unsigned long long trav[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpy(trav, ctx->d_trav, sizeof(trav), hipMemcpyDeviceToHost));
    ctx->stats.probe_rays = ctx->probe_rays;
    ctx->stats.probe_node_visits = trav[0];
    ctx->stats.probe_prim_tests = trav[1];
  }
  *out = ctx->stats;
  out->lib_alloc_calls = rti::g_alloc_calls.load();
  return RT_OK;
}

// ------------------------------------------------------------------------------------------------ resolve
int rt_resolve_device(rt_ctx* ctx, const float* d_sum, uint32_t width, uint32_t height, uint32_t spp, double gamma,
                      double* d_out_rgb, uint8_t* d_out_rgb8, void* hip_stream) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!d_sum || width == 0 || height == 0 || spp == 0 || !(gamma > 0))
    return fail(ctx, RT_E_INVALID_ARG, "rt_resolve: bad argument");
  if ((uint64_t)width * height > 0x3FFFFFFFull) return fail(ctx, RT_E_UNSUPPORTED, "rt_resolve: image too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  const uint32_t n = width * height * 3;
  const uint32_t blocks = std::min<uint32_t>((n + RT_BLOCK - 1) / RT_BLOCK, (uint32_t)ctx->n_cu * 8);
  hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(RT_BLOCK), 0, st, d_sum, n, 1.0 / (double)spp, 1.0 / gamma,
                     d_out_rgb, d_out_rgb8);
  HIP_TRY(ctx, hipGetLastError());
  return RT_OK;
}

int rt_resolve(rt_ctx* ctx, const float* sum, uint32_t width, uint32_t height, uint32_t spp, double gamma,
               double* out_rgb, uint8_t* out_rgb8) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!sum || width == 0 || height == 0) return fail(ctx, RT_E_INVALID_ARG, "rt_resolve: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)width * height * 3;
  int rc = ensure_stage(ctx, 0, n * 4);
  if (rc == RT_OK && out_rgb) rc = ensure_stage(ctx, 1, n * 8);
  if (rc == RT_OK && out_rgb8) rc = ensure_stage(ctx, 2, n);
  if (rc != RT_OK) return rc;
  float* d_sum = static_cast<float*>(ctx->stage[0]);
  double* d_rgb = out_rgb ? static_cast<double*>(ctx->stage[1]) : nullptr;
  uint8_t* d_u8 = out_rgb8 ? static_cast<uint8_t*>(ctx->stage[2]) : nullptr;
  hipError_t e = hipMemcpyAsync(d_sum, sum, n * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    rc = rt_resolve_device(ctx, d_sum, width, height, spp, gamma, d_rgb, d_u8, nullptr);
    if (rc == RT_OK && out_rgb) e = hipMemcpyAsync(out_rgb, d_rgb, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (rc == RT_OK && e == hipSuccess && out_rgb8) e = hipMemcpyAsync(out_rgb8, d_u8, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (rc == RT_OK && e != hipSuccess) rc = fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, std::string("rt_resolve: ") + hipGetErrorString(e));
  return rc;
}

}  // extern "C"
