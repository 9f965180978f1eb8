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
#include <vector>

#include "../../include/rt_abi.h"
#include "rt_bvh.h"
#include "rt_device.h"

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

template <bool SPHERES>
__global__ __launch_bounds__(RT_BLOCK) void k_intersect(Scene sc, uint32_t features, Queue q, Hits h,
                                                        const uint32_t* __restrict__ count) {
  __shared__ uint32_t stack[RT_STACK_DEPTH * RT_BLOCK];
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
                                                 stack + threadIdx.x, RT_BLOCK, nullptr, rng);
    else
      hit = traverse<GlobalAcc, uint32_t, false>(global_acc(sc), sc.root, features, o, d, q.time[i], RT_TMIN,
                                                 stack + threadIdx.x, RT_BLOCK, nullptr, rng);
    if (media) { q.r0[i] = rng.s0; q.r1[i] = rng.s1; }
    h.prim[i] = hit.prim;
    h.t[i] = hit.t;
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

template <uint32_t F, bool SPHERES>
__global__ __launch_bounds__(RT_BLOCK) void k_shade(Scene sc, Camera cam, Queue qi, Hits h, Queue qo,
                                                    const uint32_t* __restrict__ count_in,
                                                    uint32_t* __restrict__ count_out, float* __restrict__ Lr,
                                                    float* __restrict__ Lg, float* __restrict__ Lb, PassParams pp,
                                                    uint32_t bounce, uint32_t last_bounce) {
  const uint32_t n = *count_in;
  const uint32_t stride = gridDim.x * blockDim.x;
  // wave-uniform trip count: every lane of a wave reaches the ballot
  for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < n; base += stride) {
    const uint32_t i = base + (threadIdx.x & 63u);
    bool alive = false;
    ShadeOut so;
    uint32_t id = 0;
    double time = 0;
    PathRng rng = {0, 0};
    if (i < n) {
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
// One persistent kernel: every wave pulls work units (an 8x8-pixel tile x Ku sample indices) from a global
// counter and runs generate -> (intersect -> shade)* for them with per-lane path regeneration: a lane whose path
// ended takes the next sample of the unit's pool (wave64 ballot + prefix, no atomics), so lanes stay busy
// regardless of path length.  Radiance is summed in per-wave LDS accumulators as 2^-40 fixed point (integer
// adds are order-independent => bit-reproducible frames) and flushed to a u64 frame with global atomics.
// No path state ever touches HBM.
#define RT_UNIT_LEVELS 2
struct FusedParams {
  uint32_t K, s_begin, s_stride, Ku;
  uint32_t tiles_x, n_tiles, n_units;
  // unit schedule: level i covers units [level_unit0[i], level_unit0[i+1]) = passes over all tiles with level_ku[i]
  // sample indices per unit, starting at sample index level_k0[i]; unused levels start at n_units
  uint32_t level_unit0[RT_UNIT_LEVELS], level_k0[RT_UNIT_LEVELS], level_ku[RT_UNIT_LEVELS];
  uint32_t k0, k1;
  uint32_t stack_depth;
  uint32_t kb, kn;  // RESUME variant: hand-back / node-loop exit thresholds (traverse_resume)
  uint32_t noise_lds;  // the LDS-resident instances copy the noise jump table into LDS (scene has a noise texture)
  uint32_t kr;      // regenerate only when at least kr lanes are free (or none carries a path)
  uint32_t n_cached;  // four-wide nodes each workgroup copies into LDS (KIND_*_WIDE_HALF)
};
#define RT_FIX_SCALE 1099511627776.0f  // 2^40

enum { KIND_GENERIC = 0, KIND_SPHERES_GLOBAL = 1, KIND_SPHERES_LDS = 2, KIND_GENERIC_LDS = 3, KIND_GENERIC_WIDE = 4, KIND_SPHERES_WIDE = 5,
       KIND_GENERIC_WIDE_HALF = 6, KIND_SPHERES_WIDE_HALF = 7 };
constexpr bool kind_caches_top(int k) { return k == KIND_GENERIC_WIDE_HALF || k == KIND_SPHERES_WIDE_HALF; }
constexpr bool kind_in_lds(int k) { return k == KIND_SPHERES_LDS || k == KIND_GENERIC_LDS; }

__device__ __forceinline__ unsigned long long to_fix(float v) {
  if (!(v > 0.f)) return 0ull;  // also NaN
  v = fminf(v, 1048576.0f);
  return (unsigned long long)(v * RT_FIX_SCALE);
}

__global__ void k_store_camera(Camera cam, Camera* __restrict__ dst) {
  if (threadIdx.x == 0) *dst = cam;
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
template <uint32_t F, int KIND, int BLOCK, class StackT, bool STATS = false, bool RESUME = false>
__global__ __launch_bounds__(BLOCK, (BLOCK == 768 || (BLOCK == 256 && (F & RT_F_TEX_NOISE))) ? 3 : 4) void k_fused(Scene sc, const Camera* __restrict__ camp, FusedParams fp, uint32_t* __restrict__ unit_counter,
                                                 unsigned long long* __restrict__ fbacc,
                                                 unsigned long long* __restrict__ ray_total, float* __restrict__ dbg = nullptr) {
  // The camera block (26 doubles) stays in memory: read with scalar loads where a ray is generated instead of
  // occupying ~60 SGPRs for the whole kernel (they spilled to VGPR lanes: 68 v_readlane per regeneration).
  struct { uint32_t width, height, max_depth; } cam = {camp->width, camp->height, camp->max_depth};
  TravCounters tc = {0, 0, 0.f, 0.f, 0.f, 0.f};
  float st_regen = 0.f, st_wregen = 0.f, st_shade = 0.f, st_wshade = 0.f, st_tshade = 0.f, st_tregen = 0.f, st_ttotal = 0.f;
  uint64_t st_t0 = 0, st_tk = 0;
  if (STATS) st_tk = now_cycles();
  extern __shared__ __align__(16) unsigned char smem[];
  // LDS layout: [nodes][spheres] (KIND_SPHERES_LDS only) [accumulators][stack]
  size_t off = 0;
  unsigned char* l_nodes = smem;
  SphereRec* l_sph = nullptr;
  LdsTablePtr l_noise = nullptr;
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
    if ((F & RT_F_TEX_NOISE) && fp.noise_lds) {  // 64 KB noise jump table: 40 random 16-B lookups per noise sample
      uint4* lt = reinterpret_cast<uint4*>(smem + off);
      l_noise = (LdsTablePtr)(reinterpret_cast<u32x4*>(lt));
      off += 16 * 256 * 16;
      for (int i = threadIdx.x; i < 16 * 256; i += BLOCK) lt[i] = sc.noise_table[i];
    }
  }
  if (kind_caches_top(KIND)) {
    off += (size_t)fp.n_cached * 64;
    uint4* lc = reinterpret_cast<uint4*>(smem);
    for (uint32_t i = threadIdx.x; i < fp.n_cached * 4u; i += BLOCK) lc[i] = sc.nodes4h[i];
  }
  unsigned long long* l_acc = reinterpret_cast<unsigned long long*>(smem + off) + (threadIdx.x >> 6) * 192;
  off += (size_t)(BLOCK / 64) * 192 * 8;
  StackT* stack = reinterpret_cast<StackT*>(smem + off) + threadIdx.x;
  if (kind_in_lds(KIND) || kind_caches_top(KIND)) __syncthreads();
  typename AccOf<KIND>::type acc_init = AccOf<KIND>::make(sc, l_nodes, l_sph, l_noise);
  if constexpr (kind_caches_top(KIND)) acc_init.n_cached = (int32_t)fp.n_cached;
  const typename AccOf<KIND>::type acc = acc_init;

  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long my_rays = 0;
  // ---- wave state.  The wave works on one unit (8x8 tile x Ku sample indices) at a time; when the unit's sample pool
  // is exhausted it moves on to the next unit AT ONCE: the tile's LDS accumulators are flushed, the paths still in
  // flight become "stragglers" that add their radiance straight to the global frame, and the freed lanes are fed
  // from the new unit.  Lanes therefore never idle through a unit's drain (up to max_depth bounces of its
  // longest path) — only through the last one of the wave.
  uint32_t x0 = 0, y0 = 0, k_lo = 0, total = 0, next = 0;  // wave-uniform: current unit and its pool cursor
  bool have_unit = false, exhausted = false;               // wave-uniform
  bool active = false, straggler = false;
  d3 o = {0, 0, 0}, d = {0, 0, 0};
  double time = 0;
  f3 thr = {1.f, 1.f, 1.f};
  uint32_t depth = 0, lpix = 0;  // lpix: pixel index inside the tile, or (straggler) 3 * global pixel index
  PathRng rng = {0, 0};
  l_acc[lane * 3 + 0] = 0; l_acc[lane * 3 + 1] = 0; l_acc[lane * 3 + 2] = 0;
  __builtin_amdgcn_wave_barrier();
  // RESUME: the lane's traversal progress (node, stack, best hit) survives shade / regenerate rounds of its
  // neighbours; the per-ray constants are recomputed each round (unconditionally, so that they are not live across
  // the shading code)
  int32_t t_cur = RT_DONE, t_sp = 0, t_top = RT_DONE, t_best_prim = -1;
  double t_best_t = 0;
  bool fresh_lane = false;  // this lane holds a ray whose traversal has not started
  bool fresh = false;       // wave-uniform: some lane received a new ray since the last traversal call
  // shade one finished traversal; returns true when the path continues with a new ray in (o, d)
  auto bounce = [&](HitInfo hit) -> bool {
    if (STATS) { st_shade += 1.f; st_wshade += wave_share(); st_t0 = now_cycles(); }
    ShadeOut so = shade<F>(sc, acc, *camp, o, d, time, thr, hit, rng);
    if (STATS) st_tshade += (float)(now_cycles() - st_t0) * wave_share();
    if (!so.alive) {
      const unsigned long long r = to_fix(so.radiance.x), g = to_fix(so.radiance.y), b = to_fix(so.radiance.z);
      if (straggler) {
        if (r) atomicAdd(&fbacc[lpix + 0], r);
        if (g) atomicAdd(&fbacc[lpix + 1], g);
        if (b) atomicAdd(&fbacc[lpix + 2], b);
      } else {
        atomicAdd(&l_acc[lpix * 3 + 0], r);
        atomicAdd(&l_acc[lpix * 3 + 1], g);
        atomicAdd(&l_acc[lpix * 3 + 2], b);
      }
      return false;
    }
    depth++;
    if (depth >= cam.max_depth) return false;  // mod.rs:195-197: black
    o = so.o; d = so.d; thr = so.thr;
    return true;
  };
  while (true) {
    // ---- next unit when the pool is empty
    if (next >= total && !exhausted) {
      if (have_unit) {  // flush the finished tile's accumulators; paths still in flight go on as stragglers
        __builtin_amdgcn_wave_barrier();
        const uint32_t px = x0 + (lane & 7u), py = y0 + (lane >> 3);
        if (px < cam.width && py < cam.height) {
          const size_t p = ((size_t)py * cam.width + px) * 3;
          const unsigned long long a0 = l_acc[lane * 3], a1 = l_acc[lane * 3 + 1], a2 = l_acc[lane * 3 + 2];
          if (a0) atomicAdd(&fbacc[p], a0);
          if (a1) atomicAdd(&fbacc[p + 1], a1);
          if (a2) atomicAdd(&fbacc[p + 2], a2);
        }
        l_acc[lane * 3 + 0] = 0; l_acc[lane * 3 + 1] = 0; l_acc[lane * 3 + 2] = 0;
        __builtin_amdgcn_wave_barrier();
        if (active && !straggler) {
          straggler = true;
          lpix = ((y0 + (lpix >> 3)) * cam.width + x0 + (lpix & 7u)) * 3u;
        }
      }
      uint32_t u = 0;
      if (lane == 0) u = atomicAdd(unit_counter, 1u);
      u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);  // scalar: the unit bookkeeping below stays in SGPRs
      if (u >= fp.n_units) {
        exhausted = true;
        have_unit = false;
        total = 0; next = 0;
      } else {
        // chunk-major, tiles from the bottom row up: the units that start last are then the top rows, which in the
        // reference's scenes look at the sky (cheap paths make a short tail)
        // passes over the tiles, level by level: units get shorter towards the end of the frame (see launch_fused)
        uint32_t lv = 0;
#pragma unroll
        for (uint32_t i = 1; i < RT_UNIT_LEVELS; i++) lv = u >= fp.level_unit0[i] ? i : lv;
        const uint32_t v = u - fp.level_unit0[lv];
        const uint32_t tile = v % fp.n_tiles, pass = v / fp.n_tiles;
        x0 = (tile % fp.tiles_x) * 8; y0 = (tile / fp.tiles_x) * 8;
        k_lo = fp.level_k0[lv] + pass * fp.level_ku[lv];
        total = 64u * min(fp.level_ku[lv], fp.K - k_lo);
        next = 0;
        have_unit = true;
      }
    }
    // ---- regenerate: lanes without a path take the next samples of the pool (wave64 ballot + prefix)
    unsigned long long need = __ballot(!active);
    if (STATS) st_t0 = now_cycles();
    const uint32_t n_free = (uint32_t)__popcll(need);
    if (n_free && (n_free >= fp.kr || n_free == 64u) && next < total) {
      uint32_t j = next + wave_prefix(need);
      if (!active && j < total) {
        uint32_t pit = j & 63u, ks = j >> 6;
        uint32_t px = x0 + (pit & 7u), py = y0 + (pit >> 3);
        if (px < cam.width && py < cam.height) {
          if (STATS) { st_regen += 1.f; st_wregen += wave_share(); }
          uint32_t s = fp.s_begin + (k_lo + ks) * fp.s_stride;
          const Camera* cp = camp;
          asm volatile("" : "+s"(cp));  // keep the camera loads here (not hoisted into long-lived SGPRs)
          generate_ray(*cp, py, px, s, fp.k0, fp.k1, o, d, time, rng);
          thr = {1.f, 1.f, 1.f};
          depth = 0;
          lpix = pit;
          straggler = false;
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
      continue;
    }
    if (RESUME) {
      TravState ts;
      trav_ray_constants(ts, o, d);  // all lanes (garbage in lanes without a path is never used)
      if (fresh_lane) { t_cur = acc.root(sc.root); t_sp = 0; t_top = RT_DONE; t_best_prim = -1; t_best_t = __builtin_inf(); my_rays++; }
      fresh_lane = false;
      ts.cur = t_cur; ts.sp = t_sp; ts.top = t_top; ts.best_prim = t_best_prim; ts.best_t = t_best_t;
      ts.tmax_w = __double2float_ru(t_best_t) * RT_SLAB_WIDEN + 1e-30f;
      // ---- traverse until fewer than kb lanes are unfinished (stragglers keep their progress and go on next round
      //      beside the re-launched lanes); when nothing new was launched, run to completion
      traverse_resume<typename AccOf<KIND>::type, StackT, STATS>(acc, ts, active, F, o, d, time, RT_TMIN, stack, BLOCK,
                                                                 &tc, rng, fresh ? fp.kb : 1u, fp.kn);
      t_cur = ts.cur; t_sp = ts.sp; t_top = ts.top; t_best_prim = ts.best_prim; t_best_t = ts.best_t;
      const bool ready = active && t_cur == RT_DONE;
      fresh = __ballot(ready) != 0ull;
      if (ready) {
        HitInfo hit;
        hit.t = t_best_t;
        hit.prim = t_best_prim;
        if (bounce(hit)) fresh_lane = true;
        else active = false;
      }
    } else if (active) {
      // ---- one bounce for every lane that carries a path: closest hit, then shade
      my_rays++;
      HitInfo hit = traverse<typename AccOf<KIND>::type, StackT, STATS>(acc, acc.root(sc.root), F, o, d, time, RT_TMIN,
                                                                        stack, BLOCK, &tc, rng);
      if (!bounce(hit)) active = false;
    }
  }
  // rays traced (stats): one atomic per wave
  for (int off2 = 32; off2 > 0; off2 >>= 1) my_rays += __shfl_down(my_rays, off2, 64);
  if (lane == 0 && my_rays) atomicAdd(ray_total, my_rays);
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
  }
}

__global__ __launch_bounds__(RT_BLOCK) void k_fix_to_float(const unsigned long long* __restrict__ acc, float* __restrict__ fb,
                                                           uint32_t n) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    fb[i] = (float)((double)acc[i] * 9.094947017729282e-13);  // 2^-40
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
  __shared__ uint32_t stack[RT_STACK_DEPTH * RT_BLOCK];
  const uint32_t stride = gridDim.x * blockDim.x;
  TravCounters tc = {0, 0, 0.f, 0.f, 0.f, 0.f};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    rt_ray r = rays[i];
    d3 o = {r.o[0], r.o[1], r.o[2]}, d = {r.d[0], r.d[1], r.d[2]};
    const typename ProbeAcc<TREE>::type acc = ProbeAcc<TREE>::make(sc);
    PathRng rng = {i * 2654435761u + 1u, 0x9E3779B9u};  // only constant media draw (their probe hits are random)
    HitInfo hit = traverse<typename ProbeAcc<TREE>::type, uint32_t, true>(acc, acc.root(sc.root), features, o, d, r.time,
                                                                          RT_TMIN, stack + threadIdx.x, RT_BLOCK, &tc, rng);
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

// ================================================================================================ host side

// ---- fixed scratch of rt_render_device_noalloc (callers that cannot allocate: HIP graph capture, serving loops).
// One copy per GPU, held until the process ends: W*H*3 u64 frame accumulators for frames of up to RT_NOALLOC_MAX_PIXELS
// pixels (24 MiB) + 1 KiB of control words (camera block, ray counter, unit counter).  Host code never passes the
// array's name to a kernel: rt_init asks for its address on the ctx's device (hipGetSymbolAddress) and keeps it.
#define RT_NOALLOC_BYTES ((size_t)RT_NOALLOC_MAX_PIXELS * 3 * 8 + 1024)
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_BYTES];
static_assert(sizeof(Camera) <= 512, "camera block must fit the scratch's control area");

struct rt_scene {
  BvhNode* d_nodes = nullptr;
  float4* d_nodes4 = nullptr;  // four-wide form of the tree (HBM-resident fused instances)
  uint4* d_nodes4h = nullptr;  // ... with binary16 planes
  bool half_ok = false;        // every finite plane is inside the binary16 range
  int wide_stack = 0;          // stack entries a traversal of it needs
  Prim* d_prims = nullptr;
  Material* d_mats = nullptr;
  Texture* d_texs = nullptr;
  uint32_t* d_texels = nullptr;
  SphereRec* d_spheres = nullptr;
  Scene view{};
  int bvh_depth = 0;
  uint32_t features = 0;     // RT_F_* of what the scene contains
  bool spheres_only = false;
};

struct WorkBuffers {
  size_t cap = 0;
  void* slab = nullptr;  // one allocation holding both queues, hits and L
  Queue q[2]{};
  Hits hits{};
  float *Lr = nullptr, *Lg = nullptr, *Lb = nullptr;
};

// Device memory a fused render needs beyond its inputs and outputs.
struct FrameScratch {
  unsigned long long* fbacc = nullptr;      // W*H*3 fixed-point frame accumulators
  uint32_t* unit_counter = nullptr;         // work-unit counter of the persistent kernel
  Camera* cam = nullptr;                    // camera block of the render in flight (read with scalar loads)
  unsigned long long* ray_total = nullptr;  // rays traced (stats)
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
  int fused_kr = 16;
  std::map<std::pair<const void*, size_t>, int> fused_occupancy;  // workgroups per CU of (kernel instance, LDS bytes)
  std::map<const void*, size_t> fused_lds_opt_in;                 // dynamic LDS ceiling set per kernel instance
  int fused_tail = 4, fused_tail_ku = 0;  // the last passes' worth of samples goes out in shorter units (launch_fused)
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
  size_t fbacc_cap = 0;
  unsigned long long* d_trav = nullptr;  // probe counters: node visits, prim tests
  unsigned long long probe_rays = 0;
  uint32_t* d_counts = nullptr;
  size_t counts_cap = 0;
  uint4* d_noise_table = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_a = nullptr, ev_b = nullptr;
  bool timing_pending = false;
  rt_stats stats{};
  int n_cu = 256;
};

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
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = hipEventCreate(&ctx->ev_start)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_a)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev_b)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipMalloc(&ctx->own.ray_total, sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&ctx->own.unit_counter, sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&ctx->own.cam, sizeof(Camera))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemset(ctx->own.ray_total, 0, sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
  ctx->last_ray_total = ctx->own.ray_total;
  {
    // rt_render_device_noalloc: this device's copy of the file-scope array `scratch`.  The array's name is only a
    // host-side handle; the address the kernels get is the one hipGetSymbolAddress returns with the device current.
    void* base = nullptr;
    if ((e = hipGetSymbolAddress(&base, HIP_SYMBOL(scratch))) != hipSuccess || !base) return bail("hipGetSymbolAddress(scratch)", e);
    unsigned char* b = static_cast<unsigned char*>(base);
    ctx->fixed.fbacc = reinterpret_cast<unsigned long long*>(b);
    unsigned char* ctl = b + (size_t)RT_NOALLOC_MAX_PIXELS * 3 * 8;
    ctx->fixed.cam = reinterpret_cast<Camera*>(ctl);
    ctx->fixed.ray_total = reinterpret_cast<unsigned long long*>(ctl + 512);
    ctx->fixed.unit_counter = reinterpret_cast<uint32_t*>(ctl + 768);
  }
  // the partner workgroups' stream and fork / join events (launch_partner): created here so that no render call has to
  if ((e = hipStreamCreateWithFlags(&ctx->partner_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
  if ((e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
  if ((e = hipMalloc(&ctx->d_trav, 2 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMalloc(&ctx->d_dbg, 16 * sizeof(float))) != hipSuccess) return bail("hipMalloc", e);
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
  if ((e = hipMalloc(&ctx->d_noise_table, tab.size() * 4)) != hipSuccess) return bail("hipMalloc", e);
  if ((e = hipMemcpy(ctx->d_noise_table, tab.data(), tab.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)
    return bail("hipMemcpy", e);
  *out = ctx;
  return RT_OK;
}

static void free_work(rt_ctx* ctx) {
  if (ctx->wb.slab) (void)hipFree(ctx->wb.slab);
  ctx->wb = WorkBuffers();
}

int rt_shutdown(rt_ctx* ctx) {
  if (!ctx) return RT_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  free_work(ctx);
  if (ctx->d_counts) (void)hipFree(ctx->d_counts);
  if (ctx->own.ray_total) (void)hipFree(ctx->own.ray_total);
  if (ctx->own.unit_counter) (void)hipFree(ctx->own.unit_counter);
  if (ctx->own.cam) (void)hipFree(ctx->own.cam);
  if (ctx->own.fbacc) (void)hipFree(ctx->own.fbacc);
  if (ctx->d_trav) (void)hipFree(ctx->d_trav);
  if (ctx->d_dbg) (void)hipFree(ctx->d_dbg);
  if (ctx->d_noise_table) (void)hipFree(ctx->d_noise_table);
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
  if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
  if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->partner_stream) (void)hipStreamDestroy(ctx->partner_stream);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
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
    if (value < 0 || value > 2) return fail(ctx, RT_E_INVALID_ARG, "probe_tree must be 0, 1 or 2");
    ctx->probe_tree = (int)value;
  } else if (k == "fused_tail" || k == "fused_tail_ku") {
    if (value < 0 || value > 64) return fail(ctx, RT_E_INVALID_ARG, k + " must be in 0..64");
    (k == "fused_tail" ? ctx->fused_tail : ctx->fused_tail_ku) = (int)value;
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
  if (s->d_nodes) (void)hipFree(s->d_nodes);
  if (s->d_nodes4) (void)hipFree(s->d_nodes4);
  if (s->d_nodes4h) (void)hipFree(s->d_nodes4h);
  if (s->d_prims) (void)hipFree(s->d_prims);
  if (s->d_mats) (void)hipFree(s->d_mats);
  if (s->d_texs) (void)hipFree(s->d_texs);
  if (s->d_texels) (void)hipFree(s->d_texels);
  if (s->d_spheres) (void)hipFree(s->d_spheres);
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
      for (int a = 0; a < 3; a++) {
        q.f[a] = Q[a]; q.f[3 + a] = u[a]; q.f[6 + a] = v[a]; q.f[9 + a] = n[a] / len;
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
  rtb::BuildResult bvh = rtb::build_bvh(boxes, RT_STACK_DEPTH);
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
  HIP_TRY(ctx, hipMalloc(&sc->d_nodes, n_nodes * sizeof(BvhNode)));
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
      HIP_TRY(ctx, hipMalloc(&sc->d_nodes4h, hn.size() * sizeof(HalfNode)));
      HIP_TRY(ctx, hipMemcpy(sc->d_nodes4h, hn.data(), hn.size() * sizeof(HalfNode), hipMemcpyHostToDevice));
      sc->view.nodes4h = sc->d_nodes4h;
    }
    HIP_TRY(ctx, hipMalloc(&sc->d_nodes4, std::max<size_t>(wide.nodes.size(), 1) * sizeof(rtb::WideNode)));
    if (!wide.nodes.empty())
      HIP_TRY(ctx, hipMemcpy(sc->d_nodes4, wide.nodes.data(), wide.nodes.size() * sizeof(rtb::WideNode), hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, hipMalloc(&sc->d_prims, prims.size() * sizeof(Prim)));
  HIP_TRY(ctx, hipMemcpy(sc->d_prims, prims.data(), prims.size() * sizeof(Prim), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMalloc(&sc->d_mats, mats.size() * sizeof(Material)));
  HIP_TRY(ctx, hipMemcpy(sc->d_mats, mats.data(), mats.size() * sizeof(Material), hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMalloc(&sc->d_texs, texs.size() * sizeof(Texture)));
  HIP_TRY(ctx, hipMemcpy(sc->d_texs, texs.data(), texs.size() * sizeof(Texture), hipMemcpyHostToDevice));
  if (d->n_texel_bytes) {
    HIP_TRY(ctx, hipMalloc(&sc->d_texels, (d->n_texel_bytes + 3) & ~3ull));
    HIP_TRY(ctx, hipMemcpy(sc->d_texels, d->texels, d->n_texel_bytes, hipMemcpyHostToDevice));
  }
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
    HIP_TRY(ctx, hipMalloc(&sc->d_spheres, recs.size() * sizeof(SphereRec)));
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
int rt_intersect(rt_ctx* ctx, const rt_scene* scene, const rt_ray* rays, uint64_t n, rt_hit* out_hits) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!scene || (n && (!rays || !out_hits))) return fail(ctx, RT_E_INVALID_ARG, "rt_intersect: NULL argument");
  if (n == 0) return RT_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint64_t CH = 1u << 22;
  if (ctx->probe_tree != 0 && scene->wide_stack + 2 > RT_STACK_DEPTH)
    return fail(ctx, RT_E_UNSUPPORTED, "rt_intersect: the four-wide tree of this scene needs a deeper stack than the probe has");
  if (ctx->probe_tree == 2 && !scene->half_ok)
    return fail(ctx, RT_E_UNSUPPORTED, "rt_intersect: scene coordinates exceed the binary16 range");
  uint64_t chunk = n < CH ? n : CH;
  rt_ray* d_rays = nullptr;
  rt_hit* d_hits = nullptr;
  HIP_TRY(ctx, hipMalloc(&d_rays, chunk * sizeof(rt_ray)));
  hipError_t e = hipMalloc(&d_hits, chunk * sizeof(rt_hit));
  if (e != hipSuccess) {
    (void)hipFree(d_rays);
    return fail(ctx, RT_E_OOM, "rt_intersect: hipMalloc failed");
  }
  int rc = RT_OK;
  for (uint64_t off = 0; off < n && rc == RT_OK; off += chunk) {
    uint32_t m = (uint32_t)std::min<uint64_t>(chunk, n - off);
    uint32_t blocks = std::min<uint32_t>((m + RT_BLOCK - 1) / RT_BLOCK, (uint32_t)ctx->n_cu * 8);
    if ((e = hipMemcpyAsync(d_rays, rays + off, (size_t)m * sizeof(rt_ray), hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) { rc = RT_E_HIP; break; }
    auto kern = ctx->probe_tree == 2 ? k_probe<2> : ctx->probe_tree == 1 ? k_probe<1> : k_probe<0>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(RT_BLOCK), 0, ctx->stream, scene->view, scene->features, d_rays, m,
                       d_hits, ctx->d_trav);
    ctx->probe_rays += m;
    if ((e = hipGetLastError()) != hipSuccess) { rc = RT_E_HIP; break; }
    if ((e = hipMemcpyAsync(out_hits + off, d_hits, (size_t)m * sizeof(rt_hit), hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) { rc = RT_E_HIP; break; }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { rc = RT_E_HIP; break; }
  }
  (void)hipFree(d_rays);
  (void)hipFree(d_hits);
  if (rc != RT_OK) ctx->err = std::string("rt_intersect: ") + hipGetErrorString(e);
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
    const size_t total = 2 * per_q + capa * (4 + 8) + capa * 12;
    HIP_TRY(ctx, hipMalloc(&ctx->wb.slab, total));
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
    ctx->wb.cap = capa;
  }
  if (ctx->counts_cap < (size_t)depth + 2) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_counts) (void)hipFree(ctx->d_counts);
    ctx->d_counts = nullptr;
    ctx->counts_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_counts, ((size_t)depth + 2) * 4));
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
  tm.begin();
  hipLaunchKernelGGL((k_intersect<SPH>), dim3(blocks), dim3(RT_BLOCK), 0, st, scene->view, scene->features, qi,
                     ctx->wb.hits, ctx->d_counts + b);
  tm.end(ctx->stats.ms_intersect);
  tm.begin();
  hipLaunchKernelGGL((k_shade<F, SPH>), dim3(blocks), dim3(RT_BLOCK), 0, st, scene->view, dc, qi, ctx->wb.hits, qo,
                     ctx->d_counts + b, ctx->d_counts + b + 1, ctx->wb.Lr, ctx->wb.Lg, ctx->wb.Lb, pp, b,
                     (uint32_t)(b + 1 == depth));
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
// It runs on a second stream, pulls work units from the same counter and adds into the same frame accumulator.
static bool partner_possible(const rt_ctx* ctx, const rt_scene* scene, size_t lds_main) {
  if (!ctx->fused_partner || !scene->half_ok || scene->view.n_nodes4 >= 32768 || scene->view.n_prims >= 32768) return false;
  const size_t fixed = 4 * 192 * 8 + (size_t)(scene->wide_stack + 2) * 256 * 2;
  return lds_main + fixed + 16 * 64 + 1024 <= 160 * 1024;
}
static int launch_partner(rt_ctx* ctx, const rt_scene* scene, const FusedParams& fp_main, size_t lds_main, hipStream_t st) {
  auto kern = k_fused<RT_F_TEX_CHECKER, KIND_SPHERES_WIDE_HALF, 256, int16_t, false, true>;
  FusedParams fp = fp_main;  // same work units
  fp.stack_depth = (uint32_t)scene->wide_stack + 2u;
  const size_t fixed = 4 * 192 * 8 + (size_t)fp.stack_depth * 256 * 2;
  const size_t room = 160 * 1024 - lds_main - 1024;  // allocation granularity
  fp.n_cached = (uint32_t)std::min<size_t>((size_t)scene->view.n_nodes4, (room - fixed) / 64);
  const size_t smem = fixed + (size_t)fp.n_cached * 64;
  size_t& opted = ctx->fused_lds_opt_in[reinterpret_cast<const void*>(kern)];
  if (smem > opted) {
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    opted = smem;
  }
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->partner_stream, ctx->ev_fork, 0));
  hipLaunchKernelGGL(kern, dim3((uint32_t)ctx->n_cu), dim3(256), smem, ctx->partner_stream, scene->view, ctx->fs->cam, fp,
                     ctx->fs->unit_counter, ctx->fs->fbacc, ctx->fs->ray_total, (float*)nullptr);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->ev_join, ctx->partner_stream));
  HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
  return RT_OK;
}
template <uint32_t F, int KIND, int BLOCK, class StackT, bool STATS = false, bool RESUME = false>
static int launch_fused(rt_ctx* ctx, const rt_scene* scene, const Camera& dc, const FusedParams& fp0, size_t smem,
                        hipStream_t st) {
  auto kern = k_fused<F, KIND, BLOCK, StackT, STATS, RESUME>;
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
  const uint32_t waves = (uint32_t)ctx->n_cu * (uint32_t)per_cu * (BLOCK / 64);
  FusedParams fp = fp0;
  // work-unit size: ~48 units per resident wave for load balance at the end of the frame (a wave switches units
  // without draining, so small units cost only their flush), unless overridden
  uint32_t chunks = 1;
  if (ctx->fused_ku > 0) {
    chunks = (fp.K + (uint32_t)ctx->fused_ku - 1) / (uint32_t)ctx->fused_ku;
  } else {
    uint64_t target = (uint64_t)waves * 48;
    chunks = (uint32_t)std::min<uint64_t>(fp.K, std::max<uint64_t>(1, target / fp.n_tiles));
  }
  fp.Ku = (fp.K + chunks - 1) / chunks;
  chunks = (fp.K + fp.Ku - 1) / fp.Ku;
  // The frame ends when the last wave ends.  Unit costs spread widely (a tile of glass spheres takes 5-7 times the
  // average), and a costly unit that starts in the last pass over the tiles overruns all the others by most of its
  // length.  So units get shorter towards the end of the frame: with many passes over the tiles (C2: 15 of 7
  // samples) all but the last `fused_tail` (4) go out at full size and the rest in units of `fused_tail_ku` samples
  // (0 = half): -2.5 % on C2; single-sample units cost more than they save (+1 %).  With a single pass (C5 at 4K:
  // the tiles alone are units enough) halving sizes K/2, K/4, ... bought nothing (54.3 vs 53.9 ms) and are not used.
  // The schedule never changes the frame (a sample's key is its index, sums are integers).
  struct Level { uint32_t k0, ku, passes; } lv[RT_UNIT_LEVELS];
  int n_lv = 0;
  const uint32_t kt = ctx->fused_tail_ku > 0 ? (uint32_t)ctx->fused_tail_ku : (fp.Ku + 1) / 2;
  if (ctx->fused_tail > 0 && chunks >= 4 && kt < fp.Ku) {
    const uint32_t main_chunks = chunks - std::min<uint32_t>((uint32_t)ctx->fused_tail, chunks - 1);
    const uint32_t k_tail = main_chunks * fp.Ku;
    lv[n_lv++] = {0u, fp.Ku, main_chunks};
    lv[n_lv++] = {k_tail, kt, (fp.K - k_tail + kt - 1) / kt};
  } else {
    lv[n_lv++] = {0u, fp.Ku, chunks};
  }
  uint64_t unit0 = 0;
  for (int i = 0; i < RT_UNIT_LEVELS; i++) {
    if (i < n_lv) {
      fp.level_unit0[i] = (uint32_t)unit0; fp.level_k0[i] = lv[i].k0; fp.level_ku[i] = lv[i].ku;
      unit0 += (uint64_t)fp.n_tiles * lv[i].passes;
      if (unit0 >= 0xFFFFFFFFull) return fail(ctx, RT_E_UNSUPPORTED, "rt_render: more work units than the 32-bit unit counter holds (raise fused_ku)");
    } else {
      fp.level_unit0[i] = 0xFFFFFFFFu; fp.level_k0[i] = fp.K; fp.level_ku[i] = 1;
    }
  }
  fp.n_units = (uint32_t)unit0;
  const uint32_t blocks = std::min<uint32_t>((uint32_t)ctx->n_cu * (uint32_t)per_cu, (fp.n_units + BLOCK / 64 - 1) / (BLOCK / 64));
  // the camera travels as a kernel argument (captured at launch) into its device slot: stream-ordered, and no host
  // buffer has to outlive an asynchronous copy
  hipLaunchKernelGGL(k_store_camera, dim3(1), dim3(64), 0, st, dc, ctx->fs->cam);
  constexpr bool kHasPartner = F == RT_F_TEX_CHECKER && KIND == KIND_SPHERES_LDS && BLOCK == 1024 && RESUME && !STATS;
  // (a frame of under ~8 M samples is over in well under a millisecond: the second launch and its join cost more there)
  const bool partner = kHasPartner && per_cu == 1 && (uint64_t)fp.n_tiles * 64u * fp.K >= (8ull << 20) &&
                       partner_possible(ctx, scene, smem);
  if (partner) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));  // frame cleared, counter reset, camera stored
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), smem, st, scene->view, ctx->fs->cam, fp, ctx->fs->unit_counter, ctx->fs->fbacc,
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

static int render_fused(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, const Camera& dc, uint64_t seed,
                        uint32_t sample_begin, uint32_t sample_stride, uint32_t K, float* d_out_sum, hipStream_t st) {
  const size_t n_val = (size_t)cam->width * cam->height * 3;
  if (ctx->fs == &ctx->own && ctx->fbacc_cap < n_val) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own.fbacc) (void)hipFree(ctx->own.fbacc);
    ctx->own.fbacc = nullptr;
    ctx->fbacc_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->own.fbacc, n_val * 8));
    ctx->fbacc_cap = n_val;
  }
  HIP_TRY(ctx, hipMemsetAsync(ctx->fs->fbacc, 0, n_val * 8, st));
  HIP_TRY(ctx, hipMemsetAsync(ctx->fs->unit_counter, 0, sizeof(uint32_t), st));
  FusedParams fp;
  memset(&fp, 0, sizeof(fp));
  fp.K = K; fp.s_begin = sample_begin; fp.s_stride = sample_stride;
  fp.tiles_x = (cam->width + 7) / 8;
  fp.n_tiles = fp.tiles_x * ((cam->height + 7) / 8);
  fp.k0 = (uint32_t)seed; fp.k1 = (uint32_t)(seed >> 32);
  const uint32_t sdepth = (uint32_t)std::max(1, scene->bvh_depth + 1);
  fp.stack_depth = sdepth;
  fp.kb = (uint32_t)ctx->fused_kb; fp.kn = (uint32_t)ctx->fused_kn; fp.kr = use_resume(ctx, scene) ? (uint32_t)ctx->fused_kr : 1u;
  const bool lean = (scene->features & ~RT_F_TEX_CHECKER) == 0;
  const uint32_t tex_only = scene->features & RT_F_TEX_ANY;  // texture kinds present (kernels are specialised on them)
  const bool small_ids = (int64_t)scene->view.n_nodes * (RT_SLAB_NODE_BYTES / 16) < 32768 && scene->view.n_prims < 32000;  // int16 stack entries
  int rc;
  // LDS budget of the 1024-thread spheres-in-LDS variant: scene + accumulators + u16 stack
  const size_t lds_noise = (scene->features & RT_F_TEX_NOISE) ? (size_t)16 * 256 * 16 : 0;
  fp.noise_lds = lds_noise ? 1u : 0u;
  const size_t lds_tables = lds_noise;  // material / texture records stay in HBM/L2: LDS copies bought nothing measurable
  const size_t lds_scene = RT_SLAB_REGION_BYTES(scene->view.n_nodes) + (size_t)scene->view.n_prims * 64 + lds_tables;
  const size_t lds_1024 = lds_scene + 16 * 192 * 8 + (size_t)sdepth * 1024 * 2;
  const size_t lds_generic_1024 = RT_SLAB_REGION_BYTES(scene->view.n_nodes) + (size_t)scene->view.n_prims * 128 + lds_tables +
                                  16 * 192 * 8 + (size_t)sdepth * 1024 * 2;
  StageTimer tm(ctx, st);
  tm.begin();
  if (ctx->fused_stats) HIP_TRY(ctx, hipMemsetAsync(ctx->d_dbg, 0, 16 * sizeof(float), st));
  if (scene->spheres_only && ctx->fused_lds && small_ids && lds_1024 <= 160 * 1024) {
    if (lean && ctx->fused_stats) {
      if (use_resume(ctx, scene)) rc = launch_fused<RT_F_TEX_CHECKER, KIND_SPHERES_LDS, 1024, int16_t, true, true>(ctx, scene, dc, fp, lds_1024, st);
      else rc = launch_fused<RT_F_TEX_CHECKER, KIND_SPHERES_LDS, 1024, int16_t, true, false>(ctx, scene, dc, fp, lds_1024, st);
    } else if (lean) rc = launch_variant<RT_F_TEX_CHECKER, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
    else if (tex_only == RT_F_TEX_IMAGE) rc = launch_variant<RT_F_TEX_IMAGE, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
    else if (!(tex_only & RT_F_TEX_IMAGE)) rc = launch_variant<RT_F_TEX_CHECKER | RT_F_TEX_NOISE, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
    else rc = launch_variant<RT_F_TEX_ANY, KIND_SPHERES_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_1024, st);
  } else if (!scene->spheres_only && ctx->fused_lds && small_ids && lds_generic_1024 <= 160 * 1024) {
    if (tex_only == 0) rc = launch_variant<RT_F_QUAD | RT_F_MEDIUM, KIND_GENERIC_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_generic_1024, st);
    else rc = launch_variant<RT_F_ALL, KIND_GENERIC_LDS, 1024, int16_t>(ctx, scene, dc, fp, lds_generic_1024, st);
  } else {
    // HBM-resident scene: four-wide tree by default (fused_wide = 0: the binary tree, kept as a cross-check)
    const bool wide = ctx->fused_wide != 0;
    const uint32_t sd = wide ? (uint32_t)scene->wide_stack + 2u : sdepth;
    fp.stack_depth = sd;
    const size_t smem = 4 * 192 * 8 + (size_t)sd * 256 * 4;
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
    const uint32_t cblock = heavy ? 768u : 1024u;
    const size_t cfixed = (size_t)(cblock / 64) * 192 * 8 + (size_t)sd * cblock * 2;
    const bool ids16 = scene->view.n_nodes4 < 32768 && scene->view.n_prims < 32768;
    if (wide && ctx->fused_half && scene->half_ok && ids16 && cfixed + 16 * 64 <= 160 * 1024) {
      fp.n_cached = (uint32_t)std::min<size_t>((size_t)scene->view.n_nodes4, (160 * 1024 - cfixed) / 64);
      const size_t csmem = cfixed + (size_t)fp.n_cached * 64;
      if (scene->spheres_only && lean) rc = launch_variant<RT_F_TEX_CHECKER, KIND_SPHERES_WIDE_HALF, 1024, int16_t>(ctx, scene, dc, fp, csmem, st);
      else if (scene->spheres_only) rc = launch_variant<RT_F_TEX_ANY, KIND_SPHERES_WIDE_HALF, 768, int16_t>(ctx, scene, dc, fp, csmem, st);
      else if (ctx->fused_stats) {
        if (use_resume(ctx, scene)) rc = launch_fused<RT_F_ALL, KIND_GENERIC_WIDE_HALF, 768, int16_t, true, true>(ctx, scene, dc, fp, csmem, st);
        else rc = launch_fused<RT_F_ALL, KIND_GENERIC_WIDE_HALF, 768, int16_t, true, false>(ctx, scene, dc, fp, csmem, st);
      } else rc = launch_variant<RT_F_ALL, KIND_GENERIC_WIDE_HALF, 768, int16_t>(ctx, scene, dc, fp, csmem, st);
    } else if (wide) { RT_GLOBAL_DISPATCH(KIND_SPHERES_WIDE, KIND_GENERIC_WIDE) }
    else { RT_GLOBAL_DISPATCH(KIND_SPHERES_GLOBAL, KIND_GENERIC) }
#undef RT_GLOBAL_DISPATCH
  }
  if (rc) return rc;
  tm.end(ctx->stats.ms_intersect);
  const uint32_t blocks = std::min<uint32_t>((uint32_t)((n_val + RT_BLOCK - 1) / RT_BLOCK), (uint32_t)ctx->n_cu * 8);
  hipLaunchKernelGGL(k_fix_to_float, dim3(blocks), dim3(RT_BLOCK), 0, st, ctx->fs->fbacc, d_out_sum, (uint32_t)n_val);
  HIP_TRY(ctx, hipGetLastError());
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
      return fail(ctx, RT_E_UNSUPPORTED, "rt_render_device_noalloc: frame of " + std::to_string(n_pix) + " pixels exceeds the fixed scratch (" +
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
  HIP_TRY(ctx, hipEventRecord(ctx->ev_start, st));
  HIP_TRY(ctx, hipMemsetAsync(ctx->fs->ray_total, 0, sizeof(unsigned long long), st));
  ctx->last_ray_total = ctx->fs->ray_total;
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

// Same render with every scratch buffer (frame accumulators, unit counter, camera block, ray counter) in the
// file-scope device array `scratch`: no hipMalloc / hipFree / stream or event creation on this path.
int rt_render_device_noalloc(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
                             uint32_t sample_end, uint32_t sample_stride, float* d_out_sum, void* hip_stream) {
  return render_device_impl(ctx, scene, cam, seed, sample_begin, sample_end, sample_stride, d_out_sum, hip_stream, true);
}

int rt_render(rt_ctx* ctx, const rt_scene* scene, const rt_camera* cam, uint64_t seed, uint32_t sample_begin,
              uint32_t sample_end, uint32_t sample_stride, float* out_sum) {
  if (!ctx) return RT_E_INVALID_ARG;
  if (!scene || !cam || !out_sum) return fail(ctx, RT_E_INVALID_ARG, "rt_render: NULL argument");
  if (cam->width == 0 || cam->height == 0) return fail(ctx, RT_E_INVALID_ARG, "camera: zero image size");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)cam->width * cam->height * 3 * sizeof(float);
  float* d_out = nullptr;
  HIP_TRY(ctx, hipMalloc(&d_out, bytes));
  int rc = rt_render_device(ctx, scene, cam, seed, sample_begin, sample_end, sample_stride, d_out, nullptr);
  if (rc == RT_OK) {
    hipError_t e = hipMemcpyAsync(out_sum, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, RT_E_HIP, std::string("rt_render: ") + hipGetErrorString(e));
  }
  (void)hipFree(d_out);
  return rc;
}

int rt_get_stats(rt_ctx* ctx, rt_stats* out) {
  if (!ctx || !out) return RT_E_INVALID_ARG;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->timing_pending) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_stop));
    float ms = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    ctx->stats.ms_total = ms;
    unsigned long long rays = 0;
    HIP_TRY(ctx, hipMemcpy(&rays, ctx->last_ray_total, sizeof(rays), hipMemcpyDeviceToHost));
    ctx->stats.rays = rays;
    HIP_TRY(ctx, hipMemcpy(ctx->stats.dbg, ctx->d_dbg, 16 * sizeof(float), hipMemcpyDeviceToHost));
    unsigned long long trav[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpy(trav, ctx->d_trav, sizeof(trav), hipMemcpyDeviceToHost));
    ctx->stats.probe_rays = ctx->probe_rays;
    ctx->stats.probe_node_visits = trav[0];
    ctx->stats.probe_prim_tests = trav[1];
    ctx->timing_pending = false;
  }
  *out = ctx->stats;
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
  float* d_sum = nullptr;
  double* d_rgb = nullptr;
  uint8_t* d_u8 = nullptr;
  int rc = RT_OK;
  hipError_t e = hipMalloc(&d_sum, n * 4);
  if (e == hipSuccess && out_rgb) e = hipMalloc(&d_rgb, n * 8);
  if (e == hipSuccess && out_rgb8) e = hipMalloc(&d_u8, n);
  if (e == hipSuccess) e = hipMemcpyAsync(d_sum, sum, n * 4, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    rc = rt_resolve_device(ctx, d_sum, width, height, spp, gamma, d_rgb, d_u8, nullptr);
    if (rc == RT_OK && out_rgb) e = hipMemcpyAsync(out_rgb, d_rgb, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (rc == RT_OK && e == hipSuccess && out_rgb8) e = hipMemcpyAsync(out_rgb8, d_u8, n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  if (d_sum) (void)hipFree(d_sum);
  if (d_rgb) (void)hipFree(d_rgb);
  if (d_u8) (void)hipFree(d_u8);
  if (rc == RT_OK && e != hipSuccess) rc = fail(ctx, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, std::string("rt_resolve: ") + hipGetErrorString(e));
  return rc;
}

}  // extern "C"
