// rt_device.h — device-side data layout and functions of the MI355X path tracer (gfx950 only).
//
// Precision plan: geometry that decides WHERE a hit is — ray origin/direction, sphere
// quadratic, hit point, normal — is f64 like the reference (src/vec3.rs:11); BVH slab culling is f32 on
// conservatively padded boxes (result-neutral: closest-hit does not depend on what is culled,
// src/hittable/containers.rs:177-191); colour / throughput / scatter-direction sampling is f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define RT_BLOCK 256
#define RT_STACK_DEPTH 32  // the host BVH builder guarantees tree depth < RT_STACK_DEPTH
#define RT_TMIN 0.000001   // src/camera/mod.rs:198

namespace rtd {

// Floating-point contraction is decided HERE, per source expression (a * b + c inside one expression becomes an fma),
// not by the optimiser (hipcc's default, -ffp-contract=fast, fuses across statements as it sees fit in each place the
// code is inlined).  Every kernel instance then rounds the shading math identically, so the instance that happens to
// trace a path — LDS-resident main kernel or its partner workgroups, decided at run time — cannot change the frame:
// bit-identity holds by construction, not by luck of compilation.  (The f64 geometry below is contract(off): it follows
// the reference's operation order.)
#pragma clang fp contract(on)

// ----------------------------------------------------------------------------------------------- layout
struct __attribute__((aligned(64))) BvhNode {  // 64 B: both child boxes + child links, one fetch per step
  float l_lo[3], l_hi[3];
  float r_lo[3], r_hi[3];
  int32_t left, right;  // >= 0: inner node index; < 0: leaf, primitive index = ~child; RT_EMPTY: nothing
  int32_t pad[2];
};
#define RT_EMPTY_CHILD INT32_MIN

struct __attribute__((aligned(128))) Prim {  // 128 B
  // SPHERE / MOVING_SPHERE: f[0..2]=center0, f[3..5]=center1, f[6]=radius
  // QUAD: f[0..2]=Q, f[3..5]=v x w, f[6..8]=w x u (w = u x v / |u x v|^2: alpha = (p - Q).(v x w), beta = (p - Q).(w x u)),
  //       f[9..11]=unit normal, f[12]=D, f[13]=1/|u x v|
  // MEDIUM_SPHERE: f[0..2]=center, f[3]=f[12]=-1/density, f[6]=radius
  // MEDIUM_BOX: f[0..2]=corner, f[3..5],f[6..8],f[9..11]=edges / |edge|^2, f[12]=-1/density
  // AABOX (type 5, library-internal: six QUADs that form an axis-aligned box, merged at upload): f[0..2]=lo, f[3..5]=hi,
  //        f[6..8] reinterpreted as int32 face[6] = primitive index of the quad on face 2*axis + (0: lo plane, 1: hi plane)
  double f[14];
  uint32_t type, mat;
  int32_t obj;
  uint32_t pad;
};
// compact 64-B sphere record (spheres-only scenes; this is what the fused kernel keeps in LDS)
struct __attribute__((aligned(64))) SphereRec {
  double c0[3], c1[3], r;
  uint32_t type, mat;
};

struct Material {  // 32 B
  uint32_t kind;
  int32_t tex;  // texture index, or -1 when the texture is a solid colour inlined into `albedo` at upload
  float albedo[3];
  float fuzz, ior;
  uint32_t pad;
};
struct Texture {  // 64 B
  uint32_t kind;
  int32_t odd, even;   // CHECKER children; -1 when the child is a solid colour inlined into rgb (odd) / rgb2 (even)
  uint32_t width, height;
  uint32_t texel_off;  // in texels (RGBA8 words)
  float rgb[3];        // SOLID colour, or the inlined `odd` child of a CHECKER
  float rgb2[3];       // inlined `even` child of a CHECKER
  uint32_t uvtab_off;  // IMAGE: 32-bit words from Scene::texels to the table of sphere_texel_checked; 0 = none
  uint32_t pad;
  double inv_scale;
};

struct Scene {
  const BvhNode* nodes;
  const Prim* prims;
  const SphereRec* spheres;  // non-null iff every primitive is a (moving) sphere
  const Material* mats;
  const Texture* texs;
  const uint32_t* texels;    // RGBA8
  const uint4* noise_table;  // 16 x 256 x 128-bit GF(2) table of the xoroshiro short jump
  const float4* nodes4;      // four-wide form of the tree (128-B nodes, see WideAcc), root = node 0
  const uint4* nodes4h;      // the same with binary16 planes rounded outward (64-B nodes, see WideHalfAcc)
  int32_t n_nodes4;
  int32_t n_nodes, n_prims, n_mats, n_texs;
  int32_t root;  // >=0 node, <0 single leaf
  int32_t bvh_depth;
  // Spheres so large that (nearly) every ray has to test them — book_cover's r = 1000 ground — are kept OUT of the tree
  // and tested first, by all the lanes that start a ray together, instead of as one more leaf round that the wave's
  // node-walkers sit out (global_prims_test).  The closest hit does not depend on the order of the tests.
  // Scenes of at most eight primitives (the Cornell boxes) have no tree at all: every ray tests every primitive in turn,
  // all lanes the same one — no node steps, no divergent leaf rounds (tree_empty).
  int32_t n_global, tree_empty;
  const int32_t* global_list;  // n_global primitive indices (device memory; read with scalar loads)
};

struct Camera {
  double center[3], pixel00[3], du[3], dv[3], dfu[3], dfv[3];
  double defocus_angle;
  double grid_interval, grid_offset;  // Uniform sampler: 1/n and 1/(2n) + 0.5 (mod.rs:153-157), computed once on the host
  float background[3];
  uint32_t width, height, sampler_kind, sampler_n, max_depth, background_mode;
};

// Path state, structure of arrays (one queue).  72 B per path.
struct Queue {
  double *ox, *oy, *oz, *dx, *dy, *dz, *time;
  float *tr, *tg, *tb;
  uint32_t* id;  // pass-local path index: pixel_local * S + k
  uint32_t *r0, *r1;  // per-path RNG state
};
struct Hits {
  int32_t* prim;
  double* t;
};

// texture-kind / primitive-kind feature masks: kernels are specialised on what a scene actually contains
#define RT_F_TEX_CHECKER 1u
#define RT_F_TEX_IMAGE 2u
#define RT_F_TEX_NOISE 4u
#define RT_F_QUAD 8u
#define RT_F_MEDIUM 16u
#define RT_F_BOX 32u  // merged axis-aligned boxes (RT_PRIM_AABOX records) are present
#define RT_F_ALL 63u
#define RT_F_ALL_NOBOX 31u  // everything but merged boxes: the instance of general scenes that have none
#define RT_PRIM_AABOX 5u
#ifndef RT_AABOX_INLINE
#define RT_AABOX_INLINE __device__ __forceinline__
#endif
#define RT_F_TEX_ANY 7u

// ----------------------------------------------------------------------------------------------- RNG
// a ^ b ^ c in one instruction: gfx950's v_bitop3_b32 evaluates any three-input truth table (0x96 = odd parity); hipcc
// emits two v_xor_b32.  `k` variant: the third operand from an SGPR (one scalar operand per VOP3 instruction).
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return a ^ b ^ c;
#endif
}
__host__ __device__ __forceinline__ uint32_t xor3k(uint32_t a, uint32_t b, uint32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(k));
  return r;
#else
  return a ^ b ^ k;
#endif
}
// Philox4x32-10 (Salmon et al., SC'11).  generate_ray evaluates it once per path: counter = (pixel, sample, 0, 0), key = seed.
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
  // the key is wave-uniform: its ten bumped values are two scalar adds per round HERE.  Left to itself the compiler
  // hoists all eighteen out of the render loop, runs out of SGPRs and reads them back with a v_readlane each.
  k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k0);
  k1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k1);
  asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = xor3k(hi1, c1, k0), n1 = lo1, n2 = xor3k(hi0, c3, k1), n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Per-path stream: Philox (counter-based, keyed by pixel / sample / seed) is evaluated ONCE per path; two of its
// words feed the lens sample and two seed a xoroshiro64** generator (Blackman & Vigna) that supplies the
// path's remaining draws (time, sub-pixel jitter, scatter directions).  Streams of different (pixel, sample,
// seed) are decorrelated by Philox; draws within a path cost ~10 integer ops instead of a 10-round Philox.
struct PathRng {
  uint32_t s0, s1;
};
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ uint32_t rng_next(PathRng& g) {
  const uint32_t s0 = g.s0;
  uint32_t s1 = g.s1;
  const uint32_t r = rotl32(s0 * 0x9E3779BBu, 5) * 5u;
  s1 ^= s0;
  g.s0 = xor3(rotl32(s0, 26), s1, s1 << 9);
  g.s1 = rotl32(s1, 13);
  return r;
}

__device__ __forceinline__ float u01f(uint32_t x) { return (float)(x >> 8) * 5.9604645e-8f + 2.9802322e-8f; }  // (0,1)
// (x + 0.5) * 2^-32 as one fma: both forms are exact (x < 2^32), so the bits are the same
__device__ __forceinline__ double u01d(uint32_t x) { return __builtin_fma((double)x, 2.3283064365386963e-10, 1.1641532182693481e-10); }

// ----------------------------------------------------------------------------------------------- math
struct d3 { double x, y, z; };
struct f3 { float x, y, z; };
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 tof(d3 a) { return {(float)a.x, (float)a.y, (float)a.z}; }
__device__ __forceinline__ d3 tod(f3 a) { return {(double)a.x, (double)a.y, (double)a.z}; }
__device__ __forceinline__ f3 normalize(f3 a) {
  float inv = __builtin_amdgcn_rsqf(dot(a, a));
  return inv * a;
}

// ----------------------------------------------------------------------------------------------- slab test
// Per-ray constants of the f32 slab test.  t = plane*inv - (o*inv); the origin's f32 rounding error
// |o|*2^-24*|inv| is covered by widening o*inv per axis (nx.. for entry planes, fx.. for exit planes); the box
// coordinates' own rounding is covered by padding at build time (rt_bvh.cpp).
#define RT_SLAB_WIDEN 1.0000019f  // exit distances are widened by 16 ulp (+1e-30) before the entry <= exit test
// f32 upper bound of the best hit distance, widened like the exits: >= t * RT_SLAB_WIDEN + 1e-30 for every t >= 0.  Two
// more ulp in the factor cover the round-to-nearest conversion and the multiply (a software round-up conversion cost
// ~15 instructions per accepted hit and per round); a looser bound only culls a little later.
__device__ __forceinline__ float widen_tmax(double t) { return __builtin_fmaf((float)t, 1.0000021f, 1e-30f); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct SlabRay {
  // Per axis two register PAIRS, laid out for v_pk_fma_f32 (two f32 FMAs per issue slot — the packed form is what the
  // 157 TFLOP/s vector peak is made of; plain v_fma_f32 runs at half of it):
  //   m = {1/d, (1/d) * RT_SLAB_WIDEN}: multiplier of the entry planes / of the exit planes (widening folded in)
  //   a = {o/d widened towards later entry, (o/d widened towards earlier exit) * RT_SLAB_WIDEN - 1e-30}: what is
  //       subtracted from plane * multiplier for entry / exit planes
  f32x2 mx, my, mz;
  f32x2 ax, ay, az;
  bool sx, sy, sz;       // direction negative on axis (select-based node fetch: GlobalAcc / SphereAcc)
  uint32_t ox, oy, oz;   // byte offset of the ray's direction-ordered plane quad inside a slab node (SlabLdsAcc)
};
#define RT_SLAB_NODE_BYTES 112  // see SlabLdsAcc
#define RT_SLAB_REGION_BYTES(n_nodes) ((size_t)(n_nodes) * RT_SLAB_NODE_BYTES)
__device__ __forceinline__ SlabRay make_slab(d3 o, d3 d) {
  SlabRay s;
  float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
  const float ix = __builtin_amdgcn_rcpf((float)d.x);
  const float iy = __builtin_amdgcn_rcpf((float)d.y);
  const float iz = __builtin_amdgcn_rcpf((float)d.z);
  float ax = ox * ix, ay = oy * iy, az = oz * iz;
  const float E = 9.5367431640625e-07f;  // 2^-20: > 8 ulp, covers o / d / rcp / fma rounding
  float ex = __builtin_fabsf(ax) * E, ey = __builtin_fabsf(ay) * E, ez = __builtin_fabsf(az) * E;
  s.mx = {ix, ix * RT_SLAB_WIDEN}; s.my = {iy, iy * RT_SLAB_WIDEN}; s.mz = {iz, iz * RT_SLAB_WIDEN};
  s.ax = {ax + ex, (ax - ex) * RT_SLAB_WIDEN - 1e-30f};
  s.ay = {ay + ey, (ay - ey) * RT_SLAB_WIDEN - 1e-30f};
  s.az = {az + ez, (az - ez) * RT_SLAB_WIDEN - 1e-30f};
  s.sx = ix < 0.f; s.sy = iy < 0.f; s.sz = iz < 0.f;
  s.ox = s.sx ? 16u : 0u; s.oy = s.sy ? 48u : 32u; s.oz = s.sz ? 80u : 64u;
  return s;
}
// Both children of a node at once.  Each axis quad holds (entry L, entry R, exit L, exit R): the children's planes
// already ordered for the ray's direction.  tl / tr: entry distances clamped to 0; hit iff entry <= widened exit.
// The widening (x RT_SLAB_WIDEN + 1e-30) is folded into the exit planes' per-ray constants and into `tmax_w`
// (= tmax * RT_SLAB_WIDEN + 1e-30, kept by the caller): min(a, b) * K + c == min(a * K + c, b * K + c) for K > 0.
// v_max / v_min ignore a quiet NaN operand (0*inf axes) => conservative.
// {p.x * m.x - a.x, p.y * m.x - a.x}: two planes against the ENTRY constants (low halves of m and a, broadcast by
// op_sel).  Written as asm because the compiler, given the same thing as a vector fma of splats, hoists the four splats
// per axis out of the traversal loop as register pairs of their own (24 VGPRs of ray constants instead of 12).
__device__ __forceinline__ f32x2 pk_planes_entry(f32x2 p, f32x2 m, f32x2 a) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(p), "v"(m), "v"(a));
  return r;
}
// {p.x * m.y - a.y, p.y * m.y - a.y}: two planes against the EXIT constants (high halves of m and a, broadcast)
__device__ __forceinline__ f32x2 pk_planes_exit(f32x2 p, f32x2 m, f32x2 a) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(p), "v"(m), "v"(a));
  return r;
}
// min / max of plane distances, NaN-ignoring like fminf / fmaxf (the only NaNs here are the quiet ones of 0 * inf).
// Also asm: on the output of an asm statement the compiler would first canonicalise every operand (v_max_f32 x, x, x).
__device__ __forceinline__ float slab_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float slab_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float slab_max3_zero(float a, float b) { float r; asm("v_max3_f32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float slab_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ void slab_pair_test(const SlabRay& s, float4 qx, float4 qy, float4 qz, float tmax_w, float& tl,
                                               float& tr, bool& hl, bool& hr) {
  // twelve plane distances in six packed FMAs (same IEEE fma per component as the scalar form: results are identical)
  const f32x2 t0x = pk_planes_entry(f32x2{qx.x, qx.y}, s.mx, s.ax), t1x = pk_planes_exit(f32x2{qx.z, qx.w}, s.mx, s.ax);
  const f32x2 t0y = pk_planes_entry(f32x2{qy.x, qy.y}, s.my, s.ay), t1y = pk_planes_exit(f32x2{qy.z, qy.w}, s.my, s.ay);
  const f32x2 t0z = pk_planes_entry(f32x2{qz.x, qz.y}, s.mz, s.az), t1z = pk_planes_exit(f32x2{qz.z, qz.w}, s.mz, s.az);
  tl = slab_max3_zero(slab_max(t0x.x, t0y.x), t0z.x);
  tr = slab_max3_zero(slab_max(t0x.y, t0y.y), t0z.y);
  const float fl = slab_min3(slab_min(t1x.x, t1y.x), t1z.x, tmax_w);
  const float fr = slab_min3(slab_min(t1x.y, t1y.y), t1z.y, tmax_w);
  hl = tl <= fl;
  hr = tr <= fr;
}
// 64-B node (l_lo l_hi r_lo r_hi as three float4) -> direction-ordered plane quads by select
__device__ __forceinline__ void slab_select(const SlabRay& s, float4 q0, float4 q1, float4 q2, float4& qx, float4& qy,
                                            float4& qz) {
  qx = s.sx ? float4{q0.w, q2.y, q0.x, q1.z} : float4{q0.x, q1.z, q0.w, q2.y};
  qy = s.sy ? float4{q1.x, q2.z, q0.y, q1.w} : float4{q0.y, q1.w, q1.x, q2.z};
  qz = s.sz ? float4{q1.y, q2.w, q0.z, q2.x} : float4{q0.z, q2.x, q1.y, q2.w};
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4* LdsTablePtr;  // the noise jump table copied into LDS (k_fused)
typedef const u32x4* GlobalTablePtr;

// ----------------------------------------------------------------------------------------------- scene accessors
// Where the traversal data lives.  GlobalAcc / SphereAcc read HBM/L2 through the vector cache (64-B nodes, plane
// order by select); SlabLdsAcc / GenericLdsAcc read a per-workgroup copy of the whole scene in LDS (112-B slab
// nodes, primitive records, material / texture tables and, when the scene has one, the noise jump table).
struct PrimGeom {  // what the sphere path needs
  d3 c0, c1;
  double r;
  uint32_t type, mat;
};
struct GlobalAcc {
  const BvhNode* nodes;
  const Prim* prims;
  static constexpr bool kSpheresOnly = false;
  static constexpr bool kPrimsInHbm = true;
  static constexpr bool kWide = false;
  static constexpr bool kSlabLds = false;  // 112-byte slab nodes in LDS (node_loop_slab16)
  static constexpr bool kWideHalf = false;
  __device__ __forceinline__ GlobalTablePtr noise_table(const Scene& sc) const { return reinterpret_cast<GlobalTablePtr>(sc.noise_table); }
  __device__ __forceinline__ const Material* mats(const Scene& sc) const { return sc.mats; }
  __device__ __forceinline__ const Texture* texs(const Scene& sc) const { return sc.texs; }
  __device__ __forceinline__ int32_t root(int32_t r) const { return r; }
  __device__ __forceinline__ void node_slabs(int32_t i, const SlabRay& s, float4& qx, float4& qy, float4& qz,
                                             int2& ch) const {
    const float4* np = reinterpret_cast<const float4*>(nodes + i);
    slab_select(s, np[0], np[1], np[2], qx, qy, qz);
    ch = *reinterpret_cast<const int2*>(&np[3]);
  }
  __device__ __forceinline__ uint32_t type(int32_t pi) const { return prims[pi].type; }
  __device__ __forceinline__ PrimGeom sphere(int32_t pi) const {
    const Prim& P = prims[pi];
    return {{P.f[0], P.f[1], P.f[2]}, {P.f[3], P.f[4], P.f[5]}, P.f[6], P.type, P.mat};
  }
  __device__ __forceinline__ const Prim& full(int32_t pi) const { return prims[pi]; }
  __device__ __forceinline__ uint32_t mat(int32_t pi) const { return prims[pi].mat; }
};
// spheres-only scene in global memory: dense 64-B nodes + 64-B sphere records
struct SphereAcc {
  const float4* nodes;
  const SphereRec* spheres;
  static constexpr bool kSpheresOnly = true;
  static constexpr bool kPrimsInHbm = true;
  static constexpr bool kWide = false;
  static constexpr bool kSlabLds = false;  // 112-byte slab nodes in LDS (node_loop_slab16)
  static constexpr bool kWideHalf = false;
  __device__ __forceinline__ GlobalTablePtr noise_table(const Scene& sc) const { return reinterpret_cast<GlobalTablePtr>(sc.noise_table); }
  __device__ __forceinline__ const Material* mats(const Scene& sc) const { return sc.mats; }
  __device__ __forceinline__ const Texture* texs(const Scene& sc) const { return sc.texs; }
  __device__ __forceinline__ int32_t root(int32_t r) const { return r; }
  __device__ __forceinline__ void node_slabs(int32_t i, const SlabRay& s, float4& qx, float4& qy, float4& qz,
                                             int2& ch) const {
    const float4* np = nodes + 4 * i;
    slab_select(s, np[0], np[1], np[2], qx, qy, qz);
    float4 q3 = np[3];
    ch = {__float_as_int(q3.x), __float_as_int(q3.y)};
  }
  __device__ __forceinline__ uint32_t type(int32_t pi) const { return spheres[pi].type; }
  __device__ __forceinline__ PrimGeom sphere(int32_t pi) const {
    const SphereRec& S = spheres[pi];
    return {{S.c0[0], S.c0[1], S.c0[2]}, {S.c1[0], S.c1[1], S.c1[2]}, S.r, S.type, S.mat};
  }
  __device__ __forceinline__ const Prim& full(int32_t) const { return *reinterpret_cast<const Prim*>(spheres); }  // never called
  __device__ __forceinline__ uint32_t mat(int32_t pi) const { return spheres[pi].mat; }
};
// Whole spheres-only scene in LDS (k_fused).  Nodes are re-laid out as 112-byte "slab nodes": per axis the four
// planes of the two children are stored twice, once per ray direction, as (entry L, entry R, exit L, exit R), so a
// ray fetches its direction-ordered planes with one ds_read_b128 per axis at base + {ox, oy, oz} and spends no
// per-step selects (12 v_cndmask per step in the 64-B layout):
//   +0  x forward  [lo L, lo R, hi L, hi R]     +16 x backward [hi L, hi R, lo L, lo R]
//   +32 y forward                               +48 y backward
//   +64 z forward                               +80 z backward          +96 children (left, right)   +104 pad
// Inner-node references are pre-scaled to 16-byte units (index * 7) in the child links, the root and the stack,
// so a node's LDS address is ref << 4.
struct SlabLdsAcc {
  const unsigned char* nodes;
  const SphereRec* spheres;
  LdsTablePtr noise;  // LDS copy of the noise jump table (scenes with a noise texture)
  static constexpr bool kSpheresOnly = true;
  static constexpr bool kPrimsInHbm = false;
  static constexpr bool kWide = false;
  static constexpr bool kSlabLds = true;  // 112-byte slab nodes in LDS (node_loop_slab16)
  static constexpr bool kWideHalf = false;
  __device__ __forceinline__ const Material* mats(const Scene& sc) const { return sc.mats; }
  __device__ __forceinline__ const Texture* texs(const Scene& sc) const { return sc.texs; }
  __device__ __forceinline__ LdsTablePtr noise_table(const Scene&) const { return noise; }
  __device__ __forceinline__ int32_t root(int32_t r) const { return r >= 0 ? r * (RT_SLAB_NODE_BYTES / 16) : r; }
  __device__ __forceinline__ void node_slabs(int32_t ref, const SlabRay& s, float4& qx, float4& qy, float4& qz,
                                             int2& ch) const {
    const unsigned char* b = nodes + ((uint32_t)ref << 4);
    qx = *reinterpret_cast<const float4*>(b + s.ox);
    qy = *reinterpret_cast<const float4*>(b + s.oy);
    qz = *reinterpret_cast<const float4*>(b + s.oz);
    ch = *reinterpret_cast<const int2*>(b + 96);
  }
  __device__ __forceinline__ uint32_t type(int32_t pi) const { return spheres[pi].type; }
  __device__ __forceinline__ PrimGeom sphere(int32_t pi) const {
    const SphereRec& S = spheres[pi];
    return {{S.c0[0], S.c0[1], S.c0[2]}, {S.c1[0], S.c1[1], S.c1[2]}, S.r, S.type, S.mat};
  }
  __device__ __forceinline__ const Prim& full(int32_t) const { return *reinterpret_cast<const Prim*>(spheres); }  // never called
  __device__ __forceinline__ uint32_t mat(int32_t pi) const { return spheres[pi].mat; }
};
// Small general scene (quads, media, spheres) in LDS: slab nodes + the full 128-B primitive records.
struct GenericLdsAcc {
  const unsigned char* nodes;
  const Prim* prims;
  LdsTablePtr noise;
  static constexpr bool kSpheresOnly = false;
  static constexpr bool kPrimsInHbm = false;
  static constexpr bool kWide = false;
  static constexpr bool kSlabLds = true;  // 112-byte slab nodes in LDS (node_loop_slab16)
  static constexpr bool kWideHalf = false;
  __device__ __forceinline__ const Material* mats(const Scene& sc) const { return sc.mats; }
  __device__ __forceinline__ const Texture* texs(const Scene& sc) const { return sc.texs; }
  __device__ __forceinline__ LdsTablePtr noise_table(const Scene&) const { return noise; }
  __device__ __forceinline__ int32_t root(int32_t r) const { return r >= 0 ? r * (RT_SLAB_NODE_BYTES / 16) : r; }
  __device__ __forceinline__ void node_slabs(int32_t ref, const SlabRay& s, float4& qx, float4& qy, float4& qz,
                                             int2& ch) const {
    const unsigned char* b = nodes + ((uint32_t)ref << 4);
    qx = *reinterpret_cast<const float4*>(b + s.ox);
    qy = *reinterpret_cast<const float4*>(b + s.oy);
    qz = *reinterpret_cast<const float4*>(b + s.oz);
    ch = *reinterpret_cast<const int2*>(b + 96);
  }
  __device__ __forceinline__ uint32_t type(int32_t pi) const { return prims[pi].type; }
  __device__ __forceinline__ PrimGeom sphere(int32_t pi) const {
    const Prim& P = prims[pi];
    return {{P.f[0], P.f[1], P.f[2]}, {P.f[3], P.f[4], P.f[5]}, P.f[6], P.type, P.mat};
  }
  __device__ __forceinline__ const Prim& full(int32_t pi) const { return prims[pi]; }
  __device__ __forceinline__ uint32_t mat(int32_t pi) const { return prims[pi].mat; }
};
// HBM-resident scenes as k_fused walks them: the four-wide tree (128-B nodes: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4]
// hi.z[4] child[4] pad), primitives and tables as in the base accessor.  Each step is one cache line and one memory
// latency for four boxes; these instances are latency-bound, not issue-bound.
template <class Base>
struct WideAcc : Base {
  const float4* nodes4;
  static constexpr bool kWide = true;
  __device__ __forceinline__ int32_t root(int32_t r) const { return r >= 0 ? 0 : r; }
  // entry (e*) / exit (f*) planes of the four children for the ray's direction, by select
  __device__ __forceinline__ void wide_node(int32_t i, const SlabRay& s, float4& ex, float4& fx, float4& ey, float4& fy,
                                            float4& ez, float4& fz, int4& ch) const {
    const float4* np = nodes4 + 8 * (size_t)i;
    const float4 lox = np[0], loy = np[1], loz = np[2], hix = np[3], hiy = np[4], hiz = np[5];
    ch = *reinterpret_cast<const int4*>(np + 6);
    ex = s.sx ? hix : lox; fx = s.sx ? lox : hix;
    ey = s.sy ? hiy : loy; fy = s.sy ? loy : hiy;
    ez = s.sz ? hiz : loz; fz = s.sz ? loz : hiz;
  }
};
// Four-wide tree with binary16 planes, rounded outward when the scene is uploaded: 64-B nodes
// [lo.x[4] hi.x[4]] [lo.y[4] hi.y[4]] [lo.z[4] hi.z[4]] child[4] — four 16-byte fetches per step instead of seven
// (the HBM-resident instances are bound by the bytes their node fetches pull through the vector cache, DESIGN.md
// section 7).  Nodes are numbered breadth-first and every workgroup copies the first n_cached of them — the top of
// the tree, or all of it — into LDS; a lane's node pointer is then an LDS or a global address and the fetch is a
// FLAT load that the hardware routes per lane.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 half4_to_float4(uint32_t a, uint32_t b) {
  const h16x2 p = __builtin_bit_cast(h16x2, a), q = __builtin_bit_cast(h16x2, b);
  return float4{(float)p.x, (float)p.y, (float)q.x, (float)q.y};
}
template <class Base>
struct WideHalfAcc : Base {
  const uint4* nodes4h;
  const uint4* cache;  // generic pointer into LDS
  int32_t n_cached;
  static constexpr bool kWide = true;
  static constexpr bool kWideHalf = true;  // node_loop_wide16
  __device__ __forceinline__ int32_t root(int32_t r) const { return r >= 0 ? 0 : r; }
  __device__ __forceinline__ void wide_node(int32_t i, const SlabRay& s, float4& ex, float4& fx, float4& ey, float4& fy,
                                            float4& ez, float4& fz, int4& ch) const {
    const uint4* np = (i < n_cached ? cache : nodes4h) + 4 * (size_t)i;
    const uint4 ax = np[0], ay = np[1], az = np[2];
    const uint4 l = np[3];
    ch = int4{(int32_t)l.x, (int32_t)l.y, (int32_t)l.z, (int32_t)l.w};
    ex = half4_to_float4(s.sx ? ax.z : ax.x, s.sx ? ax.w : ax.y); fx = half4_to_float4(s.sx ? ax.x : ax.z, s.sx ? ax.y : ax.w);
    ey = half4_to_float4(s.sy ? ay.z : ay.x, s.sy ? ay.w : ay.y); fy = half4_to_float4(s.sy ? ay.x : ay.z, s.sy ? ay.y : ay.w);
    ez = half4_to_float4(s.sz ? az.z : az.x, s.sz ? az.w : az.y); fz = half4_to_float4(s.sz ? az.x : az.z, s.sz ? az.y : az.w);
  }
};
__device__ __forceinline__ void slab_node_store(unsigned char* l_nodes, int i, const BvhNode& n) {
  float4* w = reinterpret_cast<float4*>(l_nodes + (size_t)i * RT_SLAB_NODE_BYTES);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    w[2 * a] = float4{n.l_lo[a], n.r_lo[a], n.l_hi[a], n.r_hi[a]};
    w[2 * a + 1] = float4{n.l_hi[a], n.r_hi[a], n.l_lo[a], n.r_lo[a]};
  }
  const int32_t k = RT_SLAB_NODE_BYTES / 16;
  const int32_t l = n.left >= 0 ? n.left * k : n.left, r = n.right >= 0 ? n.right * k : n.right;
  w[6] = float4{__int_as_float(l), __int_as_float(r), 0.f, 0.f};
}

// ----------------------------------------------------------------------------------------------- intersection
struct HitInfo {
  double t;
  int32_t prim;  // -1 = miss
};
struct TravCounters {
  uint32_t nodes, prims;  // per-lane node steps / leaf tests
  float wnodes, wprims;   // per-lane share of the wave's loop trips (sums to one per executed trip)
  float t_nodes, t_prims; // per-lane share of the wave's cycles in the inner-node loop / leaf section
  uint32_t t_sph, t_mov, t_quad, t_med, t_box;  // leaf tests by primitive type (static sphere, moving sphere, quad, medium, merged box)
  uint32_t idle_leaf, idle_done, idle_free;  // node-loop trips this lane sat out: waiting at a leaf / finished, waiting for shading / without a path
};
__device__ __forceinline__ uint64_t now_cycles() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ float wave_share() { return 1.0f / (float)__popcll(__ballot(1)); }

// 1/x to ~1 ulp: hardware seed (v_rcp_f64) + two Newton steps; x in the normal range, either sign
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}
// n / d with the iteration of the compiler's own f64 division — v_rcp_f64, two Newton steps on the reciprocal, the
// quotient and one residual correction — without v_div_scale / v_div_fmas / v_div_fixup around it: the same bits as
// n / d whenever no operand or result sits near the ends of the exponent range (the scaling's only job) and d != 0.
// The callers divide by direction components or plane normals of magnitude >= 1e-8.  9 instructions instead of 14.
__device__ __forceinline__ double div_noscale(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double q = n * r;
  return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
// sqrt of a non-negative f64 with the iteration the compiler's own expansion uses (v_rsq_f64 start, two Goldschmidt
// steps, two residual corrections: correctly rounded in practice, the same bits as __builtin_sqrt for x >= 1e-300)
// without its range scaling and class tests: 11 instructions instead of 19.  Arguments below 1e-300 (a discriminant
// that is exactly 0: a tangent ray) give 1e-150 instead of 0.
__device__ __forceinline__ double fast_sqrt_nonneg(double x) {
  asm("v_max_f64 %0, %1, %2" : "=v"(x) : "v"(x), "s"(1e-300));  // (fmax() would canonicalise x first: one more v_max_f64)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return g;
}
// src/hittable/geometry.rs:36-57 — sphere quadratic in f64, evaluated in the reference's operation order
// with contraction off so that `t` is bit-identical to the CPU path for identical inputs.
#pragma clang fp contract(off)
__device__ __forceinline__ d3 sphere_center(const PrimGeom& g, double time) {
  if (g.type == 1u) {  // geometry.rs:108: center*(1-time) + destination*time
    double w0 = 1. - time;
    return {g.c0.x * w0 + g.c1.x * time, g.c0.y * w0 + g.c1.y * time, g.c0.z * w0 + g.c1.z * time};
  }
  return g.c0;
}
// `inv_a` = 1/a is hoisted per ray: root = (-b -/+ sqrt(disc)) * inv_a differs from the reference's division by
// at most one ulp (well inside the 1e-5 parity tolerance) and removes two f64 divisions per candidate.
__device__ __forceinline__ bool sphere_hit(const PrimGeom& g, d3 o, d3 d, double a, double inv_a, double time,
                                           double tmin, double tmax, double& t_out) {
  d3 c = sphere_center(g, time);
  double r = g.r;
  d3 oc = o - c;
  double alignment = oc.x * d.x + oc.y * d.y + oc.z * d.z;
  double surface = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - r * r;
  double disc = alignment * alignment - a * surface;
  if (disc < 0.) return false;
  double sq = fast_sqrt_nonneg(disc);
  double root = (-alignment - sq) * inv_a;
  if (!(tmin < root && root < tmax)) {
    root = (-alignment + sq) * inv_a;
    if (!(tmin < root && root < tmax)) return false;
  }
  t_out = root;
  return true;
}
// EXT quad (SURVEY aX1): plane hit + (alpha, beta) in [0,1]
__device__ __forceinline__ bool quad_hit(const Prim& P, d3 o, d3 d, double tmin, double tmax, double& t_out,
                                         double* ab = nullptr) {
  d3 n = {P.f[9], P.f[10], P.f[11]};
  double denom = n.x * d.x + n.y * d.y + n.z * d.z;
  if (__builtin_fabs(denom) < 1e-8) return false;
  double t = div_noscale(P.f[12] - (n.x * o.x + n.y * o.y + n.z * o.z), denom);  // the true quotient: edge-on rays decide by the last ulp
  if (!(tmin < t && t < tmax)) return false;
  d3 p = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
  d3 ph = {p.x - P.f[0], p.y - P.f[1], p.z - P.f[2]};
  // planar coordinates: alpha = w.(ph x v), beta = w.(u x ph) with w = n_raw / (n_raw.n_raw) in the book's form — 31
  // operations per candidate.  By the triple-product identity they are ph.(v x w) and ph.(w x u), and those two vectors
  // are constants of the quad (rt_scene_upload stores them in place of u and v): six operations.  Equal up to rounding;
  // a hit exactly on a quad's edge may go to the other side of the tie than in the cross-product form.
  double alpha = __builtin_fma(P.f[5], ph.z, __builtin_fma(P.f[4], ph.y, P.f[3] * ph.x));
  double beta = __builtin_fma(P.f[8], ph.z, __builtin_fma(P.f[7], ph.y, P.f[6] * ph.x));
  if (alpha < 0. || alpha > 1. || beta < 0. || beta > 1.) return false;
  t_out = t;
  if (ab) { ab[0] = alpha; ab[1] = beta; }
  return true;
}
// Six quads that form an axis-aligned box, tested as ONE primitive (rt_scene_upload merges them; the quads stay in the
// primitive table and a hit reports the quad of the face that was hit, so shading, uv and object ids are the quad's).
// For an axis-aligned quad quad_hit's t is exactly (plane - o[a]) / d[a]: the unit normal is (+-1, 0, 0) up to order, so
// n.o, n.d and D are single components and the zero terms add exactly — the division below is that same division, and `t`
// is bit-identical to what the face's quad_hit returns.  WHICH face (and whether any) is decided by the slab test — entry
// = latest near plane, exit = earliest far plane, hit iff entry <= exit — on reciprocals good to ~1e-14 instead of by the
// six (alpha, beta) range tests: the two agree except within rounding of an edge, where the quads' own tests tie.  Faces of
// an axis with |d[a]| < 1e-8 cannot be hit (quad_hit's parallel rejection).  Candidates like the quads': the entry face if
// its t is inside (tmin, tmax), else the exit face (ray starts inside the box).
RT_AABOX_INLINE bool aabox_hit(const Prim& P, d3 blo, d3 bhi, d3 o, d3 d, const SlabRay& s, double tmin, double tmax, double& t_out,
                                          int32_t& face_prim) {
  const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  const double lo3[3] = {blo.x, blo.y, blo.z}, hi3[3] = {bhi.x, bhi.y, bhi.z};
  const float r32[3] = {s.mx.x, s.my.x, s.mz.x};  // f32 reciprocals of the direction (make_slab)
  // per axis: distances to the two planes; the nearer is the entry (min), the farther the exit (max) — the same two
  // products the sign-selected form computes, without its eight selects; an axis the ray is parallel to takes no part
  double tn[3], tf[3];
  bool neg[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double lo = lo3[a], hi = hi3[a];
    neg[a] = dd[a] < 0.;
    if (__builtin_fabs(dd[a]) < 1e-8) {
      if (oo[a] < lo || oo[a] > hi) return false;
      tn[a] = -__builtin_inf(); tf[a] = __builtin_inf();
    } else {
      double r = (double)r32[a];
      r = __builtin_fma(__builtin_fma(-dd[a], r, 1.0), r, r);  // one Newton step: ~1e-14 relative
      const double ta = (lo - oo[a]) * r, tb = (hi - oo[a]) * r;
      tn[a] = __builtin_fmin(ta, tb); tf[a] = __builtin_fmax(ta, tb);
    }
  }
  // entry = latest near plane, exit = earliest far plane; ties keep the earlier axis (as the running comparison did)
  int fe = neg[0] ? 1 : 0, fx = neg[0] ? 0 : 1;
  double te = tn[0], tx = tf[0];
  if (tn[1] > te) { te = tn[1]; fe = 2 + (neg[1] ? 1 : 0); }
  if (tf[1] < tx) { tx = tf[1]; fx = 2 + (neg[1] ? 0 : 1); }
  if (tn[2] > te) { te = tn[2]; fe = 4 + (neg[2] ? 1 : 0); }
  if (tf[2] < tx) { tx = tf[2]; fx = 4 + (neg[2] ? 0 : 1); }
  if (!(te <= tx) || te == -__builtin_inf()) return false;  // (te = -inf: every axis parallel — a degenerate direction)
  // exact t of the chosen face: the entry face unless it is not beyond tmin, then the exit face
  int f = fe;
  double t = 0.;
#pragma unroll 1
  for (int k = 0; k < 2; k++) {
    const int a = f >> 1;
    const double plane = (f & 1) ? (a == 0 ? bhi.x : a == 1 ? bhi.y : bhi.z) : (a == 0 ? blo.x : a == 1 ? blo.y : blo.z);
    const double oa = a == 0 ? o.x : a == 1 ? o.y : o.z, da = a == 0 ? d.x : a == 1 ? d.y : d.z;
    t = div_noscale(plane - oa, da);
    if (tmin < t) break;
    f = fx;
  }
  if (!(tmin < t && t < tmax)) return false;
  t_out = t;
  // face[f]: int32 number f of the six packed into f[6..8] (read from the record only now: hits are the minority)
  face_prim = reinterpret_cast<const int32_t*>(&P.f[6])[f];
  return true;
}
#pragma clang fp contract(on)

// EXT constant-density medium with a convex boundary (SURVEY aX3, "The Next Week" §9 in the reference's
// conventions): boundary entry/exit over (-inf, inf) and (t1 + 1e-4, inf); clamp to the query range; scatter
// distance -ln(u)/density in units of |d|.  One RNG draw per test.  Sphere boundary or oriented box
// (corner + three orthogonal edges: an instance-transformed box baked at flatten time).
__device__ __forceinline__ bool medium_hit(const Prim& P, d3 o, d3 d, double a, double inv_a, double tmin, double tmax,
                                           PathRng& rng, double& t_out) {
  double t1, t2;
  // -1/density: f[12]; a spherical medium also carries it in f[3], inside the first 64 bytes of its record
  const double neg_inv_density = P.type == 3u ? P.f[3] : P.f[12];
  if (P.type == 3u) {
    d3 oc = {o.x - P.f[0], o.y - P.f[1], o.z - P.f[2]};
    double hb = dot(oc, d);
    double cc = dot(oc, oc) - P.f[6] * P.f[6];
    double disc = hb * hb - a * cc;
    if (disc < 0.) return false;
    double sq = fast_sqrt_nonneg(disc);
    t1 = (-hb - sq) * inv_a;
    t2 = (-hb + sq) * inv_a;
  } else {
    d3 oq = {o.x - P.f[0], o.y - P.f[1], o.z - P.f[2]};
    t1 = -__builtin_inf();
    t2 = __builtin_inf();
#pragma unroll
    for (int k = 0; k < 3; k++) {
      d3 e = {P.f[3 + 3 * k], P.f[4 + 3 * k], P.f[5 + 3 * k]};  // edge / |edge|^2 (scaled at upload)
      double so = dot(oq, e), sd = dot(d, e);
      if (__builtin_fabs(sd) < 1e-300) {
        if (so < 0. || so > 1.) return false;
      } else {
        const double isd = fast_rcp(sd);
        double ta = (0. - so) * isd, tb = (1. - so) * isd;
        t1 = fmax(t1, fmin(ta, tb));
        t2 = fmin(t2, fmax(ta, tb));
      }
    }
    if (!(t1 <= t2)) return false;
  }
  if (!(t2 > t1 + 0.0001)) return false;  // second boundary hit searched over (t1 + 1e-4, inf)
  t1 = __builtin_fmax(t1, tmin);  // (one v_max_f64 / v_min_f64 each instead of a compare and two selects; t1, t2 are not NaN here)
  t2 = __builtin_fmin(t2, tmax);
  if (t1 >= t2) return false;
  t1 = __builtin_fmax(t1, 0.);
  double len = fast_sqrt_nonneg(a);
  double dist_inside = (t2 - t1) * len;
  // free path -ln(u)/density: u is a 32-bit draw, so the hardware f32 logarithm (v_log_f32, ~1 ulp) loses nothing
  double hit_distance = neg_inv_density * (double)(__builtin_amdgcn_logf(u01f(rng_next(rng))) * 0.69314718056f);
  if (hit_distance > dist_inside) return false;
  t_out = t1 + hit_distance * (len * inv_a);  // / len
  return true;
}

// Resumable traversal state: one ray's walk through the BVH can be advanced one inner-node step or one leaf test
// at a time, so a wave can interleave lanes that are at different stages (k_fused's scheduler) or simply run it
// to completion (traverse()).
#define RT_DONE RT_EMPTY_CHILD
// The "finished" marker of a traversal must survive a round trip through the stack (it sits at the bottom as a
// sentinel): with 16-bit stack entries it is -32768 (leaf references ~prim stay above it: prim < 32 000), with 32-bit
// entries RT_DONE itself.
template <class StackT> struct TravDone { static constexpr int32_t v = sizeof(StackT) == 2 ? -32768 : RT_DONE; };
struct TravState {
  int32_t cur;  // >= 0 inner node, < 0 leaf (~prim), TravDone<StackT>::v finished
  int32_t sp;   // LDS byte address of the lane's next free stack entry (entries below it are held in LDS, the one above them is cached in `top`)
  int32_t top;  // top-of-stack kept in a register: a pop never waits on LDS (its refill load overlaps the next step)
  int32_t best_prim;
  float tmax_w;  // f32 upper bound of best_t, widened: ru(best_t) * RT_SLAB_WIDEN + 1e-30
  double best_t;
  double a, inv_a;  // |d|^2 (geometry.rs:43) and its reciprocal
  SlabRay s;
};
// the part of the state that is a function of the ray alone (recomputed rather than kept across shading)
__device__ __forceinline__ void trav_ray_constants(TravState& ts, d3 o, d3 d) {
  ts.a = d.x * d.x + d.y * d.y + d.z * d.z;
  ts.inv_a = fast_rcp(ts.a);
  ts.s = make_slab(o, d);
}
// the part that records progress: current node, stack, best hit so far
// LDS address <-> pointer (the traversal stacks live in LDS in every kernel; the stack pointer is kept as the 32-bit
// address itself, so that a push or a pop needs no address arithmetic beyond its own increment)
template <class T> __device__ __forceinline__ int32_t lds_addr(T* p) {
  return (int32_t)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) T*)p;
}
template <class T> __device__ __forceinline__ __attribute__((address_space(3))) T* lds_ptr(int32_t a) {
  return (__attribute__((address_space(3))) T*)(uintptr_t)(uint32_t)a;
}
__device__ __forceinline__ void trav_restart(TravState& ts, int32_t root, int32_t done, int32_t stack_addr) {
  ts.cur = root;
  ts.sp = stack_addr;
  ts.top = done;
  ts.best_prim = -1;
  ts.best_t = __builtin_inf();
  ts.tmax_w = __builtin_inff();
}
__device__ __forceinline__ void trav_init(TravState& ts, int32_t root, int32_t done, int32_t stack_addr, d3 o, d3 d) {
  trav_restart(ts, root, done, stack_addr);
  trav_ray_constants(ts, o, d);
}
// Stack with its top entry cached in a register and a sentinel at the bottom.  `sp` is the LDS address of the lane's
// next free entry (row k of the lane's column = entry k, rows `stride` entries apart); the register `top` is the entry
// above the stored ones.  A fresh traversal starts with
// sp = the column's first row and top = the "finished" marker, so the first push parks that marker in row 0 and the pop that empties the
// stack brings it back as the current node: no emptiness test anywhere.  A pop from the empty stack (the traversal's last) reads
// the row BELOW the column's first — the kernels keep readable LDS there (k_fused: the tile accumulators; the probe
// kernels: a guard row) — and leaves a dead `top` that the next restart overwrites.
// Push: one LDS store + one add; pop: one add + one LDS load — no multiply by the row stride (768-thread workgroups:
// a quarter-rate v_mad_u64_u32 per push before) and no shift-add.
template <class StackT>
__device__ __forceinline__ void trav_push(TravState& ts, int32_t v, StackT*, int stride) {
  *lds_ptr<StackT>(ts.sp) = (StackT)ts.top;
  ts.top = v;
  ts.sp += stride * (int)sizeof(StackT);
}
template <class StackT>
__device__ __forceinline__ void trav_pop(TravState& ts, StackT*, int stride) {
  ts.cur = ts.top;
  ts.sp -= stride * (int)sizeof(StackT);
  ts.top = (int32_t)*lds_ptr<StackT>(ts.sp);
}
// One step through a four-wide node: test the four boxes, continue with the nearest one hit and push the others
// farthest first (5-exchange sorting network on (entry distance, link) pairs).  Unused slots carry an inverted
// (+inf, -inf) box, which no ray with a finite non-zero direction component enters, and a copy of slot 0's link, so
// that even a degenerate ray only repeats work: the step has no test for them.
template <class Acc, class StackT>
__device__ __forceinline__ void node_step_wide(const Acc& acc, TravState& ts, StackT* stack, int stride) {
  const SlabRay& s = ts.s;
  float4 ex, fx, ey, fy, ez, fz;  // entry / exit planes for the ray's direction
  int4 ch;
  acc.wide_node(ts.cur, s, ex, fx, ey, fy, ez, fz, ch);
  const float INF = __builtin_inff();
  float t[4];
  int32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
  // (scalar FMAs here: the HBM-resident instances that walk wide nodes are bound by fetch latency and short of
  // registers — packed pairs cost them spills: C5 69 -> 75 ms)
  const float ix = s.mx.x, iy = s.my.x, iz = s.mz.x, ixw = s.mx.y, iyw = s.my.y, izw = s.mz.y;
  const float nx = s.ax.x, ny = s.ay.x, nz = s.az.x, ox = s.ax.y, oy = s.ay.y, oz = s.az.y;
#define RT_WIDE_SLOT(k, m)                                                                                          \
  {                                                                                                                 \
    const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(ex.m, ix, -nx), __builtin_fmaf(ey.m, iy, -ny)),               \
                                 __builtin_fmaf(ez.m, iz, -nz)), 0.f);                                              \
    const float tf = fminf(fminf(fminf(__builtin_fmaf(fx.m, ixw, -ox), __builtin_fmaf(fy.m, iyw, -oy)),             \
                                 __builtin_fmaf(fz.m, izw, -oz)), ts.tmax_w);                                       \
    t[k] = tn <= tf ? tn : INF;                                                                                     \
  }
  RT_WIDE_SLOT(0, x) RT_WIDE_SLOT(1, y) RT_WIDE_SLOT(2, z) RT_WIDE_SLOT(3, w)
#undef RT_WIDE_SLOT
#define RT_WIDE_CAS(i, j)                                  \
  {                                                        \
    const bool sw = t[j] < t[i];                           \
    const float ti = t[i], tj = t[j];                      \
    const int32_t ci = c[i], cj = c[j];                    \
    t[i] = sw ? tj : ti; t[j] = sw ? ti : tj;              \
    c[i] = sw ? cj : ci; c[j] = sw ? ci : cj;              \
  }
  RT_WIDE_CAS(0, 1) RT_WIDE_CAS(2, 3) RT_WIDE_CAS(0, 2) RT_WIDE_CAS(1, 3) RT_WIDE_CAS(1, 2)
#undef RT_WIDE_CAS
  if (t[3] < INF) trav_push(ts, c[3], stack, stride);
  if (t[2] < INF) trav_push(ts, c[2], stack, stride);
  if (t[1] < INF) trav_push(ts, c[1], stack, stride);
  if (t[0] < INF) ts.cur = c[0];
  else trav_pop(ts, stack, stride);
}
// precondition: ts.cur >= 0
template <class Acc, class StackT>
__device__ __forceinline__ void node_step(const Acc& acc, TravState& ts, StackT* stack, int stride) {
  if constexpr (Acc::kWide) {
    node_step_wide(acc, ts, stack, stride);
  } else {
    float4 qx, qy, qz;
    int2 ch;
    acc.node_slabs(ts.cur, ts.s, qx, qy, qz, ch);
    float tl, tr;
    bool hl, hr;
    slab_pair_test(ts.s, qx, qy, qz, ts.tmax_w, tl, tr, hl, hr);
    const bool left_first = tl <= tr;
    // bitwise on purpose: lane masks combined on the scalar unit and ONE select (the short-circuit form compiled to two
    // nested selects); when neither child is hit the pop overwrites `cur`
    const bool take_right = hr & (!hl | !left_first);
    if (hl && hr) trav_push(ts, left_first ? ch.y : ch.x, stack, stride);
    ts.cur = take_right ? ch.y : ch.x;
    if (!hl && !hr) trav_pop(ts, stack, stride);
  }
}
// The inner-node loop of the LDS-resident slab-node traversals (SlabLdsAcc / GenericLdsAcc, 16-bit stacks) as ONE asm
// statement: `while (at least `need` lanes are at an inner node) those lanes take a node_step`.  Same instructions on
// the same operands as node_step() — results are bit-identical — but the loop-carried values (cur, sp, top) stay in the
// registers they live in.  Compiled from the C++ form, every step carried six v_mov of loop-exit copies (cur, sp and top
// each existed twice) and waited for the pop's refill load before the next node's fetches were issued: 33 VALU
// instructions per step, 27 here, and the refill overlaps the next fetch.  The kernel's time IS its VALU instruction
// count, and this loop is half of it.
// Temporaries are v80..v93 by name (the plane quads' halves must be addressable as register pairs, which asm operands
// are not): qx v[80:83], qy v[84:87], qz v[88:91], child links / node address v[92:93].
// Hazards (no assembler help inside asm): every VALU read of a VALU-written SGPR pair is >= 2 instructions behind its
// v_cmp (gfx940 rule); LDS returns in order, so lgkmcnt(3..0) after the four fetches also covers the push / refill of
// the step before; exec is restored before the back edge and on exit.
#ifndef RT_ASM_NODE_LOOP
#define RT_ASM_NODE_LOOP 1
#endif
// (the sphere-only wide instances run at 96 VGPRs beside the LDS-resident kernel — the partner workgroups — and have no
// room for the loop's 28 named registers: 18 spilled VGPRs, C2 +0.7 %; they keep the compiled loop)
#ifndef RT_ASM_WIDE_FOR
#define RT_ASM_WIDE_FOR(Acc) (!Acc::kSpheresOnly)
#endif
#ifndef RT_ASM_TRAVERSAL
#define RT_ASM_TRAVERSAL 1
#endif
template <class Acc>
__device__ __forceinline__ void node_loop_slab16(const Acc& acc, TravState& ts, int stride_bytes, uint32_t need) {
  const uint32_t base = (uint32_t)lds_addr(acc.nodes);
  unsigned long long m_walk, m_save, m_hr, m_lf, m_t;
  uint32_t n_walk;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n"
      "1:\n"
      "v_cmp_lt_i32_e64 %[mw], -1, %[cur]\n"
      "s_bcnt1_i32_b64 %[nw], %[mw]\n"
      "s_cmp_lt_u32 %[nw], %[need]\n"
      "s_cbranch_scc1 2f\n"
      "s_and_saveexec_b64 %[sv], %[mw]\n"
      "v_lshl_add_u32 v92, %[cur], 4, %[base]\n"
      "v_add_u32_e32 v93, %[ox], v92\n"
      "ds_read_b128 v[80:83], v93\n"
      "v_add_u32_e32 v93, %[oy], v92\n"
      "ds_read_b128 v[84:87], v93\n"
      "v_add_u32_e32 v93, %[oz], v92\n"
      "ds_read_b128 v[88:91], v93\n"
      "ds_read_b64 v[92:93], v92 offset:96\n"
      "s_waitcnt lgkmcnt(3)\n"
      "v_pk_fma_f32 v[80:81], v[80:81], %[mx], %[ax] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_pk_fma_f32 v[82:83], v[82:83], %[mx], %[ax] op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_pk_fma_f32 v[84:85], v[84:85], %[my], %[ay] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_pk_fma_f32 v[86:87], v[86:87], %[my], %[ay] op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_pk_fma_f32 v[88:89], v[88:89], %[mz], %[az] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_pk_fma_f32 v[90:91], v[90:91], %[mz], %[az] op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_max_f32_e32 v80, v80, v84\n"   // entry distances: left child in v80, right in v81
      "v_max_f32_e32 v81, v81, v85\n"
      "v_min_f32_e32 v82, v82, v86\n"   // exit distances: left v82, right v83
      "v_min_f32_e32 v83, v83, v87\n"
      "v_max3_f32 v80, v80, v88, 0\n"
      "v_max3_f32 v81, v81, v89, 0\n"
      "v_min3_f32 v82, v82, v90, %[tmax]\n"
      "v_min3_f32 v83, v83, v91, %[tmax]\n"
      "v_cmp_le_f32_e32 vcc, v80, v82\n"         // hl
      "v_cmp_le_f32_e64 %[mb], v81, v83\n"       // hr
      "v_cmp_le_f32_e64 %[mc], v80, v81\n"       // left_first
      "s_waitcnt lgkmcnt(0)\n"
      "s_and_b64 %[mt], vcc, %[mb]\n"            // both hit: push the farther child
      "s_mov_b64 exec, %[mt]\n"
      "ds_write_b16 %[sp], %[top]\n"
      "v_add_u32_e32 %[sp], %[stride], %[sp]\n"
      "v_cndmask_b32_e64 %[top], v92, v93, %[mc]\n"
      "s_and_b64 %[mt], vcc, %[mc]\n"
      "s_andn2_b64 %[mt], %[mb], %[mt]\n"        // take_right = hr & !(hl & left_first)
      "s_mov_b64 exec, %[mw]\n"
      "v_cndmask_b32_e64 %[cur], v92, v93, %[mt]\n"
      "s_or_b64 %[mt], vcc, %[mb]\n"
      "s_andn2_b64 exec, %[mw], %[mt]\n"         // neither hit: pop
      "v_mov_b32_e32 %[cur], %[top]\n"
      "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n"
      "ds_read_i16 %[top], %[sp]\n"
      "s_mov_b64 exec, %[sv]\n"
      "s_branch 1b\n"
      "2:\n"
      "s_waitcnt lgkmcnt(0)\n"
      : [cur] "+v"(ts.cur), [sp] "+v"(ts.sp), [top] "+v"(ts.top), [mw] "=&s"(m_walk), [sv] "=&s"(m_save), [mb] "=&s"(m_hr),
        [mc] "=&s"(m_lf), [mt] "=&s"(m_t), [nw] "=&s"(n_walk)
      : [need] "s"(need), [base] "s"(base), [stride] "s"(stride_bytes), [ox] "v"(ts.s.ox), [oy] "v"(ts.s.oy), [oz] "v"(ts.s.oz),
        [mx] "v"(ts.s.mx), [my] "v"(ts.s.my), [mz] "v"(ts.s.mz), [ax] "v"(ts.s.ax), [ay] "v"(ts.s.ay), [az] "v"(ts.s.az),
        [tmax] "v"(ts.tmax_w)
      : "vcc", "scc", "memory", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93");
}
// The inner-node loop of the four-wide binary16 traversals (WideHalfAcc, 16-bit stacks: the HBM-resident render
// instances and the partner workgroups) as one asm statement, like node_loop_slab16: the operations of node_step_wide()
// on the same operands in the same order (same v_fma_mix_f32 / v_max3 / v_min3 per plane, the same five-exchange
// network with strict `<` swaps, pushes farthest first) — results are bit-identical — with what the compiled form spends
// around them removed: the LDS-or-global node address as a generic pointer (10 instructions -> 2: cached lanes read with
// ds_read, the others with global_load off a scalar base), two register copies per push and three per pop, the copies
// between the loop levels.  ~101 VALU per step instead of ~123.
// Named registers: v60..v82 — node words A v[60:63] B v[64:67] C v[68:71] (x / y / z: lo01 lo23 hi01 hi23), links
// v[72:75], entry planes by direction v[76:81] (exit planes stay in A / B / C's upper halves; v76 first holds the node's
// address), +inf v82; the low halves of A / B / C hold the entry distances and scratch once the planes are selected, and
// the sorting network renames into whatever is free.  s80..s98: masks, count.
template <class Acc>
__device__ __forceinline__ void node_loop_wide16(const Acc& acc, TravState& ts, int stride_bytes, uint32_t need) {
  const uint32_t cache_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lds_addr(acc.cache));
  const uint32_t n_cached = (uint32_t)__builtin_amdgcn_readfirstlane(acc.n_cached);
  need = (uint32_t)__builtin_amdgcn_readfirstlane((int)need);
  // (by value: the asm's inputs are twelve scalars, not a view of the aggregate)
  const float c_ix = ts.s.mx.x, c_ixw = ts.s.mx.y, c_iy = ts.s.my.x, c_iyw = ts.s.my.y, c_iz = ts.s.mz.x, c_izw = ts.s.mz.y;
  const float c_nx = ts.s.ax.x, c_ox = ts.s.ax.y, c_ny = ts.s.ay.x, c_oy = ts.s.ay.y, c_nz = ts.s.az.x, c_oz = ts.s.az.y;
// one child: entry distance -> TN, exit distance -> v68, hit mask (entry <= exit) -> HIT; v69 is scratch
#define RT_WIDE_CHILD(TN, HIT, EX, EY, EZ, FX, FY, FZ, SEL)                                                      \
  "v_fma_mix_f32 " TN ", " EX ", %[ix], -%[nx] " SEL "\n"                                                      \
  "v_fma_mix_f32 v69, " EY ", %[iy], -%[ny] " SEL "\n"                                                         \
  "v_max_f32_e32 " TN ", " TN ", v69\n"                                                                        \
  "v_fma_mix_f32 v69, " EZ ", %[iz], -%[nz] " SEL "\n"                                                         \
  "v_max3_f32 " TN ", " TN ", v69, 0\n"                                                                        \
  "v_fma_mix_f32 v68, " FX ", %[ixw], -%[ox] " SEL "\n"                                                        \
  "v_fma_mix_f32 v69, " FY ", %[iyw], -%[oy] " SEL "\n"                                                        \
  "v_min_f32_e32 v68, v68, v69\n"                                                                              \
  "v_fma_mix_f32 v69, " FZ ", %[izw], -%[oz] " SEL "\n"                                                        \
  "v_min3_f32 v68, v68, v69, %[tmax]\n"                                                                        \
  "v_cmp_le_f32_e64 " HIT ", " TN ", v68\n"
// compare-exchange of (distance, link) pairs: swap iff TJ < TI (strictly), results renamed into fresh registers
#define RT_WIDE_CAS(M, TI, TJ, CI, CJ, TLO, THI, CLO, CHI) \
  "v_cmp_lt_f32_e64 " M ", " TJ ", " TI "\n"                \
  "v_min_f32_e32 " TLO ", " TI ", " TJ "\n"                 \
  "v_max_f32_e32 " THI ", " TI ", " TJ "\n"                 \
  "v_cndmask_b32_e64 " CLO ", " CI ", " CJ ", " M "\n"      \
  "v_cndmask_b32_e64 " CHI ", " CJ ", " CI ", " M "\n"
  asm volatile(
      "v_cmp_gt_f32_e64 s[84:85], 0, %[ix]\n"  // direction negative on the axis (SlabRay::sx, sy, sz)
      "v_cmp_gt_f32_e64 s[86:87], 0, %[iy]\n"
      "v_cmp_gt_f32_e64 s[88:89], 0, %[iz]\n"
      "v_mov_b32_e32 v82, 0x7f800000\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      "1:\n"
      "v_cmp_lt_i32_e64 s[80:81], -1, %[cur]\n"
      "s_bcnt1_i32_b64 s98, s[80:81]\n"
      "s_cmp_lt_u32 s98, %[need]\n"
      "s_cbranch_scc1 9f\n"
      "s_and_saveexec_b64 s[82:83], s[80:81]\n"
      // ---- the node's 64 bytes: from the LDS copy of the top of the tree, else from memory
      "v_cmp_gt_i32_e32 vcc, %[ncached], %[cur]\n"
      "s_and_b64 exec, s[80:81], vcc\n"
      "v_lshl_add_u32 v76, %[cur], 6, %[cbase]\n"
      "ds_read_b128 v[60:63], v76\n"
      "ds_read_b128 v[64:67], v76 offset:16\n"
      "ds_read_b128 v[68:71], v76 offset:32\n"
      "ds_read_b128 v[72:75], v76 offset:48\n"
      "s_andn2_b64 exec, s[80:81], vcc\n"
      "s_cbranch_execz 2f\n"
      "v_lshlrev_b32_e32 v76, 6, %[cur]\n"
      "global_load_dwordx4 v[60:63], v76, %[nodes]\n"
      "global_load_dwordx4 v[64:67], v76, %[nodes] offset:16\n"
      "global_load_dwordx4 v[68:71], v76, %[nodes] offset:32\n"
      "global_load_dwordx4 v[72:75], v76, %[nodes] offset:48\n"
      "2:\n"
      "s_mov_b64 exec, s[80:81]\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
      // ---- entry / exit planes for the ray's direction
      "v_cndmask_b32_e64 v76, v60, v62, s[84:85]\n"
      "v_cndmask_b32_e64 v77, v61, v63, s[84:85]\n"
      "v_cndmask_b32_e64 v62, v62, v60, s[84:85]\n"
      "v_cndmask_b32_e64 v63, v63, v61, s[84:85]\n"
      "v_cndmask_b32_e64 v78, v64, v66, s[86:87]\n"
      "v_cndmask_b32_e64 v79, v65, v67, s[86:87]\n"
      "v_cndmask_b32_e64 v66, v66, v64, s[86:87]\n"
      "v_cndmask_b32_e64 v67, v67, v65, s[86:87]\n"
      "v_cndmask_b32_e64 v80, v68, v70, s[88:89]\n"
      "v_cndmask_b32_e64 v81, v69, v71, s[88:89]\n"
      "v_cndmask_b32_e64 v70, v70, v68, s[88:89]\n"
      "v_cndmask_b32_e64 v71, v71, v69, s[88:89]\n"
      // ---- four slab tests: entry distances v60 v61 v64 v65 (the node words' low halves are free now), hit masks
      RT_WIDE_CHILD("v60", "vcc", "v76", "v78", "v80", "v62", "v66", "v70", "op_sel_hi:[1,0,0]")
      RT_WIDE_CHILD("v61", "s[90:91]", "v76", "v78", "v80", "v62", "v66", "v70", "op_sel:[1,0,0] op_sel_hi:[1,0,0]")
      RT_WIDE_CHILD("v64", "s[92:93]", "v77", "v79", "v81", "v63", "v67", "v71", "op_sel_hi:[1,0,0]")
      RT_WIDE_CHILD("v65", "s[94:95]", "v77", "v79", "v81", "v63", "v67", "v71", "op_sel:[1,0,0] op_sel_hi:[1,0,0]")
      "v_cndmask_b32_e32 v60, v82, v60, vcc\n"         // t[k] = hit ? entry distance : +inf
      "v_cndmask_b32_e64 v61, v82, v61, s[90:91]\n"
      "v_cndmask_b32_e64 v64, v82, v64, s[92:93]\n"
      "v_cndmask_b32_e64 v65, v82, v65, s[94:95]\n"
      // ---- sort by distance: (0,1) (2,3) (0,2) (1,3) (1,2), every result in a register that is free by then
      RT_WIDE_CAS("s[90:91]", "v60", "v61", "v72", "v73", "v62", "v63", "v76", "v77")   // -> t v62 v63, c v76 v77
      RT_WIDE_CAS("s[92:93]", "v64", "v65", "v74", "v75", "v66", "v67", "v78", "v79")   // -> t v66 v67, c v78 v79
      RT_WIDE_CAS("s[90:91]", "v62", "v66", "v76", "v78", "v60", "v64", "v72", "v74")   // (0,2) -> t v60 v64, c v72 v74
      RT_WIDE_CAS("s[92:93]", "v63", "v67", "v77", "v79", "v61", "v65", "v73", "v75")   // (1,3) -> t v61 v65, c v73 v75
      RT_WIDE_CAS("s[90:91]", "v61", "v64", "v73", "v74", "v62", "v63", "v76", "v77")   // (1,2) -> t v62 v63, c v76 v77
      // sorted: t = v60 v62 v63 v65, links = v72 v76 v77 v75
      "v_cmp_lt_f32_e64 s[90:91], v65, v82\n"
      "v_cmp_lt_f32_e64 s[92:93], v63, v82\n"
      "v_cmp_lt_f32_e64 s[94:95], v62, v82\n"
      "v_cmp_lt_f32_e32 vcc, v60, v82\n"
      "s_mov_b64 exec, s[90:91]\n"                     // push the hit children, farthest first
      "ds_write_b16 %[sp], %[top]\n"
      "v_add_u32_e32 %[sp], %[stride], %[sp]\n"
      "v_mov_b32_e32 %[top], v75\n"
      "s_mov_b64 exec, s[92:93]\n"
      "ds_write_b16 %[sp], %[top]\n"
      "v_add_u32_e32 %[sp], %[stride], %[sp]\n"
      "v_mov_b32_e32 %[top], v77\n"
      "s_mov_b64 exec, s[94:95]\n"
      "ds_write_b16 %[sp], %[top]\n"
      "v_add_u32_e32 %[sp], %[stride], %[sp]\n"
      "v_mov_b32_e32 %[top], v76\n"
      "s_mov_b64 exec, s[80:81]\n"
      "v_mov_b32_e32 %[cur], v72\n"                    // continue with the nearest ...
      "s_andn2_b64 exec, s[80:81], vcc\n"
      "v_mov_b32_e32 %[cur], %[top]\n"                 // ... or pop when nothing was hit
      "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n"
      "ds_read_i16 %[top], %[sp]\n"
      "s_mov_b64 exec, s[82:83]\n"
      "s_branch 1b\n"
      "9:\n"
      "s_waitcnt lgkmcnt(0)\n"
      : [cur] "+v"(ts.cur), [sp] "+v"(ts.sp), [top] "+v"(ts.top)
      : [need] "s"(need), [stride] "s"(stride_bytes), [ncached] "s"(n_cached), [cbase] "s"(cache_base), [nodes] "s"(acc.nodes4h),
        [tmax] "v"(ts.tmax_w), [ix] "v"(c_ix), [ixw] "v"(c_ixw), [iy] "v"(c_iy), [iyw] "v"(c_iyw), [iz] "v"(c_iz),
        [izw] "v"(c_izw), [nx] "v"(c_nx), [ox] "v"(c_ox), [ny] "v"(c_ny), [oy] "v"(c_oy), [nz] "v"(c_nz), [oz] "v"(c_oz)
      : "vcc", "scc", "memory", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74",
        "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89",
        "s90", "s91", "s92", "s93", "s94", "s95", "s98");
#undef RT_WIDE_CHILD
#undef RT_WIDE_CAS
}
// The WHOLE resumable traversal of a spheres-only LDS-resident scene (SlabLdsAcc, 16-bit stack) as one asm statement:
// traverse_resume()'s loop — busy / at-node ballots, the inner-node loop above, the f64 sphere test of
// sphere_hit() for the lanes at a leaf, the pop — with the same operations in the same order (every f64 operation is
// the one the compiler emits for the C++ form, contraction off: frames stay bit-identical to the C++ traversal, which
// the partner workgroups and the parity probes still run), but without the copies the compiler keeps between the two
// loop levels (cur, sp, top, best_t each lived in two registers: ~13 v_mov per leaf round) and with the sphere test's
// temporaries in place.
// Named registers: VGPR temporaries v66..v93, SGPR temporaries s80..s97.
//   node step   qx v[80:83]  qy v[84:87]  qz v[88:91]  links / address v[92:93]
//   leaf test   centre -> oc v[66:71]   r -> r^2 -> a*surface v[72:73]   type v74   prim v75   address v76
//               alignment v[78:79]   products / root2 v[80:81]   disc v[84:85]   sqrt g v[86:87] h v[88:89] tmp v[90:91]   root v[92:93]
//   s[80:81] at-node mask   s[82:83] saved exec (step / leaf)   s[84:85] s[86:87] s[88:89] lane masks   s90 count   s91 need
//   s92 node base   s93 sphere base   s[94:95] tmin   s[96:97] 1e-300   s98 the "finished" marker
// Hazards handled by hand: s_nop after v_rsq_f64 (gfx940 trans forwarding), >= 2 instructions between a v_cmp and a
// v_cndmask that reads its mask, LDS completes in order (lgkmcnt arithmetic as in node_loop_slab16).
template <int STRIDE_BYTES>
__device__ __forceinline__ void trav_resume_spheres_asm(const SlabLdsAcc& acc, TravState& ts, d3 o, d3 d, double time, uint32_t min_busy,
                                                        uint32_t kn) {
  // (wave-uniform by construction; said explicitly so that the "s" operands are SGPRs in every instantiation)
  const uint32_t bases = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)((uint32_t)lds_addr(acc.nodes) | ((uint32_t)lds_addr(acc.spheres) << 16)));  // LDS addresses < 64 Ki
  min_busy = (uint32_t)__builtin_amdgcn_readfirstlane((int)min_busy);
  kn = (uint32_t)__builtin_amdgcn_readfirstlane((int)kn);
  const double omt = 1. - time;  // geometry.rs:108
  asm volatile(
      "s_and_b32 s92, %[bases], 0xffff\n"
      "s_lshr_b32 s93, %[bases], 16\n"
      "s_mov_b32 s94, 0xa0b5ed8d\n"  // RT_TMIN = 1e-6
      "s_mov_b32 s95, 0x3eb0c6f7\n"
      "s_mov_b32 s96, 0xc2f8f359\n"  // 1e-300
      "s_mov_b32 s97, 0x01a56e1f\n"
      "s_mov_b32 s98, 0xffff8000\n"  // TravDone<int16_t>::v
      "s_waitcnt lgkmcnt(0)\n"
      // ---- while at least min_busy lanes are unfinished
      "10:\n"
      "v_cmp_ne_u32_e32 vcc, 0xffff8000, %[cur]\n"
      "s_bcnt1_i32_b64 s90, vcc\n"
      "s_cmp_lt_u32 s90, %[minbusy]\n"
      "s_cbranch_scc1 19f\n"
      "v_cmp_lt_i32_e64 s[80:81], -1, %[cur]\n"
      "s_cmp_lg_u64 s[80:81], vcc\n"
      "s_cselect_b32 s91, %[kn], 1\n"  // leave the node loop early only while some lane waits at a leaf
      "s_branch 12f\n"
      // ---- inner-node loop (node_loop_slab16)
      "11:\n"
      "v_cmp_lt_i32_e64 s[80:81], -1, %[cur]\n"
      "12:\n"
      "s_bcnt1_i32_b64 s90, s[80:81]\n"
      "s_cmp_lt_u32 s90, s91\n"
      "s_cbranch_scc1 13f\n"
      "s_and_saveexec_b64 s[82:83], s[80:81]\n"
      "v_lshl_add_u32 v92, %[cur], 4, s92\n"
      "v_add_u32_e32 v93, %[ox], v92\n"
      "ds_read_b128 v[80:83], v93\n"
      "v_add_u32_e32 v93, %[oy], v92\n"
      "ds_read_b128 v[84:87], v93\n"
      "v_add_u32_e32 v93, %[oz], v92\n"
      "ds_read_b128 v[88:91], v93\n"
      "ds_read_b64 v[92:93], v92 offset:96\n"
      "s_waitcnt lgkmcnt(3)\n"
      "v_pk_fma_f32 v[80:81], v[80:81], %[mx], %[ax] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_pk_fma_f32 v[82:83], v[82:83], %[mx], %[ax] op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "s_waitcnt lgkmcnt(2)\n"
      "v_pk_fma_f32 v[84:85], v[84:85], %[my], %[ay] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_pk_fma_f32 v[86:87], v[86:87], %[my], %[ay] op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "s_waitcnt lgkmcnt(1)\n"
      "v_pk_fma_f32 v[88:89], v[88:89], %[mz], %[az] op_sel_hi:[1,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_pk_fma_f32 v[90:91], v[90:91], %[mz], %[az] op_sel:[0,1,1] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n"
      "v_max_f32_e32 v80, v80, v84\n"
      "v_max_f32_e32 v81, v81, v85\n"
      "v_min_f32_e32 v82, v82, v86\n"
      "v_min_f32_e32 v83, v83, v87\n"
      "v_max3_f32 v80, v80, v88, 0\n"
      "v_max3_f32 v81, v81, v89, 0\n"
      "v_min3_f32 v82, v82, v90, %[tmax]\n"
      "v_min3_f32 v83, v83, v91, %[tmax]\n"
      "v_cmp_le_f32_e32 vcc, v80, v82\n"
      "v_cmp_le_f32_e64 s[84:85], v81, v83\n"
      "v_cmp_le_f32_e64 s[86:87], v80, v81\n"
      "s_waitcnt lgkmcnt(0)\n"
      "s_and_b64 s[88:89], vcc, s[84:85]\n"
      "s_mov_b64 exec, s[88:89]\n"
      "ds_write_b16 %[sp], %[top]\n"
      "v_add_u32_e32 %[sp], %[stride], %[sp]\n"
      "v_cndmask_b32_e64 %[top], v92, v93, s[86:87]\n"
      "s_and_b64 s[88:89], vcc, s[86:87]\n"
      "s_andn2_b64 s[88:89], s[84:85], s[88:89]\n"
      "s_mov_b64 exec, s[80:81]\n"
      "v_cndmask_b32_e64 %[cur], v92, v93, s[88:89]\n"
      "s_or_b64 s[88:89], vcc, s[84:85]\n"
      "s_andn2_b64 exec, s[80:81], s[88:89]\n"
      "v_mov_b32_e32 %[cur], %[top]\n"
      "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n"
      "ds_read_i16 %[top], %[sp]\n"
      "s_mov_b64 exec, s[82:83]\n"
      "s_branch 11b\n"
      // ---- lanes at a leaf: sphere_hit() + pop
      "13:\n"
      "v_cmp_gt_i32_e32 vcc, 0, %[cur]\n"
      "v_cmp_ne_u32_e64 s[84:85], s98, %[cur]\n"
      "s_and_b64 s[84:85], vcc, s[84:85]\n"
      "s_and_saveexec_b64 s[82:83], s[84:85]\n"
      "s_cbranch_execz 18f\n"
      "v_not_b32_e32 v75, %[cur]\n"
      "v_lshl_add_u32 v76, v75, 6, s93\n"
      "ds_read_b128 v[66:69], v76\n"
      "ds_read2_b64 v[70:73], v76 offset0:2 offset1:6\n"
      "ds_read_b32 v74, v76 offset:56\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_cmp_eq_u32_e32 vcc, 1, v74\n"
      "s_and_saveexec_b64 s[86:87], vcc\n"
      "s_cbranch_execz 14f\n"
      "ds_read2_b64 v[80:83], v76 offset0:3 offset1:4\n"
      "ds_read_b64 v[84:85], v76 offset:40\n"
      "v_mul_f64 v[66:67], v[66:67], %[omt]\n"
      "v_mul_f64 v[68:69], v[68:69], %[omt]\n"
      "v_mul_f64 v[70:71], v[70:71], %[omt]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_mul_f64 v[80:81], v[80:81], %[time]\n"
      "v_mul_f64 v[82:83], v[82:83], %[time]\n"
      "v_mul_f64 v[84:85], v[84:85], %[time]\n"
      "v_add_f64 v[66:67], v[66:67], v[80:81]\n"
      "v_add_f64 v[68:69], v[68:69], v[82:83]\n"
      "v_add_f64 v[70:71], v[70:71], v[84:85]\n"
      "14:\n"
      "s_or_b64 exec, exec, s[86:87]\n"
      "v_add_f64 v[66:67], %[o_x], -v[66:67]\n"
      "v_add_f64 v[68:69], %[o_y], -v[68:69]\n"
      "v_add_f64 v[70:71], %[o_z], -v[70:71]\n"
      "v_mul_f64 v[78:79], %[d_x], v[66:67]\n"
      "v_mul_f64 v[80:81], %[d_y], v[68:69]\n"
      "v_mul_f64 v[66:67], v[66:67], v[66:67]\n"
      "v_mul_f64 v[68:69], v[68:69], v[68:69]\n"
      "v_add_f64 v[66:67], v[66:67], v[68:69]\n"
      "v_mul_f64 v[68:69], v[70:71], v[70:71]\n"
      "v_add_f64 v[78:79], v[78:79], v[80:81]\n"
      "v_mul_f64 v[80:81], %[d_z], v[70:71]\n"
      "v_add_f64 v[66:67], v[66:67], v[68:69]\n"
      "v_mul_f64 v[72:73], v[72:73], v[72:73]\n"
      "v_add_f64 v[78:79], v[78:79], v[80:81]\n"      // alignment
      "v_add_f64 v[72:73], v[66:67], -v[72:73]\n"     // surface
      "v_mul_f64 v[66:67], v[78:79], v[78:79]\n"
      "v_mul_f64 v[72:73], %[a], v[72:73]\n"
      "v_add_f64 v[84:85], v[66:67], -v[72:73]\n"     // discriminant
      "v_cmp_ngt_f64_e32 vcc, 0, v[84:85]\n"
      "s_and_saveexec_b64 s[86:87], vcc\n"
      "s_cbranch_execz 17f\n"
      "v_max_f64 v[84:85], v[84:85], s[96:97]\n"      // fast_sqrt_nonneg()
      "v_rsq_f64_e32 v[88:89], v[84:85]\n"
      "s_nop 0\n"
      "v_mul_f64 v[86:87], v[84:85], v[88:89]\n"
      "v_mul_f64 v[88:89], v[88:89], 0.5\n"
      "v_fma_f64 v[90:91], -v[88:89], v[86:87], 0.5\n"
      "v_fma_f64 v[86:87], v[86:87], v[90:91], v[86:87]\n"
      "v_fma_f64 v[88:89], v[88:89], v[90:91], v[88:89]\n"
      "v_fma_f64 v[90:91], -v[86:87], v[86:87], v[84:85]\n"
      "v_fma_f64 v[86:87], v[90:91], v[88:89], v[86:87]\n"
      "v_fma_f64 v[90:91], -v[86:87], v[86:87], v[84:85]\n"
      "v_fma_f64 v[86:87], v[90:91], v[88:89], v[86:87]\n"
      "v_add_f64 v[92:93], -v[78:79], -v[86:87]\n"    // nearer root
      "v_mul_f64 v[92:93], %[inv_a], v[92:93]\n"
      "v_cmp_lt_f64_e32 vcc, s[94:95], v[92:93]\n"
      "v_cmp_lt_f64_e64 s[88:89], v[92:93], %[bt]\n"
      "s_and_b64 s[88:89], vcc, s[88:89]\n"           // accepted at the nearer root
      "s_mov_b64 s[80:81], exec\n"
      "s_andn2_b64 exec, exec, s[88:89]\n"
      "s_cbranch_execz 15f\n"
      "v_add_f64 v[92:93], v[86:87], -v[78:79]\n"     // farther root
      "v_mul_f64 v[92:93], %[inv_a], v[92:93]\n"
      "v_cmp_lt_f64_e32 vcc, s[94:95], v[92:93]\n"
      "v_cmp_lt_f64_e64 s[80:81], v[92:93], %[bt]\n"
      "s_and_b64 s[80:81], vcc, s[80:81]\n"
      "s_or_b64 s[88:89], s[88:89], s[80:81]\n"
      "15:\n"
      "s_mov_b64 exec, s[88:89]\n"                    // lanes with a new best hit
      "v_mov_b64_e32 %[bt], v[92:93]\n"
      "v_mov_b32_e32 %[bp], v75\n"
      "v_cvt_f32_f64_e32 v90, v[92:93]\n"
      "v_mov_b32_e32 v91, 0x0da24260\n"               // 1e-30f
      "v_fmamk_f32 %[tmax], v90, 0x3f800012, v91\n"   // widen_tmax()
      "17:\n"
      "s_mov_b64 exec, s[84:85]\n"                    // every lane that was at a leaf pops
      "v_mov_b32_e32 %[cur], %[top]\n"
      "v_subrev_u32_e32 %[sp], %[stride], %[sp]\n"
      "ds_read_i16 %[top], %[sp]\n"
      "18:\n"
      "s_mov_b64 exec, s[82:83]\n"
      "s_branch 10b\n"
      "19:\n"
      "s_waitcnt lgkmcnt(0)\n"
      : [cur] "+v"(ts.cur), [sp] "+v"(ts.sp), [top] "+v"(ts.top), [bp] "+v"(ts.best_prim), [bt] "+v"(ts.best_t), [tmax] "+v"(ts.tmax_w)
      : [minbusy] "s"(min_busy), [kn] "s"(kn), [bases] "s"(bases), [stride] "n"(STRIDE_BYTES), [ox] "v"(ts.s.ox), [oy] "v"(ts.s.oy),
        [oz] "v"(ts.s.oz), [mx] "v"(ts.s.mx), [my] "v"(ts.s.my), [mz] "v"(ts.s.mz), [ax] "v"(ts.s.ax), [ay] "v"(ts.s.ay),
        [az] "v"(ts.s.az), [o_x] "v"(o.x), [o_y] "v"(o.y), [o_z] "v"(o.z), [d_x] "v"(d.x), [d_y] "v"(d.y), [d_z] "v"(d.z),
        [a] "v"(ts.a), [inv_a] "v"(ts.inv_a), [time] "v"(time), [omt] "v"(omt)
      : "vcc", "scc", "memory", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v78", "v79", "v80", "v81",
        "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "s80", "s81", "s82", "s83", "s84", "s85",
        "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98");
}
// One primitive against the ray: on a hit closer than the best so far, the best hit becomes this one.
template <class Acc>
__device__ __forceinline__ void prim_test(const Acc& acc, TravState& ts, int32_t pi, uint32_t features, d3 o, d3 d, double time,
                                          double tmin, PathRng& rng) {
  double t;
  bool h = false;
  if (Acc::kSpheresOnly) {
    PrimGeom g = acc.sphere(pi);
    h = sphere_hit(g, o, d, ts.a, ts.inv_a, time, tmin, ts.best_t, t);
  } else if (Acc::kPrimsInHbm && !(features & RT_F_BOX)) {
    // one round trip: fetch the whole 128-B record (one cache line) before looking at its type, instead of
    // type -> branch -> fields (two dependent L1/L2 latencies per leaf on a latency-bound instance)
    const Prim P = acc.full(pi);
    const uint32_t ty = P.type;
    if (ty <= 1u) {
      PrimGeom g = {{P.f[0], P.f[1], P.f[2]}, {P.f[3], P.f[4], P.f[5]}, P.f[6], P.type, P.mat};
      h = sphere_hit(g, o, d, ts.a, ts.inv_a, time, tmin, ts.best_t, t);
    } else if ((features & RT_F_QUAD) && ty == 2u) {
      h = quad_hit(P, o, d, tmin, ts.best_t, t);
    } else if ((features & RT_F_MEDIUM) && ty >= 3u) {
      h = medium_hit(P, o, d, ts.a, ts.inv_a, tmin, ts.best_t, rng, t);
    }
  } else if (Acc::kPrimsInHbm) {
    // Scenes with merged boxes (their quads are few): one round trip for the common types — the first 56 bytes of the
    // record (sphere: centres + radius; merged box: both corners; spherical medium: centre, -1/density, radius) and its
    // (type, material) word are requested together, before the type is looked at.  Quads and box media read the rest of
    // their record inside their branch.  (All 128 bytes up front hold 32 registers through every leaf test; with the
    // merged-box code that tipped the final-scene instance into spilling inside the traversal loop: 46 -> 96 ms.)
    const Prim& R = acc.full(pi);
    const double f0 = R.f[0], f1 = R.f[1], f2 = R.f[2], f3 = R.f[3], f4 = R.f[4], f5 = R.f[5], f6 = R.f[6];
    const uint32_t ty = R.type, mat = R.mat;
    if (ty <= 1u) {
      PrimGeom g = {{f0, f1, f2}, {f3, f4, f5}, f6, ty, mat};
      h = sphere_hit(g, o, d, ts.a, ts.inv_a, time, tmin, ts.best_t, t);
    } else if ((features & RT_F_QUAD) && ty == 2u) {
      h = quad_hit(R, o, d, tmin, ts.best_t, t);
    } else if ((features & RT_F_BOX) && ty == RT_PRIM_AABOX) {
      h = aabox_hit(R, d3{f0, f1, f2}, d3{f3, f4, f5}, o, d, ts.s, tmin, ts.best_t, t, pi);
    } else if ((features & RT_F_MEDIUM) && ty >= 3u) {
      h = medium_hit(R, o, d, ts.a, ts.inv_a, tmin, ts.best_t, rng, t);
    }
  } else {
    uint32_t ty = acc.type(pi);
    if (ty <= 1u) {
      PrimGeom g = acc.sphere(pi);
      h = sphere_hit(g, o, d, ts.a, ts.inv_a, time, tmin, ts.best_t, t);
    } else if ((features & RT_F_QUAD) && ty == 2u) {
      h = quad_hit(acc.full(pi), o, d, tmin, ts.best_t, t);
    } else if ((features & RT_F_BOX) && ty == RT_PRIM_AABOX) {
      const Prim& R = acc.full(pi);
      h = aabox_hit(R, d3{R.f[0], R.f[1], R.f[2]}, d3{R.f[3], R.f[4], R.f[5]}, o, d, ts.s, tmin, ts.best_t, t, pi);
    } else if ((features & RT_F_MEDIUM) && ty >= 3u) {
      h = medium_hit(acc.full(pi), o, d, ts.a, ts.inv_a, tmin, ts.best_t, rng, t);
    }
  }
  if (h) {
    ts.best_t = t;
    ts.best_prim = pi;
    ts.tmax_w = widen_tmax(t);
  }
}
// precondition: ts.cur < 0 && ts.cur != TravDone<StackT>::v
template <class Acc, class StackT>
__device__ __forceinline__ void leaf_step(const Acc& acc, TravState& ts, uint32_t features, d3 o, d3 d, double time,
                                          double tmin, StackT* stack, int stride, PathRng& rng) {
  prim_test(acc, ts, ~ts.cur, features, o, d, time, tmin, rng);
  trav_pop(ts, stack, stride);
}

#define RT_MAX_GLOBAL 8
struct GlobalPrims {
  int32_t n;           // primitives tested up-front, outside the tree
  int32_t tree_empty;  // ... and they are ALL the scene's primitives (tiny scenes: no tree at all)
  const int32_t* list;
};
__device__ __forceinline__ GlobalPrims global_prims(const Scene& sc) { return {sc.n_global, sc.tree_empty, sc.global_list}; }
// precondition: ts holds a fresh traversal (trav_restart) and the ray's constants.  The primitive index is the same in
// every lane (a scalar load), so the type dispatch inside prim_test does not diverge.
template <class Acc>
__device__ __forceinline__ void global_prims_test(const Acc& acc, const GlobalPrims& gp, TravState& ts, uint32_t features, d3 o,
                                                  d3 d, double time, double tmin, PathRng& rng) {
  // (a loop, not eight copies of the test: the list's length is a run-time value and the code stays in the instruction cache)
  // (the list is read through the constant address space: a scalar load the compiler may issue early, the index and
  // everything derived from it in SGPRs — a plain load became a vector load the wave waited on in every round)
  const __attribute__((address_space(4))) int32_t* list = (const __attribute__((address_space(4))) int32_t*)(uintptr_t)gp.list;
#pragma unroll 1
  for (int g = 0; g < gp.n; g++) prim_test(acc, ts, list[g], features, o, d, time, tmin, rng);
}

template <class Acc>
__device__ __forceinline__ void count_leaf_type(const Acc& acc, int32_t pi, TravCounters* cnt) {
  const uint32_t ty = acc.type(pi);
  cnt->t_sph += ty == 0u; cnt->t_mov += ty == 1u;
  if constexpr (!Acc::kSpheresOnly) {  // (sphere scenes: two live counters instead of five in a register-capped kernel)
    cnt->t_quad += ty == 2u; cnt->t_med += ty == 3u || ty == 4u; cnt->t_box += ty == RT_PRIM_AABOX;
  }
}

// Closest hit over t in (tmin, +inf) exclusive.  `stack` points at this lane's column of the LDS short
// stack (entries `stride` elements apart => bank-conflict free).  "while-while" form: lanes walk inner
// nodes until they hold a leaf, then the wave runs the (expensive, f64) leaf test together.
// Replaces BVHNode::hit / HittableList::hit (src/hittable/containers.rs:29-40,177-191) + AABB::hit
// (src/hittable/aabb.rs:48-71) + Sphere/MovingSphere::hit (geometry.rs:36-110).
template <class Acc, class StackT, bool COUNT>
__device__ __forceinline__ HitInfo traverse(const Acc& acc, int32_t root, uint32_t features, d3 o, d3 d, double time,
                                            double tmin, StackT* stack, int stride, TravCounters* cnt, PathRng& rng,
                                            GlobalPrims gp = GlobalPrims{0, 0, nullptr}) {
  constexpr int32_t kDone = TravDone<StackT>::v;
  TravState ts;
  trav_init(ts, root, kDone, lds_addr(stack), o, d);
  if (gp.n) global_prims_test(acc, gp, ts, features, o, d, time, tmin, rng);
  if (gp.tree_empty) ts.cur = kDone;
  while (ts.cur != kDone) {
    uint64_t t0 = 0;
    if (COUNT) t0 = now_cycles();
    if constexpr (RT_ASM_NODE_LOOP && !COUNT && !Acc::kWide && Acc::kSlabLds && sizeof(StackT) == 2) {
      node_loop_slab16(acc, ts, stride * 2, 1u);
    } else if constexpr (RT_ASM_NODE_LOOP && !COUNT && Acc::kWideHalf && RT_ASM_WIDE_FOR(Acc) && sizeof(StackT) == 2) {
      node_loop_wide16(acc, ts, stride * 2, 1u);
    } else {
      while (ts.cur >= 0) {
        if (COUNT) { cnt->nodes++; cnt->wnodes += wave_share(); }
        node_step(acc, ts, stack, stride);
      }
    }
    if (COUNT) { uint64_t t1 = now_cycles(); cnt->t_nodes += (float)(t1 - t0) * wave_share(); t0 = t1; }
    if (ts.cur != kDone) {
      if (COUNT) { cnt->prims++; cnt->wprims += wave_share(); count_leaf_type(acc, ~ts.cur, cnt); }
      leaf_step(acc, ts, features, o, d, time, tmin, stack, stride, rng);
      if (COUNT) { uint64_t t1 = now_cycles(); cnt->t_prims += (float)(t1 - t0) * wave_share(); }
    }
  }
  HitInfo best;
  best.t = ts.best_t;
  best.prim = ts.best_prim;
  return best;
}

// Resumable form of traverse() for a wave whose lanes carry independent traversals (k_fused, RESUME variant).
// Advances the participating lanes' traversals and hands the wave back once fewer than `min_busy` of them are
// unfinished (min_busy = 1: run everything to completion).  Inside, the inner-node loop is left as soon as fewer
// than `kn` lanes still walk inner nodes while some lane waits at a leaf (kn = 1: plain while-while).
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
template <class Acc, class StackT, bool COUNT, int STRIDE = 0>  // STRIDE: `stride` as a compile-time value where the caller has one
__device__ __forceinline__ void traverse_resume(const Acc& acc, TravState& ts, bool participating, uint32_t features,
                                                d3 o, d3 d, double time, double tmin, StackT* stack, int stride,
                                                TravCounters* cnt, PathRng& rng, uint32_t min_busy, uint32_t kn) {
  constexpr int32_t kDone = TravDone<StackT>::v;
  if (!participating) ts.cur = kDone;  // lanes without a path: nothing to walk (keeps the loop predicates to one compare)
  if constexpr (RT_ASM_TRAVERSAL && !COUNT && STRIDE > 0 && sizeof(StackT) == 2 && std::is_same<Acc, SlabLdsAcc>::value) {
    trav_resume_spheres_asm<STRIDE * 2>(acc, ts, o, d, time, min_busy, kn);  // (tmin is RT_TMIN on every path that renders)
    return;
  }
  while (true) {
    const unsigned long long m_busy = wave_ballot(ts.cur != kDone);
    if ((uint32_t)__popcll(m_busy) < min_busy) break;
    bool at_node = ts.cur >= 0;
    unsigned long long m_node = wave_ballot(at_node);
    const uint32_t need = (m_node != m_busy) ? kn : 1u;
    uint64_t t0 = 0;
    if (COUNT) t0 = now_cycles();
    if constexpr (RT_ASM_NODE_LOOP && !COUNT && !Acc::kWide && Acc::kSlabLds && sizeof(StackT) == 2) {
      node_loop_slab16(acc, ts, stride * 2, need);
    } else if constexpr (RT_ASM_NODE_LOOP && !COUNT && Acc::kWideHalf && RT_ASM_WIDE_FOR(Acc) && sizeof(StackT) == 2) {
      node_loop_wide16(acc, ts, stride * 2, need);
    } else {
      while ((uint32_t)__popcll(m_node) >= need) {
        if (at_node) {
          if (COUNT) { cnt->nodes++; cnt->wnodes += wave_share(); }
          node_step(acc, ts, stack, stride);
        } else if (COUNT) {
          if (!participating) cnt->idle_free++;
          else if (ts.cur == kDone) cnt->idle_done++;
          else cnt->idle_leaf++;
        }
        at_node = ts.cur >= 0;
        m_node = wave_ballot(at_node);
      }
    }
    if (COUNT) { uint64_t t1 = now_cycles(); cnt->t_nodes += (float)(t1 - t0) * wave_share(); t0 = t1; }
    if (ts.cur < 0 && ts.cur != kDone) {
      if (COUNT) { cnt->prims++; cnt->wprims += wave_share(); count_leaf_type(acc, ~ts.cur, cnt); }
      leaf_step(acc, ts, features, o, d, time, tmin, stack, stride, rng);
      if (COUNT) { uint64_t t1 = now_cycles(); cnt->t_prims += (float)(t1 - t0) * wave_share(); }
    }
  }
}

// Full hit record of (prim, t) — geometry.rs:58-70.
struct HitRec {
  d3 p, n, outward;
  bool front;
  double u, v;
  bool uv_valid;
};
#pragma clang fp contract(off)
__device__ __forceinline__ void sphere_uv(d3 on, double& u, double& v) {  // geometry.rs:30-34
  const double PI = 3.14159265358979323846;
#ifdef RT_EXP_F32_UV
  double theta = (double)acosf((float)-on.y);
  double phi = (double)atan2f((float)-on.z, (float)on.x) + PI;
#else
  double theta = acos(-on.y);
  double phi = atan2(-on.z, on.x) + PI;
#endif
  u = div_noscale(phi, 2. * PI);  // (operands in the normal range: the scaling-free division rounds like `/`)
  v = div_noscale(theta, PI);
}
template <class Acc>
__device__ __forceinline__ HitRec make_hit(const Acc& acc, int32_t pi, d3 o, d3 d, double time, double t, bool want_uv) {
  HitRec h;
  h.p = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};  // ray.at (ray.rs:17-19)
  h.u = 0; h.v = 0; h.uv_valid = false;
  if (Acc::kSpheresOnly || acc.type(pi) <= 1u) {
    PrimGeom g = acc.sphere(pi);
    d3 c = sphere_center(g, time);
    h.outward = {div_noscale(h.p.x - c.x, g.r), div_noscale(h.p.y - c.y, g.r), div_noscale(h.p.z - c.z, g.r)};
    if (want_uv) { sphere_uv(h.outward, h.u, h.v); h.uv_valid = true; }
  } else if (acc.type(pi) >= 3u) {  // medium scatter event: arbitrary normal, front face (book convention)
    h.outward = {1., 0., 0.};
    h.front = true;
    h.n = h.outward;
    return h;
  } else {
    const Prim& P = acc.full(pi);
    h.outward = {P.f[9], P.f[10], P.f[11]};
    if (want_uv) {
      double ab[2] = {0, 0}, tt;
      quad_hit(P, o, d, -__builtin_inf(), __builtin_inf(), tt, ab);
      h.u = ab[0]; h.v = ab[1]; h.uv_valid = true;
    }
  }
  h.front = (d.x * h.outward.x + d.y * h.outward.y + d.z * h.outward.z) < 0.;
  double sgn = h.front ? 1. : -1.;
  h.n = {sgn * h.outward.x, sgn * h.outward.y, sgn * h.outward.z};
  return h;
}
#pragma clang fp contract(on)

// ----------------------------------------------------------------------------------------------- textures
// xoroshiro128+ short jump as a GF(2)-linear map: state' = XOR over the 16 state bytes of table[byte][value]
// (the jump of random.rs:37-57 is linear, so its 128 conditional-XOR steps collapse to 16 table lookups).
// XOR of the four table rows row0..row0+3 selected by the bytes of `word` (one 32-bit quarter of the state)
template <class TabT>
__device__ __noinline__ uint4 noise_jump_part(TabT table, uint32_t word, int row0) {
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; i++) acc ^= table[(row0 + i) * 256 + ((word >> (i * 8)) & 0xffu)];
  return uint4{acc.x, acc.y, acc.z, acc.w};
}
__device__ __forceinline__ uint4 xor4(uint4 a, uint4 b) { return {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; }
__device__ __forceinline__ double noise_value(uint4 st) {
  uint64_t s0 = (uint64_t)st.x | ((uint64_t)st.y << 32), s1 = (uint64_t)st.z | ((uint64_t)st.w << 32);
  uint64_t r = s0 + s1;                       // next_u64 (random.rs:18-28)
  return (double)r / 18446744073709551616.0;  // next_f64 (random.rs:30-32), u64->f64 RNE
}
// texture.rs:148-154 for the 8 lattice corners of a cell at once.  Seed words (texture.rs:149-150):
//   a = x as u64 (sign-extended), b = y as u64 + (z as u64 << 32)  =>  b.lo = y, b.hi = z + (y < 0 ? -1 : 0).
// The jump is linear over GF(2), so the state splits into parts that depend on x only (2 values), on y only
// (2 values) and on (y's sign, z) (4 values): 40 table lookups for the 8 corners instead of 128.
// Corner k: x bit 0, z bit 1, y bit 2 (order of texture.rs:130-137).
#ifndef RT_NOISE_INLINE
#define RT_NOISE_INLINE 1
#endif
// The same with the row fetches in line (the compiler overlaps the 40 LDS reads instead of waiting after every four) and
// the XOR trees in three-input steps: per state word 28 v_bitop3 / v_xor instead of 48 v_xor.
template <class TabT>
__device__ __forceinline__ u32x4 noise_rows(TabT table, uint32_t word, int row0) {  // the 4 rows the bytes of `word` select
  const u32x4 a = table[(row0 + 0) * 256 + (word & 0xffu)], b = table[(row0 + 1) * 256 + ((word >> 8) & 0xffu)];
  const u32x4 c = table[(row0 + 2) * 256 + ((word >> 16) & 0xffu)], d = table[(row0 + 3) * 256 + (word >> 24)];
  return u32x4{xor3(a.x, b.x, c.x) ^ d.x, xor3(a.y, b.y, c.y) ^ d.y, xor3(a.z, b.z, c.z) ^ d.z, xor3(a.w, b.w, c.w) ^ d.w};
}
__device__ __forceinline__ double noise_value_words(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  // next_u64 = s0 + s1 with s0 = x | y << 32, s1 = z | w << 32 (random.rs:18-28); next_f64 = that / 2^64 (random.rs:30-32):
  // hi * 2^-32 + lo * 2^-64 in one fma rounds once, exactly like the u64 -> f64 conversion followed by the exact scaling
  const uint32_t lo = x + z;
  const uint32_t hi = y + w + (lo < x ? 1u : 0u);
  return __builtin_fma((double)hi, 0x1p-32, (double)lo * 0x1p-64);
}
template <class TabT>
__device__ __forceinline__ void noise_corners_inline(TabT table, int32_t ix, int32_t iy, int32_t iz, double c[8]) {
  u32x4 tx[2], ty[2], tz[4];
#pragma unroll
  for (int dx = 0; dx < 2; dx++) {
    const int32_t cx = (int32_t)((uint32_t)ix + (uint32_t)dx);
    // the coordinate's four bytes, and its sign extension: ONE row (the host folds positions 4..7, rt_init; value 0 -> 0)
    const uint32_t w = (uint32_t)cx;
    const u32x4 a = table[0 * 256 + (w & 0xffu)], b = table[1 * 256 + ((w >> 8) & 0xffu)];
    const u32x4 c = table[2 * 256 + ((w >> 16) & 0xffu)], e = table[3 * 256 + (w >> 24)];
    const u32x4 g = table[4 * 256 + (cx < 0 ? 255u : 0u)];
    tx[dx] = u32x4{xor3(xor3(a.x, b.x, c.x), e.x, g.x), xor3(xor3(a.y, b.y, c.y), e.y, g.y),
                   xor3(xor3(a.z, b.z, c.z), e.z, g.z), xor3(xor3(a.w, b.w, c.w), e.w, g.w)};
  }
#pragma unroll
  for (int dy = 0; dy < 2; dy++) {
    const int32_t cy = (int32_t)((uint32_t)iy + (uint32_t)dy);
    ty[dy] = noise_rows(table, (uint32_t)cy, 8);
#pragma unroll
    for (int dz = 0; dz < 2; dz++) {
      const uint32_t cz = (uint32_t)iz + (uint32_t)dz;
      tz[dy * 2 + dz] = noise_rows(table, cz + (cy < 0 ? 0xffffffffu : 0u), 12);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int dx = k & 1, dz = (k >> 1) & 1, dy = (k >> 2) & 1;
    const u32x4 &a = tx[dx], &b = ty[dy], &d = tz[dy * 2 + dz];
    c[k] = noise_value_words(xor3(a.x, b.x, d.x), xor3(a.y, b.y, d.y), xor3(a.z, b.z, d.z), xor3(a.w, b.w, d.w));
  }
}
// INLINE: the kernels whose only textures are checkers and noise (C3b); the general instances, where a noise lookup is
// one shading path among many, keep the out-of-line rows and their registers.
template <bool INLINE, class TabT>
__device__ __forceinline__ void noise_corners_dev(TabT table, int32_t ix, int32_t iy, int32_t iz,
                                                  double c[8]) {
  if constexpr (RT_NOISE_INLINE && INLINE) {
    noise_corners_inline(table, ix, iy, iz, c);
    return;
  }
  uint4 tx[2], ty[2], tz[4];
#pragma unroll
  for (int dx = 0; dx < 2; dx++) {
    const int32_t cx = (int32_t)((uint32_t)ix + (uint32_t)dx);
    tx[dx] = xor4(noise_jump_part(table, (uint32_t)cx, 0), noise_jump_part(table, cx < 0 ? 0xffffffffu : 0u, 4));
  }
#pragma unroll
  for (int dy = 0; dy < 2; dy++) {
    const int32_t cy = (int32_t)((uint32_t)iy + (uint32_t)dy);
    ty[dy] = noise_jump_part(table, (uint32_t)cy, 8);
#pragma unroll
    for (int dz = 0; dz < 2; dz++) {
      const uint32_t cz = (uint32_t)iz + (uint32_t)dz;
      tz[dy * 2 + dz] = noise_jump_part(table, cz + (cy < 0 ? 0xffffffffu : 0u), 12);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int dx = k & 1, dz = (k >> 1) & 1, dy = (k >> 2) & 1;
    c[k] = noise_value(xor4(xor4(tx[dx], ty[dy]), tz[dy * 2 + dz]));
  }
}
// Rust `as i32`: saturating, NaN -> 0 — which is exactly what v_cvt_i32_f64 does (out-of-range values and infinities
// saturate, NaN converts to 0).  The C++ cast is undefined out of range, so spelled out it was three compares and
// three branches per coordinate, three coordinates per checker level.
__device__ __forceinline__ int32_t sat_i32(double v) {
  int32_t r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ double rem_euclid1(double a) {  // f64::rem_euclid(1.0)
  double r = a - trunc(a);  // == fmod(a, 1.0) exactly
  return r < 0.0 ? r + 1.0 : r;
}
// atan2(y, x) / 2 pi in f32: octant reduction, t * P(t^2) with P fitted to atan(t) / (2 pi t) on [0, 1] (|error| 3e-8 of a
// revolution in f32 evaluation); the sign of the result is y's sign bit, like atan2's.  x = y = 0 gives NaN.
__device__ __forceinline__ float fast_atan2_rev(float y, float x) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float t = fminf(ax, ay) * __builtin_amdgcn_rcpf(fmaxf(ax, ay));
  const float s = t * t;
  float p = -0.0007689043413847685f;
  p = fmaf(p, s, 0.003940164111554623f);
  p = fmaf(p, s, -0.009584171697497368f);
  p = fmaf(p, s, 0.01586444303393364f);
  p = fmaf(p, s, -0.02234557457268238f);
  p = fmaf(p, s, 0.03178910166025162f);
  p = fmaf(p, s, -0.05305000767111778f);
  p = fmaf(p, s, 0.15915493667125702f);
  float r = p * t;
  r = ay > ax ? 0.25f - r : r;
  r = x < 0.f ? 0.5f - r : r;
  return __builtin_copysignf(r, y);
}
// Texel (i, j) of a W x H image under a sphere's outward normal `on`, without the f64 acos and atan2 of get_sphere_uv
// (geometry.rs:30-34) + ImageTexture::value (texture.rs:79-93), which are two thirds of what a path on the earth scene
// costs.  The texel those formulas select is a function of WHERE `on` points: column i iff the longitude of (on.x, -on.z)
// lies in [b_i, b_i+1), b_i = 2 pi i / W - pi; row j iff cos(pi (j+1) / H) < on.y <= cos(pi j / H).  So: GUESS (i, j) from
// f32 arctangents, then CHECK the guess in f64 against the boundaries themselves — two cross products with the unit
// vectors of b_i, b_i+1 and two comparisons with the cosines, all read from a table made at scene upload.  Returns false
// when the check fails (the guess was a texel off: ~1e-4 of the hits; the normal is ON a boundary or the seam or a pole;
// NaN) — the caller then evaluates the formulas, so every special case keeps the formulas' answer.  A passed check and
// the formulas can disagree only where the formulas' own rounding decides, within a few ulps of a boundary (for a
// 1024 x 512 image ~1e-13 of the hits) — the same class as the last-ulp differences between two libm's atan2.
__device__ __forceinline__ bool sphere_texel_checked(d3 on, const double* __restrict__ tab, uint32_t W, uint32_t H, uint32_t& i,
                                                     uint32_t& j) {
  const float x = (float)on.x, y = (float)on.y, z = (float)on.z;
  const float u = fast_atan2_rev(-z, x) + 0.5f;                                 // (phi + pi) / 2 pi
  const float th = fast_atan2_rev(__builtin_amdgcn_sqrtf(fmaf(x, x, z * z)), -y);  // theta / 2 pi, theta = acos(-y) in [0, pi]
  i = min((uint32_t)(u * (float)W), W - 1u);
  j = min((uint32_t)((1.f - 2.f * th) * (float)H), H - 1u);
  const double2* ut = reinterpret_cast<const double2*>(tab);
  const double* vt = tab + 2 * ((size_t)W + 1);
  const double2 b0 = ut[i], b1 = ut[i + 1];
  const double c0 = vt[j], c1 = vt[j + 1];
  const double px = on.x, py = -on.z;
  const double lo = b0.x * py - b0.y * px, hi = b1.x * py - b1.y * px;  // > 0: counter-clockwise of the boundary
  return lo >= 0. && hi < 0. && on.y <= c0 && on.y > c1;
}
#ifndef RT_UV_FORMULA_CALL
#define RT_UV_FORMULA_CALL 1
#endif
// get_sphere_uv + ImageTexture::value's index arithmetic (the formulas themselves) as a function of the outward normal,
// OUT OF LINE: the failed checks of sphere_texel_checked call it, so the f64 acos / atan2 code and its two dozen live
// registers are not part of the shading code's register budget.  Returns i | j << 32.
__device__ __noinline__ uint64_t sphere_texel_formula(double nx, double ny, double nz, uint32_t W, uint32_t H) {
  double u, v;
  sphere_uv(d3{nx, ny, nz}, u, v);
  u = fmin(fmax(u, 0.0), 1.0);
  v = 1.0 - fmin(fmax(v, 0.0), 1.0);
  uint32_t i = (uint32_t)(u * (double)W), j = (uint32_t)(v * (double)H);
  i = i > W - 1 ? W - 1 : i;  // deviation: clamp where the reference would panic
  j = j > H - 1 ? H - 1 : j;
  return (uint64_t)i | ((uint64_t)j << 32);
}
template <uint32_t F, class Acc>
__device__ __forceinline__ f3 tex_value(const Scene& sc, const Acc& acc, int32_t ti, int32_t pi, const HitRec& h0, d3 o,
                                        d3 d, double time, double t) {
  // texture.rs:55-66: a checker selects a child by the parity of floor(p*inv_scale); children may nest.
#pragma unroll 1
  for (int guard = 0; guard < 8; guard++) {
    const Texture& T = acc.texs(sc)[ti];
    const uint32_t kind = T.kind;
    if (kind == 0u) return {T.rgb[0], T.rgb[1], T.rgb[2]};  // texture.rs:32-36
    if ((F & RT_F_TEX_CHECKER) && kind == 1u) {
      int32_t x = sat_i32(floor(h0.p.x * T.inv_scale));
      int32_t y = sat_i32(floor(h0.p.y * T.inv_scale));
      int32_t z = sat_i32(floor(h0.p.z * T.inv_scale));
      uint32_t s = (uint32_t)x + (uint32_t)y + (uint32_t)z;
      const bool is_odd = (s & 1u) == 0u;  // rem_euclid(2) == 0 -> `odd` field (texture.rs:60-61)
      ti = is_odd ? T.odd : T.even;
      if (ti < 0) return is_odd ? f3{T.rgb[0], T.rgb[1], T.rgb[2]} : f3{T.rgb2[0], T.rgb2[1], T.rgb2[2]};
      continue;
    }
    if ((F & RT_F_TEX_IMAGE) && kind == 2u) {  // texture.rs:79-93
      if (T.width == 0 || T.height == 0) return {0.f, 1.f, 1.f};
      uint32_t i = 0, j = 0;
      if (RT_UV_FORMULA_CALL && !h0.uv_valid && (Acc::kSpheresOnly || acc.type(pi) <= 1u)) {
        const HitRec h = make_hit(acc, pi, o, d, time, t, false);  // (the outward normal in f64, as the reference has it)
        bool have = false;
        if (T.uvtab_off)
          have = sphere_texel_checked(h.outward, reinterpret_cast<const double*>(sc.texels + T.uvtab_off), T.width, T.height, i, j);
        if (!have) {  // ~1e-4 of the hits (or no table): out of line, so that its registers are not the branch's
          const uint64_t ij = sphere_texel_formula(h.outward.x, h.outward.y, h.outward.z, T.width, T.height);
          i = (uint32_t)ij; j = (uint32_t)(ij >> 32);
        }
      } else {
        bool have = false;
        if (T.uvtab_off && !h0.uv_valid && (Acc::kSpheresOnly || acc.type(pi) <= 1u)) {
          const HitRec h = make_hit(acc, pi, o, d, time, t, false);
          have = sphere_texel_checked(h.outward, reinterpret_cast<const double*>(sc.texels + T.uvtab_off), T.width, T.height, i, j);
        }
        if (!have) {
          double u = h0.u, v = h0.v;
          if (!h0.uv_valid) {
            HitRec h = make_hit(acc, pi, o, d, time, t, true);
            u = h.u; v = h.v;
          }
          u = fmin(fmax(u, 0.0), 1.0);
          v = 1.0 - fmin(fmax(v, 0.0), 1.0);
          i = (uint32_t)(u * (double)T.width); j = (uint32_t)(v * (double)T.height);
          i = i > T.width - 1 ? T.width - 1 : i;  // deviation: clamp where the reference would panic
          j = j > T.height - 1 ? T.height - 1 : j;
        }
      }
      uint32_t px = sc.texels[T.texel_off + j * T.width + i];
      const float k = 1.0f / 255.0f;
      return {(float)(px & 0xffu) * k, (float)((px >> 8) & 0xffu) * k, (float)((px >> 16) & 0xffu) * k};
    }
    if ((F & RT_F_TEX_NOISE) && kind == 3u) {  // texture.rs:108-146 (grey noise_at variant)
      double x = h0.p.x * T.inv_scale, y = h0.p.y * T.inv_scale, z = h0.p.z * T.inv_scale;
      int32_t ix = sat_i32(floor(x)), iy = sat_i32(floor(y)), iz = sat_i32(floor(z));
      double fx = rem_euclid1(x), fy = rem_euclid1(y), fz = rem_euclid1(z);
      double xb = fx * fx * (3.0 - 2.0 * fx), yb = fy * fy * (3.0 - 2.0 * fy), zb = fz * fz * (3.0 - 2.0 * fz);
      double c[8];
      noise_corners_dev<(F & ~(RT_F_TEX_CHECKER | RT_F_TEX_NOISE)) == 0u>(acc.noise_table(sc), ix, iy, iz, c);
      auto lerp = [](double a, double b, double w) { return (1.0 - w) * a + w * b; };  // color.rs:62-64
      double m00 = lerp(c[0], c[1], xb), m01 = lerp(c[2], c[3], xb);
      double m10 = lerp(c[4], c[5], xb), m11 = lerp(c[6], c[7], xb);
      double o0 = lerp(m00, m10, yb), o1 = lerp(m01, m11, yb);
      float g = (float)lerp(o0, o1, zb);
      return {g, g, g};
    }
    break;
  }
  return {0.f, 1.f, 1.f};
}

// ----------------------------------------------------------------------------------------------- sampling
// Square roots of the SAMPLING code (directions on the unit sphere, lens disk, refraction): the bare v_sqrt_f32 (1 ulp),
// like the v_sin / v_cos beside them; sqrtf() expands to 14 instructions of correction and denormal scaling.
#ifndef RT_RAW_SQRT
#define RT_RAW_SQRT 1
#endif
__device__ __forceinline__ float samp_sqrt(float x) { return RT_RAW_SQRT ? __builtin_amdgcn_sqrtf(x) : __builtin_sqrtf(x); }
// vec3.rs:39-47 — uniform point on the unit sphere (theta uniform, z uniform).  v_sin/v_cos take revolutions.
__device__ __forceinline__ f3 sample_unit_sphere(uint32_t r0, uint32_t r1) {
  float u = u01f(r0);
  float z = 2.f * u01f(r1) - 1.f;
  float r = samp_sqrt(fmaxf(0.f, 1.f - z * z));
  return {r * __builtin_amdgcn_cosf(u), r * __builtin_amdgcn_sinf(u), z};
}

// Camera::sample_point (src/camera/mod.rs:176-186) + sample_pixel's sub-pixel position (mod.rs:150-172) for
// sample index s of pixel (j = row, i = column).
__device__ __forceinline__ void generate_ray(const Camera& cam, uint32_t j, uint32_t i, uint32_t s, uint32_t k0,
                                             uint32_t k1, d3& o, d3& d, double& time, PathRng& rng) {
  const uint32_t pixel = j * cam.width + i;
  uint32_t r[4];
  philox4x32_10(pixel, s, 0u, 0u, k0, k1, r);
  rng.s0 = r[2];
  rng.s1 = r[3] | ((r[2] | r[3]) == 0u ? 1u : 0u);  // the all-zero state is a fixed point
  double dx, dy;
  {
#pragma clang fp contract(off)
    if (cam.sampler_kind == 0u) {  // Uniform(n): fixed grid, offset as in mod.rs:153-157
      const double interval = cam.grid_interval, offset = cam.grid_offset;
      // s / n without the 22-instruction integer division: (s + 0.5) * (1/n) is at least 0.5/n away from an integer, far
      // more than its f64 rounding error for any 32-bit s and n <= 65535 — the quotient is exact
      uint32_t n = cam.sampler_n;
      uint32_t yi = (uint32_t)(((double)s + 0.5) * interval), xi = s - yi * n;
      dy = (double)j + (double)yi * interval - offset;
      dx = (double)i + (double)xi * interval - offset;
    } else {  // Random(n): dy then dx uniform in [-0.5, 0.5) (mod.rs:166-167)
      dy = (double)j + (u01d(rng_next(rng)) - 0.5);
      dx = (double)i + (u01d(rng_next(rng)) - 0.5);
    }
    d3 pc = {cam.pixel00[0] + dx * cam.du[0] + dy * cam.dv[0], cam.pixel00[1] + dx * cam.du[1] + dy * cam.dv[1],
             cam.pixel00[2] + dx * cam.du[2] + dy * cam.dv[2]};
    if (cam.defocus_angle <= 0.0) {
      o = {cam.center[0], cam.center[1], cam.center[2]};
    } else {
      // uniform point in the unit disk (same distribution as the rejection loop of vec3.rs:48-59)
      float rr = samp_sqrt(u01f(r[0]));
      float u = u01f(r[1]);
      double lx = (double)(rr * __builtin_amdgcn_cosf(u)), ly = (double)(rr * __builtin_amdgcn_sinf(u));
      o = {cam.center[0] + cam.dfu[0] * lx + cam.dfv[0] * ly, cam.center[1] + cam.dfu[1] * lx + cam.dfv[1] * ly,
           cam.center[2] + cam.dfu[2] * lx + cam.dfv[2] * ly};
    }
    d = {pc.x - o.x, pc.y - o.y, pc.z - o.z};
  }
  time = u01d(rng_next(rng));  // mod.rs:184
}

// ----------------------------------------------------------------------------------------------- shading
struct ShadeOut {
  bool alive;   // a scattered ray continues (o, d, thr updated); otherwise `radiance` is the path's contribution
  f3 radiance;  // already multiplied by the throughput
  d3 o, d;
  f3 thr;
};
__device__ __forceinline__ f3 sky(d3 d) {  // mod.rs:205-207
  f3 u = normalize(tof(d));
  float a = 0.5f * (u.y + 1.0f);
  return {(1.f - a) + a * 0.5f, (1.f - a) + a * 0.7f, (1.f - a) + a * 1.0f};
}

// Hit point and shading normal as the render kernels form them (geometry.rs:58-64): the point stays f64 (it is the next
// ray's origin); the normal is only ever used by f32 shading math, so it is the f64 difference p - c scaled by an f32
// reciprocal radius, flipped against the ray.  Also what rt_intersect reports for the production tree forms (probe_tree
// 3 / 4), so that the parity tests hold the render path's own normal against the oracle's.
template <class Acc>
__device__ __forceinline__ void shading_frame(const Acc& acc, int32_t prim, d3 o, d3 d, double time, double t, d3& p, f3& n,
                                              bool& front) {
  p = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
  if (Acc::kSpheresOnly || acc.type(prim) <= 1u) {
    PrimGeom g = acc.sphere(prim);
    d3 c = sphere_center(g, time);
    n = __builtin_amdgcn_rcpf((float)g.r) * tof(p - c);
  } else if (acc.type(prim) == 2u) {
    const Prim& P = acc.full(prim);
    n = {(float)P.f[9], (float)P.f[10], (float)P.f[11]};
  } else {
    n = -1.0f * normalize(tof(d));  // medium: the isotropic phase function ignores the normal
  }
  front = dot(tof(d), n) < 0.f;
  if (!front) n = -1.0f * n;
}

// One bounce of ray_color_inner (src/camera/mod.rs:187-210) given the closest hit.
template <uint32_t F, class Acc>
__device__ __forceinline__ ShadeOut shade(const Scene& sc, const Acc& acc, const Camera& cam, d3 o, d3 d, double time,
                                          f3 thr, HitInfo hit, PathRng& rng) {
  ShadeOut out;
  out.alive = false;
  out.radiance = {0.f, 0.f, 0.f};
  out.thr = thr;
  out.o = o; out.d = d;
  if (hit.prim < 0) {  // miss: sky gradient or solid background (EXT)
    f3 c;
    if (cam.background_mode == 0u) c = sky(d);
    else c = {cam.background[0], cam.background[1], cam.background[2]};
    out.radiance = {thr.x * c.x, thr.y * c.y, thr.z * c.z};
    return out;
  }
  const Material& M = acc.mats(sc)[acc.mat(hit.prim)];
  const uint32_t kind = M.kind;
  HitRec h;
  h.u = 0; h.v = 0; h.uv_valid = false;
  f3 n;
  shading_frame(acc, hit.prim, o, d, time, hit.t, h.p, n, h.front);
  uint32_t r[2] = {rng_next(rng), rng_next(rng)};
  // one unit-sphere sample for whichever material needs it (Lambertian, Metal fuzz, isotropic): computed once for
  // the wave instead of once per divergent material branch
  const f3 u = sample_unit_sphere(r[0], r[1]);
  f3 att, dir;
  if (kind == 0u) {  // Lambertian (materials.rs:28-43)
    dir = n + u;
    // near_zero (vec3.rs:89-92, threshold 1e-9 on f64 components): the guard exists so that a direction that cancelled
    // to rounding noise is not sent out.  n and u are unit vectors in f32 here, so the sum's noise floor is ~6e-8 per
    // component, not 1e-16: a literal 1e-9 could never fire while a ~1e-8-long garbage direction went out.  Scaled to
    // f32: 16 ulp of 1.0.  (Probability of either event ~1e-13 per bounce: statistically invisible, as in the reference.)
    const float th = 9.5367431640625e-07f;
    if (__builtin_fabsf(dir.x) < th && __builtin_fabsf(dir.y) < th && __builtin_fabsf(dir.z) < th) dir = n;
    if (M.tex < 0) att = {M.albedo[0], M.albedo[1], M.albedo[2]};
    else att = tex_value<F>(sc, acc, M.tex, hit.prim, h, o, d, time, hit.t);
  } else if (kind == 1u) {  // Metal (materials.rs:73-80): reflects the UN-normalised direction, adds fuzz * unit
    f3 df = tof(d);
    f3 refl = df - (2.0f * dot(df, n)) * n;
    dir = refl + M.fuzz * u;
    att = {M.albedo[0], M.albedo[1], M.albedo[2]};
  } else if (kind == 2u) {  // Dielectric (materials.rs:104-141)
    float ratio = h.front ? __builtin_amdgcn_rcpf(M.ior) : M.ior;
    f3 ud = normalize(tof(d));
    float cos_t = fminf(-dot(ud, n), 1.0f);
    float sin_t = samp_sqrt(fmaxf(0.f, 1.0f - cos_t * cos_t));
    bool cannot = ratio * sin_t > 1.0f;
    float r0 = (1.f - ratio) * __builtin_amdgcn_rcpf(1.f + ratio);
    r0 = r0 * r0;
    float m = 1.f - cos_t;
    float refl_p = r0 + (1.f - r0) * (m * m * m * m * m);
    if (cannot || refl_p > u01f(r[0])) {
      dir = ud - (2.0f * dot(ud, n)) * n;
    } else {  // refract (materials.rs:131-136)
      f3 perp = ratio * (ud + cos_t * n);
      float par = -samp_sqrt(__builtin_fabsf(1.0f - dot(perp, perp)));
      dir = perp + par * n;
    }
    att = {1.f, 1.f, 1.f};
  } else if (kind == 3u) {  // EXT diffuse light: emits, never scatters
    f3 e;
    if (cam.background_mode == 0u) e = sky(d);
    else if (M.tex < 0) e = {M.albedo[0], M.albedo[1], M.albedo[2]};
    else e = tex_value<F>(sc, acc, M.tex, hit.prim, h, o, d, time, hit.t);  // reference integrator: a `None` scatter falls through to the sky (mod.rs:203-207)
    out.radiance = {thr.x * e.x, thr.y * e.y, thr.z * e.z};
    return out;
  } else {  // EXT isotropic
    dir = u;
    if (M.tex < 0) att = {M.albedo[0], M.albedo[1], M.albedo[2]};
    else att = tex_value<F>(sc, acc, M.tex, hit.prim, h, o, d, time, hit.t);
  }
  out.thr = {thr.x * att.x, thr.y * att.y, thr.z * att.z};
  out.o = h.p;
  out.d = tod(dir);
  out.alive = true;
  return out;
}

}  // namespace rtd
