// rt_device.h — device-side data layout and functions of the MI355X path tracer (gfx950 only).
//
// Precision plan: geometry that decides WHERE a hit is — ray origin/direction, sphere
// quadratic, hit point, normal — is f64 like the reference (src/vec3.rs:11); BVH slab culling is f32 on
// conservatively padded boxes (result-neutral: closest-hit does not depend on what is culled,
// src/hittable/containers.rs:177-191); colour / throughput / scatter-direction sampling is f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_BLOCK 256
#define RT_STACK_DEPTH 32  // the host BVH builder guarantees tree depth <= RT_STACK_DEPTH
#define RT_TMIN 0.000001   // src/camera/mod.rs:198

namespace rtd {

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
  // QUAD: f[0..2]=Q, f[3..5]=u, f[6..8]=v, f[9..11]=unit normal, f[12]=D, f[13]=1/|u x v|
  // MEDIUM_SPHERE: f[0..2]=center, f[6]=radius, f[13]=density
  // MEDIUM_BOX: f[0..2]=corner, f[3..5],f[6..8],f[9..11]=edges, f[13]=density
  double f[14];
  uint32_t type, mat;
  int32_t obj;
  uint32_t pad;
};

struct Material {  // 32 B
  uint32_t kind;
  int32_t tex;
  float albedo[3];
  float fuzz, ior;
  uint32_t pad;
};
struct Texture {  // 48 B
  uint32_t kind;
  int32_t odd, even;
  uint32_t width, height;
  uint32_t texel_off;  // in texels (RGBA8 words)
  float rgb[3];
  uint32_t pad;
  double inv_scale;
};

struct Scene {
  const BvhNode* nodes;
  const Prim* prims;
  const Material* mats;
  const Texture* texs;
  const uint32_t* texels;    // RGBA8
  const uint4* noise_table;  // 16 x 256 x 128-bit GF(2) table of the xoroshiro short jump (or null)
  int32_t n_nodes, n_prims, n_mats, n_texs;
  int32_t root;  // >=0 node, <0 single leaf
};

struct Camera {
  double center[3], pixel00[3], du[3], dv[3], dfu[3], dfv[3];
  double defocus_angle;
  float background[3];
  uint32_t width, height, sampler_kind, sampler_n, max_depth, background_mode;
};

// Path state, structure of arrays (one queue).  72 B per path.
struct Queue {
  double *ox, *oy, *oz, *dx, *dy, *dz, *time;
  float *tr, *tg, *tb;
  uint32_t* id;  // pass-local path index: pixel_local * S + k
};
struct Hits {
  int32_t* prim;
  double* t;
};

// ----------------------------------------------------------------------------------------------- RNG
// Philox4x32-10 (Salmon et al., SC'11).  Counter = (pixel, sample, bounce, purpose), key = seed.
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// purposes (counter word 3)
#define RT_RNG_CAMERA 0x10000u   // subpixel x2, lens x2
#define RT_RNG_CAMERA2 0x10001u  // time
#define RT_RNG_SCATTER 0u        // + nothing: counter word 2 is the bounce

__device__ __forceinline__ float u01f(uint32_t x) { return (float)(x >> 8) * 5.9604645e-8f + 2.9802322e-8f; }  // (0,1)
__device__ __forceinline__ double u01d(uint32_t x) { return ((double)x + 0.5) * 2.3283064365386963e-10; }

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

// ----------------------------------------------------------------------------------------------- intersection
struct HitInfo {
  double t;
  int32_t prim;  // -1 = miss
};

// src/hittable/geometry.rs:36-57 — sphere quadratic in f64, evaluated in the reference's operation order
// with contraction off so that `t` is bit-identical to the CPU path for identical inputs.
// Returns true and updates tmax/best when a root lies in (tmin, tmax) exclusive.
#pragma clang fp contract(off)
__device__ __forceinline__ bool sphere_hit(const Prim& P, bool moving, d3 o, d3 d, double a, double time, double tmin,
                                           double tmax, double& t_out) {
  d3 c;
  if (moving) {  // geometry.rs:108: center*(1-time) + destination*time
    double w0 = 1. - time;
    c = {P.f[0] * w0 + P.f[3] * time, P.f[1] * w0 + P.f[4] * time, P.f[2] * w0 + P.f[5] * time};
  } else {
    c = {P.f[0], P.f[1], P.f[2]};
  }
  double r = P.f[6];
  d3 oc = o - c;
  double alignment = oc.x * d.x + oc.y * d.y + oc.z * d.z;
  double surface = (oc.x * oc.x + oc.y * oc.y + oc.z * oc.z) - r * r;
  double disc = alignment * alignment - a * surface;
  if (disc < 0.) return false;
  double sq = __builtin_sqrt(disc);
  double root = (-alignment - sq) / a;
  if (!(tmin < root && root < tmax)) {
    root = (-alignment + sq) / a;
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
  double t = (P.f[12] - (n.x * o.x + n.y * o.y + n.z * o.z)) / denom;
  if (!(tmin < t && t < tmax)) return false;
  d3 p = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
  d3 ph = {p.x - P.f[0], p.y - P.f[1], p.z - P.f[2]};
  d3 u = {P.f[3], P.f[4], P.f[5]}, v = {P.f[6], P.f[7], P.f[8]};
  // w = n_raw / (n_raw . n_raw) = unit_n * inv_len
  d3 w = {n.x * P.f[13], n.y * P.f[13], n.z * P.f[13]};
  d3 phxv = {ph.y * v.z - ph.z * v.y, ph.z * v.x - ph.x * v.z, ph.x * v.y - ph.y * v.x};
  d3 uxph = {u.y * ph.z - u.z * ph.y, u.z * ph.x - u.x * ph.z, u.x * ph.y - u.y * ph.x};
  double alpha = w.x * phxv.x + w.y * phxv.y + w.z * phxv.z;
  double beta = w.x * uxph.x + w.y * uxph.y + w.z * uxph.z;
  if (alpha < 0. || alpha > 1. || beta < 0. || beta > 1.) return false;
  t_out = t;
  if (ab) { ab[0] = alpha; ab[1] = beta; }
  return true;
}
#pragma clang fp contract(fast)

// Per-ray constants of the f32 slab test.  t = plane*inv - (o*inv); the origin's f32 rounding error
// |o|*2^-24*|inv| is covered by widening o*inv per axis (oiN for entry planes, oiF for exit planes); the box
// coordinates' own rounding is covered by padding at build time.
struct SlabRay {
  float ix, iy, iz;        // 1/d
  float nx, ny, nz;        // o*inv + slack   (subtracted from entry planes)
  float fx, fy, fz;        // o*inv - slack   (subtracted from exit planes)
  bool sx, sy, sz;         // direction negative on axis
};
__device__ __forceinline__ SlabRay make_slab(d3 o, d3 d) {
  SlabRay s;
  float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
  s.ix = __builtin_amdgcn_rcpf((float)d.x);
  s.iy = __builtin_amdgcn_rcpf((float)d.y);
  s.iz = __builtin_amdgcn_rcpf((float)d.z);
  float ax = ox * s.ix, ay = oy * s.iy, az = oz * s.iz;
  const float E = 9.5367431640625e-07f;  // 2^-20: > 8 ulp, covers o / d / rcp / fma rounding
  float ex = __builtin_fabsf(ax) * E, ey = __builtin_fabsf(ay) * E, ez = __builtin_fabsf(az) * E;
  s.nx = ax + ex; s.ny = ay + ey; s.nz = az + ez;
  s.fx = ax - ex; s.fy = ay - ey; s.fz = az - ez;
  s.sx = s.ix < 0.f; s.sy = s.iy < 0.f; s.sz = s.iz < 0.f;
  return s;
}
// returns entry distance (clamped to 0) in tn; hit iff tn <= tf
__device__ __forceinline__ bool slab_test(const SlabRay& s, const float* lo, const float* hi, float tmax, float& tn) {
  float nx = s.sx ? hi[0] : lo[0], fx = s.sx ? lo[0] : hi[0];
  float ny = s.sy ? hi[1] : lo[1], fy = s.sy ? lo[1] : hi[1];
  float nz = s.sz ? hi[2] : lo[2], fz = s.sz ? lo[2] : hi[2];
  float t0x = __builtin_fmaf(nx, s.ix, -s.nx), t1x = __builtin_fmaf(fx, s.ix, -s.fx);
  float t0y = __builtin_fmaf(ny, s.iy, -s.ny), t1y = __builtin_fmaf(fy, s.iy, -s.fy);
  float t0z = __builtin_fmaf(nz, s.iz, -s.nz), t1z = __builtin_fmaf(fz, s.iz, -s.fz);
  // fmaxf/fminf ignore NaN (0*inf axes) => conservative
  tn = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), 0.f);
  float tf = fminf(fminf(fminf(t1x, t1y), t1z), tmax);
  // relative widening of the far side covers the entry/exit rounding of near-degenerate intervals
  return tn <= tf * 1.0000019f + 1e-30f;
}

// Closest hit over t in (tmin, +inf) exclusive.  `stack` points at this lane's column of the LDS short
// stack (entries `stride` words apart => bank-conflict free).  Replaces BVHNode::hit / HittableList::hit
// (src/hittable/containers.rs:29-40,177-191) + AABB::hit (src/hittable/aabb.rs:48-71).
__device__ __forceinline__ HitInfo traverse(const Scene& sc, d3 o, d3 d, double time, double tmin, uint32_t* stack,
                                            int stride) {
  HitInfo best;
  best.t = __builtin_inf();
  best.prim = -1;
  const double a = d.x * d.x + d.y * d.y + d.z * d.z;  // squared_raydir_magnitude (geometry.rs:43)
  SlabRay s = make_slab(o, d);
  float tmax32 = __builtin_inff();
  int sp = 0;
  int32_t cur = sc.root;
  while (true) {
    if (cur >= 0) {
      const float4* np = reinterpret_cast<const float4*>(sc.nodes + cur);
      float4 q0 = np[0], q1 = np[1], q2 = np[2];
      int2 ch = *reinterpret_cast<const int2*>(&np[3]);
      float llo[3] = {q0.x, q0.y, q0.z}, lhi[3] = {q0.w, q1.x, q1.y};
      float rlo[3] = {q1.z, q1.w, q2.x}, rhi[3] = {q2.y, q2.z, q2.w};
      float tl, tr;
      bool hl = slab_test(s, llo, lhi, tmax32, tl);
      bool hr = slab_test(s, rlo, rhi, tmax32, tr);
      if (hl && hr) {
        bool left_first = tl <= tr;
        int32_t nearc = left_first ? ch.x : ch.y, farc = left_first ? ch.y : ch.x;
        stack[sp * stride] = (uint32_t)farc;
        sp++;
        cur = nearc;
        continue;
      } else if (hl) {
        cur = ch.x;
        continue;
      } else if (hr) {
        cur = ch.y;
        continue;
      }
    } else if (cur != RT_EMPTY_CHILD) {
      const int32_t pi = ~cur;
      const Prim& P = sc.prims[pi];
      double t;
      bool h = false;
      uint32_t ty = P.type;
      if (ty <= 1u) h = sphere_hit(P, ty == 1u, o, d, a, time, tmin, best.t, t);
      else if (ty == 2u) h = quad_hit(P, o, d, tmin, best.t, t);
      if (h) {
        best.t = t;
        best.prim = pi;
        tmax32 = __double2float_ru(t);
      }
    }
    if (sp == 0) break;
    sp--;
    cur = (int32_t)stack[sp * stride];
  }
  return best;
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
  double theta = acos(-on.y);
  double phi = atan2(-on.z, on.x) + PI;
  u = phi / (2. * PI);
  v = theta / PI;
}
__device__ __forceinline__ HitRec make_hit(const Prim& P, d3 o, d3 d, double time, double t, bool want_uv) {
  HitRec h;
  h.p = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};  // ray.at (ray.rs:17-19)
  h.u = 0; h.v = 0; h.uv_valid = false;
  if (P.type <= 1u) {
    d3 c;
    if (P.type == 1u) {
      double w0 = 1. - time;
      c = {P.f[0] * w0 + P.f[3] * time, P.f[1] * w0 + P.f[4] * time, P.f[2] * w0 + P.f[5] * time};
    } else {
      c = {P.f[0], P.f[1], P.f[2]};
    }
    double r = P.f[6];
    h.outward = {(h.p.x - c.x) / r, (h.p.y - c.y) / r, (h.p.z - c.z) / r};
    if (want_uv) { sphere_uv(h.outward, h.u, h.v); h.uv_valid = true; }
  } else {
    h.outward = {P.f[9], P.f[10], P.f[11]};
    if (want_uv) {
      double ab[2], tt;
      quad_hit(P, o, d, -__builtin_inf(), __builtin_inf(), tt, ab);
      h.u = ab[0]; h.v = ab[1]; h.uv_valid = true;
    }
  }
  h.front = (d.x * h.outward.x + d.y * h.outward.y + d.z * h.outward.z) < 0.;
  double sgn = h.front ? 1. : -1.;
  h.n = {sgn * h.outward.x, sgn * h.outward.y, sgn * h.outward.z};
  return h;
}
#pragma clang fp contract(fast)

// ----------------------------------------------------------------------------------------------- textures
// xoroshiro128+ short jump as a GF(2)-linear map: state' = XOR over the 16 state bytes of table[byte][value]
// (the jump of random.rs:37-57 is linear, so the 128 conditional XOR steps collapse to 16 lookups).
__device__ __forceinline__ double noise_at_dev(const Scene& sc, int32_t x, int32_t y, int32_t z) {  // texture.rs:148-154
  uint64_t a = (uint64_t)(int64_t)x;
  uint64_t b = (uint64_t)(int64_t)y + ((uint64_t)(int64_t)z << 32);
  uint32_t w[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
  uint4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; i++) {
    uint32_t byte = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
    uint4 e = sc.noise_table[i * 256 + byte];
    acc.x ^= e.x; acc.y ^= e.y; acc.z ^= e.z; acc.w ^= e.w;
  }
  uint64_t s0 = (uint64_t)acc.x | ((uint64_t)acc.y << 32), s1 = (uint64_t)acc.z | ((uint64_t)acc.w << 32);
  uint64_t r = s0 + s1;                            // next_u64 (random.rs:18-28)
  return (double)r / 18446744073709551616.0;       // next_f64 (random.rs:30-32), u64->f64 RNE
}
__device__ __forceinline__ int32_t sat_i32(double v) {  // Rust `as i32`
  if (!(v == v)) return 0;
  if (v <= -2147483648.0) return INT32_MIN;
  if (v >= 2147483647.0) return INT32_MAX;
  return (int32_t)v;
}
__device__ __forceinline__ double rem_euclid1(double a) {  // f64::rem_euclid(1.0)
  double r = fmod(a, 1.0);
  return r < 0.0 ? r + 1.0 : r;
}
__device__ __forceinline__ f3 tex_value(const Scene& sc, int32_t ti, const Prim& P, const HitRec& h0, d3 o, d3 d,
                                        double time, double t) {
  // texture.rs:55-66: a checker selects a child by the parity of floor(p*inv_scale); children may nest.
  for (int guard = 0; guard < 8; guard++) {
    const Texture& T = sc.texs[ti];
    if (T.kind == 0u) return {T.rgb[0], T.rgb[1], T.rgb[2]};  // texture.rs:32-36
    if (T.kind == 1u) {
      int32_t x = sat_i32(floor(h0.p.x * T.inv_scale));
      int32_t y = sat_i32(floor(h0.p.y * T.inv_scale));
      int32_t z = sat_i32(floor(h0.p.z * T.inv_scale));
      uint32_t s = (uint32_t)x + (uint32_t)y + (uint32_t)z;
      ti = ((s & 1u) == 0u) ? T.odd : T.even;  // rem_euclid(2) == 0 -> `odd` field (texture.rs:60-61)
      continue;
    }
    if (T.kind == 2u) {  // texture.rs:79-93
      if (T.width == 0 || T.height == 0) return {0.f, 1.f, 1.f};
      double u = h0.u, v = h0.v;
      if (!h0.uv_valid) {
        HitRec h = make_hit(P, o, d, time, t, true);
        u = h.u; v = h.v;
      }
      u = fmin(fmax(u, 0.0), 1.0);
      v = 1.0 - fmin(fmax(v, 0.0), 1.0);
      uint32_t i = (uint32_t)(u * (double)T.width), j = (uint32_t)(v * (double)T.height);
      i = i > T.width - 1 ? T.width - 1 : i;  // deviation: clamp where the reference would panic
      j = j > T.height - 1 ? T.height - 1 : j;
      uint32_t px = sc.texels[T.texel_off + j * T.width + i];
      const float k = 1.0f / 255.0f;
      return {(float)(px & 0xffu) * k, (float)((px >> 8) & 0xffu) * k, (float)((px >> 16) & 0xffu) * k};
    }
    // NOISE — texture.rs:108-146 (grey noise_at variant)
    double x = h0.p.x * T.inv_scale, y = h0.p.y * T.inv_scale, z = h0.p.z * T.inv_scale;
    int32_t ix = sat_i32(floor(x)), iy = sat_i32(floor(y)), iz = sat_i32(floor(z));
    double fx = rem_euclid1(x), fy = rem_euclid1(y), fz = rem_euclid1(z);
    double xb = fx * fx * (3.0 - 2.0 * fx), yb = fy * fy * (3.0 - 2.0 * fy), zb = fz * fz * (3.0 - 2.0 * fz);
    int32_t ix1 = (int32_t)((uint32_t)ix + 1u), iy1 = (int32_t)((uint32_t)iy + 1u), iz1 = (int32_t)((uint32_t)iz + 1u);
    auto lerp = [](double a, double b, double w) { return (1.0 - w) * a + w * b; };  // color.rs:62-64
    double m00 = lerp(noise_at_dev(sc, ix, iy, iz), noise_at_dev(sc, ix1, iy, iz), xb);
    double m01 = lerp(noise_at_dev(sc, ix, iy, iz1), noise_at_dev(sc, ix1, iy, iz1), xb);
    double m10 = lerp(noise_at_dev(sc, ix, iy1, iz), noise_at_dev(sc, ix1, iy1, iz), xb);
    double m11 = lerp(noise_at_dev(sc, ix, iy1, iz1), noise_at_dev(sc, ix1, iy1, iz1), xb);
    double o0 = lerp(m00, m10, yb), o1 = lerp(m01, m11, yb);
    float g = (float)lerp(o0, o1, zb);
    return {g, g, g};
  }
  return {0.f, 1.f, 1.f};
}

// ----------------------------------------------------------------------------------------------- sampling
// vec3.rs:39-47 — uniform point on the unit sphere (theta uniform, z uniform).  v_sin/v_cos take revolutions.
__device__ __forceinline__ f3 sample_unit_sphere(uint32_t r0, uint32_t r1) {
  float u = u01f(r0);
  float z = 2.f * u01f(r1) - 1.f;
  float r = __builtin_sqrtf(fmaxf(0.f, 1.f - z * z));
  return {r * __builtin_amdgcn_cosf(u), r * __builtin_amdgcn_sinf(u), z};
}

// Camera::sample_point (src/camera/mod.rs:176-186) + sample_pixel's sub-pixel position (mod.rs:150-172) for
// sample index s of pixel (j = row, i = column).
__device__ __forceinline__ void generate_ray(const Camera& cam, uint32_t pixel, uint32_t s, uint32_t k0, uint32_t k1,
                                             d3& o, d3& d, double& time) {
  uint32_t j = pixel / cam.width, i = pixel - j * cam.width;
  uint32_t r[4];
  philox4x32_10(pixel, s, 0u, RT_RNG_CAMERA, k0, k1, r);
  double dx, dy;
  {
#pragma clang fp contract(off)
    if (cam.sampler_kind == 0u) {  // Uniform(n): fixed grid, offset as in mod.rs:153-157
      uint32_t n = cam.sampler_n;
      uint32_t yi = s / n, xi = s - yi * n;
      double interval = 1.0 / (double)n;
      double offset = interval / 2.0 + 0.5;
      dy = (double)j + (double)yi * interval - offset;
      dx = (double)i + (double)xi * interval - offset;
    } else {  // Random(n): dy then dx uniform in [-0.5, 0.5) (mod.rs:166-167)
      dy = (double)j + (u01d(r[0]) - 0.5);
      dx = (double)i + (u01d(r[1]) - 0.5);
    }
    d3 pc = {cam.pixel00[0] + dx * cam.du[0] + dy * cam.dv[0], cam.pixel00[1] + dx * cam.du[1] + dy * cam.dv[1],
             cam.pixel00[2] + dx * cam.du[2] + dy * cam.dv[2]};
    if (cam.defocus_angle <= 0.0) {
      o = {cam.center[0], cam.center[1], cam.center[2]};
    } else {
      // uniform point in the unit disk (same distribution as the rejection loop of vec3.rs:48-59)
      float rr = __builtin_sqrtf(u01f(r[2]));
      float u = u01f(r[3]);
      double lx = (double)(rr * __builtin_amdgcn_cosf(u)), ly = (double)(rr * __builtin_amdgcn_sinf(u));
      o = {cam.center[0] + cam.dfu[0] * lx + cam.dfv[0] * ly, cam.center[1] + cam.dfu[1] * lx + cam.dfv[1] * ly,
           cam.center[2] + cam.dfu[2] * lx + cam.dfv[2] * ly};
    }
    d = {pc.x - o.x, pc.y - o.y, pc.z - o.z};
  }
  uint32_t r2[4];
  philox4x32_10(pixel, s, 0u, RT_RNG_CAMERA2, k0, k1, r2);
  time = u01d(r2[0]);  // mod.rs:184
}

// ----------------------------------------------------------------------------------------------- shading
struct ShadeOut {
  bool alive;       // a scattered ray continues
  f3 radiance;      // contribution when the path ends here (already multiplied by throughput)
  bool terminated;  // path ended with a (possibly zero) contribution
  d3 o, d;          // scattered ray
  f3 thr;           // updated throughput
};

// One bounce of ray_color_inner (src/camera/mod.rs:187-210) given the closest hit.
__device__ __forceinline__ ShadeOut shade(const Scene& sc, const Camera& cam, d3 o, d3 d, double time, f3 thr,
                                          HitInfo hit, uint32_t pixel, uint32_t s, uint32_t bounce, uint32_t k0,
                                          uint32_t k1) {
  ShadeOut out;
  out.alive = false;
  out.terminated = true;
  out.radiance = {0.f, 0.f, 0.f};
  out.thr = thr;
  out.o = o; out.d = d;
  if (hit.prim < 0) {  // miss: sky gradient (mod.rs:205-207) or solid background (EXT)
    f3 c;
    if (cam.background_mode == 0u) {
      f3 u = normalize(tof(d));
      float a = 0.5f * (u.y + 1.0f);
      c = {(1.f - a) + a * 0.5f, (1.f - a) + a * 0.7f, (1.f - a) + a * 1.0f};
    } else {
      c = {cam.background[0], cam.background[1], cam.background[2]};
    }
    out.radiance = {thr.x * c.x, thr.y * c.y, thr.z * c.z};
    return out;
  }
  const Prim& P = sc.prims[hit.prim];
  const Material& M = sc.mats[P.mat];
  HitRec h = make_hit(P, o, d, time, hit.t, false);
  uint32_t r[4];
  philox4x32_10(pixel, s, bounce, RT_RNG_SCATTER, k0, k1, r);
  f3 n = tof(h.n);
  f3 att, dir;
  switch (M.kind) {
    case 0u: {  // Lambertian (materials.rs:28-43)
      f3 u = sample_unit_sphere(r[0], r[1]);
      dir = n + u;
      const float th = 1e-9f;
      if (__builtin_fabsf(dir.x) < th && __builtin_fabsf(dir.y) < th && __builtin_fabsf(dir.z) < th) dir = n;
      att = tex_value(sc, M.tex, P, h, o, d, time, hit.t);
      break;
    }
    case 1u: {  // Metal (materials.rs:73-80): reflects the UN-normalised direction, adds fuzz * unit
      f3 df = tof(d);
      f3 refl = df - (2.0f * dot(df, n)) * n;
      f3 u = sample_unit_sphere(r[0], r[1]);
      dir = refl + M.fuzz * u;
      att = {M.albedo[0], M.albedo[1], M.albedo[2]};
      break;
    }
    case 2u: {  // Dielectric (materials.rs:104-141)
      float ratio = h.front ? 1.0f / M.ior : M.ior;
      f3 ud = normalize(tof(d));
      float cos_t = fminf(-dot(ud, n), 1.0f);
      float sin_t = __builtin_sqrtf(fmaxf(0.f, 1.0f - cos_t * cos_t));
      bool cannot = ratio * sin_t > 1.0f;
      float r0 = (1.f - ratio) / (1.f + ratio);
      r0 = r0 * r0;
      float m = 1.f - cos_t;
      float refl_p = r0 + (1.f - r0) * (m * m * m * m * m);
      if (cannot || refl_p > u01f(r[0])) {
        dir = ud - (2.0f * dot(ud, n)) * n;
      } else {  // refract (materials.rs:131-136)
        f3 perp = ratio * (ud + cos_t * n);
        float par = -__builtin_sqrtf(__builtin_fabsf(1.0f - dot(perp, perp)));
        dir = perp + par * n;
      }
      att = {1.f, 1.f, 1.f};
      break;
    }
    case 3u: {  // EXT diffuse light: emits, no scatter
      f3 e = tex_value(sc, M.tex, P, h, o, d, time, hit.t);
      // the reference integrator (background_mode 0) has no emission term
      if (cam.background_mode != 0u) out.radiance = {thr.x * e.x, thr.y * e.y, thr.z * e.z};
      else {
        // a `None` scatter falls through to the sky in the reference (mod.rs:203-207)
        f3 u = normalize(tof(d));
        float a = 0.5f * (u.y + 1.0f);
        out.radiance = {thr.x * ((1.f - a) + a * 0.5f), thr.y * ((1.f - a) + a * 0.7f), thr.z * ((1.f - a) + a)};
      }
      return out;
    }
    default: {  // EXT isotropic
      dir = sample_unit_sphere(r[0], r[1]);
      att = tex_value(sc, M.tex, P, h, o, d, time, hit.t);
      break;
    }
  }
  out.thr = {thr.x * att.x, thr.y * att.y, thr.z * att.z};
  out.o = h.p;
  out.d = tod(dir);
  out.alive = true;
  out.terminated = false;
  return out;
}

}  // namespace rtd
