// Issue cost of VALU instruction kinds on gfx950, in units of the plain v_fma_f32 (one wave64 instruction per 4 cycles
// and SIMD): kernel time of 16 independent chains x 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  float a[16]; double d[8]; unsigned long long q[8];
  for (int i = 0; i < 16; i++) a[i] = lane + i;
  for (int i = 0; i < 8; i++) { d[i] = 1.0 + 1e-9 * (lane + i); q[i] = lane + i; }
  const float m = 1.0001f, c = 0.5f;
  if (KIND == 33) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
  if (KIND == 31) asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 1) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 2) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 3) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i & 7]) : "v"(m), "v"(c) : "vcc");
        REP16(X)
#undef X
      } else if (KIND == 4) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        REP16(X)
#undef X
      } else if (KIND == 5) {
#define X(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
        REP16(X)
#undef X
      } else if (KIND == 6) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
        REP16(X)
#undef X
      } else if (KIND == 7) {
#define X(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i & 7]));
        REP16(X)
#undef X
      } else if (KIND == 8) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 9) {
#define X(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i & 7]));
        REP16(X)
#undef X
      } else if (KIND == 11) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
        REP16(X)
#undef X
      } else if (KIND == 12) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 13) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 14) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
        REP16(X)
#undef X
      } else if (KIND == 15) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 16) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 17) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 18) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 19) {
#define X(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 20) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 21) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 22) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 23) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
        REP16(X)
#undef X
      } else if (KIND == 24) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
        REP16(X)
#undef X
      } else if (KIND == 25) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 26) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[i]));
        REP16(X)
#undef X
      } else if (KIND == 27) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 28) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
        REP16(X)
#undef X
      } else if (KIND == 29) {
#define X(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i & 7]), "v"(d[(i + 1) & 7]) : "vcc");
        REP16(X)
#undef X
      } else if (KIND == 30) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 31) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(m) : "s20", "s21");
        REP16(X)
#undef X
      } else if (KIND == 32) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
        REP16(X)
#undef X
      } else if (KIND == 33) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 34) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 35) {
#define X(i) asm volatile("v_bfe_i32 %0, %0, 0, 16" : "+v"(a[i]));
        REP16(X)
#undef X
      } else if (KIND == 36) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 37) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 38) {
#define X(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 39) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 40) {
#define X(i) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(a[i]));
        REP16(X)
#undef X
      } else if (KIND == 41) {
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 42) {
#define X(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 43) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(m), "v"(c));
        REP16(X)
#undef X
      } else if (KIND == 44) {
#define X(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
        REP16(X)
#undef X
      } else if (KIND == 45) {
#define X(i) asm volatile("v_max_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 46) {
#define X(i) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        REP16(X)
#undef X
      } else if (KIND == 10) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        REP16(X)
#undef X
      }
    }
  }
  float s = 0; for (int i = 0; i < 16; i++) s += a[i]; for (int i = 0; i < 8; i++) s += (float)d[i] + (float)q[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
int run(const char* name, float* d, int blocks, double& base) {
  const int iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, 8);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ns = ms * 1e6 / (4.0 * 64.0 * iters);  // per wave-instruction per SIMD
  if (KIND == 0) base = ns;
  printf("%-14s %.2f ns per wave-instruction per SIMD = %.2f x v_fma_f32\n", name, ns, ns / base);
  return 0;
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount;
  float* d; CHECK(hipMalloc(&d, (size_t)blocks * 1024 * 4));
  double base = 1;
  if (run<0>("v_fma_f32", d, blocks, base)) return 1;
  if (run<1>("v_mul_lo_u32", d, blocks, base)) return 1;
  if (run<2>("v_mul_hi_u32", d, blocks, base)) return 1;
  if (run<3>("v_mad_u64_u32", d, blocks, base)) return 1;
  if (run<8>("v_mul_u32_u24", d, blocks, base)) return 1;
  if (run<4>("v_rcp_f32", d, blocks, base)) return 1;
  if (run<10>("v_sqrt_f32", d, blocks, base)) return 1;
  if (run<5>("v_sin_f32", d, blocks, base)) return 1;
  if (run<6>("v_fma_f64", d, blocks, base)) return 1;
  if (run<7>("v_rsq_f64", d, blocks, base)) return 1;
  if (run<9>("v_cvt_f32_f64", d, blocks, base)) return 1;
  if (run<11>("v_pk_fma_f32", d, blocks, base)) return 1;
  if (run<12>("v_max_f32", d, blocks, base)) return 1;
  if (run<13>("v_max3_f32", d, blocks, base)) return 1;
  if (run<14>("v_cmp_le_f32", d, blocks, base)) return 1;
  if (run<15>("v_cndmask_b32", d, blocks, base)) return 1;
  if (run<16>("v_lshl_add_u32", d, blocks, base)) return 1;
  if (run<17>("v_add_u32", d, blocks, base)) return 1;
  if (run<18>("v_mov_b32", d, blocks, base)) return 1;
  if (run<19>("v_max_u32", d, blocks, base)) return 1;
  if (run<20>("v_add_f32", d, blocks, base)) return 1;
  if (run<21>("v_mul_f32", d, blocks, base)) return 1;
  if (run<22>("v_fmac_f32", d, blocks, base)) return 1;
  if (run<23>("v_add_f64", d, blocks, base)) return 1;
  if (run<24>("v_mul_f64", d, blocks, base)) return 1;
  if (run<25>("v_xor_b32", d, blocks, base)) return 1;
  if (run<26>("v_alignbit_b32", d, blocks, base)) return 1;
  if (run<27>("v_min3_f32", d, blocks, base)) return 1;
  if (run<28>("v_pk_mul_f32", d, blocks, base)) return 1;
  if (run<29>("v_cmp_lt_f64", d, blocks, base)) return 1;
  if (run<30>("v_and_b32", d, blocks, base)) return 1;
  if (run<31>("cndmask_e64_sgpr", d, blocks, base)) return 1;
  if (run<32>("cmp+cndmask", d, blocks, base)) return 1;
  if (run<33>("cndmask_vcc_init", d, blocks, base)) return 1;
  if (run<34>("v_min_f32", d, blocks, base)) return 1;
  if (run<35>("v_bfe_i32", d, blocks, base)) return 1;
  if (run<36>("v_and_or_b32", d, blocks, base)) return 1;
  if (run<37>("v_min_u32", d, blocks, base)) return 1;
  if (run<38>("v_sub_f32", d, blocks, base)) return 1;
  if (run<39>("v_fma_f32_neg", d, blocks, base)) return 1;
  if (run<40>("v_lshlrev_b32", d, blocks, base)) return 1;
  if (run<41>("v_add3_u32", d, blocks, base)) return 1;
  if (run<42>("v_or_b32", d, blocks, base)) return 1;
  if (run<43>("v_fma_mix_f32", d, blocks, base)) return 1;
  if (run<44>("v_mov_b64", d, blocks, base)) return 1;
  if (run<45>("v_max_f32_e64", d, blocks, base)) return 1;
  if (run<46>("v_mul_f32_e64", d, blocks, base)) return 1;
  return 0;
}
