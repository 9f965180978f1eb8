// VALU cost as a function of the number of ACTIVE lanes of the issuing wave (gfx950).
// exec_mask.hip showed a 4x cliff at <= 8 active lanes with v_pk_fma_f32; this one sweeps instruction kinds (inline asm,
// 16 independent chains per wave) and mixes wave kinds on one SIMD: three waves of each SIMD run with nA lanes, the
// fourth with nB, and every wave reports its own shader-cycle count.
//   hipcc --offload-arch=gfx950 -O3 lane_count.hip -o lane_count && ./lane_count
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__device__ __forceinline__ void body(float (&a)[16], double (&d)[8], float m, float c) {
  if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    REP16(X)
#undef X
  } else if (KIND == 1) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
    REP16(X)
#undef X
  } else if (KIND == 2) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
    REP16(X)
#undef X
  } else if (KIND == 3) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
    REP16(X)
#undef X
  } else if (KIND == 4) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : );
    REP16(X)
#undef X
  } else if (KIND == 5) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
    REP16(X)
#undef X
  }
}

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int nA, int nB, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (wave >> 2) == 3 ? nB : nA;
  float a[16]; double d[8];
  for (int i = 0; i < 16; i++) a[i] = lane + i;
  for (int i = 0; i < 8; i++) d[i] = 1.0 + 1e-9 * (lane + i);
  const float m = 1.0001f, c = 0.5f;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  if (((lane * 37) & 63) < n) {
    for (int i = 0; i < iters; i++) { body<KIND>(a, d, m, c); body<KIND>(a, d, m, c); body<KIND>(a, d, m, c); body<KIND>(a, d, m, c); }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 16; i++) s += a[i]; for (int i = 0; i < 8; i++) s += (float)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 7 && lane == 0) clk[wave] = c1 - c0;
}

template <int KIND>
int run(const char* name, float* d, unsigned long long* clk, int blocks, int iters) {
  const int cases[][2] = {{64, 64}, {32, 32}, {16, 16}, {9, 9}, {8, 8}, {4, 4}, {1, 1}, {64, 4}, {64, 8}, {64, 9}, {4, 64}, {16, 4}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, cs[0], cs[1], iters, clk);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[16]; CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double a = 0, b = 0;
    for (int w = 0; w < 12; w++) a += h[w] / 12.0;
    for (int w = 12; w < 16; w++) b += h[w] / 4.0;
    const double per = 64.0 * iters;  // wave-instructions per wave
    printf("%-12s 3 waves/SIMD with %2d lanes: %6.2f cycles per instruction per wave | 1 wave/SIMD with %2d lanes: %6.2f\n", name, cs[0], a / per, cs[1], b / per);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount;
  float* d; CHECK(hipMalloc(&d, (size_t)blocks * 1024 * 4));
  unsigned long long* clk; CHECK(hipMalloc(&clk, 16 * 8));
  const int iters = 4000;
  if (run<0>("v_fma_f32", d, clk, blocks, iters)) return 1;
  if (run<1>("v_add_u32", d, clk, blocks, iters)) return 1;
  if (run<2>("v_max_f32", d, clk, blocks, iters)) return 1;
  if (run<3>("v_fma_f64", d, clk, blocks, iters)) return 1;
  if (run<4>("v_cndmask", d, clk, blocks, iters)) return 1;
  if (run<5>("v_mov_b32", d, clk, blocks, iters)) return 1;
  return 0;
}
