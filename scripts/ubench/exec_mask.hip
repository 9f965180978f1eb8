// Does a wave64 VALU instruction get cheaper when whole 16-lane quarters (or the upper half) of EXEC are off?
// One 256-thread workgroup per SIMD slot x 4 waves per SIMD, a long chain of independent v_fma_f32 under `lane < n`
// or `(lane & 15) < n` masks.   hipcc --offload-arch=gfx950 -O3 exec_mask.hip -o exec_mask && ./exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int n_active, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  bool on;
  if (MODE == 0) on = lane < n_active;                 // contiguous low lanes
  else if (MODE == 1) on = (lane & 15) < n_active;     // the same count spread over all four quarters
  else on = ((lane * 37) & 63) < n_active;             // scattered
  float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
  const float m = 1.0001f, c = 0.5f;
  if (on) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
        a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

template <int MODE>
int run(const char* name, float* d, int blocks, int iters) {
  unsigned long long* clk; CHECK(hipMalloc(&clk, 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int n : {64, 32, 16, 12, 9, 8, 4, 1, 64}) {
    const int nn = MODE == 1 ? (n + 3) / 4 : n;  // per quarter
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, nn, 16, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, nn, iters, clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // 4 waves per SIMD, 64 FMA wave-instructions per iteration each
    const double instr_per_simd = 4.0 * 64.0 * iters;
    unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("%-10s active %2d/64: %.3f ms  -> %.2f ns per FMA per SIMD; wave 0: %llu shader cycles, %llu wall ticks (100 MHz) -> %.0f MHz\n", name,
           MODE == 1 ? nn * 4 : n, ms, ms * 1e6 / instr_per_simd, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0));
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount;  // one 1024-thread workgroup per CU = 4 waves per SIMD
  float* d; CHECK(hipMalloc(&d, (size_t)blocks * 1024 * 4));
  printf("%s, %d CUs, clock %d MHz\n", p.name, blocks, p.clockRate / 1000);
  if (run<0>("low-lanes", d, blocks, 20000)) return 1;
  if (run<1>("quarters", d, blocks, 20000)) return 1;
  if (run<2>("scattered", d, blocks, 20000)) return 1;
  CHECK(hipFree(d));
  return 0;
}
