// Latency of a DEPENDENT chain of VALU instructions as a function of the wave's active-lane count (gfx950): is the
// <= 8-lane throughput cliff of lane_count.hip also a latency cliff (what a straggler wave at a frame's end would feel)?
//   hipcc --offload-arch=gfx950 -O3 lane_latency.hip -o lane_latency && ./lane_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP32(X) X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X X

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int n, int iters, int waves_on, unsigned long long* clk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a = lane; double d = 1.0 + lane * 1e-9; const float m = 1.0001f, c = 0.5f;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter();
  if (wave < waves_on && ((lane * 37) & 63) < n) {
    for (int i = 0; i < iters; i++) {
      if (KIND == 0) { REP32(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));) }
      else if (KIND == 1) { REP32(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(m));) }
      else if (KIND == 2) { REP32(asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d));) }
      else { REP32(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(m));) }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + (float)d;
  if (blockIdx.x == 3 && threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int KIND>
int run(const char* name, float* d, unsigned long long* clk, int blocks) {
  const int iters = 2000;
  for (int waves_on : {4, 16}) {   // 1 or 4 waves per SIMD
    for (int n : {64, 16, 9, 8, 4, 1}) {
      hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, n, iters, waves_on, clk);
      CHECK(hipDeviceSynchronize());
      unsigned long long h; CHECK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
      printf("%-10s %d wave(s)/SIMD, %2d lanes: %6.2f cycles per dependent instruction\n", name, waves_on / 4, n, (double)h / (32.0 * iters));
    }
  }
  return 0;
}
int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount;
  float* d; CHECK(hipMalloc(&d, (size_t)blocks * 1024 * 4));
  unsigned long long* clk; CHECK(hipMalloc(&clk, 8));
  if (run<0>("v_fma_f32", d, clk, blocks)) return 1;
  if (run<1>("v_add_u32", d, clk, blocks)) return 1;
  if (run<2>("v_fma_f64", d, clk, blocks)) return 1;
  if (run<3>("v_max_f32", d, clk, blocks)) return 1;
  return 0;
}
