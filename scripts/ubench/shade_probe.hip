// Isolates rtd::shade on synthetic hits (one lane each): prints what it returns for a Lambertian quad, a light quad and an
// isotropic medium hit.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<dir of rt_device.h> shade_probe.hip -o shade_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "rt_device.h"
using namespace rtd;
struct Out { float thr[3], rad[3]; double o[3], d[3]; int alive; };
template <uint32_t F>
__global__ void k(Scene sc, Camera cam, Out* out) {
  const int i = threadIdx.x;
  if (i >= 3) return;
  GlobalAcc acc{sc.nodes, sc.prims};
  d3 o = {278., 278., -800.}, d = {0.01, 0.02, 1.0};
  f3 thr = {0.5f, 0.25f, 1.0f};
  HitInfo hit; hit.prim = i; hit.t = 900.0;
  PathRng rng = {12345u + i, 6789u};
  ShadeOut so = shade<F>(sc, acc, cam, o, d, 0.5, thr, hit, rng);
  Out r;
  r.thr[0] = so.thr.x; r.thr[1] = so.thr.y; r.thr[2] = so.thr.z;
  r.rad[0] = so.radiance.x; r.rad[1] = so.radiance.y; r.rad[2] = so.radiance.z;
  r.o[0] = so.o.x; r.o[1] = so.o.y; r.o[2] = so.o.z; r.d[0] = so.d.x; r.d[1] = so.d.y; r.d[2] = so.d.z;
  r.alive = so.alive;
  out[i] = r;
}
int main() {
  Prim hp[3]; memset(hp, 0, sizeof(hp));
  hp[0].type = 2; hp[0].mat = 0; hp[0].f[11] = -1.0;  // quad, normal (0,0,-1)
  hp[1].type = 2; hp[1].mat = 1; hp[1].f[11] = -1.0;
  hp[2].type = 4; hp[2].mat = 2;                       // medium box
  Material hm[3]; memset(hm, 0, sizeof(hm));
  hm[0].kind = 0; hm[0].tex = -1; hm[0].albedo[0] = 0.73f; hm[0].albedo[1] = 0.5f; hm[0].albedo[2] = 0.25f;
  hm[1].kind = 3; hm[1].tex = -1; hm[1].albedo[0] = 7.f; hm[1].albedo[1] = 6.f; hm[1].albedo[2] = 5.f;
  hm[2].kind = 4; hm[2].tex = -1; hm[2].albedo[0] = 0.f; hm[2].albedo[1] = 0.125f; hm[2].albedo[2] = 1.f;
  Texture ht[1]; memset(ht, 0, sizeof(ht));
  Scene sc; memset(&sc, 0, sizeof(sc));
  Prim* dp; Material* dm; Texture* dt; Out* dout;
  hipMalloc(&dp, sizeof(hp)); hipMalloc(&dm, sizeof(hm)); hipMalloc(&dt, sizeof(ht)); hipMalloc(&dout, 3 * sizeof(Out));
  hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice); hipMemcpy(dm, hm, sizeof(hm), hipMemcpyHostToDevice);
  hipMemcpy(dt, ht, sizeof(ht), hipMemcpyHostToDevice);
  sc.prims = dp; sc.mats = dm; sc.texs = dt; sc.n_prims = 3; sc.n_mats = 3; sc.n_texs = 1;
  Camera cam; memset(&cam, 0, sizeof(cam)); cam.background_mode = 1;
  for (int f = 0; f < 2; f++) {
    if (f == 0) hipLaunchKernelGGL(k<RT_F_QUAD | RT_F_MEDIUM>, dim3(1), dim3(64), 0, 0, sc, cam, dout);
    else hipLaunchKernelGGL(k<RT_F_ALL>, dim3(1), dim3(64), 0, 0, sc, cam, dout);
    Out h[3]; hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[3] = {"lambertian", "light", "isotropic"};
    for (int i = 0; i < 3; i++)
      printf("F=%s %-10s alive %d thr %.6g %.6g %.6g rad %.6g %.6g %.6g o %.10g %.10g %.10g d %.8g %.8g %.8g\n", f ? "ALL" : "24", nm[i], h[i].alive,
             h[i].thr[0], h[i].thr[1], h[i].thr[2], h[i].rad[0], h[i].rad[1], h[i].rad[2], h[i].o[0], h[i].o[1], h[i].o[2], h[i].d[0], h[i].d[1], h[i].d[2]);
  }
  return 0;
}
