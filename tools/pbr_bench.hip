// Standalone timing harness of k_proj_bwd_r (tools/experiments/k_proj_bwd_roles.h: wave roles) beside k_proj_bwd_t at BASELINE config 2's shape
// (64 samples x 64 channels x 128 x 128): random operands, HIP-event time per launch, and with -DPBQ_TRACE the average cycles
// between the phase stamps of workgroup 0 per wave.  Results are NOT checked here (tools/pbq_check.py does that through the
// library).   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPBQ_TRACE] tools/pbq_bench.hip -o /tmp/pbq_bench && /tmp/pbq_bench
#ifdef USE_R3
#include "experiments/k_proj_bwd_roles3.h"
#else
#include "experiments/k_proj_bwd_roles.h"
#endif
#if defined(USE_R3)
#define KERN k_proj_bwd_r3<256, false>
#define KNAME "k_proj_bwd_r3 (two vector + one matrix wave per SIMD)"
#define NTHREADS 768
#elif !defined(USE_T)
#define KERN k_proj_bwd_r<256, false>
#define KNAME "k_proj_bwd_r"
#else
#define KERN k_proj_bwd_t<64, 256, false, 2>
#define KNAME "k_proj_bwd_t"
#endif
#ifndef NTHREADS
#define NTHREADS 512
#endif
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.f - 1.f); }
  float* d; if (hipMalloc(&d, n * 4) != hipSuccess) exit(1);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}
int main(int argc, char** argv) {
  const int B = 64, C = 64, W = 128, P = 128, PW = W * P, HID = 256, K2 = 6, NJ = 1;
  const int grid = argc > 1 ? atoi(argv[1]) : 256;
  ProjBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.x = dev_rand((size_t)B * C * PW, 2.f, 1); a.dy = dev_rand((size_t)B * PW, 1.f, 2);
  a.w1 = dev_rand((size_t)HID * C, 0.125f, 3); a.b1 = dev_rand(HID, 0.1f, 4); a.w2 = dev_rand(HID, 0.06f, 5);
  a.gout = dev_rand((size_t)B * C * PW, 0.f, 6); a.x1g = dev_rand((size_t)B * P * K2 * C * 2, 0.f, 7);
  a.tfwd = dev_rand((size_t)16 * NJ * W, 1.f, 8);
  a.dw1_part = dev_rand((size_t)grid * HID * C, 0.f, 9); a.db1_part = dev_rand((size_t)grid * 8 * HID, 0.f, 10);
  a.dw2_part = dev_rand((size_t)grid * 8 * HID, 0.f, 11);
  float am[4] = {0.f, 1.f, 0.125f, 0.06f}, xm = 2.f;
  float* amax; CK(hipMalloc(&amax, 16)); CK(hipMemcpy(amax, am, 16, hipMemcpyHostToDevice));
  float* xmax; CK(hipMalloc(&xmax, 4)); CK(hipMemcpy(xmax, &xm, 4, hipMemcpyHostToDevice));
  float* gmax; CK(hipMalloc(&gmax, 4)); CK(hipMemset(gmax, 0, 4));
  a.amax = amax; a.xmax = xmax; a.gmax_out = gmax;
  unsigned short *wa1, *wa3;
  CK(hipMalloc(&wa1, (size_t)(HID / 32) * (C / 16) * 2 * 64 * 16)); CK(hipMalloc(&wa3, (size_t)(HID / 16) * (C / 32) * 2 * 64 * 16));
  {
    const int nitems = (HID / 32) * (C / 16) * 64 + (HID / 16) * (C / 32) * 64;
    hipLaunchKernelGGL(k_pack_w1_t<2>, dim3((nitems + 255) / 256), dim3(256), 0, 0, a.w1, wa1, wa3, HID, C, (const float*)(amax + 2));
  }
  a.wa1 = wa1; a.wa3 = wa3;
  a.PW = PW; a.W = W; a.P = P; a.K2out = K2; a.NJ = NJ; a.CO = 1; a.act_in = 0;
  a.tiles_per_plane = PW / 128; a.ntiles = B * a.tiles_per_plane;
  #if defined(USE_R3)
  const size_t lds = proj_bwd_r3_lds(W, NJ, true);
#elif defined(USE_T)
  const size_t lds = (size_t)2 * C * 256 + 2 * 2 * 64 * 256 + 128 * 4 + (size_t)16 * NJ * (W + 4) * 4;      // pbwd_t_lds (fno_abi.hip)
#else
  const size_t lds = proj_bwd_r_lds(W, NJ, true);
#endif
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((KERN), dim3(grid), dim3(NTHREADS), lds, 0, a);
  CK(hipDeviceSynchronize());
  const int N = 20;
  hipEventRecord(e0);
  for (int it = 0; it < N; ++it) hipLaunchKernelGGL((KERN), dim3(grid), dim3(NTHREADS), lds, 0, a);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.4f ms per launch (grid %d, LDS %zu)\n", KNAME, ms / N, grid, lds);

#ifdef PBR_TRACE
  {
    std::vector<unsigned long long> tr(12 * 8 * 8 * 4);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_pbr), tr.size() * 8));
    for (int w : {0, 3, 4, 7, NTHREADS / 64 - 1}) {
      printf("wave %d (%s):\n", w, w < 4 ? "matrix" : "vector");
      for (int t = 2; t < 5; ++t) {
        const unsigned long long* r = &tr[((w * 8 + t) * 8) * 4];
        const unsigned long long* rn = &tr[((w * 8 + t + 1) * 8) * 4];
        printf("  tile %d: total %6lld |", t, (long long)(rn[0] - r[0]));
        for (int s = 0; s < 6; ++s) printf(" s%d wait %5lld work %5lld |", s, (long long)(r[s * 4 + 1] - r[s * 4]), (long long)(r[s * 4 + 2] - r[s * 4 + 1]));
        printf(" tail %5lld\n", (long long)(rn[0] - r[6 * 4]));
      }
    }
  }
#endif
  return 0;
}
