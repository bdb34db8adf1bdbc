// Standalone timing / phase-trace harness of k_proj_bwd_q (k_projection3.h) and k_proj_bwd_t at BASELINE config 2's shape
// (64 samples x 64 channels x 128 x 128): random operands, HIP-event time per launch, and with -DPBQ_TRACE the average cycles
// between the phase stamps of workgroup 0 per wave.  Results are NOT checked here (tools/pbq_check.py does that through the
// library).   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPBQ_TRACE] tools/pbq_bench.hip -o /tmp/pbq_bench && /tmp/pbq_bench
#include "../pde_policylearning_amd/csrc/k_projection3.h"
#include "../pde_policylearning_amd/csrc/k_projection2.h"
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
  a.wa1 = (const unsigned short*)a.w1; a.wa3 = (const unsigned short*)a.w1;      // (flags only for the q kernel)
  a.PW = PW; a.W = W; a.P = P; a.K2out = K2; a.NJ = NJ; a.CO = 1; a.act_in = 0;
  a.tiles_per_plane = PW / 128; a.ntiles = B * a.tiles_per_plane;
  const size_t lds = proj_bwd_q_lds(HID, W, NJ, true);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_proj_bwd_q<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k_proj_bwd_q<256, false>), dim3(grid), dim3(1024), lds, 0, a);
  CK(hipDeviceSynchronize());
  const int N = 20;
  hipEventRecord(e0);
  for (int it = 0; it < N; ++it) hipLaunchKernelGGL((k_proj_bwd_q<256, false>), dim3(grid), dim3(1024), lds, 0, a);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("k_proj_bwd_q: %.4f ms per launch (grid %d, LDS %zu)\n", ms / N, grid, lds);
#ifdef PBQ_TRACE
  std::vector<unsigned long long> tr(16 * 32 * 16);
  CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_pbq), tr.size() * 8));
  const char* nm[11] = {"commit", "B1 wait", "recompute", "E + dW1", "B2 wait", "dx / idle", "B4 wait", "epilogue", "B5 wait", "(x1g)", "row DFT"};
  for (int w : {0, 1, 3, 4, 7, 8, 12, 15}) {
    double acc[11] = {0}; double tot = 0; int n = 0;
    for (int ht = 4; ht < 31; ++ht) {
      const unsigned long long* r = &tr[(w * 32 + ht) * 16];
      const unsigned long long* rn = &tr[(w * 32 + ht + 1) * 16];
      for (int k = 0; k < 10; ++k) acc[k] += (double)(r[k + 1] - r[k]);
      tot += (double)(rn[0] - r[0]); ++n;
    }
    printf("wave %2d: half tile %7.0f cycles |", w, tot / n);
    for (int k = 0; k < 10; ++k) printf(" %s %5.0f |", nm[k], acc[k] / n);
    double e0 = 0, d0 = 0, e1 = 0, d1 = 0;
    for (int ht = 4; ht < 31; ++ht) {
      const unsigned long long* r = &tr[(w * 32 + ht) * 16];
      e0 += (double)(r[11] - r[3]); d0 += (double)(r[12] - r[11]); e1 += (double)(r[13] - r[12]); d1 += (double)(r[14] - r[13]);
    }
    printf("\n          E(s=0) %5.0f  dW1(s=0) %5.0f  E(s=1) %5.0f  dW1(s=1) %5.0f\n", e0 / n, d0 / n, e1 / n, d1 / n);
  }
#endif
  return 0;
}
