// Standalone check + timing of the third-generation block forward (csrc/k_block_fwd3.h: independent strip waves fed by LDS-DMA)
// against the second generation (k_blk_fwd_t, two-term fp16 mode) on random data: u must be BIT-IDENTICAL, x1 within 1e-6.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bf3_test.bin tools/bf3_test.hip && tools/bf3_test.bin [B] [W] [K2] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <vector>
#include "../pde_policylearning_amd/csrc/k_block_fwd2.h"
#include "../pde_policylearning_amd/csrc/k_block_fwd3.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }
template <typename T> static T* dev(const std::vector<T>& h) { T* d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
static double rel(const std::vector<float>& a, const std::vector<float>& b) {
  double n = 0, d = 0; for (size_t i = 0; i < a.size(); ++i) { n += (double)(a[i] - b[i]) * (a[i] - b[i]); d += (double)b[i] * b[i]; } return std::sqrt(n / (d > 0 ? d : 1));
}
template <int C, bool LIFT, bool AIN, int EPI>
static void run(int B, int W, int K2, int reps) {
  const int P = 128 * 128 / W, PW = P * W, NJ = (2 * K2 + 15) / 16, CL = 3;
  const size_t nact = (size_t)B * C * PW;
  std::vector<float> x(LIFT ? (size_t)B * CL * PW : nact), w((size_t)C * C), bias(C), z((size_t)B * P * K2 * C * 2), tinv((size_t)2 * K2 * W),
      tfwd((size_t)16 * NJ * W, 0.f), lw((size_t)C * CL), lb(C);
  for (auto& v : x) v = 1.5f * frand();
  for (auto& v : w) v = 0.2f * frand();
  for (auto& v : bias) v = 0.1f * frand();
  for (auto& v : z) v = 0.3f * frand();
  for (auto& v : tinv) v = 0.5f * frand();
  for (int j = 0; j < 2 * K2; ++j) for (int i = 0; i < W; ++i) tfwd[(size_t)j * W + i] = 0.1f * frand();
  for (auto& v : lw) v = frand();
  for (auto& v : lb) v = frand();
  PwFwdArgs a; memset(&a, 0, sizeof(a));
  a.x = dev(x); a.w = dev(w); a.bias = dev(bias); a.z = dev(z); a.tinv = dev(tinv); a.tfwd = dev(tfwd);
  float *u0, *u1, *x10, *x11; const size_t nx1 = (size_t)B * P * K2 * C * 2;
  CK(hipMalloc(&u0, nact * 4)); CK(hipMalloc(&u1, nact * 4)); CK(hipMalloc(&x10, nx1 * 4)); CK(hipMalloc(&x11, nx1 * 4));
  CK(hipMemset(u0, 0, nact * 4)); CK(hipMemset(u1, 0, nact * 4)); CK(hipMemset(x10, 0, nx1 * 4)); CK(hipMemset(x11, 0, nx1 * 4));
  a.PW = PW; a.W = W; a.P = P; a.K2in = K2; a.K2out = K2; a.NJ = NJ; a.act_in = AIN; a.act_out = EPI == 2;
  a.tiles_per_plane = PW / 128; a.ntiles = B * a.tiles_per_plane;
  a.loose = getenv("STAG") ? atoi(getenv("STAG")) : 0;
  { float mx = 0.f; for (auto v : x) mx = std::max(mx, std::fabs(v)); std::vector<float> m1(4, mx); a.xmax = dev(m1); std::vector<float> z4(4, 0.f); a.umax = dev(z4); a.ubound = a.umax + 2; }
  if (LIFT) { a.lw = dev(lw); a.lb = dev(lb); a.CL = CL; }
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const size_t lds_old = blk_fwd_t_lds_bytes(C, W, K2, NJ, true, EPI != 0), lds_new = blk_fwd_s_lds_bytes(K2, EPI != 0) + (getenv("LDSPAD") ? atoi(getenv("LDSPAD")) : 0);      // (LDSPAD: occupancy experiment)
  auto kold = k_blk_fwd_t<C, LIFT, false, AIN, EPI, false, 1, 2>;
  auto knew = k_blk_fwd_s<AIN, EPI, LIFT>;
  CK(hipFuncSetAttribute((const void*)kold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
  CK(hipFuncSetAttribute((const void*)knew, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
  const int nthr = (C / 32) * 2 * 64;
  const int gold = std::min(a.ntiles, 2 * ncu), gnew = std::min(a.ntiles, (getenv("GRID") ? atoi(getenv("GRID")) : 2) * ncu);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_old = 0, ms_new = 0;
  for (int pass = 0; pass < 2; ++pass) {
    a.u = u0; a.x1 = EPI ? x10 : nullptr; a.share32 = (getenv("SHARE") ? atoi(getenv("SHARE")) : 18);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kold, dim3(gold), dim3(nthr), lds_old, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_old, e0, e1));
    a.u = u1; a.x1 = EPI ? x11 : nullptr; a.share32 = (getenv("SHARE3") ? atoi(getenv("SHARE3")) : 18);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(knew, dim3(gnew), dim3(nthr), lds_new, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
  }
  CK(hipGetLastError());
  if (getenv("CHAIN")) {      // producer (old kernel, x -> u0) then consumer (new kernel, u0 -> u1): the consumer's time, forward / reverse tile order
    for (int rev = 0; rev < 2; ++rev) {
      float tot = 0.f;
      for (int r = 0; r < reps; ++r) {
        PwFwdArgs p1 = a; p1.u = u0; p1.x1 = EPI ? x10 : nullptr; p1.share32 = 18;
        hipLaunchKernelGGL(kold, dim3(gold), dim3(nthr), lds_old, 0, p1);
        PwFwdArgs p2 = a; p2.x = u0; p2.u = u1; p2.x1 = EPI ? x11 : nullptr; p2.share32 = 18; p2.rev = rev;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(knew, dim3(gnew), dim3(nthr), lds_new, 0, p2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
      }
      printf("   chain: consumer rev=%d %.1f us\n", rev, 1e3 * tot / reps);
    }
    // restore u1 for the comparison below
    a.u = u1; a.x1 = EPI ? x11 : nullptr; a.share32 = 18;
    hipLaunchKernelGGL(knew, dim3(gnew), dim3(nthr), lds_new, 0, a);
    CK(hipDeviceSynchronize());
  }
#ifdef FNO_TRACE
  {   // per-phase shader-clock stamps of workgroup 0 of the LAST launch (the new kernel): cycles per phase, averaged over its tiles
    std::vector<unsigned long long> tr(16 * 256);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_trace), tr.size() * 8));
    const int ntl = std::min(a.ntiles / gnew, 30);
    const char* nm[8] = {"commit", "wait B1", "issue+GEMM", "wait B2", "epilogue", "wait B3", "row DFT", "wait B4"};
    for (int wv = 0; wv < nthr / 64; ++wv) {
      double ph[8] = {0};
      for (int t = 1; t + 1 < ntl; ++t)
        for (int k = 0; k < 8; ++k) {
          const unsigned long long t0 = tr[wv * 256 + 8 * t + k], t1 = tr[wv * 256 + 8 * t + k + 1];
          if (EPI == 0 && k >= 5) { if (k == 5) ph[k] += (double)(tr[wv * 256 + 8 * t + 8] - t0); continue; }
          ph[k] += (double)(t1 - t0);
        }
      double tot = 0; printf("   wave %d cycles/tile:", wv);
      for (int k = 0; k < 8; ++k) { printf(" %s %.0f", nm[k], ph[k] / (ntl - 2)); tot += ph[k] / (ntl - 2); }
      printf("  | total %.0f\n", tot);
    }
  }
#endif
  std::vector<float> h0(nact), h1(nact), g0(nx1), g1(nx1);
  CK(hipMemcpy(h0.data(), u0, nact * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), u1, nact * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g0.data(), x10, nx1 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), x11, nx1 * 4, hipMemcpyDeviceToHost));
  // CPU reference on the first and the last tile (double accumulation)
  double eo = 0, en = 0, den = 0;
  if (!LIFT) {
    const int tiles[2] = {0, a.ntiles - 1};
    for (int tt = 0; tt < 2; ++tt) {
      const int b = tiles[tt] / a.tiles_per_plane, px0 = (tiles[tt] % a.tiles_per_plane) * 128;
      for (int o = 0; o < C; ++o)
        for (int px = 0; px < 128; ++px) {
          double acc = bias[o];
          for (int c = 0; c < C; ++c) {
            double v = x[((size_t)b * C + c) * PW + px0 + px];
            if (AIN) v = 0.5 * v * (1.0 + std::erf(v / std::sqrt(2.0)));
            acc += (double)w[o * C + c] * v;
          }
          const int row = (px0 + px) / W, wc = (px0 + px) % W;
          for (int s2 = 0; s2 < K2; ++s2)
            for (int ri = 0; ri < 2; ++ri)
              acc += (double)z[((((size_t)b * P + row) * K2 + s2) * C + o) * 2 + ri] * tinv[(size_t)(2 * s2 + ri) * W + wc];
          const size_t idx = ((size_t)b * C + o) * PW + px0 + px;
          eo += (h0[idx] - acc) * (h0[idx] - acc); en += (h1[idx] - acc) * (h1[idx] - acc); den += acc * acc;
        }
    }
    printf("   vs CPU (2 tiles): old %.2e  new %.2e\n", std::sqrt(eo / den), std::sqrt(en / den));
  }
  size_t bad = 0, first = 0; for (size_t i = 0; i < nact; ++i) if (memcmp(&h0[i], &h1[i], 4) != 0) { if (!bad) first = i; ++bad; }
  printf("C=%d LIFT=%d AIN=%d EPI=%d B=%d W=%d K2=%d lds old/new %zu/%zu: u rel %.2e (bitwise-different %zu, first at b %zu c %zu px %zu: %g vs %g)  x1 rel %.2e | old %.1f us  new %.1f us\n",
         C, LIFT, AIN, EPI, B, W, K2, lds_old, lds_new, rel(h1, h0), bad, first / ((size_t)C * PW), (first / PW) % C, first % PW, bad ? h1[first] : 0.f,
         bad ? h0[first] : 0.f, EPI ? rel(g1, g0) : 0.0, 1e3 * ms_old / reps, 1e3 * ms_new / reps);
  hipFree((void*)a.x); hipFree((void*)a.w); hipFree((void*)a.z); hipFree(u0); hipFree(u1); hipFree(x10); hipFree(x11);
}
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, W = 128, K2 = argc > 2 ? atoi(argv[2]) : 7, reps = argc > 3 ? atoi(argv[3]) : 20;
  const int which = argc > 4 ? atoi(argv[4]) : -1;      // run one configuration only (profiling)
  if (which < 0 || which == 0) run<64, false, true, 2>(B, W, K2, reps);
  if (which < 0 || which == 1) run<64, false, true, 1>(B, W, K2, reps);
  if (which < 0 || which == 2) run<64, false, false, 0>(B, W, K2, reps);
  if (which < 0 || which == 3) run<64, false, true, 0>(B, W, K2, reps);
  if (which < 0 || which == 4) run<64, false, false, 2>(B, W, K2, reps);
  if (which < 0 || which == 5) run<64, true, false, 2>(B, W, K2, reps);
  return 0;
}
