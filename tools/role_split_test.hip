// Can ONE vector wave per SIMD do the projection backward's whole GELU' / split phase at the SIMD's full issue rate while a
// SECOND wave of the same SIMD keeps the matrix pipe busy?  (Premise of a producer / consumer split of k_proj_bwd_t: round 5.)
// 512-thread workgroups, one per CU: waves 0-3 ("matrix") issue fp16 MFMAs from registers, waves 4-7 ("vector") run the E phase
// of the projection backward on P1 values read from LDS (fp32, 16 per lane and step) and write the two fp16 terms of dP1 back,
// ILP = 1, 2 or 4 groups of four values in flight.  Prints cycles per element-step of the vector waves alone, beside the
// matrix waves, and the MFMA rate.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/role_split_test.hip -o /tmp/role_split && /tmp/role_split
#include "../pde_policylearning_amd/csrc/fno_dev.h"
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void __launch_bounds__(512) k(int mode, int iters, float* out, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8192; i += 512) smem[i] = 0.001f * (float)((i * 37) % 4001) - 2.f;
  __syncthreads();
  const bool matrix = wave < 4;
  const long long t0 = __builtin_readcyclecounter();
  float acc = 0.f;
  if (matrix) {
    if (mode & 1) {
      f32x16 a0 = {0}, a1 = {0};
      f16x8 fa, fb;
      for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(1 + lane % 7); fb[i] = (_Float16)(1 + i); }
      // per vector step (16 values per lane = a 32 x 32 tile): 12 (recompute) + 12 (dx) + 12 (dW1) products
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 18; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, a1, 0, 0, 0);
        }
      for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r];
    }
  } else if (mode & 2) {
    const float* src = smem + (wave - 4) * 1024 + lane * 16;
    unsigned* dst = reinterpret_cast<unsigned*>(smem + 4096 + (wave - 4) * 1024) + lane * 8;
    const float b1v = 0.01f * lane, w2v = 0.3f, sd = 2048.f;
    float sdb = 0.f, sdw = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g0 = 0; g0 < 4; g0 += ILP) {
        float4 p[ILP], dg[ILP];
#pragma unroll
        for (int g = 0; g < ILP; ++g) {
          p[g] = ld4(src + 4 * (g0 + g));
          p[g] = make_float4(fmaf(p[g].x, 1.0001f, b1v), fmaf(p[g].y, 1.0001f, b1v), fmaf(p[g].z, 1.0001f, b1v), fmaf(p[g].w, 1.0001f, b1v));
        }
#pragma unroll
        for (int g = 0; g < ILP; ++g) gelu_both4(p[g], dg[g]);
#pragma unroll
        for (int g = 0; g < ILP; ++g) {
          const float dyv[4] = {0.5f, -0.25f, 0.125f, 1.f};
          const float gl4[4] = {p[g].x, p[g].y, p[g].z, p[g].w}, dg4[4] = {dg[g].x, dg[g].y, dg[g].z, dg[g].w};
          float dp[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { dp[j] = dg4[j] * (w2v * dyv[j]); sdw = fmaf(gl4[j], dyv[j], sdw); sdb += dp[j]; }
#ifdef SCALAR_SPLIT
          _Float16 hh[4], ll[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = dp[j] * sd;
            asm volatile("" : "+v"(v));
            hh[j] = (_Float16)v;
            float r = v - (float)hh[j];
            asm volatile("" : "+v"(r));
            ll[j] = (_Float16)r;
          }
          const f16x2 h0 = {hh[0], hh[1]}, h1 = {hh[2], hh[3]}, l0 = {ll[0], ll[1]}, l1 = {ll[2], ll[3]};
#else
          const f32x2 v0 = natural_pair(dp[0], dp[1]) * f32x2{sd, sd}, v1 = natural_pair(dp[2], dp[3]) * f32x2{sd, sd};
          const f16x2 h0 = __builtin_convertvector(v0, f16x2), h1 = __builtin_convertvector(v1, f16x2);
          const f16x2 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, f32x2), f16x2);
          const f16x2 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, f32x2), f16x2);
#endif
          *reinterpret_cast<uint2*>(dst + 2 * ((g0 + g) & 1)) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
          *reinterpret_cast<uint2*>(dst + 4 + 2 * ((g0 + g) & 1)) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
        }
      }
      asm volatile("" : "+v"(sdb), "+v"(sdw));
    }
    acc = sdb + sdw;
  }
  const long long t1 = __builtin_readcyclecounter();
  out[(size_t)blockIdx.x * 512 + tid] = acc;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int ILP>
static void run(float* out, long long* cyc) {
  const int iters = 2000;
  for (int mode : {2, 1, 3}) {
    hipLaunchKernelGGL(k<ILP>, dim3(256), dim3(512), 32768, 0, mode, iters, out, cyc);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    m /= 1024; v /= 1024;
    printf("ILP %d  %-22s vector wave: %7.1f cycles per step of 16 values (%.2f per value)   matrix wave: %6.1f cycles per MFMA\n", ILP,
           mode == 2 ? "vector waves alone" : mode == 1 ? "matrix waves alone" : "both", v / iters, v / iters / 16, m / iters / 36);
  }
}
int main() {
  float* out; long long* cyc;
  if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess || hipMalloc(&cyc, 256 * 8 * 8) != hipSuccess) return 1;
  run<1>(out, cyc);
  run<2>(out, cyc);
  run<4>(out, cyc);
  return 0;
}
