// Do matrix instructions of DIFFERENT input types issued by the two waves of a SIMD disturb each other?
// Workgroups of 8 waves (two per SIMD): waves 0-3 run chains of v_mfma_f32_32x32x16_f16, waves 4-7 chains of
// v_mfma_f32_32x32x16_bf16 (mode 1), or all f16 (mode 0), or f16 with fragments re-read from LDS every step (mode 2 / 3:
// same / mixed types) - the access pattern of the block-forward kernel's GEMM loop.  Every product is exact in fp32
// (small integers), so any accumulator that differs from its expected value is a hardware / hazard effect.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_mix tools/mfma_mix_test.hip ; run: /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NT>
__global__ void __launch_bounds__(NT) k(int iters, unsigned* bad, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[2][64 * 8 * 8];           // per type: 8 fragments x 64 lanes x 8 halves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool use_bf = (MODE & 1) && (NT == 512 ? wave >= 4 : (blockIdx.x & 1));      // 4-wave workgroups: every second workgroup
  // fragments: value (1 + (j & 1)) in both formats (1.0 / 2.0)
  for (int i = tid; i < 64 * 8 * 8; i += NT) {
    const int j = i & 7, f = i / (64 * 8);
    if (MODE >= 2) {      // fragment f holds the value 1 + f % 3 in every element: a fragment read too late / too early changes the sum
      const int v = 1 + f % 3;
      img[0][i] = v == 1 ? 0x3c00 : v == 2 ? 0x4000 : 0x4200;      // fp16 1, 2, 3
      img[1][i] = v == 1 ? 0x3f80 : v == 2 ? 0x4000 : 0x4040;      // bf16 1, 2, 3
    } else {
      img[0][i] = (j & 1) ? 0x4000 : 0x3c00;      // fp16 2.0 / 1.0
      img[1][i] = (j & 1) ? 0x4000 : 0x3f80;      // bf16 2.0 / 1.0
    }
  }
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const unsigned short* src = img[use_bf ? 1 : 0] + lane * 8;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(src), b = *reinterpret_cast<const bf16x8*>(src + 64 * 8);
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 2) {      // re-read the fragments right behind the products that used them
      const int f = (it & 3) * 2;
      a = *reinterpret_cast<const bf16x8*>(src + f * 64 * 8);
      b = *reinterpret_cast<const bf16x8*>(src + (f + 1) * 64 * 8);
    }
    if (use_bf) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
    } else {
      const f16x8 ah = __builtin_bit_cast(f16x8, a), bh = __builtin_bit_cast(f16x8, b);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
    }
  }
  // each product: sum over k = 16 of a_k b_k with a_k = b_k in {1, 2} alternating per k -> 8 * 1 + 8 * 4 = 40
  float e0 = 80.f * iters, e1 = 40.f * iters;
  if (MODE >= 2) {
    e0 = 0.f; e1 = 0.f;
    for (int it = 0; it < iters; ++it) {
      const int f = (it & 3) * 2;
      const float p = 16.f * (1 + f % 3) * (1 + (f + 1) % 3);
      e0 += 2.f * p; e1 += p;
    }
  }
  unsigned nb = 0;
  for (int r = 0; r < 16; ++r) nb += (acc0[r] != e0) + (acc1[r] != e1);
  if (nb) atomicAdd(bad + (use_bf ? 1 : 0), nb);
  if (blockIdx.x == 0 && tid == 0) { out[0] = acc0[0]; out[1] = e0; }
}

// one wave ALTERNATES fp16 and bf16 products (independent accumulator chains), fragments re-read from LDS behind them, as the
// block-forward kernel's two-term variant does (main GEMM fp16, spectral extension bf16 interleaved by the scheduler)
template <int NT>
__global__ void __launch_bounds__(NT) k_alt(int iters, unsigned* bad, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[2][64 * 8 * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 8 * 8; i += NT) {
    const int f = i / (64 * 8), v = 1 + f % 3;
    img[0][i] = v == 1 ? 0x3c00 : v == 2 ? 0x4000 : 0x4200;
    img[1][i] = v == 1 ? 0x3f80 : v == 2 ? 0x4000 : 0x4040;
  }
  __syncthreads();
  f32x16 acc0, acc1, acc2;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
  const unsigned short* sh = img[0] + lane * 8;
  const unsigned short* sb = img[1] + lane * 8;
  float e0 = 0.f, e1 = 0.f, e2 = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int f = (it & 3) * 2;
    const f16x8 ah = *reinterpret_cast<const f16x8*>(sh + f * 64 * 8), bh = *reinterpret_cast<const f16x8*>(sh + (f + 1) * 64 * 8);
    const bf16x8 ab = *reinterpret_cast<const bf16x8*>(sb + f * 64 * 8), bb = *reinterpret_cast<const bf16x8*>(sb + (f + 1) * 64 * 8);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc2, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
    const float p = 16.f * (1 + f % 3) * (1 + (f + 1) % 3);
    e0 += 2.f * p; e1 += 2.f * p; e2 += p;
  }
  unsigned nb = 0;
  for (int r = 0; r < 16; ++r) nb += (acc0[r] != e0) + (acc1[r] != e1) + (acc2[r] != e2);
  if (nb) atomicAdd(bad, nb);
  if (blockIdx.x == 0 && tid == 0) { out[0] = acc1[0]; out[1] = e1; }
}

int main() {
  unsigned* bad; float* out;
  hipMalloc(&bad, 8); hipMalloc(&out, 8);
  const int iters = 4000, grid = 256 * 1, reps = 50;
  for (int wg4 = 0; wg4 < 2; ++wg4)
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(bad, 0, 8);
    for (int r = 0; r < reps; ++r) {
      if (wg4) {
        if (mode == 0) hipLaunchKernelGGL((k<0, 256>), dim3(2 * grid), dim3(256), 0, 0, iters, bad, out);
        if (mode == 1) hipLaunchKernelGGL((k<1, 256>), dim3(2 * grid), dim3(256), 0, 0, iters, bad, out);
        if (mode == 2) hipLaunchKernelGGL((k<2, 256>), dim3(2 * grid), dim3(256), 0, 0, iters, bad, out);
        if (mode == 3) hipLaunchKernelGGL((k<3, 256>), dim3(2 * grid), dim3(256), 0, 0, iters, bad, out);
      } else {
        if (mode == 0) hipLaunchKernelGGL((k<0, 512>), dim3(grid), dim3(512), 0, 0, iters, bad, out);
        if (mode == 1) hipLaunchKernelGGL((k<1, 512>), dim3(grid), dim3(512), 0, 0, iters, bad, out);
        if (mode == 2) hipLaunchKernelGGL((k<2, 512>), dim3(grid), dim3(512), 0, 0, iters, bad, out);
        if (mode == 3) hipLaunchKernelGGL((k<3, 512>), dim3(grid), dim3(512), 0, 0, iters, bad, out);
      }
    }
    hipDeviceSynchronize();
    unsigned hb[2]; float ho[2];
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 8, hipMemcpyDeviceToHost);
    printf("%s, mode %d (%s%s): wrong accumulator values: f16 waves %u, other waves %u  (sample %g expected %g)\n", wg4 ? "two 4-wave workgroups per CU" : "one 8-wave workgroup per CU", mode,
           (mode & 1) ? "f16 beside bf16" : "f16 beside f16", mode >= 2 ? ", fragments re-read from LDS" : "", hb[0], hb[1], ho[0], ho[1]);
  }
  for (int wg4 = 0; wg4 < 2; ++wg4) {
    hipMemset(bad, 0, 8);
    for (int r = 0; r < reps; ++r) {
      if (wg4) hipLaunchKernelGGL((k_alt<256>), dim3(2 * grid), dim3(256), 0, 0, iters, bad, out);
      else hipLaunchKernelGGL((k_alt<512>), dim3(grid), dim3(512), 0, 0, iters, bad, out);
    }
    hipDeviceSynchronize();
    unsigned hb[2]; float ho[2];
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 8, hipMemcpyDeviceToHost);
    printf("%s, every wave alternating fp16 / bf16 products: wrong accumulator values %u  (sample %g expected %g)\n",
           wg4 ? "two 4-wave workgroups per CU" : "one 8-wave workgroup per CU", hb[0], ho[0], ho[1]);
  }
  return 0;
}
