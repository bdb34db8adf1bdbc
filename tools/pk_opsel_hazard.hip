// Reproducer of a gfx950 (MI355X, ROCm 7.2) hazard: a packed-fp32 VOP3P instruction whose op_sel takes the LOW result's
// operand from the HIGH dword of a VGPR pair in SRC1 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 .. op_sel:[0,1(,0)]) now and
// then computes lanes 48-63 with that operand read as ZERO while another wave keeps the SIMD's matrix pipe busy.  hipcc emits
// the form by itself whenever the register allocator holds a pair in swapped order; it caused the sporadic wrong patches of
// the two-term fp16 block forward.  Testers (waves 0-3) evaluate each form next to the natural-order
// form of the same numbers and count bitwise mismatches; waves 4-7 (their SIMD partners) issue fp16 MFMAs back to back.
// The instruction sequence matters (the swapped pair comes out of a v_pk_mov_b32 right ahead, as in hipcc's own code; scalar
// instructions in between make the failures rare) - keep the tester loop as it is.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_opsel_hazard.bin pk_opsel_hazard.hip      run: ./pk_opsel_hazard.bin [mfma 0|1] [wgs per CU] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static const char* kForm[8] = {"mul  src1 lo<-hi, hi<-lo", "mul  src1 lo<-hi", "mul  src1 hi<-lo (broadcast)", "fma  src1 lo<-hi, hi<-lo",
                               "add  src1 lo<-hi, hi<-lo", "v_pk_mov_b32 swap", "mul  SGPR pair broadcast", "mul  src0 lo<-hi, hi<-lo"};

__global__ void __launch_bounds__(512, 2) k(unsigned* out, int mode, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();
  if (wave < 4) {                      // ---- tester
    unsigned seed = (blockIdx.x * 512u + tid) * 2654435761u + 12345u;
    unsigned bad[8] = {0, 0, 0, 0, 0, 0, 0, 0}, first = 0;
    const f32x2 ss = {1024.f, 1024.f};
    for (int it = 0; it < iters; ++it) {
      seed = seed * 1664525u + 1013904223u;
      const float a = (float)(int)(seed >> 8) * (1.f / 8388608.f) + 0.25f;
      seed = seed * 1664525u + 1013904223u;
      const float b = (float)(int)(seed >> 8) * (1.f / 8388608.f) + 0.25f;
      const f32x2 nat = {a, b}, swp = {b, a}, aa = {a, a}, c = {0.5f, 0.75f};
      f32x2 r[8], e[8];
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r[0]) : "v"(ss), "v"(swp));
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=&v"(e[0]) : "v"(ss), "v"(nat));
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(r[1]) : "v"(ss), "v"(swp));
      e[1] = f32x2{e[0][0], 1024.f * a};
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(r[2]) : "v"(ss), "v"(nat));
      e[2] = f32x2{e[0][0], e[0][0]};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(r[3]) : "v"(ss), "v"(swp), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=&v"(e[3]) : "v"(ss), "v"(nat), "v"(c));
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(r[4]) : "v"(ss), "v"(swp));
      asm volatile("v_pk_add_f32 %0, %1, %2" : "=&v"(e[4]) : "v"(ss), "v"(nat));
      asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=&v"(r[5]) : "v"(swp));
      e[5] = nat;
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=&v"(r[6]) : "v"(nat), "s"(0x4480000044800000ull));
      e[6] = e[0];
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=&v"(r[7]) : "v"(swp), "v"(ss));
      e[7] = e[0];
      unsigned what = 0;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const bool m = __builtin_bit_cast(unsigned, r[f][0]) != __builtin_bit_cast(unsigned, e[f][0]) ||
                       __builtin_bit_cast(unsigned, r[f][1]) != __builtin_bit_cast(unsigned, e[f][1]);
        bad[f] += m; what |= (unsigned)m << f;
      }
      if (what) atomicAdd(&out[8 + (it == 0 ? 0 : it < 16 ? 1 : it < 256 ? 2 : 3)], 1u);
      if (what && !first) { first = 1; atomicAdd(&out[12], 1u); 
        atomicAdd(&out[16 + (lane >> 4)], 1u);
        if (((what & 1) && r[0][0] == 0.f) || ((what & 2) && r[1][0] == 0.f) || ((what & 8) && r[3][0] == -0.5f)) atomicAdd(&out[20], 1u);
      }
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) if (bad[f]) atomicAdd(&out[f], bad[f]);
  } else {                             // ---- SIMD partner: bursts of eight fp16 MFMAs
    f32x16 acc = {0};
    float facc = 0.f;
    f16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(lane * 0.01f + j); fb[j] = (_Float16)(j * 0.5f - lane * 0.02f); }
    for (int it = 0; it < iters * 2; ++it) {
      if (mode & 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
      }
    }
    if (acc[0] + acc[5] + facc == 123.456f) out[15] = 1;
  }
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1, per_cu = argc > 2 ? atoi(argv[2]) : 2, iters = argc > 3 ? atoi(argv[3]) : 20000;
  unsigned* d; unsigned h[32] = {0};
  if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMemset(d, 0, sizeof(h)) != hipSuccess) return 2;
  hipLaunchKernelGGL(k, dim3(256 * per_cu), dim3(512), 0, 0, d, mode, iters);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  printf("partner MFMAs %s, %d workgroups per CU, %d iterations x %d tester lanes; mismatches against the natural-order form:\n", (mode & 1) ? "ON" : "off", per_cu, iters,
         256 * per_cu * 256);
  unsigned total = 0;
  for (int f = 0; f < 8; ++f) { printf("   %-28s %9u\n", kForm[f], h[f]); total += h[f]; }
  printf("   by iteration [0, 1-15, 16-255, 256+]: %u %u %u %u; lanes hit %u, by quarter wave [0-15, 16-31, 32-47, 48-63]: %u %u %u %u; first hits with the zero-operand value: %u\n",
         h[8], h[9], h[10], h[11], h[12], h[16], h[17], h[18], h[19], h[20]);
  return total ? 1 : 0;
}
