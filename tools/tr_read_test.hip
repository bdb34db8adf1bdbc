// Semantics probe of ds_read_b64_tr_b16 (gfx950): per group of 16 lanes a 4 x 16 block of 16-bit elements is delivered
// column-major.  Lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i, row q in element q.
//   hipcc --offload-arch=gfx950 -O3 tools/tr_read_test.hip -o tools/tr_read_test.bin && tools/tr_read_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;      // element (row, col) of a 64 x 64 array = 64 row + col
  __syncthreads();
  const int lane = threadIdx.x;
  const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
  // group g: rows 8g + q (q = 0..3), columns 16 + 4p .. 16 + 4p + 3
  auto ptr = (__attribute__((address_space(3))) s16x4*)(lds + (8 * g + q) * 64 + 16 + 4 * p);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
    const int g = lane >> 4, i = lane & 15;
    const int want = (8 * g + j) * 64 + 16 + i;       // row 8g + j, column 16 + i
    if (h[lane * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got (row %d, col %d) want (row %d, col %d)\n", lane, j, h[lane*4+j] / 64, h[lane*4+j] % 64, want / 64, want % 64); ++bad; }
  }
  printf("tr16_b64 semantics: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as documented", bad);
  return 0;
}
