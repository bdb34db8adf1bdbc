// Prints the resident workgroups/CU the runtime reports for each hot kernel at its launch geometry.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../pde_policylearning_amd/csrc/k_pointwise.h"
#include "../pde_policylearning_amd/csrc/k_block_bwd.h"
#include "../pde_policylearning_amd/csrc/k_projection.h"
template <typename K> void occ(const char* name, K kern, int threads, size_t lds) {
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nb = -1;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kern);
  printf("%-28s threads=%4d lds=%6zu regs=%3d static_lds=%zu -> blocks/CU=%d (%s)\n", name, threads, lds, fa.numRegs, fa.sharedSizeBytes, nb, hipGetErrorString(e));
}
int main() {
  occ("k_pw_fwd<64,64,128>", k_pw_fwd<64, 64, 128>, 512, pw_fwd_lds_bytes(64, 64, 128, 128, 6, 1, true, true));
  occ("k_pw_fwd<3,64,128>", k_pw_fwd<3, 64, 128>, 512, pw_fwd_lds_bytes(3, 64, 128, 128, 0, 1, false, true));
  occ("k_block_bwd<64,128>", k_block_bwd<64, 128>, 512, (size_t)(2 * 64 * 132 + 12 * 128 + 6 * 64 * 2) * 4);
  occ("k_proj_fwd<64,256,128,1>", k_proj_fwd<64, 256, 128, 1>, 512, (size_t)64 * 132 * 4 + (256 + 256 + 128) * 4);
  occ("k_proj_bwd<64,256,128,1>", k_proj_bwd<64, 256, 128, 1>, 512, (size_t)(64 + 128) * 132 * 4 + (128 + 256 + 256) * 4);
  return 0;
}
