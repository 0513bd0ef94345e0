// experiment: where does the 256x256 kernel's K loop spend its time?  variants via -DG2_VARIANT
//   0 full   1 no DMA after the first step (LDS reads + MFMA only)   2 no LDS fragment reads (DMA + MFMA)   3 MFMA only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../dpr_scale_amd/csrc/gemm256.h"
using namespace dprhot;
struct EpiNull {
  float* out;
  struct BigRegs { int x; };
  __device__ BigRegs big_load(int, int, int) const { return BigRegs{0}; }
  __device__ void big_store(const BigRegs&, int, int, int*, int) const {}
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ int big_state(const TileCtx&, const int*) const { return 0; }
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ void finish(f32x4 (&acc)[TM][TN], const TileCtx& c, int) const {
    float s = 0.f;
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
    if (s == 12345.678f) out[c.tid] = s;
  }
};
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
int main(int argc, char** argv) {
  const int M = 1024, N = 65536, K = 768;
  uint16_t *A, *B; float* out;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&out, 4096));
  std::vector<uint16_t> h((size_t)N * K);
  unsigned s = 1; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (uint16_t)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (uint16_t)((s >> 3) & 0x8000); }
  CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
  GemmArgs a{A, B, M, N, K, K, K, K};
  EpiNull epi{out};
  auto kern = gemm256_kernel<EpiNull, true>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2_lds_total));
  const int nbx = N / 256, nby = M / 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int grid : {256, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), g2_lds_total, 0, a, epi, nbx, nby);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      if (rep == 2) printf("variant %d grid %4d: %.1f us  %.0f TFLOP/s\n", G2_VARIANT, grid, ms * 1e3, 2.0 * M * N * K / ms * 1e-9);
    }
  }
  return 0;
}
