// experiment: where the streaming top-k kernel spends its time when it starts from an empty state (build with hipcc -O3; not shipped)
#define TK_TIMING 1
#include <hip/hip_runtime.h>
#include "../dpr_scale_amd/csrc/dprhot.hip"
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(float* s, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    // roughly normal: sum of 4 uniforms
    s[i] = ((x & 255) + ((x >> 8) & 255) + ((x >> 16) & 255) + (x >> 24)) / 128.0f - 4.0f;
  }
}
int main(int argc, char** argv) {
  const int rows = 1024, k = argc > 1 ? atoi(argv[1]) : 100;
  const int colsv[3] = {65536, 8192, 4096};
  float *S, *vals; int64_t* idx;
  CK(hipMalloc(&S, (size_t)rows * 65536 * 4)); CK(hipMalloc(&vals, rows * 1024 * 4)); CK(hipMalloc(&idx, rows * 1024 * 8));
  fill<<<4096, 256>>>(S, (size_t)rows * 65536, 7u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int ci = 0; ci < 3; ++ci) for (int first = 1; first >= 0; --first) {
    const int cols = colsv[ci];
    float ms = 0;
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0));
      int rc = dprhot_topk_update(S, rows, cols, 65536, first ? 0 : 1000000, k, vals, idx, first, nullptr);
      if (rc) { printf("rc=%d %s\n", rc, dprhot_last_error()); return 1; }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> t(4096 * 8);
    unsigned long long* dptr; CK(hipGetSymbolAddress((void**)&dptr, HIP_SYMBOL(dprhot::g_tk_tm)));
    CK(hipMemcpy(t.data(), dptr, t.size() * 8, hipMemcpyDeviceToHost));
    double a[8] = {0};
    for (int r = 0; r < rows; ++r) for (int i = 0; i < 8; ++i) a[i] += t[(size_t)r * 8 + i];
    printf("cols %6d first %d k %d: %.1f us | per workgroup avg us: tail %.2f load+count %.2f barriers %.2f append %.2f flush %.2f slow %.2f | flushes %.2f slow windows %.2f\n",
           cols, first, k, ms * 1000, a[0] / rows * 0.01, a[1] / rows * 0.01, a[2] / rows * 0.01, a[3] / rows * 0.01, a[4] / rows * 0.01,
           a[5] / rows * 0.01, a[6] / rows, a[7] / rows);
  }
  return 0;
}
