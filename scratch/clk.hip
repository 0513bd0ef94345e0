#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void probe(unsigned long long* out, int spin) {
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (unsigned long long)x; }
}
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void heavy(float* p, int n) { float x = threadIdx.x; for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f; if (x == 123.f) p[0] = x; }
int main() {
  unsigned long long* d; float* f; CK(hipMalloc(&d, 64)); CK(hipMalloc(&f, 64)); CK(hipMemset(f, 0, 64));
  unsigned long long h[3];
  int rate = 0; CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0)); printf("wall clock rate %d kHz\n", rate);
  auto run = [&](const char* tag) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 20000); hipDeviceSynchronize(); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("%-28s shader cycles %llu wall ticks %llu -> %.0f MHz\n", tag, h[0], h[1], (double)h[0] / ((double)h[1] / (rate * 1e3)) * 1e-6); };
  run("cold");
  run("second");
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny, dim3(8), dim3(256), 0, 0, f);
  run("after 2000 tiny launches");
  hipLaunchKernelGGL(heavy, dim3(2048), dim3(256), 0, 0, f, 2000000); run("after heavy (all CUs ~ms)");
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny, dim3(8), dim3(256), 0, 0, f);
  run("tiny burst after heavy");
  // time tiny kernels back-to-back
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0); for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(tiny, dim3(8), dim3(256), 0, 0, f); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("1000 tiny launches: %.2f us each\n", ms);
  }
  return 0;
}
