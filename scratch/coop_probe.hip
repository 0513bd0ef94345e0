// Probe: what does a cooperative launch (grid-wide sync inside one kernel) cost on this box, and does it survive stream capture?
//   hipcc --offload-arch=gfx950 -O3 scratch/coop_probe.hip -o scratch/coop_probe && ./scratch/coop_probe
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void coop_kernel(float* part, float* out, int nsync) {
  extern __shared__ float smem[];
  cg::grid_group grid = cg::this_grid();
  float acc = 0.f;
  for (int s = 0; s < nsync; ++s) {
    if (threadIdx.x == 0) part[s * gridDim.x + blockIdx.x] = (float)(blockIdx.x + s);
    grid.sync();
    float v = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) v += part[s * gridDim.x + i];
    smem[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < 256; ++i) t += smem[i]; acc += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256, 2) void plain_kernel(float* part, float* out, int s) {
  extern __shared__ float smem[];
  if (threadIdx.x == 0) part[s * gridDim.x + blockIdx.x] = (float)(blockIdx.x + s);
  if (threadIdx.x == 0) out[blockIdx.x] = 1.f;
}

int main() {
  const int nb = 256, lds = 72 * 1024;
  float *part, *out;
  CK(hipMalloc(&part, 64 * nb * sizeof(float)));
  CK(hipMalloc(&out, nb * sizeof(float)));
  CK(hipFuncSetAttribute((const void*)coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void*)plain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, coop_kernel, 256, lds));
  printf("occupancy blocks/CU at 72 KiB LDS: %d\n", occ);
  for (int nsync = 1; nsync <= 3; ++nsync) {
    void* args[] = {&part, &out, &nsync};
    for (int w = 0; w < 5; ++w) CK(hipLaunchCooperativeKernel((const void*)coop_kernel, dim3(nb), dim3(256), args, lds, st));
    CK(hipStreamSynchronize(st));
    const int it = 200;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < it; ++i) CK(hipLaunchCooperativeKernel((const void*)coop_kernel, dim3(nb), dim3(256), args, lds, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h(nb);
    CK(hipMemcpy(h.data(), out, nb * sizeof(float), hipMemcpyDeviceToHost));
    float expect = 0.f; for (int s = 0; s < nsync; ++s) expect += nb * (nb - 1) / 2.0f + (float)s * nb;
    printf("cooperative, %d grid syncs: %.2f us per launch (eager, back to back); out[0]=%.0f out[255]=%.0f expect %.0f\n", nsync, ms * 1e3 / it, h[0], h[255], expect);
  }
  {
    const int it = 200;
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(plain_kernel, dim3(nb), dim3(256), lds, st, part, out, 0);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(plain_kernel, dim3(nb), dim3(256), lds, st, part, out, 0);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("plain launch: %.2f us per launch (eager, back to back)\n", ms * 1e3 / it);
  }
  {  // stream capture
    int nsync = 1;
    void* args[] = {&part, &out, &nsync};
    hipGraph_t g; hipGraphExec_t ge;
    hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    printf("begin capture: %s\n", hipGetErrorString(e));
    for (int i = 0; i < 20 && e == hipSuccess; ++i) e = hipLaunchCooperativeKernel((const void*)coop_kernel, dim3(nb), dim3(256), args, lds, st);
    printf("cooperative launch under capture: %s\n", hipGetErrorString(e));
    hipError_t e2 = hipStreamEndCapture(st, &g);
    printf("end capture: %s\n", hipGetErrorString(e2));
    if (e == hipSuccess && e2 == hipSuccess) {
      e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      printf("instantiate: %s\n", hipGetErrorString(e));
      if (e == hipSuccess) {
        for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph of 20 cooperative launches: %.2f us per launch\n", ms * 1e3 / 400);
      }
    }
  }
  return 0;
}
