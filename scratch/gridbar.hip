// How much does an in-kernel grid barrier (release/acquire across the 8 XCD L2s) cost on gfx950, next to a kernel
// boundary?  Each workgroup writes a line, barrier, reads the line of a workgroup on another XCD and checks it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void bar_kernel(unsigned* ctr, float* buf, int iters, int* bad, int payload) {
  const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  int errors = 0;
  for (int it = 0; it < iters; ++it) {
    for (int p = tid; p < payload; p += 256) buf[(size_t)wg * payload + p] = (float)(it * 1000 + wg);
    grid_barrier(ctr, (unsigned)G * (it + 1));
    const int other = (wg + 1) % G;  // consecutive workgroups sit on different XCDs
    for (int p = tid; p < payload; p += 256) errors += buf[(size_t)other * payload + p] != (float)(it * 1000 + other);
    grid_barrier(ctr + 32, (unsigned)G * (it + 1));  // so that nobody overwrites before everybody has read
  }
  if (errors) atomicAdd(bad, errors);
}

__global__ __launch_bounds__(256) void step_kernel(float* buf, int it, int* bad, int payload) {
  const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  const int other = (wg + 1) % G;
  int errors = 0;
  if (it > 0) for (int p = tid; p < payload; p += 256) errors += buf[(size_t)(it & 1 ? 0 : 1) * G * payload + (size_t)other * payload + p] != (float)((it - 1) * 1000 + other);
  for (int p = tid; p < payload; p += 256) buf[(size_t)(it & 1) * G * payload + (size_t)wg * payload + p] = (float)(it * 1000 + wg);
  if (errors) atomicAdd(bad, errors);
}

int main() {
  unsigned* ctr; float* buf; int* bad;
  const int payload = 1024;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&buf, 2 * 512 * payload * 4)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {8, 32, 64, 128, 256}) {
    const int iters = 200;
    float best = 1e9;
    int hbad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(bad, 0, 4));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(bar_kernel, dim3(G), dim3(256), 0, 0, ctr, buf, iters, bad, payload);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    }
    printf("G=%3d: in-kernel  %.2f us per (write 4 KB, barrier, read neighbour, barrier)  -> %.2f us per barrier+trip, mismatches %d\n", G,
           best * 1e3 / iters, best * 1e3 / iters / 2, hbad);
    best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(bad, 0, 4));
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(step_kernel, dim3(G), dim3(256), 0, 0, buf, it, bad, payload);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    }
    printf("       kernel per step %.2f us (read neighbour of previous launch, write), mismatches %d\n", best * 1e3 / iters, hbad);
  }
  return 0;
}
