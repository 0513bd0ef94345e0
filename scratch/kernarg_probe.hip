// kernarg_probe -- what does a kernel pay for reading its arguments?  Chains of dependent trivial launches (one workgroup per CU, every
// thread's first action depends on an argument), arguments as a by-value struct (s_load from the kernarg segment) or as scalars that
// hipcc preloads into SGPRs when built with -mllvm -amdgpu-kernarg-preload-count=16.
// build: hipcc --offload-arch=gfx950 -O3 scratch/kernarg_probe.hip -o scratch/kernarg_probe          (plain)
//        hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 scratch/kernarg_probe.hip -o scratch/kernarg_probe_pl
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Args { float* out; const float* in; int a, b, c, d, e, f, g, h; float s, t; };
__global__ __launch_bounds__(256) void k_struct(Args p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  p.out[i] = p.in[i] * p.s + (float)(p.a + p.b + p.c + p.d + p.e + p.f + p.g + p.h) + p.t;
}
__global__ __launch_bounds__(256) void k_scalar(float* out, const float* in, int a, int b, int c, int d, int e, int f, int g, int h, float s, float t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  out[i] = in[i] * s + (float)(a + b + c + d + e + f + g + h) + t;
}
int main() {
  float *x, *y;
  hipMalloc(&x, 256 * 256 * 4); hipMalloc(&y, 256 * 256 * 4);
  hipMemset(x, 0, 256 * 256 * 4); hipMemset(y, 0, 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipGraph_t g; hipGraphExec_t ge; hipStream_t st; hipStreamCreate(&st);
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int i = 0; i < 100; ++i) {
        float* o = (i & 1) ? x : y; const float* in = (i & 1) ? y : x;
        if (mode == 0) { Args a{o, in, 1, 2, 3, 4, 5, 6, 7, 8, 0.5f, 1.f}; hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, st, a); }
        else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, st, o, in, 1, 2, 3, 4, 5, 6, 7, 8, 0.5f, 1.f);
      }
      hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, st);
      hipEventRecord(e0, st);
      for (int w = 0; w < 20; ++w) hipGraphLaunch(ge, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s args: %.3f us per dependent launch (graph of 100, 20 replays)\n", mode == 0 ? "struct" : "scalar", ms * 1e3 / 2000);
      hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
    }
  return 0;
}
