// Per-CU ingest microbenchmark: how fast can ONE workgroup per CU pull bytes, by path and access pattern?
//   mode 0: global_load_dwordx4 -> registers (xor-reduced)      mode 1: global_load_lds_dwordx4 -> LDS ring
//   hot 0: every workgroup reads its own region (HBM, cold)     hot 1: all workgroups read the same region (L2 hits)
//   seg: contiguous bytes per row piece (1024 = fully contiguous; 256 / 128 = row pieces at a 1536-byte stride)
// build: hipcc --offload-arch=gfx950 -O3 scratch/ingest.hip -o scratch/ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_ptr;
typedef __attribute__((address_space(1))) const void gbl_ptr;

template <int MODE, int SEG, int THREADS>
__global__ __launch_bounds__(THREADS) void ingest(const uint4* __restrict__ src, size_t region16, int kb, int hot, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = THREADS / 64;
  const uint4* base = src + (hot ? 0 : (size_t)blockIdx.x * region16);
  const int ninstr = kb;  // wave-instructions of 1 KiB in total
  uint4 acc = make_uint4(0, 0, 0, 0);
  // instruction i (1 KiB = 64 lanes x 16 B): rows of SEG bytes, row stride 1536 B when SEG < 1024
  constexpr int LPR = SEG / 16;       // lanes per row piece
  constexpr int RPI = 64 / LPR;       // row pieces per instruction
  for (int i = wave; i < ninstr; i += NW) {
    size_t off16;
    if (SEG == 1024) off16 = (size_t)i * 64 + lane;
    else {
      const int piece = i * RPI + lane / LPR;       // global piece index
      const int per_row = 1536 / SEG;               // pieces per 1536-byte row
      off16 = (size_t)(piece / per_row) * 96 + (size_t)(piece % per_row) * LPR + lane % LPR;
    }
    if (MODE == 0) {
      const uint4 v = base[off16];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    } else {
      __builtin_amdgcn_global_load_lds((gbl_ptr*)(base + off16), (lds_ptr*)(smem + ((i / NW) % 128) * 1024 + wave * 0), 16, 0, 0);
    }
  }
  if (MODE == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc.x = ((unsigned*)smem)[tid];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int MODE, int SEG, int THREADS>
float run(const uint4* src, size_t region16, int kb, int hot, int grid, unsigned* out, int lds) {
  auto k = ingest<MODE, SEG, THREADS>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, src, region16, kb, hot, out);
  hipEventRecord(e0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, 0, src, region16, kb, hot, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main() {
  const size_t region = 1 << 20;  // 1 MiB per workgroup (rows of 1536 B: kb KiB of pieces span kb*1536/SEG... bounded below)
  const int maxgrid = 512;
  uint4* src; unsigned* out;
  hipMalloc(&src, region * maxgrid); hipMalloc(&out, 64);
  hipMemset(src, 1, region * maxgrid);
  printf("mode seg thr hot grid KiB/wg   us    GB/s/CU  chipTB/s\n");
  for (int hot = 0; hot < 2; ++hot)
    for (int grid : {256, 512})
      for (int kb : {64, 128, 256}) {
#define RUN(M, S, T, LDS) { const float us = run<M, S, T>(src, region / 16, kb, hot, grid, out, LDS); \
        const double cu = (grid > 256 ? 2.0 : 1.0) * kb * 1024.0 / us * 1e-3; \
        printf("%4d %4d %3d %3d %4d %6d %7.2f %8.1f %8.2f\n", M, S, T, hot, grid, kb, us, cu, cu * 256 * 1e-3); }
        if (kb * (1024 / 1024) * 1024 <= (int)region) { RUN(0, 1024, 256, 0) RUN(1, 1024, 256, 65536) }
        if ((size_t)kb * 1024 / 256 / 6 * 1536 + 1536 <= region) { RUN(0, 256, 256, 0) RUN(1, 256, 256, 65536) }
        if ((size_t)kb * 1024 / 128 / 12 * 1536 + 1536 <= region) { RUN(0, 128, 256, 0) RUN(1, 128, 256, 65536) }
        RUN(0, 1024, 512, 0) RUN(1, 1024, 512, 65536)
      }
  return 0;
}
