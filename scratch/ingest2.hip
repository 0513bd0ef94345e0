// ingest2 -- how fast does ONE sim unit's single-use operand arrive, by path and access SHAPE?  (round 5, VERDICT r4 #2a)
// Emulates sk_sim_kernel's C traffic at cfg3 per rank: unit (rb, ct) of 4 x 64 pulls the C tile [128 contexts][768] bf16 (196 KiB)
// in 12 chunks of [128 rows][64 k] = 16 KiB; the four row-block units of a tile read the SAME tile (XCD-contiguous numbering as in
// the kernel), so the footprint is 12.6 MB and every line is fetched 4x through one XCD's L2.
//   mode 0  LDS-DMA, 1-KiB pieces of 8 rows x 128 B (what skinny.h does)                       -> LDS ring of 4 slots
//   mode 1  global_load_dwordx4 in MFMA B-fragment shape: one instruction = 16 rows x 64 B     -> VGPRs (xor-reduced)
//   mode 2  global_load_dwordx4 in whole lines: one instruction = 8 rows x 128 B               -> VGPRs (xor-reduced)
//   mode 3  as 1, but the lane's two k halves of a row adjacent (32 contiguous bytes per lane and row: the k-permuted fragment)
// NW = 4 | 8 waves.  DEPTH = chunks in flight for the register modes (loads of chunk c + DEPTH issued before chunk c is consumed).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/ingest2.hip -o scratch/ingest2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_ptr;
typedef __attribute__((address_space(1))) const void gbl_ptr;

constexpr int D = 768, COLS = 128, NCH = D / 64, NTILE = 64, NRB = 4;

__device__ __forceinline__ int xcd_order(int wg, int nwg) {
  const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
}

template <int MODE, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64) void ingest(const uint16_t* __restrict__ C, unsigned* out, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int unit = xcd_order(blockIdx.x, gridDim.x);
  const int ct = unit / NRB;
  const uint16_t* tile = C + (size_t)ct * COLS * D;
  const unsigned long long t0 = wall_clock64();
  uint4 acc = make_uint4(0, 0, 0, 0);
  if constexpr (MODE == 0) {
    constexpr int IPC = COLS * 64 * 2 / 1024 / NW;  // pieces per wave and chunk (4 | 2)
    constexpr int SLOTS = 4;
    unsigned cof[IPC];
#pragma unroll
    for (int j = 0; j < IPC; ++j) {
      const int row = (wave * IPC + j) * 8 + (lane >> 3);
      cof[j] = (unsigned)row * D + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    auto issue = [&](int kc, int slot) {
#pragma unroll
      for (int j = 0; j < IPC; ++j)
        __builtin_amdgcn_global_load_lds((gbl_ptr*)(tile + cof[j] + kc * 64), (lds_ptr*)(smem + slot * 16384 + (wave * IPC + j) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int c = 0; c < SLOTS - 1; ++c) issue(c, c);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // this wave's share of chunk c has landed; then everybody's
      const int younger = (c + SLOTS - 2 < NCH - 1 ? c + SLOTS - 2 : NCH - 1) - c;
      if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPC) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPC) : "memory");
      __builtin_amdgcn_s_barrier();
      if (c + SLOTS - 1 < NCH) issue(c + SLOTS - 1, (c + SLOTS - 1) % SLOTS);
    }
    acc.x = ((unsigned*)smem)[tid];
  } else {
    // register modes: wave w owns rows [w * 128 / NW, ...) of the tile; per chunk it needs [rows][64 k] = rows * 128 B
    constexpr int RPW = COLS / NW;               // rows per wave (32 | 16)
    constexpr int LPC = RPW * 128 / 1024;         // load instructions per wave and chunk (4 | 2)
    auto addr = [&](int c, int j) -> const uint4* {
      int row, byte;
      if (MODE == 1) {        // fragment shape: instruction j = (row block j >> 1, k half j & 1); lane (i16, g4) -> row i16, bytes g4 * 16
        if (RPW == 32) { row = (j >> 1) * 16 + (lane & 15); byte = (j & 1) * 64 + (lane >> 4) * 16; }
        else { row = lane & 15; byte = j * 64 + (lane >> 4) * 16; }
      } else if (MODE == 3) {  // k-permuted fragment: lane (i16, g4) holds bytes [g4 * 32, + 32) of its row: instruction j & 1 = which half
        if (RPW == 32) { row = (j >> 1) * 16 + (lane & 15); byte = (lane >> 4) * 32 + (j & 1) * 16; }
        else { row = lane & 15; byte = (lane >> 4) * 32 + j * 16; }
      } else {                 // whole lines: instruction j = rows j * 8 .. + 8, lane -> row l >> 3, 16-byte chunk l & 7
        row = j * 8 + (lane >> 3); byte = (lane & 7) * 16;
      }
      return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tile + (size_t)(wave * RPW + row) * D + c * 64) + byte);
    };
    uint4 v[DEPTH][LPC];
#pragma unroll
    for (int c = 0; c < DEPTH; ++c)
#pragma unroll
      for (int j = 0; j < LPC; ++j) v[c][j] = *addr(c, j);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int j = 0; j < LPC; ++j) {
        const uint4 x = v[c % DEPTH][j];
        acc.x ^= x.x; acc.y ^= x.y; acc.z ^= x.z; acc.w ^= x.w;
        if (c + DEPTH < NCH) v[c % DEPTH][j] = *addr(c + DEPTH, j);
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  if (tid == 0 && stamps) stamps[blockIdx.x] = t1 - t0;
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int MODE, int NW, int DEPTH>
void run(const uint16_t* C, unsigned* out, unsigned long long* stamps, const char* label) {
  auto k = ingest<MODE, NW, DEPTH>;
  const int lds = MODE == 0 ? 65536 : 0;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = NTILE * NRB;
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, C, out, stamps);
  hipEventRecord(e0);
  const int reps = 50;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, C, out, stamps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  unsigned long long h[256];
  hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0, mx = 0;
  for (int i = 0; i < grid; ++i) { mean += (double)h[i]; mx = h[i] > mx ? (double)h[i] : mx; }
  mean /= grid;
  const double unit_us = mean / 100.0, bytes = (double)COLS * D * 2;  // wall clock = 100 MHz
  printf("%-44s launch %6.2f us | unit mean %5.2f max %5.2f us | %6.1f GB/s per CU = %5.1f B/clk @2.4GHz | chip %5.2f TB/s (launch)\n", label, us, unit_us,
         mx / 100.0, bytes / unit_us * 1e-3, bytes / unit_us * 1e-3 / 2.4, bytes * grid / us * 1e-6);
}

int main() {
  uint16_t* C; unsigned* out; unsigned long long* stamps;
  const size_t bytes = (size_t)NTILE * COLS * D * 2;
  hipMalloc(&C, bytes); hipMalloc(&out, 64); hipMalloc(&stamps, 256 * 8);
  hipMemset(C, 1, bytes);
  printf("cfg3-per-rank sim unit's C tile: 196 KiB per unit, 256 units, footprint %.1f MB (each tile read by 4 units of one XCD)\n", bytes * 1e-6);
  run<0, 4, 0>(C, out, stamps, "LDS-DMA 8x128B pieces, 4 waves, 3 in flight");
  run<0, 8, 0>(C, out, stamps, "LDS-DMA 8x128B pieces, 8 waves, 3 in flight");
  run<1, 4, 3>(C, out, stamps, "dwordx4 fragment 16x64B, 4 waves, depth 3");
  run<1, 4, 6>(C, out, stamps, "dwordx4 fragment 16x64B, 4 waves, depth 6");
  run<1, 4, 12>(C, out, stamps, "dwordx4 fragment 16x64B, 4 waves, depth 12");
  run<1, 8, 6>(C, out, stamps, "dwordx4 fragment 16x64B, 8 waves, depth 6");
  run<1, 8, 12>(C, out, stamps, "dwordx4 fragment 16x64B, 8 waves, depth 12");
  run<3, 4, 6>(C, out, stamps, "dwordx4 k-permuted 16x(4x32B), 4 waves, d6");
  run<3, 8, 12>(C, out, stamps, "dwordx4 k-permuted 16x(4x32B), 8 waves, d12");
  run<2, 4, 3>(C, out, stamps, "dwordx4 lines 8x128B, 4 waves, depth 3");
  run<2, 4, 6>(C, out, stamps, "dwordx4 lines 8x128B, 4 waves, depth 6");
  run<2, 4, 12>(C, out, stamps, "dwordx4 lines 8x128B, 4 waves, depth 12");
  run<2, 8, 12>(C, out, stamps, "dwordx4 lines 8x128B, 8 waves, depth 12");
  return 0;
}
