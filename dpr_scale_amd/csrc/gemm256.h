// Large-shape bf16 GEMM tile for gfx950: D[M,N] = A[M,K] * B[N,K]^T, both operands k-major (the similarity GEMM of
// dpr_task.py:98-105 at evaluation / retrieval / large-batch sizes, run_retrieval_pytorch.py:148).
//
// Why a second structure next to gemm_bf16.h: with four 64x64 wave tiles (128x128 per workgroup) every k needs
// (64+64)*2 B of LDS reads per wave for 64*64*2 flops = 32 flop/B, and 128 B/clk of LDS bandwidth against
// 4096 flop/clk of MFMA per CU is exactly 32 flop/B: the LDS pipe is as busy as the MFMA pipe and the kernel tops out
// near a quarter of the peak (measured 588-660 TFLOP/s).  Here: 256x256 per workgroup, 8 waves (2 x 4) of 128x64
// (43 flop/B), K step 64, operands DMA'd global -> LDS (global_load_lds_dwordx4: no staging registers, no ds_write
// pass), two LDS buffers (128 KiB), one barrier per K step, one workgroup per CU.
//
// LDS image = TileGeom<256, 64, true>: dense 128-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
// (conflict-free for the ds_read_b128 fragment reads, see gemm_bf16.h).  The DMA writes lane-linearly (wave-uniform
// base + lane * 16), so the swizzle is applied on the SOURCE side: lane l of the instruction that fills rows
// r0..r0+7 fetches chunk (l & 7) ^ ((row >> 1) & 7) of row r0 + (l >> 3) -- each 8-lane group still reads one whole
// 128-byte line.
#pragma once
#include "gemm_bf16.h"
#ifndef G2_VARIANT
#define G2_VARIANT 0  // scratch/g2probe.hip builds 1..3 to take the K loop apart; the library is always 0
#endif

namespace dprhot {

constexpr int G2_B = 256;      // BM = BN
constexpr int G2_BK = 64;
constexpr int G2_THREADS = 512;
constexpr int G2_TILE = G2_B * G2_BK;  // elements of one operand tile
constexpr size_t g2_lds_bytes = (size_t)2 * 2 * G2_TILE * sizeof(uint16_t);  // 2 buffers x (A + B) = 128 KiB

typedef __attribute__((address_space(3))) void g2_lds_ptr;
typedef __attribute__((address_space(1))) const void g2_gbl_ptr;

// Workgroup id -> tile.  Linear ids are dealt round-robin to the 8 XCDs (each with its own 4 MiB L2): first give
// every XCD a contiguous range of tile numbers (bijective also when the count is not a multiple of 8), then walk the
// tiles of a range in groups of 8 tile rows x all columns, row fastest, so that the ~32 workgroups an XCD runs at a
// time form an 8 x 4 patch sharing 8 A tiles and 4 B tiles (4.7 MB) instead of 1 + 32.
__device__ __forceinline__ void g2_tile_of(int wg, int nbx, int nby, int& bx, int& by) {
  const int nwg = nbx * nby;
  const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  constexpr int GM = 8;
  const int per_group = GM * nbx;
  const int grp = t / per_group, in = t - grp * per_group;
  const int rows = min(GM, nby - grp * GM);
  by = grp * GM + in % rows;
  bx = in / rows;
}

// Persistent: the grid is min(#tiles, #CUs) workgroups, workgroup w computes tiles w, w + grid, ... and the K-step
// pipeline runs straight across tile boundaries: the first K step of the next tile is DMA'd during the last K step
// (and the epilogue) of the current one.  With K = d = 768 a tile is only 12 K steps; without this every tile pays
// one exposed memory latency (~2 us against ~5 us of MFMA work -- one workgroup per CU, nobody else to hide it).
// The epilogue's per-row / per-column inputs (labels, mask bytes, top-k thresholds) are fetched a tile ahead into
// LDS (Epi::big_load / big_store / big_state), so the K loop carries accumulators and fragments only.
// LDS: 2 x (A + B) tile buffers | epilogue scratch (max, sum per row per wave column) | 2 x 512 ints of tile metadata
constexpr size_t g2_scratch_bytes = (size_t)G2_B * 4 * 2 * sizeof(float);
constexpr size_t g2_meta_bytes = (size_t)2 * 2 * G2_B * sizeof(int);
constexpr size_t g2_lds_total = g2_lds_bytes + g2_scratch_bytes + g2_meta_bytes;

template <class Epi, bool PERSIST>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm256_kernel(GemmArgs p, Epi epi, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int WM = 2, WN = 4, TM = 8, TN = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = p.K / G2_BK;  // the launcher guarantees K % 64 == 0
  const int ntiles = nbx * nby;
  float* const scratch = reinterpret_cast<float*>(smem + 4 * G2_TILE);
  int* const meta0 = reinterpret_cast<int*>(scratch + G2_B * 4 * 2);

  // each wave fills rows [wave*32, wave*32 + 32) of both tiles: 4 DMA instructions of 8 rows per operand per K step.
  // 32-bit element offsets from the (uniform) operand bases: the launcher guarantees both operands are < 4 GiB.
  unsigned oa[4], ob[4];
  auto aim = [&](int bx, int by) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 32 + j * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      oa[j] = (unsigned)min(by * G2_B + row, p.M - 1) * (unsigned)p.lda + c * 8;
      ob[j] = (unsigned)min(bx * G2_B + row, p.N - 1) * (unsigned)p.ldb + c * 8;
    }
  };
  auto issue = [&](int t, int buf) {
    uint16_t* As = smem + buf * 2 * G2_TILE;
    uint16_t* Bs = As + G2_TILE;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.A + oa[j] + t * G2_BK), (g2_lds_ptr*)(As + (wave * 32 + j * 8) * G2_BK), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.B + ob[j] + t * G2_BK), (g2_lds_ptr*)(Bs + (wave * 32 + j * 8) * G2_BK), 16, 0, 0);
    }
  };

  int tile = blockIdx.x, bx, by, buf = 0, par = 0;
  g2_tile_of(tile, nbx, nby, bx, by);
  aim(bx, by);
  issue(0, 0);
  if (tid < G2_B) {
    const auto r = epi.big_load(by * G2_B, bx * G2_B, tid);
    epi.big_store(r, bx * G2_B, tid, meta0, G2_B);
  }
  while (tile < ntiles) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int next = PERSIST ? tile + (int)gridDim.x : ntiles;  // !PERSIST: one tile per workgroup
    int nbx_ = bx, nby_ = by;
    if (next < ntiles) g2_tile_of(next, nbx, nby, nbx_, nby_);

    for (int t = 0; t < nt; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the step has landed
      __syncthreads();                                    // ... everybody's has; and all waves are past the previous step
      typename Epi::BigRegs mr{};
      const bool last = t + 1 == nt;
      if (G2_VARIANT == 1 || G2_VARIANT == 3) {
      } else if (!last) {
        issue(t + 1, buf ^ 1);  // overwrites the buffer the previous step was read from
      } else if (next < ntiles) {
        aim(nbx_, nby_);
        issue(0, buf ^ 1);      // first K step of the next tile
        if (tid < G2_B) mr = epi.big_load(nby_ * G2_B, nbx_ * G2_B, tid);
      }
      const uint16_t* Ac = smem + buf * 2 * G2_TILE;
      const uint16_t* Bc = Ac + G2_TILE;
#pragma unroll
      for (int kk = 0; kk < G2_BK / 32; ++kk) {
        bf16x8 af[TM], bfr[TN];
        if (G2_VARIANT >= 2) {
#pragma unroll
          for (int b = 0; b < TN; ++b) bfr[b] = bf16x8{(short)(lane + t), 1, 2, 3, 4, 5, 6, (short)b};
#pragma unroll
          for (int a = 0; a < TM; ++a) af[a] = bf16x8{(short)(lane - t), 1, 2, 3, 4, 5, 6, (short)a};
        } else {
#pragma unroll
          for (int b = 0; b < TN; ++b) bfr[b] = load_frag<G2_B, G2_BK, true, false>(Bc, wn * 64 + b * 16, kk, lane);
#pragma unroll
          for (int a = 0; a < TM; ++a) af[a] = load_frag<G2_B, G2_BK, true, false>(Ac, wm * 128 + a * 16, kk, lane);
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
      if (last && next < ntiles && tid < G2_B) epi.big_store(mr, nbx_ * G2_B, tid, meta0 + (par ^ 1) * 2 * G2_B, G2_B);
      buf ^= 1;
    }
    __syncthreads();  // metadata of this tile visible (first tile); previous epilogue's scratch reads are over
    const TileCtx ctx{by * G2_B, bx * G2_B, wm, wn, lane, tid, bx, nbx, 0, scratch};
    const auto est = epi.template big_state<G2_B, G2_B, WM, WN, TM, TN>(ctx, meta0 + par * 2 * G2_B);
    epi.template finish<G2_B, G2_B, WM, WN, TM, TN>(acc, ctx, est);
    tile = next;
    bx = nbx_;
    by = nby_;
    par ^= 1;
  }
}

}  // namespace dprhot
