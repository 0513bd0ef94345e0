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

// ---- backward GEMMs at MFMA-bound sizes: operands that are mn-major in HBM ------------------------------------------
// dC_part = G^T Q reads both operands, dQ = G C its B operand, with the contraction index as the ROW of the matrix in
// HBM (dpr_task.py:98-105 backward).  Tile image [64 k][256 mn] bf16 (512-byte rows); the MFMA fragment (16 mn values x
// 32 k) comes out of ds_read_b64_tr_b16.  32 lanes of such a read are served together and touch rows k0..k0+3 and
// k0+8..k0+11 of ONE 32-byte column group, so group cg of row k is stored at slot cg ^ mswz(k) (8 distinct slots of the
// 256-byte bank row; the XOR stays inside the group's half row).  The DMA writes lane-linearly -- one instruction =
// two k rows -- so lane l fetches the column group that belongs at its slot: source swizzle, as for k-major tiles.
__device__ __forceinline__ bf16x8 g2_tr_frag(const uint16_t* T, int r0, int kk, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int k = kk * 32 + g * 8 + (i >> 2);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const uint16_t* p = T + k * G2_B + (((r0 >> 4) ^ mswz(k)) << 4) + (i & 3) * 4;  // mswz(k + 4) == mswz(k)
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p + 4 * G2_B));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// One 256x256 tile over the K range of slice bz (p.kchunk, a multiple of 64; K % 64 == 0), one workgroup per tile.
template <bool A_KMAJOR, bool B_KMAJOR, class Epi>
__device__ __forceinline__ void g2_tile_once(const GemmArgs& p, const Epi& epi, int bx, int by, int bz, int nbx, uint16_t* smem) {
  constexpr int WM = 2, WN = 4, TM = 8, TN = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = by * G2_B, n0 = bx * G2_B;
  const int kbeg = bz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nt = (kend - kbeg) / G2_BK;
  // one workgroup per tile: when finish() runs (behind the barrier that ends the K loop) the operand images are dead, and the whole
  // tile memory is the epilogue's scratch (EpiScaleF32 stages 8 waves x 16 x 68 floats there)
  float* const scratch = reinterpret_cast<float*>(smem);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const TileCtx ctx{m0, n0, wm, wn, lane, tid, bx, nbx, bz, scratch};
  const auto eraw = epi.template begin<G2_B, G2_B, WM, WN, TM, TN>(ctx);

  // per-lane source offsets (elements) of this wave's 4 DMA instructions per operand; the K step adds a uniform term
  unsigned oa[4], ob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if constexpr (A_KMAJOR) {
      const int row = wave * 32 + j * 8 + (lane >> 3);
      oa[j] = (unsigned)min(m0 + row, p.M - 1) * (unsigned)p.lda + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    } else {
      const int krow = (wave * 4 + j) * 2 + (lane >> 5), pos = lane & 31;
      const int col = ((((pos >> 1) ^ mswz(krow))) << 4) + (pos & 1) * 8;
      oa[j] = (unsigned)krow * (unsigned)p.lda + (unsigned)min(m0 + col, p.M - 8);
    }
    if constexpr (B_KMAJOR) {
      const int row = wave * 32 + j * 8 + (lane >> 3);
      ob[j] = (unsigned)min(n0 + row, p.N - 1) * (unsigned)p.ldb + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    } else {
      const int krow = (wave * 4 + j) * 2 + (lane >> 5), pos = lane & 31;
      const int col = ((((pos >> 1) ^ mswz(krow))) << 4) + (pos & 1) * 8;
      ob[j] = (unsigned)krow * (unsigned)p.ldb + (unsigned)min(n0 + col, p.N - 8);
    }
  }
  auto issue = [&](int t, int buf) {
    uint16_t* As = smem + buf * 2 * G2_TILE;
    uint16_t* Bs = As + G2_TILE;
    const int k0 = kbeg + t * G2_BK;
    const uint16_t* Ab = p.A + (A_KMAJOR ? (size_t)k0 : (size_t)k0 * p.lda);
    const uint16_t* Bb = p.B + (B_KMAJOR ? (size_t)k0 : (size_t)k0 * p.ldb);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(Ab + oa[j]), (g2_lds_ptr*)(As + (wave * 4 + j) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(Bb + ob[j]), (g2_lds_ptr*)(Bs + (wave * 4 + j) * 512), 16, 0, 0);
    }
  };
  issue(0, 0);
  const auto est = epi.template settle<G2_B, G2_B, WM, WN, TM, TN>(ctx, eraw);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
    const uint16_t* Ac = smem + (t & 1) * 2 * G2_TILE;
    const uint16_t* Bc = Ac + G2_TILE;
#pragma unroll
    for (int kk = 0; kk < G2_BK / 32; ++kk) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        if constexpr (B_KMAJOR) bfr[b] = load_frag<G2_B, G2_BK, true, false>(Bc, wn * 64 + b * 16, kk, lane);
        else bfr[b] = g2_tr_frag(Bc, wn * 64 + b * 16, kk, lane);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        if constexpr (A_KMAJOR) af[a] = load_frag<G2_B, G2_BK, true, false>(Ac, wm * 128 + a * 16, kk, lane);
        else af[a] = g2_tr_frag(Ac, wm * 128 + a * 16, kk, lane);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  __syncthreads();
  epi.template finish<G2_B, G2_B, WM, WN, TM, TN>(acc, ctx, est);
}

// dC_part = G^T Q (both operands mn-major) and dQ = G C (A k-major, B mn-major, split over K) in ONE launch: neither
// fills the chip alone when d is a few hundred (d / 256 column tiles).  Unit u -> XCD-contiguous order, then
// [dC tiles: context block major, d block minor | dQ units: (K slice, query block) major, d block minor], so that the
// units an XCD runs together share their G block in its L2.
template <class Epi>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm256_bwd_kernel(GemmArgs p1, Epi e1, int nbx1, int nby1, GemmArgs p2, Epi e2, int nbx2,
                                                                    int nby2, int splits2) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int n1 = nbx1 * nby1, n2 = nbx2 * nby2 * splits2, nwg = n1 + n2;
  const int wg = blockIdx.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
  int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  if (t < n1) {
    g2_tile_once<false, false, Epi>(p1, e1, t % nbx1, t / nbx1, 0, nbx1, smem);
  } else {
    t -= n1;
    const int bx = t % nbx2, rest = t / nbx2;
    g2_tile_once<true, false, Epi>(p2, e2, bx, rest % nby2, rest / nby2, nbx2, smem);
  }
}

}  // namespace dprhot
