// gemm2w.h -- the similarity GEMM as TWO independent 4-wave workgroups per CU (dpr_task.py:98-105 at evaluation / retrieval /
// large-batch sizes): D[M,N] = A[M,K] * B[N,K]^T, both operands k-major bf16, persistent workgroups, tile 256 x 128, each wave a
// 128 x 64 quarter on v_mfma_f32_32x32x16_bf16 (the accumulator layout of gemm8p.h: its epilogues run here unchanged).
//
// Why.  gemm8p.h keeps the matrix pipe at 66 % because its two waves per SIMD hand the pipe over through s_barrier twice per 16
// MFMAs, and because the epilogue (exponentials, stores and -- vmcnt retires in order -- the wait for those stores' acknowledgement)
// stops all eight waves.  A single wave per SIMD with no hand-over (scratch/negative/gemm1w.h) is no better: nobody covers its own
// barrier and issue gaps.  Here the two waves of a SIMD belong to DIFFERENT workgroups: no barrier couples them, the hardware
// arbiter interleaves their MFMAs, and whatever one workgroup is stalled on -- its k-half barrier, its LDS-DMA, its epilogue's
// VALU work, the drain of its G / logit stores -- is covered by the other's K loop.
//
// Per workgroup: ring of 3 k-halves (32 deep) x {A [256][32] 16 KiB, B [128][32] 8 KiB} = 72 KiB (+ 4 KiB of epilogue input words:
// two workgroups fill the CU's 160 KiB).  Per k-half s, in the slot s % 3:
//   BLOCK0(s): 8 MFMAs on F0 (k slice 0)    reads F1 <- (s, slice 1)      DMA: B image of k-half s+2
//   s_waitcnt vmcnt(6) [k-half s+1 has landed, s+2 stays in flight], lgkmcnt(0), s_barrier
//   BLOCK1(s): 8 MFMAs on F1 (k slice 1)    reads F0 <- (s+1, slice 0)    DMA: A image of k-half s+3 -> the slot of k-half s
// Fragment reads and DMA instructions sit BETWEEN the MFMAs of a block (an in-order wave issues them in the shadow of the MFMA it
// just issued); the reads are inline asm (hipcc would order plain LDS reads behind the DMA ring with vmcnt(0) at the loop back
// edge) and each block opens with the lgkmcnt(0) for the set it multiplies, issued a whole block earlier.
// Image layout: 64-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 2) & 3) (conflict-free for ds_read_b128's lane
// groups); the DMA writes lane-linearly, the permutation is applied on its source address.
#pragma once
#include "../../dpr_scale_amd/csrc/gemm8p.h"

namespace dprhot {

constexpr int W2_THREADS = 256;
constexpr int W2_BN = 128;                                  // tile columns
constexpr int W2_IMGA = 256 * 32, W2_IMGB = 128 * 32;       // elements of the A / B image of a k-half
constexpr int W2_SLOT = W2_IMGA + W2_IMGB;                  // 24 KiB
constexpr size_t w2_ring_bytes = (size_t)3 * W2_SLOT * 2;   // 72 KiB
constexpr size_t w2_meta_bytes = (size_t)1024 * sizeof(int);
constexpr size_t w2_lds_total = w2_ring_bytes + w2_meta_bytes;  // 76 KiB: two workgroups per CU

template <int OFF>
__device__ __forceinline__ bf16x8 w2_ds_read(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// Workgroup id -> tile of 256 x 128: g2_tile_of on the (nbx x nby) grid of such tiles -- every XCD gets a contiguous range of tile
// numbers, walked in groups of 8 tile rows x all columns, row fastest: the ~64 workgroups an XCD runs at a time form an 8 x 8 patch
// (8 A panels of 256 rows, 8 B panels of 128).
__device__ __forceinline__ void w2_tile_of(int wg, int nbx, int nby, int& bx, int& by) { g2_tile_of(wg, nbx, nby, bx, by); }

template <class Epi, int VAR = 0>
__global__ __launch_bounds__(W2_THREADS, 2) void gemm2w_kernel(GemmArgs p, Epi epi, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nh = p.K / 32;  // k-halves per tile (K % 32 == 0, nh >= 4)
  const int ntiles = nbx * nby;
  int* const meta = reinterpret_cast<int*>(smem + 3 * W2_SLOT);

  // LDS byte addresses of this lane's fragment chunks in slot 0: row (lane & 31) of the wave's first 32-row block, k slice kk
  // (chunk kk*2 + (lane >> 5), swizzled).  Row blocks add immediates, the slot a scalar.
  unsigned fa0, fa1, fb0, fb1;
  {
    const unsigned s0 = g8_lds_addr(smem);
    const int ra = wm * 128 + (lane & 31), rb = wn * 64 + (lane & 31);
    fa0 = s0 + (unsigned)(ra * 64 + (((0 + (lane >> 5)) ^ ((ra >> 2) & 3)) << 4));
    fa1 = s0 + (unsigned)(ra * 64 + (((2 + (lane >> 5)) ^ ((ra >> 2) & 3)) << 4));
    fb0 = s0 + (unsigned)(W2_IMGA * 2 + rb * 64 + (((0 + (lane >> 5)) ^ ((rb >> 2) & 3)) << 4));
    fb1 = s0 + (unsigned)(W2_IMGA * 2 + rb * 64 + (((2 + (lane >> 5)) ^ ((rb >> 2) & 3)) << 4));
  }
  // Byte offsets of this lane's source chunk for the wave's DMA instructions (16 rows of 64 bytes each: row (lane >> 2), the lane
  // at position (lane & 3) fetches chunk (lane & 3) ^ ((row >> 2) & 3)): A image rows (wave*4 + j)*16.., j < 4; B image rows
  // (wave*2 + j)*16.., j < 2.
  unsigned oa0, oa1, oa2, oa3, ob0, ob1;
  auto aim = [&](int bx, int by) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));  // opaque: nothing derived here stays live across the K loop
    const int ln = t_ & 63;
    const unsigned la = (unsigned)p.lda * 2u, lb = (unsigned)p.ldb * 2u;
    const int ra = (wave * 4) * 16 + (ln >> 2), rb = (wave * 2) * 16 + (ln >> 2);
    const int c = ln & 3;
#define W2_OFF(R, ROWBASE, DIM, LD) ((unsigned)min((ROWBASE) + (R), (DIM) - 1) * (LD) + (unsigned)((c ^ (((R) >> 2) & 3)) * 16))
    oa0 = W2_OFF(ra, by * 256, p.M, la); oa1 = W2_OFF(ra + 16, by * 256, p.M, la);
    oa2 = W2_OFF(ra + 32, by * 256, p.M, la); oa3 = W2_OFF(ra + 48, by * 256, p.M, la);
    ob0 = W2_OFF(rb, bx * W2_BN, p.N, lb); ob1 = W2_OFF(rb + 16, bx * W2_BN, p.N, lb);
#undef W2_OFF
  };
  auto fetch_meta = [&](int bx, int by) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(by * 256, bx * W2_BN, t_), (g2_lds_ptr*)(meta + wave * 64), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(by * 256, bx * W2_BN, t_ + 256), (g2_lds_ptr*)(meta + 256 + wave * 64), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(by * 256, bx * W2_BN, t_ + 512), (g2_lds_ptr*)(meta + 512 + wave * 64), 4, 0, 0);
  };

  // DMA front: k-half index inside its tile (dk), its slot (dslot), byte offset of the k-half in a row (dkb)
  int dk = 0, dslot = 0;
  bool dma_off = false;  // VAR & 8
  // instruction J of the A image (4 per wave) / B image (2 per wave) of the DMA front's k-half
#define W2_DMA_A(J, OA)                                                                                                   \
  if (!((VAR & 8) && dma_off))                                                                                            \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(reinterpret_cast<const char*>(p.A) + (size_t)(dk * 64) + (size_t)(OA)), \
                                     (g2_lds_ptr*)(smem + dslot * W2_SLOT + (wave * 4 + (J)) * 512), 16, 0, 0);             \
  __builtin_amdgcn_sched_barrier(0);
#define W2_DMA_B(J, OB)                                                                                                   \
  if (!((VAR & 8) && dma_off))                                                                                            \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(reinterpret_cast<const char*>(p.B) + (size_t)(dk * 64) + (size_t)(OB)), \
                                     (g2_lds_ptr*)(smem + dslot * W2_SLOT + W2_IMGA + (wave * 2 + (J)) * 512), 16, 0, 0);  \
  __builtin_amdgcn_sched_barrier(0);
  // fragment I (0..3: A row blocks, 4..5: B column blocks) of k slice KK of the k-half in the slot at byte offset SOFF -> F[I]
#define W2_RD1(F, I, KK, SOFF)                                                                                            \
  if constexpr (!(VAR & 16)) {                                                                                            \
    if ((I) < 4) F[I] = w2_ds_read<((I) & 3) * 2048>(((KK) ? fa1 : fa0) + (SOFF));                                        \
    else F[I] = w2_ds_read<((I) & 1) * 2048>(((KK) ? fb1 : fb0) + (SOFF));                                                \
  }                                                                                                                       \
  __builtin_amdgcn_sched_barrier(0);
  // MFMA J (0..7) of a block: row block J >> 1, column block J & 1
#define W2_MF(F, J)                                                                                                       \
  acc.v[(J) >> 1][(J) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[4 + ((J) & 1)], F[(J) >> 1], acc.v[(J) >> 1][(J) & 1], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);

  int tile = blockIdx.x, bx, by;
  w2_tile_of(tile, nbx, nby, bx, by);
  aim(bx, by);
  int next = tile + (int)gridDim.x, nbx_ = bx, nby_ = by;
  bool has_next = next < ntiles;
  if (has_next) w2_tile_of(next, nbx, nby, nbx_, nby_);
  // the DMA front moves on to the next k-half; past the tile's last one it enters the workgroup's next tile (past the last tile
  // it keeps fetching the same tile again into slots nobody reads: never a branch around a load)
  auto advance = [&]() {
    ++dk;
    dslot = dslot == 2 ? 0 : dslot + 1;
    if (dk == nh) {
      dk = 0;
      if (has_next) aim(nbx_, nby_);
    }
  };

  bf16x8 F0[6], F1[6];
  G8Acc acc;
  acc.zero();

  // ---- prologue: the first tile's epilogue words, k-halves 0 and 1 whole, the B image of k-half 2; fragments of (0, slice 0)
  fetch_meta(bx, by);
  W2_DMA_A(0, oa0); W2_DMA_A(1, oa1); W2_DMA_A(2, oa2); W2_DMA_A(3, oa3); W2_DMA_B(0, ob0); W2_DMA_B(1, ob1);
  advance();
  W2_DMA_A(0, oa0); W2_DMA_A(1, oa1); W2_DMA_A(2, oa2); W2_DMA_A(3, oa3); W2_DMA_B(0, ob0); W2_DMA_B(1, ob1);
  advance();
  W2_DMA_A(0, oa0); W2_DMA_A(1, oa1); W2_DMA_A(2, oa2); W2_DMA_A(3, oa3);
  g8_wait_vm<10>();  // k-half 0 (and the epilogue words) have landed
  g8_bar();
  {
    const unsigned so = 0;
    W2_RD1(F0, 0, 0, so); W2_RD1(F0, 1, 0, so); W2_RD1(F0, 2, 0, so); W2_RD1(F0, 3, 0, so); W2_RD1(F0, 4, 0, so); W2_RD1(F0, 5, 0, so);
  }
  dma_off = true;

  int cslot = 0;  // slot of the k-half being multiplied
  while (true) {
    for (int s = 0; s < nh; ++s) {
      const unsigned so = (unsigned)(cslot * W2_SLOT * 2);
      const int nslot = cslot == 2 ? 0 : cslot + 1;
      const unsigned sn = (unsigned)(nslot * W2_SLOT * 2);
      // BLOCK0: slice 0 (F0); reads F1 <- slice 1 of this k-half; DMA: the B image of the front's k-half (A went out a block ago)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      W2_MF(F0, 0); W2_RD1(F1, 0, 1, so);
      W2_MF(F0, 1); W2_RD1(F1, 4, 1, so);
      W2_MF(F0, 2); W2_RD1(F1, 1, 1, so);
      W2_MF(F0, 3); W2_RD1(F1, 5, 1, so);
      W2_MF(F0, 4); W2_RD1(F1, 2, 1, so);
      W2_MF(F0, 5); W2_RD1(F1, 3, 1, so);
      W2_MF(F0, 6); W2_DMA_B(0, ob0);
      W2_MF(F0, 7); W2_DMA_B(1, ob1);
      advance();
      // hand-over: k-half s+1 has landed for everybody; everybody is done reading k-half s
      g8_wait_vm<6>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      g8_bar();
      // BLOCK1: slice 1 (F1); reads F0 <- slice 0 of the next k-half; DMA: the A image of the front's k-half -> the slot just freed
      W2_MF(F1, 0); W2_RD1(F0, 0, 0, sn);
      W2_MF(F1, 1); W2_RD1(F0, 4, 0, sn);
      W2_MF(F1, 2); W2_RD1(F0, 1, 0, sn);
      W2_MF(F1, 3); W2_RD1(F0, 5, 0, sn);
      W2_MF(F1, 4); W2_RD1(F0, 2, 0, sn); W2_DMA_A(0, oa0);
      W2_MF(F1, 5); W2_RD1(F0, 3, 0, sn); W2_DMA_A(1, oa1);
      W2_MF(F1, 6); W2_DMA_A(2, oa2);
      W2_MF(F1, 7); W2_DMA_A(3, oa3);
      cslot = nslot;
    }
    // ---- epilogue of (bx, by)
    int te = tid;
    asm volatile("" : "+v"(te));  // opaque: the epilogue's lane arithmetic starts here, not above the K loop
    Tile8 tc{by * 256, bx * W2_BN, wm, wn, te & 63, te, bx, by, nbx, nullptr, meta};
    tc.ncol = W2_BN;
    tc.nthr = W2_THREADS;
    if (!has_next) g8_wait_vm<0>();  // nothing of this workgroup may still be writing LDS when its place on the CU is handed on
    epi.finish(acc, tc);
    acc.zero();
    if (!has_next) break;
    tile = next;
    bx = nbx_;
    by = nby_;
    next = tile + (int)gridDim.x;
    has_next = next < ntiles;
    if (has_next) w2_tile_of(next, nbx, nby, nbx_, nby_);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g8_bar();             // every wave is done with this tile's epilogue words
    fetch_meta(bx, by);   // the next epilogue's: a whole tile of time to land (older than every DMA wait that follows)
  }
}

}  // namespace dprhot
