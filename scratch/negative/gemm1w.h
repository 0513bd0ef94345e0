// gemm1w.h -- the 256x256 similarity GEMM with ONE wave per SIMD (dpr_task.py:98-105 at evaluation / retrieval / large-batch
// sizes): D[M,N] = A[M,K] * B[N,K]^T, both operands k-major bf16, persistent workgroups of 4 waves, each wave a 128 x 128
// quarter of the tile on v_mfma_f32_32x32x16_bf16 (16 accumulators of 16 registers = 256 of the wave's 512 registers).
//
// Why: with two waves per SIMD (gemm256.h, gemm8p.h) the matrix pipe is handed from wave to wave through s_barrier -- every
// 256 MFMA cycles cost ~90 cycles of hand-over, and the barrier-free variant of the same loop measures 69-72 % MFMA-busy
// cycles at best (profiles/r02_g8_ablation.txt).  A wave alone on its SIMD needs no hand-over: it issues the fragment reads of
// the NEXT k slice, then the 16 MFMAs of the current one; the reads and the LDS-DMA run in the shadow of its own MFMAs, and the
// only barrier left is the one that hands LDS images between the DMA and the readers, once per 32 MFMAs (1024 MFMA cycles),
// with the first MFMA behind it fed from registers.  LDS traffic drops by a third as well (each wave re-reads 128 + 128 rows per
// k instead of 128 + 64: 64 flop per LDS byte).
//
// K pipeline: the unit is a k-half (32 deep).  Ring of 4 slots x {A [256][32], B [256][32]} (16 KiB images, 128 KiB).
//   BLOCK1(s-1): reads F[0] <- (s, kk 0)     DMA first half of k-half s+3     16 MFMAs on F[1]   (k-half s-1, k slice 1)
//   BLOCK0(s)  : reads F[1] <- (s, kk 1)     DMA second half of k-half s+3    16 MFMAs on F[0]   (k-half s,   k slice 0)
//   s_waitcnt vmcnt(16) -- this wave's share of k-half s+1 has landed (s+2, s+3 stay in flight) -- lgkmcnt(0), s_barrier
//   BLOCK1(s)  : reads F[0] <- (s+1, kk 0) ...
// k-half s+3 goes into the slot of k-half s-1, whose last reads (s-1, kk 1) were waited for before the barrier between
// BLOCK0(s-1) and BLOCK1(s-1); it is read ~5 blocks (2500 MFMA cycles) after it was requested.  k-halves beyond the tile's last
// belong to the workgroup's next tile.
//
// Image layout: 64-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 2) & 3): the 16 lanes a ds_read_b128 serves
// together ({0-3,12-15,20-27}, ...) then hit 16 distinct 16-byte slots of the 256-byte bank row.  The DMA writes lane-linearly,
// so the permutation is applied on the source address.
//
// Accumulator layout (MFMA operands swapped, as gemm8p.h MF = 32): acc[a][b] (a, b < 4), lane (i = lane & 31, h = lane >> 5),
// register r: row wm*128 + a*32 + i, column wn*128 + b*32 + (r >> 2)*8 + h*4 + (r & 3).
#pragma once
#include "../../dpr_scale_amd/csrc/gemm8p.h"

namespace dprhot {

constexpr int W1_THREADS = 256;
constexpr int W1_IMG = 256 * 32;                                   // elements of one operand image of a k-half
constexpr size_t w1_ring_bytes = (size_t)4 * 2 * W1_IMG * 2;       // 128 KiB
constexpr size_t w1_scratch_bytes = 8 * 1024;
constexpr size_t w1_meta_bytes = (size_t)2 * 1024 * sizeof(int);
constexpr size_t w1_lds_total = w1_ring_bytes + w1_scratch_bytes + w1_meta_bytes;  // 144 KiB

struct Tile1w {
  int m0, n0;       // tile origin
  int wm, wn;       // wave position (2 x 2): rows wm * 128.., columns wn * 128..
  int lane, tid;
  int bx, by, nbx;
  float* scratch;   // LDS, w1_scratch_bytes
  int* meta;        // LDS, this tile's epilogue inputs: meta[0..255] column-wise, meta[256..511] row-wise raw words (Epi::meta_src)
};
struct W1Acc {
  f32x16 v[4][4];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[a][b][r] = 0.f;
  }
};

// Epilogue interface (all members __device__):
//   const void* meta_src(int m0, int n0, int e) const     e < 512: 4-byte aligned, always valid source address of entry e of the
//                                                         tile's inputs (e < 256: per column n0 + e, else per row m0 + e - 256);
//                                                         fetched by LDS-DMA while the previous tile is being computed
//   void finish(W1Acc& acc, const Tile1w&) const          every wave executes the same number of barriers in it; it runs with the
//                                                         next tile's first k-halves in flight: LDS only through g8_lds_* (asm)

// Fragment reads go through inline asm: hipcc orders a plain LDS read behind every LDS-DMA it cannot tell apart from the read's
// image with s_waitcnt vmcnt(0) (here: at the loop back edge, once per two K steps -- the whole DMA ring drained).  An asm read is
// invisible to its counters: every block opens with lgkmcnt(0) (W1_SETTLE / the hand-over) for the set it is about to multiply --
// read a whole block earlier, so the wait is free -- and only then issues the reads of the next set.
// base = LDS byte address of the lane's chunk in image row 0 .. 31 of slot 0 / 2 (ds offsets are 16 bits: two bases cover 4 slots).
template <int OFF>
__device__ __forceinline__ bf16x8 w1_ds_read(unsigned base) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(OFF));
  return v;
}

// VAR (scratch/g1probe.hip only): 8 no DMA after the prologue, 16 no fragment reads.
template <class Epi, int VAR = 0>
__global__ __launch_bounds__(W1_THREADS) void gemm1w_kernel(GemmArgs p, Epi epi, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nh = p.K / 32;  // k-halves per tile; the launcher guarantees K % 128 == 0 (nh % 4 == 0)
  const int ntiles = nbx * nby;
  float* const scratch = reinterpret_cast<float*>(smem + 8 * W1_IMG);
  int* const meta0 = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + w1_scratch_bytes);

  // LDS byte addresses of this lane's fragment chunk: row (lane & 31) of the wave's first 32-row block, k slice kk (chunk kk*2 +
  // (lane >> 5), swizzled); [kk][0]: slots 0 and 1, [kk][1]: slots 2 and 3.  Row blocks a / b and the slot add immediates.
  unsigned fa[2][2], fb[2][2];
  {
    const unsigned s0 = g8_lds_addr(smem);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ra = wm * 128 + (lane & 31), rb = wn * 128 + (lane & 31);
      const unsigned ca = (unsigned)(ra * 64 + (((kk * 2 + (lane >> 5)) ^ ((ra >> 2) & 3)) << 4));
      const unsigned cb = (unsigned)(rb * 64 + (((kk * 2 + (lane >> 5)) ^ ((rb >> 2) & 3)) << 4)) + W1_IMG * 2;
      fa[kk][0] = s0 + ca; fa[kk][1] = s0 + ca + 4 * W1_IMG * 2;
      fb[kk][0] = s0 + cb; fb[kk][1] = s0 + cb + 4 * W1_IMG * 2;
    }
  }
  // Byte offsets of this lane's source chunk for the wave's 4 DMA instructions per operand image: instruction j covers the
  // image rows (wave * 4 + j) * 16 + (lane >> 2), the lane at position (lane & 3) fetches chunk (lane & 3) ^ ((row >> 2) & 3).
  unsigned oa0, oa1, oa2, oa3, ob0, ob1, ob2, ob3;
  auto aim = [&](int bx, int by) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));  // opaque: nothing derived here stays live across the K loop
    const int ln = t_ & 63;
    const unsigned la = (unsigned)p.lda * 2u, lb = (unsigned)p.ldb * 2u;
    const int r0 = (wave * 4) * 16 + (ln >> 2);
    const int c = (ln & 3);
#define W1_OFF(J, ROWBASE, DIM, LD) ((unsigned)min((ROWBASE) + r0 + (J) * 16, (DIM) - 1) * (LD) + (unsigned)((c ^ (((r0 + (J) * 16) >> 2) & 3)) * 16))
    oa0 = W1_OFF(0, by * G2_B, p.M, la); oa1 = W1_OFF(1, by * G2_B, p.M, la); oa2 = W1_OFF(2, by * G2_B, p.M, la); oa3 = W1_OFF(3, by * G2_B, p.M, la);
    ob0 = W1_OFF(0, bx * G2_B, p.N, lb); ob1 = W1_OFF(1, bx * G2_B, p.N, lb); ob2 = W1_OFF(2, bx * G2_B, p.N, lb); ob3 = W1_OFF(3, bx * G2_B, p.N, lb);
#undef W1_OFF
  };
  auto slotA = [&](int slot) { return smem + slot * 2 * W1_IMG; };
  auto slotB = [&](int slot) { return smem + slot * 2 * W1_IMG + W1_IMG; };
  bool dma_off = false;
  // The DMA never branches: past the workgroup's last tile it keeps fetching (the same tile's k-halves again, into slots nobody
  // reads) -- a scalar branch between two MFMAs costs more than the load it skips; the kernel drains vmcnt before it ends.
  // Instruction J (0..3) of one half of the DMA of k-half U (relative to the tile being computed; U >= nh: next tile): HALF 0 = the
  // A image, 1 = the B image.  One instruction at a time, because a block interleaves them with its MFMAs (below).
#define W1_DMA1(HALF, U, SLOT, J, OA, OB)                                                                                              \
  if ((VAR & 8) && dma_off) {                                                                                                          \
  } else {                                                                                                                             \
    const size_t kb_ = (size_t)(((U) >= nh ? (U) - nh : (U)) * 64);                                                                    \
    if ((HALF) == 0) {                                                                                                                 \
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(reinterpret_cast<const char*>(p.A) + kb_ + (size_t)(OA)),                         \
                                       (g2_lds_ptr*)(slotA(SLOT) + (wave * 4 + (J)) * 512), 16, 0, 0);                                 \
    } else {                                                                                                                           \
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(reinterpret_cast<const char*>(p.B) + kb_ + (size_t)(OB)),                         \
                                       (g2_lds_ptr*)(slotB(SLOT) + (wave * 4 + (J)) * 512), 16, 0, 0);                                 \
    }                                                                                                                                  \
  }                                                                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define W1_DMA(HALF, U, SLOT)            \
  W1_DMA1(HALF, U, SLOT, 0, oa0, ob0);   \
  W1_DMA1(HALF, U, SLOT, 1, oa1, ob1);   \
  W1_DMA1(HALF, U, SLOT, 2, oa2, ob2);   \
  W1_DMA1(HALF, U, SLOT, 3, oa3, ob3);
  // fragment I (0..3: A row blocks, 4..7: B column blocks) of k slice KK of the k-half in SLOT -> F[I]  (asm read: settled by the
  // lgkmcnt(0) that opens the block which multiplies F)
#define W1_RD1(F, I, SLOT, KK)                                                                                                         \
  if constexpr (!(VAR & 16)) {                                                                                                         \
    F[I] = w1_ds_read<((SLOT) & 1) * 2 * W1_IMG * 2 + ((I) & 3) * 2048>((I) < 4 ? fa[KK][(SLOT) >> 1] : fb[KK][(SLOT) >> 1]);          \
  }                                                                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
#define W1_READ(F, SLOT, KK)                                                                                       \
  W1_RD1(F, 0, SLOT, KK); W1_RD1(F, 1, SLOT, KK); W1_RD1(F, 2, SLOT, KK); W1_RD1(F, 3, SLOT, KK);                  \
  W1_RD1(F, 4, SLOT, KK); W1_RD1(F, 5, SLOT, KK); W1_RD1(F, 6, SLOT, KK); W1_RD1(F, 7, SLOT, KK);
#define W1_SETTLE()                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
  __builtin_amdgcn_sched_barrier(0);
  // MFMA J (0..15) of a block: row block J >> 2, column block J & 3
#define W1_MF(F, J)                                                                                                                    \
  acc.v[(J) >> 2][(J) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[4 + ((J) & 3)], F[(J) >> 2], acc.v[(J) >> 2][(J) & 3], 0, 0, 0); \
  __builtin_amdgcn_sched_barrier(0);
  // One block: the 16 MFMAs of register set FC, with the 8 fragment reads of the next set FN and the 4 DMA instructions issued
  // BETWEEN them -- a wave issues in order, and everything it issues between two MFMAs is free as long as it fits in the 32 cycles
  // the matrix pipe spends on the first; issued ahead of the block instead, the same instructions leave the pipe idle (measured:
  // 104 vs 81 us for the 8192^2 x 768 problem).
#define W1_BLOCK(FC, FN, RSLOT, RKK, DHALF, DU, DSLOT)                                                             \
  W1_MF(FC, 0);  W1_RD1(FN, 0, RSLOT, RKK);                                                                        \
  W1_MF(FC, 1);  W1_RD1(FN, 4, RSLOT, RKK);                                                                        \
  W1_MF(FC, 2);  W1_RD1(FN, 1, RSLOT, RKK);                                                                        \
  W1_MF(FC, 3);  W1_RD1(FN, 5, RSLOT, RKK);                                                                        \
  W1_MF(FC, 4);  W1_RD1(FN, 2, RSLOT, RKK);                                                                        \
  W1_MF(FC, 5);  W1_RD1(FN, 6, RSLOT, RKK);                                                                        \
  W1_MF(FC, 6);  W1_RD1(FN, 3, RSLOT, RKK);                                                                        \
  W1_MF(FC, 7);  W1_RD1(FN, 7, RSLOT, RKK);                                                                        \
  W1_MF(FC, 8);  W1_DMA1(DHALF, DU, DSLOT, 0, oa0, ob0);                                                           \
  W1_MF(FC, 9);  W1_DMA1(DHALF, DU, DSLOT, 1, oa1, ob1);                                                           \
  W1_MF(FC, 10); W1_DMA1(DHALF, DU, DSLOT, 2, oa2, ob2);                                                           \
  W1_MF(FC, 11); W1_DMA1(DHALF, DU, DSLOT, 3, oa3, ob3);                                                           \
  W1_MF(FC, 12); W1_MF(FC, 13); W1_MF(FC, 14); W1_MF(FC, 15);
  // the barrier that hands k-half s+1 to the readers and the slot of k-half s to the DMA
#define W1_HANDOVER()                                            \
  g8_wait_vm<16>();                                              \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             \
  g8_bar();

  int tile = blockIdx.x, bx, by;
  g2_tile_of(tile, nbx, nby, bx, by);
  aim(bx, by);
  int next = tile, nbx_ = bx, nby_ = by;
  bool has_next = false;
  int par = 0;  // meta buffer of the current tile

  bf16x8 F0[8], F1[8];
  W1Acc acc;
  acc.zero();

  // ---- prologue: the first tile's epilogue inputs, k-halves 0, 1, 2 and the first half of k-half 3, the fragments of (0, kk 0)
  __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(by * G2_B, bx * G2_B, tid), (g2_lds_ptr*)(meta0 + wave * 64), 4, 0, 0);
  __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(by * G2_B, bx * G2_B, tid + 256), (g2_lds_ptr*)(meta0 + 256 + wave * 64), 4, 0, 0);
  W1_DMA(0, 0, 0); W1_DMA(1, 0, 0);
  W1_DMA(0, 1, 1); W1_DMA(1, 1, 1);
  W1_DMA(0, 2, 2); W1_DMA(1, 2, 2);
  W1_DMA(0, 3, 3);
  g8_wait_vm<20>();  // k-half 0 (and everything older) has landed
  g8_bar();
  W1_READ(F0, 0, 0);
  dma_off = true;

  while (true) {
    for (int s = 0; s < nh; s += 4) {
      // position Q (0..3) of the unrolled group: k-half s + Q in slot Q
#define W1_KHALF(Q, SWITCH)                                                                                       \
  /* BLOCK0: k slice 0 of k-half s + Q (F0); reads F1 <- its k slice 1; DMA: B image of k-half s + Q + 3 */      \
  W1_SETTLE();                                                                                                    \
  W1_BLOCK(F0, F1, Q, 1, 1, s + (Q) + 3, ((Q) + 3) & 3);                                                          \
  W1_HANDOVER();                                                                                                  \
  /* BLOCK1: k slice 1 (F1); reads F0 <- k slice 0 of k-half s + Q + 1; DMA: A image of k-half s + Q + 4 */      \
  if (SWITCH) {                                                                                                   \
    /* from here on the DMA front is in the workgroup's next tile */                                             \
    next = tile + (int)gridDim.x;                                                                                 \
    has_next = next < ntiles;                                                                                     \
    if (has_next) {                                                                                               \
      g2_tile_of(next, nbx, nby, nbx_, nby_);                                                                     \
      aim(nbx_, nby_);                                                                                            \
      int te_ = tid;                                                                                              \
      asm volatile("" : "+v"(te_));                                                                               \
      int* mn_ = meta0 + (par ^ 1) * 1024;                                                                        \
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(nby_ * G2_B, nbx_ * G2_B, te_), (g2_lds_ptr*)(mn_ + wave * 64), 4, 0, 0); \
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(nby_ * G2_B, nbx_ * G2_B, te_ + 256), (g2_lds_ptr*)(mn_ + 256 + wave * 64), 4, 0, 0); \
    }                                                                                                             \
  }                                                                                                               \
  W1_BLOCK(F1, F0, ((Q) + 1) & 3, 0, 0, s + (Q) + 4, Q);
      W1_KHALF(0, s == nh - 4);  // k-half s + 4 = nh is the first one of the next tile
      W1_KHALF(1, false);
      W1_KHALF(2, false);
      W1_KHALF(3, false);
    }
    // ---- epilogue of (bx, by)
    int te = tid;
    asm volatile("" : "+v"(te));  // opaque: the epilogue's lane arithmetic starts here, not above the K loop
    const Tile1w tc{by * G2_B, bx * G2_B, wm, wn, te & 63, te, bx, by, nbx, scratch, meta0 + par * 1024};
    epi.finish(acc, tc);
    acc.zero();
    if (!has_next) break;
    par ^= 1;
    tile = next;
    bx = nbx_;
    by = nby_;
  }
  g8_wait_vm<0>();  // nothing of this workgroup may still be writing LDS when its slot on the CU is handed on
}

// ---- epilogues -------------------------------------------------------------------------------------------------------------
// Inputs shared by the sim epilogues.  Raw words arrive by LDS-DMA (meta_src); meta_fix() turns them, one entry per thread, into
//   meta[e]        1 where tile column e is masked or outside the matrix, else 0
//   meta[256 + e]  gold column of tile row e (global column index), -1: none / row outside the matrix
//   meta[512 + w]  != 0 when a gold column of the rows fixed by wave w falls inside this tile
struct Epi1wBase {
  EpiSim sim;         // mask source (colmask / packed layout), M, N, inv_T, y, y_offset, gold
  const void* dummy;  // any valid device address: source of the entries that have no input (no mask, no labels)

  __device__ __forceinline__ const uint8_t* mask_byte(int n) const {
    if (sim.packed != nullptr) {
      const int r = n / sim.p_rows_c, j = n - r * sim.p_rows_c;
      return sim.packed + (size_t)(r * sim.p_rows_c + sim.p_n_ctx) * sim.p_row_bytes + min(j, sim.p_n_ctx - 1);
    }
    return sim.colmask != nullptr ? sim.colmask + n : nullptr;
  }
  __device__ __forceinline__ const void* meta_src(int m0, int n0, int e) const {
    if (e < 256) {
      const uint8_t* b = mask_byte(min(n0 + e, sim.N - 1));
      return b != nullptr ? reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(b) & ~(uintptr_t)3) : dummy;
    }
    return sim.y != nullptr ? static_cast<const void*>(reinterpret_cast<const int*>(sim.y) + 2 * min(m0 + e - 256, sim.M - 1)) : dummy;
  }
  __device__ __forceinline__ void meta_fix(const Tile1w& t) const {
    const int e = t.tid;  // 256 threads: column entry e and row entry e
    const int n = t.n0 + e;
    const int raw = g8_lds_read(t.meta + e);
    const uint8_t* b = mask_byte(min(n, sim.N - 1));
    int flag = n >= sim.N ? 1 : 0;
    if (b != nullptr) flag |= ((raw >> ((reinterpret_cast<uintptr_t>(b) & 3) * 8)) & 0xff) != 0 ? 1 : 0;
    if (sim.packed != nullptr) flag |= (min(n, sim.N - 1) % sim.p_rows_c) >= sim.p_n_ctx ? 1 : 0;
    const int rawy = g8_lds_read(t.meta + 256 + e);
    const int yi = (sim.y != nullptr && t.m0 + e < sim.M) ? rawy + (int)sim.y_offset : -1;
    g8_lds_write(t.meta + e, flag);
    g8_lds_write(t.meta + 256 + e, yi);
    const bool hit = yi >= t.n0 && yi < t.n0 + G2_B;
    const unsigned long long any = __ballot(hit);
    if (t.lane == 0) g8_lds_write(t.meta + 512 + (t.tid >> 6), any != 0ull ? 1 : 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g8_bar();
  }
  __device__ __forceinline__ bool tile_has_gold(const Tile1w& t) const {
    const g8_i32x4 f = g8_lds_read4(t.meta + 512);
    return (f[0] | f[1] | f[2] | f[3]) != 0;
  }
};

// fp32 logits (sim_score with a caller buffer): S = acc / T, masked columns -inf   (dpr_task.py:104,211)
struct Epi1wStore : Epi1wBase {
  float* S;
  __device__ __forceinline__ void finish(W1Acc& acc, const Tile1w& t) const {
    meta_fix(t);
    const int i = t.lane & 31, h = t.lane >> 5;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cn = t.wn * 128 + b * 32 + q * 8 + h * 4;  // tile column of the lane's 4 consecutive values
        const g8_i32x4 f = g8_lds_read4(t.meta + cn);
        const int n = t.n0 + cn;
        if (n >= sim.N) continue;  // N % 4 == 0: the four columns are inside or outside together
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int m = t.m0 + t.wm * 128 + a * 32 + i;
          if (m >= sim.M) continue;
          float4 v;
          v.x = f[0] != 0 ? -INFINITY : acc.v[a][b][q * 4 + 0] * sim.inv_T;
          v.y = f[1] != 0 ? -INFINITY : acc.v[a][b][q * 4 + 1] * sim.inv_T;
          v.z = f[2] != 0 ? -INFINITY : acc.v[a][b][q * 4 + 2] * sim.inv_T;
          v.w = f[3] != 0 ? -INFINITY : acc.v[a][b][q * 4 + 3] * sim.inv_T;
          *reinterpret_cast<float4*>(S + (size_t)m * sim.N + n) = v;
        }
      }
  }
};

}  // namespace dprhot
