// gemm_bf16.h -- LDS-tiled bf16 MFMA GEMM for gfx950 (CDNA4), fp32 accumulate, the one GEMM engine behind
//   sim   S  = Q  x C^T      A k-major, B k-major      (dpr_task.py:99-101  torch.matmul(q, c.T))
//   dQ       = G  x C        A k-major, B mn-major     (autograd of the above wrt q)
//   dC_part  = G^T x Q       A mn-major, B mn-major    (autograd of the above wrt c)
//
// D[M,N] = sum_k A(m,k) * B(k,n).  "k-major" operand: stored [rows][K], K contiguous (rows = M for A, N for
// B).  "mn-major" operand: stored [K][rows], rows contiguous.  No operand is ever transposed in HBM: an
// mn-major tile is staged row-for-row into LDS and handed to the matrix core through the gfx950 LDS
// transpose read (ds_read_b64_tr_b16).
//
// Geometry: 256 threads = 4 wave64 as WM x WN; block tile BM x BN, K step BK (64 for big tiles, up to 256 for
// the small-M tiles of the training shapes so that a whole K range is in flight at once -- those launches
// are latency-bound, not bandwidth-bound); v_mfma_f32_16x16x32_bf16.
// Staging is global -> VGPR -> LDS through PF register stages (the loads of 2-4 K steps are in flight at
// once), two LDS buffers, one barrier per K step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef DPRHOT_TIMING
extern __device__ unsigned long long g_dprhot_tm[64];
#define DPRHOT_TM(i)                                                                       \
  do {                                                                                     \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_dprhot_tm[i] = wall_clock64(); \
  } while (0)
// per-workgroup stamps (scratch/sk_timing.hip): slot i of workgroup blockIdx.x
extern __device__ unsigned long long g_dprhot_tmb[4 * 4096 * 8];
#define DPRHOT_TMB(k, i)                                                                                      \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_dprhot_tmb[((k) * 4096 + blockIdx.x) * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define DPRHOT_TM(i) do {} while (0)
#define DPRHOT_TMB(k, i) do {} while (0)
#endif

namespace dprhot {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KPAD = 0;    // k-major LDS rows are dense; bank conflicts are removed by an XOR swizzle of the 16-byte chunks
constexpr int MNPAD = 16;  // mn-major LDS row = R + 16 bf16

struct GemmArgs {
  const uint16_t* A;
  const uint16_t* B;
  int M, N, K;
  int lda, ldb;  // leading dimensions in elements
  int kchunk;    // K range per split (multiple of BK); >= K when not split
  uint16_t* Acopy = nullptr;  // fp32 operands only: where the staged tile is also stored as bf16 (same layout), or null
  uint16_t* Bcopy = nullptr;
};

template <int R, int BK, bool KMAJOR>
struct TileGeom {
  static constexpr int kRowStride = KMAJOR ? (BK + KPAD) : (R == 128 ? R : R + MNPAD);  // elements
  static constexpr int kRows = KMAJOR ? R : BK;
  static constexpr int kElems = kRows * kRowStride;
  static constexpr int kChunks = R * BK / 8;  // 16-byte chunks in the tile
  static constexpr int kIters = (kChunks + 255) / 256;
};

// F32: the operand lives in HBM as fp32 (encoder output); a chunk of 8 values is two 16-byte loads, rounded to
// bf16 (RNE) on its way into LDS -- the separate cast launch and its round trip through HBM disappear.
template <int R, int BK, bool KMAJOR, bool F32 = false>
struct StageRegs {
  uint4 v[TileGeom<R, BK, KMAJOR>::kIters * (F32 ? 2 : 1)];
};

// two fp32 -> one dword of two bf16, round-to-nearest-even: a single v_cvt_pk_bf16_f32 on gfx950
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pack_bf16_rne(uint32_t a, uint32_t b) {  // two fp32 bit patterns -> bf16 pair
  return cvt_pk_bf16(__uint_as_float(a), __uint_as_float(b));
}

// k-major LDS image: row r holds BK bf16 = BK/8 chunks of 16 bytes; chunk c of row r lives at chunk position
// swz(r, c).  ds_read_b128 serves a wave in four 16-lane groups over a 256-byte bank row (16 slots of 16 bytes); a
// fragment read touches 16 consecutive rows at one chunk column, so the swizzle must spread 16 rows over the 16
// slots: BK=64 (128-byte rows, two rows per bank row): c ^ ((r >> 1) & 7);  BK>=128 (row = k * 256 bytes): c ^ (r & 15).
// Both are conflict-free for the b128 lane groups (checked against the group table of MI355X_MICROARCH.md) and keep
// the 8-lane write groups of ds_write_b128 inside one row.  (The padded layout it replaces measured 32 % extra
// LDS cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.)
// mn-major LDS image of a 128-wide tile: k row = 256 bytes = one bank row = eight 32-byte column groups (the unit one
// 4-lane quarter of a ds_read_b64_tr_b16 group fetches).  The 32 lanes served together read rows k0..k0+3 and
// k0+8..k0+11 of ONE column group, so group cg of row k is stored at slot cg ^ mswz(k): 8 distinct slots.
__device__ __forceinline__ int mswz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

template <int BK>
__device__ __forceinline__ int kswz(int row, int chunk) {
  if constexpr (BK == 64) return chunk ^ ((row >> 1) & 7);
  else return chunk ^ (row & 15);
}

// Per-thread staging plan, computed ONCE per tile: element offset of each of the thread's chunks at K step 0 and its
// LDS position.  Per K step only a wave-uniform offset is added (the earlier per-step 64-bit index arithmetic was
// ~40 % of all VALU instructions of a large GEMM).
template <int R, int BK, bool KMAJOR>
struct StagePlan {
  size_t eoff[TileGeom<R, BK, KMAJOR>::kIters];  // element offset in the operand (row clamped into the matrix)
  int lds[TileGeom<R, BK, KMAJOR>::kIters];      // element offset in the LDS tile
  int kpos[TileGeom<R, BK, KMAJOR>::kIters];     // k position of the chunk inside a K step
};

template <int R, int BK, bool KMAJOR>
__device__ __forceinline__ void stage_plan(StagePlan<R, BK, KMAJOR>& pl, int ld, int r0, int rdim, int tid) {
  using G = TileGeom<R, BK, KMAJOR>;
#pragma unroll
  for (int it = 0; it < G::kIters; ++it) {
    const int c = tid + it * 256;
    if constexpr (KMAJOR) {
      constexpr int CPR = BK / 8;  // chunks per row
      const int row = c / CPR, kc = c % CPR;
      pl.eoff[it] = (size_t)min(r0 + row, rdim - 1) * ld + kc * 8;
      pl.lds[it] = row * G::kRowStride + kswz<BK>(row, kc) * 8;
      pl.kpos[it] = kc * 8;
    } else {
      constexpr int CPR = R / 8;  // chunks per k-row
      const int krow = c / CPR, mc = c % CPR;
      pl.eoff[it] = (size_t)krow * ld + min(r0 + mc * 8, rdim - 8);
      if constexpr (R == 128) pl.lds[it] = krow * G::kRowStride + (((mc >> 1) ^ mswz(krow)) << 4) + (mc & 1) * 8;
      else pl.lds[it] = krow * G::kRowStride + mc * 8;
      pl.kpos[it] = krow;
    }
  }
}

// global -> registers.  Branch-free: every load executes, on an in-range address.  Rows of the M/N dimension beyond
// the matrix only feed output rows/columns that are never stored (clamped in the plan); positions beyond the K
// range would enter every sum: their address falls back to the operand base and stage_store zeroes the chunk on its
// way into LDS (nothing here *uses* a loaded value -- that would park an s_waitcnt right behind the load).
// kbase = first k of this K step; koff = kbase (k-major) or kbase * ld (mn-major) elements, wave-uniform.
template <int R, int BK, bool KMAJOR, bool F32>
__device__ __forceinline__ void stage_load(StageRegs<R, BK, KMAJOR, F32>& regs, const StagePlan<R, BK, KMAJOR>& pl,
                                           const uint16_t* __restrict__ P, size_t koff, int kbase, int kend, int tid) {
  using G = TileGeom<R, BK, KMAJOR>;
  constexpr int ES = F32 ? 2 : 1;  // element size in uint16 units
#pragma unroll
  for (int it = 0; it < G::kIters; ++it) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u), w = make_uint4(0u, 0u, 0u, 0u);
    if (G::kChunks % 256 == 0 || tid + it * 256 < G::kChunks) {
      const bool kok = kbase + pl.kpos[it] < kend;
      const uint16_t* src = kok ? P + (pl.eoff[it] + koff) * ES : P;
      v = *reinterpret_cast<const uint4*>(src);
      if constexpr (F32) w = *reinterpret_cast<const uint4*>(src + 8);
    }
    if constexpr (F32) {
      regs.v[2 * it] = v;
      regs.v[2 * it + 1] = w;
    } else {
      regs.v[it] = v;
    }
  }
}

// registers -> LDS tile (fp32 operands are rounded to bf16 here; copy != nullptr: the bf16 image of the chunk is also
// written to HBM at the operand's own position -- the backward GEMMs read that copy)
template <int R, int BK, bool KMAJOR, bool F32>
__device__ __forceinline__ void stage_store(const StageRegs<R, BK, KMAJOR, F32>& regs, const StagePlan<R, BK, KMAJOR>& pl, uint16_t* T,
                                            int tid, uint16_t* copy, size_t koff, int kbase, int kend) {
  using G = TileGeom<R, BK, KMAJOR>;
#pragma unroll
  for (int it = 0; it < G::kIters; ++it) {
    if (G::kChunks % 256 == 0 || tid + it * 256 < G::kChunks) {
      uint4 v;
      if constexpr (F32) {
        const uint4 a = regs.v[2 * it], b = regs.v[2 * it + 1];
        v = make_uint4(pack_bf16_rne(a.x, a.y), pack_bf16_rne(a.z, a.w), pack_bf16_rne(b.x, b.y), pack_bf16_rne(b.z, b.w));
      } else {
        v = regs.v[it];
      }
      const bool kok = kbase + pl.kpos[it] < kend;
      if (!kok) v = make_uint4(0u, 0u, 0u, 0u);  // beyond the K range: contributes nothing
      *reinterpret_cast<uint4*>(T + pl.lds[it]) = v;
      if constexpr (F32) {
        if (copy != nullptr && kok) *reinterpret_cast<uint4*>(copy + pl.eoff[it] + koff) = v;
      }
    }
  }
}

// MFMA operand fragment for the 16 rows r0..r0+15 and the 32-wide k slice kk of an LDS tile.
// Lane l = (g = l >> 4, i = l & 15) receives row r0 + i, k = kk*32 + g*8 + {0..7}.
template <int R, int BK, bool KMAJOR, bool USE_TR>
__device__ __forceinline__ bf16x8 load_frag(const uint16_t* T, int r0, int kk, int lane) {
  using G = TileGeom<R, BK, KMAJOR>;
  const int i = lane & 15, g = lane >> 4;
  if constexpr (KMAJOR) {
    const int row = r0 + i;
    return *reinterpret_cast<const bf16x8*>(T + row * G::kRowStride + kswz<BK>(row, kk * 4 + g) * 8);
  } else if constexpr (USE_TR) {
    // ds_read_b64_tr_b16: within each 16-lane group, source lane s supplies 4 contiguous bf16 =
    // row (s >> 2), columns 4*(s & 3).. of a 4 x 16 block; result lane i receives column i of that block.
    // (verified on MI355X by csrc/selftest "trdump")
    const int k = kk * 32 + g * 8 + (i >> 2);
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const uint16_t *p, *ph;
    if constexpr (R == 128) {  // swizzled 32-byte column groups (mswz(k + 4) == mswz(k): one swizzle for both reads)
      p = T + k * G::kRowStride + (((r0 >> 4) ^ mswz(k)) << 4) + (i & 3) * 4;
      ph = p + 4 * G::kRowStride;
    } else {
      p = T + k * G::kRowStride + r0 + (i & 3) * 4;
      ph = p + 4 * G::kRowStride;
    }
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(ph));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  } else {
    // plain 16-bit gathers (slow; kept as the cross-check of the transpose read)
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kk * 32 + g * 8 + j;
      if constexpr (R == 128) r[j] = (short)T[k * G::kRowStride + (((r0 >> 4) ^ mswz(k)) << 4) + i];
      else r[j] = (short)T[k * G::kRowStride + r0 + i];
    }
    return r;
  }
}

template <int BM, int BN, int BK, bool A_KMAJOR, bool B_KMAJOR>
constexpr size_t gemm_lds_bytes() {
  const size_t tiles = 2 * (size_t)(TileGeom<BM, BK, A_KMAJOR>::kElems + TileGeom<BN, BK, B_KMAJOR>::kElems) * sizeof(uint16_t);
  const size_t epi = (size_t)BM * 4 * 2 * sizeof(float);  // epilogue scratch: (max,sum) per row per wave column
  return tiles > epi ? tiles : epi;
}

// What an epilogue sees of one workgroup's tile.
struct TileCtx {
  int m0, n0;      // tile origin
  int wm, wn;      // this wave's position in the WM x WN grid
  int lane, tid;
  int bx, nbx;     // column-tile index and count
  int bz;          // split-K slab
  float* scratch;  // the LDS tile memory, free for reuse (all waves are past the last barrier)
};

// One workgroup tile: D[m0.., n0..] over K range of split bz; then epi.finish(acc, ctx).
template <int BM, int BN, int BK, int WM, int WN, bool A_KMAJOR, bool B_KMAJOR, bool USE_TR, class Epi, bool A_F32 = false,
          bool B_F32 = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, const Epi& epi, int bx, int by, int bz, int nbx, uint16_t* smem) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  using GA = TileGeom<BM, BK, A_KMAJOR>;
  using GB = TileGeom<BN, BK, B_KMAJOR>;
  uint16_t* const As0 = smem;
  uint16_t* const Bs0 = smem + 2 * GA::kElems;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nt = (kend - kbeg + BK - 1) / BK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the epilogue's own global reads (labels, mask bytes, the grad_output scalar) are issued now, so that their
  // latency hides under the K loop instead of trailing it
  DPRHOT_TM(0);
  const TileCtx ctx{m0, n0, wm, wn, lane, tid, bx, nbx, bz, reinterpret_cast<float*>(smem)};
  const auto eraw = epi.template begin<BM, BN, WM, WN, TM, TN>(ctx);

  // PF register stages: the loads of K steps t .. t+PF-1 are in flight together (a small-M launch is bound by
  // the latency of dependent loads, not by bandwidth); LDS is double-buffered, one barrier per K step.
  constexpr int kStageRegs = GA::kIters * (A_F32 ? 2 : 1) + GB::kIters * (B_F32 ? 2 : 1);
  // 128x128 tiles keep 64 accumulator registers per lane: two stages keep VGPR+AGPR <= 256, i.e. two waves per
  // SIMD (the third stage cost a whole wave of occupancy: -25 % measured); small tiles are latency-bound and take
  // as many stages as cover their K range.
  constexpr int PF = (BM * BN >= 128 * 128) ? 2
                     : (kStageRegs <= 6 ? 4 : ((kStageRegs <= 8 || (BM * BN <= 32 * 32 && kStageRegs <= 16)) ? 3 : 2));
  StageRegs<BM, BK, A_KMAJOR, A_F32> ra[PF];
  StageRegs<BN, BK, B_KMAJOR, B_F32> rb[PF];
  StagePlan<BM, BK, A_KMAJOR> pa;
  StagePlan<BN, BK, B_KMAJOR> pb;
  stage_plan<BM, BK, A_KMAJOR>(pa, p.lda, m0, p.M, tid);
  stage_plan<BN, BK, B_KMAJOR>(pb, p.ldb, n0, p.N, tid);
  const size_t astep = A_KMAJOR ? (size_t)1 : (size_t)p.lda;  // operand elements per unit of k
  const size_t bstep = B_KMAJOR ? (size_t)1 : (size_t)p.ldb;
  uint16_t* const acopy = (A_F32 && bx == 0) ? p.Acopy : nullptr;  // each operand row is copied by exactly one tile column/row
  uint16_t* const bcopy = (B_F32 && by == 0) ? p.Bcopy : nullptr;
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nt) {
      const int kb = kbeg + u * BK;
      stage_load<BM, BK, A_KMAJOR, A_F32>(ra[u], pa, p.A, kb * astep, kb, kend, tid);
      stage_load<BN, BK, B_KMAJOR, B_F32>(rb[u], pb, p.B, kb * bstep, kb, kend, tid);
    }

  // The epilogue's prefetched words are consumed HERE: they were issued before the tile loads, so this wait is a
  // counted vmcnt(<tile loads>) that costs nothing extra.  Consuming them only in the epilogue would force a
  // vmcnt(0) there (the counter is in-order and also counts stores): a full store round trip, +1.7 us measured.
  const auto est = epi.template settle<BM, BN, WM, WN, TM, TN>(ctx, eraw);

  for (int t0 = 0; t0 < nt; t0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int t = t0 + u;
      if (t < nt) {  // uniform across the workgroup
        const int cur = t & 1;
        uint16_t* Ac = As0 + cur * GA::kElems;
        uint16_t* Bc = Bs0 + cur * GB::kElems;
        if (t == 0) DPRHOT_TM(1);
        const int kb = kbeg + t * BK;
        stage_store<BM, BK, A_KMAJOR, A_F32>(ra[u], pa, Ac, tid, acopy, kb * astep, kb, kend);
        stage_store<BN, BK, B_KMAJOR, B_F32>(rb[u], pb, Bc, tid, bcopy, kb * bstep, kb, kend);
        if (t == 0) DPRHOT_TM(2);
        __syncthreads();  // tile t visible; every wave is past its MFMAs on this buffer (step t-2)
        if (t == 0) DPRHOT_TM(3);
        if (t + PF < nt) {
          const int kn = kbeg + (t + PF) * BK;
          stage_load<BM, BK, A_KMAJOR, A_F32>(ra[u], pa, p.A, kn * astep, kn, kend, tid);
          stage_load<BN, BK, B_KMAJOR, B_F32>(rb[u], pb, p.B, kn * bstep, kn, kend, tid);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
          bf16x8 af[TM], bfr[TN];
#pragma unroll
          for (int a = 0; a < TM; ++a) af[a] = load_frag<BM, BK, A_KMAJOR, USE_TR>(Ac, wm * (BM / WM) + a * 16, kk, lane);
#pragma unroll
          for (int b = 0; b < TN; ++b) bfr[b] = load_frag<BN, BK, B_KMAJOR, USE_TR>(Bc, wn * (BN / WN) + b * 16, kk, lane);
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  }
  DPRHOT_TM(4);
  __syncthreads();  // the epilogue reuses the tile memory as scratch
  DPRHOT_TM(5);

  epi.template finish<BM, BN, WM, WN, TM, TN>(acc, ctx, est);
  DPRHOT_TM(6);
}

template <int BM, int BN, int BK, int WM, int WN, bool A_KMAJOR, bool B_KMAJOR, bool USE_TR, class Epi, bool A_F32 = false,
          bool B_F32 = false>
__global__ __launch_bounds__(256, (A_F32 && B_F32) ? 1 : 2) void gemm_bf16_kernel(GemmArgs p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  gemm_tile<BM, BN, BK, WM, WN, A_KMAJOR, B_KMAJOR, USE_TR, Epi, A_F32, B_F32>(p, epi, blockIdx.x, blockIdx.y, blockIdx.z,
                                                                                gridDim.x, smem);
}

template <int BM_, int BN_, int BK_, bool AK_, bool BKM_, bool TR_>
struct GemmCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr bool AK = AK_, BKM = BKM_, TR = TR_;
  static constexpr size_t lds = gemm_lds_bytes<BM_, BN_, BK_, AK_, BKM_>();
};

// Two independent GEMMs in ONE launch (horizontal fusion): linear block ids [0, n1) run problem 1, the rest
// problem 2 (with split-K slabs).  Used for the backward pair dC_part = G^T Q and dQ = G C, which share only
// their input G.
template <class Cfg1, class Cfg2, class Epi1, class Epi2>
__global__ __launch_bounds__(256, 2) void gemm_pair_kernel(GemmArgs p1, Epi1 e1, int nbx1, int nby1, GemmArgs p2, Epi2 e2, int nbx2,
                                                        int nby2) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int n1 = nbx1 * nby1;
  // (An XCD-contiguous renumbering with the shorter tile dimension fastest -- so that tiles re-reading the same operand tile meet
  //  in one L2 -- was measured on the router-width backward and lost: 63.3 -> 70.7 us.  Neighbouring d tiles then store to the same
  //  rows of dC at the same time from one XCD; the plain order spreads every row's stores over the XCDs.)
  int id = blockIdx.x;
  if (id < n1) {
    gemm_tile<Cfg1::BM, Cfg1::BN, Cfg1::BK, 2, 2, Cfg1::AK, Cfg1::BKM, Cfg1::TR, Epi1>(p1, e1, id % nbx1, id / nbx1, 0, nbx1, smem);
  } else {
    id -= n1;
    const int per = nbx2 * nby2;
    const int bz = id / per;
    id -= bz * per;
    gemm_tile<Cfg2::BM, Cfg2::BN, Cfg2::BK, 2, 2, Cfg2::AK, Cfg2::BKM, Cfg2::TR, Epi2>(p2, e2, id % nbx2, id / nbx2, bz, nbx2, smem);
  }
}

// ---- epilogues --------------------------------------------------------------------------------------------
// C/D layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg

// Reductions over the 16 lanes of a DPP row (= the 16 columns of one MFMA fragment row) with row_ror: pure VALU
// data movement.  (__shfl_xor lowers to ds_bpermute + s_waitcnt lgkmcnt(0): ~120 cycles per step, serialised --
// 32 of them cost the stats epilogue 1.6 us before this.)
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_row_ror<8>(v));
  v = fmaxf(v, dpp_row_ror<4>(v));
  v = fmaxf(v, dpp_row_ror<2>(v));
  v = fmaxf(v, dpp_row_ror<1>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_row_ror<8>(v);
  v += dpp_row_ror<4>(v);
  v += dpp_row_ror<2>(v);
  v += dpp_row_ror<1>(v);
  return v;
}

__device__ __forceinline__ void ms_merge(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  if (M == -INFINITY) { m = M; s = 0.f; return; }
  s = s * __expf(m - M) + s2 * __expf(m2 - M);
  m = M;
}

// sim_score epilogue (dpr_task.py:104,211): * inv_T, masked columns -> -inf, fp32 store (optional) and, for the
// fused training forward, the per-(row, column tile) softmax statistics and the gold logit, so that the row
// logsumexp never needs a second pass over S.
struct EpiSim {
  float* S;                // [M,N] or nullptr
  const uint8_t* colmask;  // [N] or nullptr
  int M, N;
  float inv_T;
  float* part_m;     // [M][nbx] running max per column tile, or nullptr
  float* part_s;     // [M][nbx] sum exp(S - max)
  const int64_t* y;  // [M] gold column (minus y_offset)
  int64_t y_offset;
  float* gold;       // [M] gold logit
  unsigned long long* zero_words;  // words to clear for the next kernel (tile (0,0) clears them), or nullptr
  int n_zero;
  size_t slab_stride = 0;  // split-K (statistics off): split bz stores its partial logits at S + bz * slab_stride
  // Packed multi-rank layout (dprhot_pack_ctx): when `packed` is set the column mask is read straight from the gathered
  // buffer -- column n = (rank r, local j): j >= p_n_ctx (mask/padding rows) is masked, else the byte j of rank r's
  // mask row block -- and the separate unpack launch disappears.
  const uint8_t* packed = nullptr;
  int p_rows_c = 1, p_n_ctx = 0, p_row_bytes = 0;

  __device__ __forceinline__ uint8_t mask_at(int n) const {
    if (packed != nullptr) {
      const int r = n / p_rows_c, j = n - r * p_rows_c;
      const uint8_t b = packed[(size_t)(r * p_rows_c + p_n_ctx) * p_row_bytes + min(j, p_n_ctx - 1)];
      return j >= p_n_ctx ? (uint8_t)1 : b;
    }
    return colmask != nullptr ? colmask[n] : (uint8_t)0;
  }

  // the byte alone (a load whose value nothing in begin() looks at) and what the packed layout adds to it without a load: mask_at's
  // select on the loaded byte made begin() wait for each of its TN mask loads in turn (four trips to memory before the first tile load)
  __device__ __forceinline__ uint8_t mask_raw(int n) const {
    if (packed != nullptr) {
      const int r = n / p_rows_c, j = n - r * p_rows_c;
      return packed[(size_t)(r * p_rows_c + p_n_ctx) * p_row_bytes + min(j, p_n_ctx - 1)];
    }
    return colmask != nullptr ? colmask[n] : (uint8_t)0;
  }
  __device__ __forceinline__ bool mask_pad(int n) const { return packed != nullptr && n - (n / p_rows_c) * p_rows_c >= p_n_ctx; }

  // begin(): raw loads only, issued back to back BEFORE the tile loads (nothing is used here, or the compiler
  // parks one s_waitcnt per load at the top of the kernel); settle(): turned into what finish() needs, after the
  // tile loads have been issued.
  template <int TM, int TN>
  struct Raw {
    int64_t yraw[TM][4];  // label of each row this lane holds (clamped row index)
    uint8_t mraw[TN];     // mask byte of each column this lane holds (clamped column index)
  };
  template <int TM, int TN>
  struct State {
    int yi[TM][4];    // gold column of each row (-1: none)
    bool masked[TN];  // column masked or outside the matrix
  };

  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ State<TM, TN> settle(const TileCtx& c, const Raw<TM, TN>& raw) const {
    const int i = c.lane & 15;
    State<TM, TN> st;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = c.n0 + c.wn * (BN / WN) + b * 16 + i;
      st.masked[b] = n >= N || raw.mraw[b] != 0 || mask_pad(min(n, N - 1));
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) st.yi[a][r] = part_m != nullptr ? (int)(raw.yraw[a][r] + y_offset) : -1;
    return st;
  }

  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ Raw<TM, TN> begin(const TileCtx& c) const {
    const int i = c.lane & 15, g = c.lane >> 4;
    Raw<TM, TN> st;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = min(c.n0 + c.wn * (BN / WN) + b * 16 + i, N - 1);
      st.mraw[b] = mask_raw(n);
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = min(c.m0 + c.wm * (BM / WM) + a * 16 + g * 4 + r, M - 1);
        st.yraw[a][r] = part_m != nullptr ? y[m] : (int64_t)-1;
      }
    return st;
  }

  // gemm256.h keeps no epilogue state in registers across its K loop: thread t < BM fetches the label of row t and
  // the mask byte of column t of the NEXT tile (big_load, a tile ahead), parks them in LDS (big_store) and the
  // epilogue rebuilds its State from there (big_state).
  struct BigRegs { int64_t y; uint8_t m; };
  __device__ __forceinline__ BigRegs big_load(int m0, int n0, int t) const {
    BigRegs r;
    r.y = part_m != nullptr ? y[min(m0 + t, M - 1)] : (int64_t)-1;
    r.m = mask_at(min(n0 + t, N - 1));
    return r;
  }
  __device__ __forceinline__ void big_store(const BigRegs& r, int n0, int t, int* meta, int BM) const {
    meta[t] = part_m != nullptr ? (int)(r.y + y_offset) : -1;
    meta[BM + t] = (n0 + t >= N || r.m != 0) ? 1 : 0;
  }
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ State<TM, TN> big_state(const TileCtx& c, const int* meta) const {
    const int i = c.lane & 15, g = c.lane >> 4;
    State<TM, TN> st;
#pragma unroll
    for (int b = 0; b < TN; ++b) st.masked[b] = meta[BM + c.wn * (BN / WN) + b * 16 + i] != 0;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) st.yi[a][r] = meta[c.wm * (BM / WM) + a * 16 + g * 4 + r];
    return st;
  }

  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ void finish(f32x4 (&acc)[TM][TN], const TileCtx& c, const State<TM, TN>& st) const {
    const int i = c.lane & 15, g = c.lane >> 4;
    if (zero_words != nullptr && c.bx == 0 && c.m0 == 0 && c.bz == 0 && c.tid < n_zero) zero_words[c.tid] = 0ull;
    const bool (&masked)[TN] = st.masked;
    float* const S = this->S != nullptr ? this->S + (size_t)c.bz * slab_stride : nullptr;
    float* red_m = c.scratch;            // [BM][WN]
    float* red_s = c.scratch + BM * WN;  // [BM][WN]
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = c.wm * (BM / WM) + a * 16 + g * 4 + r;
        const int m = c.m0 + lrow;
        float v[TN];
        float mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          v[b] = masked[b] ? -INFINITY : acc[a][b][r] * inv_T;
          mx = fmaxf(mx, v[b]);
        }
        if (S != nullptr && m < M) {
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const int n = c.n0 + c.wn * (BN / WN) + b * 16 + i;
            if (n < N) S[(size_t)m * N + n] = v[b];
          }
        }
        if (part_m != nullptr) {
          mx = row16_max(mx);  // row max over this wave's columns (all 16 lanes of the row get it)
          float sm = 0.f;
          if (mx != -INFINITY) {
#pragma unroll
            for (int b = 0; b < TN; ++b) sm += __expf(v[b] - mx);
          }
          sm = row16_sum(sm);
          if (i == 0) { red_m[lrow * WN + c.wn] = mx; red_s[lrow * WN + c.wn] = sm; }
          if (m < M) {
#pragma unroll
            for (int b = 0; b < TN; ++b)
              if (st.yi[a][r] == c.n0 + c.wn * (BN / WN) + b * 16 + i) gold[m] = v[b];
          }
        }
      }
    }
    if (part_m != nullptr) {
      __syncthreads();
      if (c.tid < BM && c.m0 + c.tid < M) {
        float mx = red_m[c.tid * WN], sm = red_s[c.tid * WN];
#pragma unroll
        for (int w = 1; w < WN; ++w) ms_merge(mx, sm, red_m[c.tid * WN + w], red_s[c.tid * WN + w]);
        part_m[(size_t)(c.m0 + c.tid) * c.nbx + c.bx] = mx;
        part_s[(size_t)(c.m0 + c.tid) * c.nbx + c.bx] = sm;
      }
    }
  }
};

// Training forward in ONE pass on the 128 x 128 tile (round 6; the 256 x 256 family's Epi8StatsP, gemm8p.h, for the shapes whose 256-wide
// tiles would leave most of the chip idle: B = 256 .. 1024 rows against 8192 contexts are 32 .. 128 of them).  Each wave owns one
// statistics strip of 64 columns per row: it publishes (strip max, sum exp(S - strip max)) at part[m][n / 64] and the numerators
// exp(S - strip max) * 2^14 as fp16 into the buffer that will hold G -- the format g8_lse_p2g_kernel rescales in place.  The numerators
// leave through the tile memory (free after the K loop): a lane holds single columns of 16 rows, the store wants 16-byte runs of a row.
struct EpiSimP : EpiSim {
  uint16_t* P = nullptr;  // [M][N] fp16 bit patterns
  int npart = 0;          // strips per row: cdiv(N, 64)

  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ void finish(f32x4 (&acc)[TM][TN], const TileCtx& c, const State<TM, TN>& st) const {
    static_assert(TN * 16 == 64, "one 64-column statistics strip per wave");
    constexpr int WR = TM * 16, WC = TN * 16, LDP = WC + 8;  // (row pitch 144 bytes: 16-byte aligned, the four row groups of a write on distinct banks)
    const int i = c.lane & 15, g = c.lane >> 4;
    uint16_t* stage = reinterpret_cast<uint16_t*>(c.scratch) + (c.wm * WN + c.wn) * WR * LDP;
    const int mw = c.m0 + c.wm * WR, nw = c.n0 + c.wn * WC;
    // After the row reductions all 16 lanes of a row group hold the row's (max, sum): lane i keeps those of row (a, r) = (i >> 2, i & 3)
    // of its group, and the strip statistics leave in ONE store per array and lane instead of 32 four-lane stores per wave.
    float keep_m = -INFINITY, keep_s = 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = a * 16 + g * 4 + r;
        const int m = mw + lrow;
        float v[TN];
        float mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          v[b] = st.masked[b] ? -INFINITY : acc[a][b][r] * inv_T;
          mx = fmaxf(mx, v[b]);
        }
        mx = row16_max(mx);
        const float mref = mx == -INFINITY ? 0.f : mx;
        float sm = 0.f;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const float e = __expf(v[b] - mref);  // exp(-inf) == 0 at masked columns
          sm += e;
          stage[lrow * LDP + b * 16 + i] = __builtin_bit_cast(uint16_t, (_Float16)(e * 16384.f));
        }
        sm = row16_sum(sm);
        if (i == a * 4 + r) {
          keep_m = mx;
          keep_s = sm;
        }
        // the gold logit: only a row whose label falls into this wave's 64 columns looks at them
        const int rel = st.yi[a][r] - nw - i;  // == b * 16 for the lane that holds the gold column
        if ((rel & ~48) == 0 && m < M) {
#pragma unroll
          for (int b = 0; b < TN; ++b)
            if (rel == b * 16) gold[m] = v[b];
        }
      }
    }
    {
      const int m = mw + (i >> 2) * 16 + g * 4 + (i & 3);
      if (TM * 4 > i && m < M && nw < N) {
        part_m[(size_t)m * npart + (nw >> 6)] = keep_m;
        part_s[(size_t)m * npart + (nw >> 6)] = keep_s;
      }
    }
    // (wave-private staging: program order within the wave is all the synchronisation it needs)
#pragma unroll
    for (int it = 0; it < WR / 8; ++it) {
      const int row = it * 8 + (c.lane >> 3), c8 = c.lane & 7;
      const uint4 w = *reinterpret_cast<const uint4*>(stage + row * LDP + c8 * 8);
      const int m = mw + row, n = nw + c8 * 8;
      if (m < M && n < N) *reinterpret_cast<uint4*>(P + (size_t)m * N + n) = w;
    }
  }
};

// Retrieval epilogue (run_retrieval_pytorch.py:149-150 without the score matrix): a score only leaves the tile when it
// ranks ahead of the row's current k-th best (score desc, passage id asc); such scores are appended to the row's
// candidate list (value + local column), which the top-k merge kernel folds into the state.  Once a few chunks have
// been seen almost nothing qualifies, so the GEMM writes next to nothing.
struct EpiFilter {
  const float* kth_val;     // state values  [M][k]
  const int64_t* kth_idx;   // state ids     [M][k]  (-1: slot unfilled)
  int k;
  int M, N;
  long long col_offset;     // passage id of column 0
  int* cnt;                 // [M] candidates appended so far (the merge kernel resets it)
  float* cand_v;            // [M][N]
  int* cand_j;              // [M][N]

  template <int TM, int TN>
  struct Raw {
    float tv[TM][4];  // the id of the k-th best is only needed on an exact score tie: fetched there
  };
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ Raw<TM, TN> begin(const TileCtx& c) const {
    const int g = c.lane >> 4;
    Raw<TM, TN> st;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = min(c.m0 + c.wm * (BM / WM) + a * 16 + g * 4 + r, M - 1);
        st.tv[a][r] = kth_val[(size_t)m * k + k - 1];
      }
    return st;
  }
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ Raw<TM, TN> settle(const TileCtx&, const Raw<TM, TN>& raw) const {
    return raw;
  }
  struct BigRegs { float tv; };
  __device__ __forceinline__ BigRegs big_load(int m0, int n0, int t) const {
    return BigRegs{kth_val[(size_t)min(m0 + t, M - 1) * k + k - 1]};
  }
  __device__ __forceinline__ void big_store(const BigRegs& r, int n0, int t, int* meta, int BM) const {
    meta[t] = __float_as_int(r.tv);
  }
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ Raw<TM, TN> big_state(const TileCtx& c, const int* meta) const {
    const int g = c.lane >> 4;
    Raw<TM, TN> st;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) st.tv[a][r] = __int_as_float(meta[c.wm * (BM / WM) + a * 16 + g * 4 + r]);
    return st;
  }

  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ void finish(f32x4 (&acc)[TM][TN], const TileCtx& c, const Raw<TM, TN>& st) const {
    const int i = c.lane & 15, g = c.lane >> 4;
    // Branch-free screen first: once the thresholds have risen almost no tile holds a score that reaches its row's
    // k-th best, and 2 x TM*TN*4 divergent branches per lane would cost more than the K loop of a d=768 tile.
    bool any = false;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < TN; ++b) any |= acc[a][b][r] >= st.tv[a][r];
    if (!any) return;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = c.m0 + c.wm * (BM / WM) + a * 16 + g * 4 + r;
        if (m >= M) continue;
        const float tv = st.tv[a][r];
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int n = c.n0 + c.wn * (BN / WN) + b * 16 + i;
          const float v = acc[a][b][r];
          if (!(n < N && v >= tv)) continue;
          bool take = v > tv;
          if (!take) {  // exact tie with the k-th best -> lower passage id wins (-1: slot unfilled)
            const long long ti = kth_idx[(size_t)m * k + k - 1];
            take = ti < 0 || col_offset + n < ti;
          }
          if (take) {
            const int pos = atomicAdd(&cnt[m], 1);
            cand_v[(size_t)m * N + pos] = v;
            cand_j[(size_t)m * N + pos] = n;
          }
        }
      }
  }
};

// fp32 store with a scale that may live on the device (autograd grad_output); bz selects a split-K slab
struct EpiScaleF32 {
  float* out;  // [splits][M][N]
  int M, N;
  float h_scale;
  const float* d_scale;
  // Loss piggy-back for the multi-rank step: element [m][0] of every row m with m % stamp_period == stamp_row (the
  // first mask row of each rank's chunk: its gradient is dead weight) is replaced by *stamp_src, this rank's loss
  // numerator, so that the reduce-scatter of dC also delivers the sum of the losses and no all-reduce is needed.
  const float* stamp_src = nullptr;
  int stamp_period = 1, stamp_row = -1;
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ float begin(const TileCtx&) const {
    return d_scale ? *d_scale : 1.0f;  // raw prefetch (see EpiSim::State)
  }
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ float settle(const TileCtx&, float ds) const {
    return h_scale * ds;
  }
  // Through LDS, one 16-row block of the wave's tile at a time (the tile memory is free by now; each wave has its own 16 x (WN_COLS + 4)
  // fp32 patch, so no barrier is needed): the MFMA layout gives a lane one column of four rows -- stored as it is, that is a 4-byte
  // store per value in 64-byte runs (6x the time per byte of a 16-byte store, MI355X_MICROARCH.md; the router-width backward, 141 MB of
  // fp32 gradients, spent 67 us there) -- the patch is read back as float4: 16 lanes cover 256 contiguous bytes of one row.
  template <int BM, int BN, int WM, int WN, int TM, int TN>
  __device__ __forceinline__ void finish(f32x4 (&acc)[TM][TN], const TileCtx& c, float s) const {
    constexpr int WC = TN * 16, TS = WC + 4;  // columns of the wave's tile, patch row stride
    const int i = c.lane & 15, g = c.lane >> 4;
    float* o = out + (size_t)c.bz * M * N;
    float* const T = c.scratch + (c.wm * WN + c.wn) * (16 * TS);
    const int nw0 = c.n0 + c.wn * (BN / WN);
    const float stamp = stamp_src != nullptr ? *stamp_src : 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(g * 4 + r) * TS + b * 16 + i] = acc[a][b][r] * s;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      const int mb = c.m0 + c.wm * (BM / WM) + a * 16;
#pragma unroll
      for (int it = 0; it < TN; ++it) {  // 16 rows x WC / 4 float4 = 64 lanes x TN
        const int e = c.lane + it * 64, row = e / (WC / 4), cq = e % (WC / 4);
        const int m = mb + row, n = nw0 + cq * 4;
        float4 v = *reinterpret_cast<const float4*>(T + row * TS + cq * 4);
        if (m < M && n < N) {
          if (n == 0 && stamp_src != nullptr && m % stamp_period == stamp_row) v.x = stamp;
          if (n + 3 < N) {
            *reinterpret_cast<float4*>(o + (size_t)m * N + n) = v;
          } else {  // ragged N (not a multiple of 4): never for dQ / dC (N = d, a multiple of 8)
            o[(size_t)m * N + n] = v.x;
            if (n + 1 < N) o[(size_t)m * N + n + 1] = v.y;
            if (n + 2 < N) o[(size_t)m * N + n + 2] = v.z;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the patch is rewritten by the next block only after these reads
    }
  }
};

}  // namespace dprhot
