// gemm8pb.h -- the backward GEMMs at MFMA-bound sizes on the phase-interleaved schedule of gemm8p.h (autograd of
// dpr_task.py:98-105): dC_part = G^T x Q reads BOTH operands "mn-major" (the contraction index is the ROW of the matrix in HBM),
// dQ = G x C its B operand.  One workgroup per unit (a 256 x 256 tile over one K range of `kchunk`): units are long (B or a slice
// of Nc deep, dozens of K steps), so the pipeline is filled once per unit and nothing is persistent here.
//
// Schedule, half-tiles, register sets, staggered wave groups, vmcnt(10): exactly gemm8p.h (see its header).  What differs is the
// image of an mn-major half-tile and its fragment read:
//   image [64 k][128 mn] bf16 = 256-byte rows (one LDS bank row each); the 32-byte column group cg (16 mn) of row k sits at slot
//   cg ^ ((k & 3) << 1).  A v_mfma_f32_32x32x16_bf16 operand (32 mn x 16 k) is two ds_read_b64_tr_b16 per lane (k .. k+3 and
//   k+4 .. k+7 of its 8 k values, same swizzle); the 32 lanes served together read rows k0 .. k0+3 of TWO adjacent groups
//   (cg0, cg0 + 1): bit 0 of the slot tells the groups apart, bits 1-2 the rows -- 8 distinct slots, no bank conflict.
//   The DMA writes lane-linearly (one instruction = 4 k rows), so lane l at slot (l & 15) >> 1 of row k fetches the group that
//   belongs there: the permutation is applied on the source address, as for k-major images.
// A-half h holds the tile rows {wm*128 + h*64 + j}: in HBM two 128-byte pieces per k row; B-half h the columns {wn*64 + h*32 + j}:
// four 64-byte pieces per k row.
#pragma once
#include "gemm8p.h"

namespace dprhot {

// fragment of an mn-major image: rows mn0 .. mn0+31 (mn0 % 32 == 0), k slice kk (16 deep).  Inline asm: hipcc would park
// s_waitcnt vmcnt(0) in front of the builtin form while LDS-DMA is in flight; settled by the lgkmcnt(0) that opens the MFMA section.
__device__ __forceinline__ bf16x8 g8_trfrag32(const uint16_t* img, int mn0, int kk, int lane) {
  const int s = lane & 15, blk = (lane >> 4) & 1, hh = lane >> 5;
  const int k = kk * 16 + hh * 8 + (s >> 2);
  const int cg = (mn0 >> 4) + blk;
  const unsigned addr = g8_lds_addr(img + k * 128 + ((cg ^ ((k & 3) << 1)) << 4) + (s & 3) * 4);
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// fp32 store of the transposed accumulators with a scale that may live on the device (autograd grad_output); bz selects a
// split-K slab.  stamp_*: the loss piggy-back of the multi-rank step (EpiScaleF32).
struct Epi8Scale {
  float* out;  // [splits][M][N]
  int M, N;
  float h_scale;
  const float* d_scale;
  const float* stamp_src = nullptr;
  int stamp_period = 1, stamp_row = -1;
  __device__ __forceinline__ void finish(G8Acc& acc, int m0, int n0, int bz, int wm, int wn, int lane) const {
    const float s = h_scale * (d_scale != nullptr ? *d_scale : 1.0f);
    const int i = lane & 31, h = lane >> 5;
    float* o = out + (size_t)bz * M * N;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int m = m0 + wm * 128 + a * 32 + i;
      if (m >= M) continue;
      const bool stamp = stamp_src != nullptr && m % stamp_period == stamp_row;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + b * 32 + q * 8 + h * 4;
          if (n >= N) continue;  // N % 4 == 0
          float4 v = make_float4(acc.v[a][b][q * 4 + 0] * s, acc.v[a][b][q * 4 + 1] * s, acc.v[a][b][q * 4 + 2] * s, acc.v[a][b][q * 4 + 3] * s);
          if (stamp && n == 0) v.x = *stamp_src;
          *reinterpret_cast<float4*>(o + (size_t)m * N + n) = v;
        }
    }
  }
};

// One unit: D[m0.., n0..] over the K range of slice bz.  A_KM: the A operand is k-major (dQ = G x C: A = G [B][Nc]); B is always
// mn-major.  K range length a multiple of 128 (an even number of K steps).
template <bool A_KM, class Epi>
__device__ __forceinline__ void g8b_unit(const GemmArgs& p, const Epi& epi, int bx, int by, int bz, uint16_t* smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = by * G2_B, n0 = bx * G2_B;
  const int kbeg = bz * p.kchunk;
  const int nt = (min(p.K, kbeg + p.kchunk) - kbeg) / G2_BK;

  // per-lane source offsets (bytes) of the wave's two DMA instructions per half-tile
  unsigned oa00, oa01, oa10, oa11, ob00, ob01, ob10, ob11;
  {
    const unsigned la = (unsigned)p.lda * 2u, lb = (unsigned)p.ldb * 2u;
    if constexpr (A_KM) {
      const int lr0 = (wave * 2 + 0) * 8 + (lane >> 3), lr1 = lr0 + 8;
      const int c0 = ((lane & 7) ^ ((lr0 >> 1) & 7)) * 16, c1 = ((lane & 7) ^ ((lr1 >> 1) & 7)) * 16;
      const int ar0 = m0 + (lr0 >> 6) * 128 + (lr0 & 63), ar1 = m0 + (lr1 >> 6) * 128 + (lr1 & 63);
      oa00 = (unsigned)min(ar0, p.M - 1) * la + c0;
      oa01 = (unsigned)min(ar1, p.M - 1) * la + c1;
      oa10 = (unsigned)min(ar0 + 64, p.M - 1) * la + c0;
      oa11 = (unsigned)min(ar1 + 64, p.M - 1) * la + c1;
    } else {
      // instruction j: k rows (wave*2 + j)*4 + (lane >> 4); slot (lane & 15) >> 1 of that row holds group slot ^ ((k & 3) << 1)
      const int k0 = (wave * 2 + 0) * 4 + (lane >> 4), k1 = k0 + 4, pos = lane & 15;
      const int cg0 = (pos >> 1) ^ ((k0 & 3) << 1), cg1 = (pos >> 1) ^ ((k1 & 3) << 1);
      const int e8 = (pos & 1) * 8;
#define G8B_ACOL(CG, H) (min(m0 + ((CG) >> 2) * 128 + (H) * 64 + ((CG) & 3) * 16 + e8, p.M - 8))
      oa00 = (unsigned)k0 * la + (unsigned)G8B_ACOL(cg0, 0) * 2u;
      oa01 = (unsigned)k1 * la + (unsigned)G8B_ACOL(cg1, 0) * 2u;
      oa10 = (unsigned)k0 * la + (unsigned)G8B_ACOL(cg0, 1) * 2u;
      oa11 = (unsigned)k1 * la + (unsigned)G8B_ACOL(cg1, 1) * 2u;
#undef G8B_ACOL
    }
    {
      const int k0 = (wave * 2 + 0) * 4 + (lane >> 4), k1 = k0 + 4, pos = lane & 15;
      const int cg0 = (pos >> 1) ^ ((k0 & 3) << 1), cg1 = (pos >> 1) ^ ((k1 & 3) << 1);
      const int e8 = (pos & 1) * 8;
#define G8B_BCOL(CG, H) (min(n0 + ((CG) >> 1) * 64 + (H) * 32 + ((CG) & 1) * 16 + e8, p.N - 8))
      ob00 = (unsigned)k0 * lb + (unsigned)G8B_BCOL(cg0, 0) * 2u;
      ob01 = (unsigned)k1 * lb + (unsigned)G8B_BCOL(cg1, 0) * 2u;
      ob10 = (unsigned)k0 * lb + (unsigned)G8B_BCOL(cg0, 1) * 2u;
      ob11 = (unsigned)k1 * lb + (unsigned)G8B_BCOL(cg1, 1) * 2u;
#undef G8B_BCOL
    }
  }
  auto img = [&](int par, int which) { return smem + (par * 4 + which) * G8_HALF; };
  // byte offset of K step TT of this unit in the operand (K steps beyond the unit's last wrap around: fetched, never read)
  // (a unit of ONE or TWO steps wraps twice: the fall-back is step 0 -- any address inside the unit's own K range will do)
  auto wrap = [&](int tt) { return tt < nt ? tt : (tt - nt < nt ? tt - nt : 0); };
  auto kofsA = [&](int tt) -> size_t {
    const size_t k = (size_t)(kbeg + wrap(tt) * G2_BK);
    return A_KM ? k * 2 : k * (size_t)p.lda * 2;
  };
  auto kofsB = [&](int tt) -> size_t { return (size_t)(kbeg + wrap(tt) * G2_BK) * (size_t)p.ldb * 2; };
#define G8B_STAGE(P, KOFS, O0, O1, IMG)                                                                                             \
  {                                                                                                                                 \
    const char* base_ = reinterpret_cast<const char*>(P) + (KOFS);                                                                  \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(base_ + (size_t)(O0)), (g2_lds_ptr*)((IMG) + (wave * 2 + 0) * 512), 16, 0, 0);   \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(base_ + (size_t)(O1)), (g2_lds_ptr*)((IMG) + (wave * 2 + 1) * 512), 16, 0, 0);   \
    g8_wait_vm<10>();                                                                                                               \
  }
#define G8B_RD_A(IMG)                                                                                                               \
  _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                              \
    if constexpr (A_KM) af[a_ * 4 + k_] = g8_frag32((IMG), wm * 64 + a_ * 32, k_, lane);                                            \
    else af[a_ * 4 + k_] = g8_trfrag32((IMG), wm * 64 + a_ * 32, k_, lane);                                                         \
  }
#define G8B_RD_B(DST, IMG) _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) DST[k_] = g8_trfrag32((IMG), wn * 32, k_, lane);
// (LIVE: false only in the second K step of the last pair when the unit has an ODD number of K steps -- round 6: column counts of the
//  packed multi-rank layout are multiples of 64, rarely of 128; the step's DMAs and fragment reads run as always, on wrapped-around
//  addresses, and only its MFMAs are skipped: a scalar branch per phase in that one step)
#define G8B_MM(AH, BH, BQ, LIVE)                                                                                                    \
  if (LIVE) {                                                                                                                       \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                               \
        acc.v[(AH) * 2 + a_][(BH)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BQ[k_], af[a_ * 4 + k_], acc.v[(AH) * 2 + a_][(BH)], 0, 0, 0); \
  }

  bf16x8 af[8], bq[2][4];
  G8Acc acc;
  acc.zero();

  // ---- prologue: the DMAs of phases -7 .. -1 in schedule order, the B0 read of phase -1 (gemm8p.h)
  G8B_STAGE(p.B, kofsB(0), ob00, ob01, img(0, 2));
  G8B_STAGE(p.A, kofsA(0), oa00, oa01, img(0, 0));
  G8B_STAGE(p.B, kofsB(0), ob10, ob11, img(0, 3));
  G8B_STAGE(p.A, kofsA(0), oa10, oa11, img(0, 1));
  G8B_STAGE(p.B, kofsB(1), ob00, ob01, img(1, 2));
  G8B_STAGE(p.A, kofsA(1), oa00, oa01, img(1, 0));  // ... vmcnt(10): B0(0) has landed
  g8_bar();
  g8_bar();
  G8B_RD_B(bq[0], img(0, 2));
  G8B_STAGE(p.B, kofsB(1), ob10, ob11, img(1, 3));  // ... A0(0) has landed
  g8_wait_lgkm0();
  g8_bar();
  g8_bar();

  if (wm == 1) g8_bar();  // this wave group runs one barrier behind from here on
  for (int t = 0; t < nt; t += 2) {
#define G8B_KSTEP(PAR, T, LIVE)                                                          \
  {                                                                                      \
    /* p0 */                                                                             \
    G8B_RD_A(img(PAR, 0));                                                               \
    G8B_STAGE(p.A, kofsA((T) + 1), oa10, oa11, img((PAR) ^ 1, 1));                       \
    g8_bar();                                                                            \
    g8_wait_lgkm0();                                                                     \
    G8B_MM(0, 0, bq[PAR], LIVE);                                                         \
    g8_bar();                                                                            \
    /* p1 */                                                                             \
    G8B_RD_B(bq[(PAR) ^ 1], img(PAR, 3));                                                \
    G8B_STAGE(p.B, kofsB((T) + 2), ob00, ob01, img(PAR, 2));                             \
    g8_bar();                                                                            \
    g8_wait_lgkm0();                                                                     \
    G8B_MM(0, 1, bq[(PAR) ^ 1], LIVE);                                                   \
    g8_bar();                                                                            \
    /* p2 */                                                                             \
    G8B_RD_A(img(PAR, 1));                                                               \
    G8B_STAGE(p.A, kofsA((T) + 2), oa00, oa01, img(PAR, 0));                             \
    g8_bar();                                                                            \
    g8_wait_lgkm0();                                                                     \
    G8B_MM(1, 1, bq[(PAR) ^ 1], LIVE);                                                   \
    g8_bar();                                                                            \
    /* p3 */                                                                             \
    G8B_RD_B(bq[(PAR) ^ 1], img((PAR) ^ 1, 2));                                          \
    G8B_STAGE(p.B, kofsB((T) + 2), ob10, ob11, img(PAR, 3));                             \
    g8_bar();                                                                            \
    g8_wait_lgkm0();                                                                     \
    G8B_MM(1, 0, bq[PAR], LIVE);                                                         \
    g8_bar();                                                                            \
  }
    G8B_KSTEP(0, t, true);
    G8B_KSTEP(1, t + 1, t + 1 < nt);
  }
  if (wm == 0) g8_bar();  // both wave groups in step again
  g8_wait_vm<0>();        // the wrap-around fetches: nothing may still be writing LDS when the workgroup ends
  epi.finish(acc, m0, n0, bz, wm, wn, lane);
}

// dC_part = G^T Q (both operands mn-major) and dQ = G C (A k-major, B mn-major, split over K) in ONE launch: neither fills the
// chip alone when d is a few hundred.  Unit u -> XCD-contiguous order, then [dC tiles: context block major, d block minor | dQ
// units: (K slice, query block) major, d block minor], so that the units an XCD runs together share their G block in its L2.
template <class Epi>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm8p_bwd_kernel(GemmArgs p1, Epi e1, int nbx1, int nby1, GemmArgs p2, Epi e2, int nbx2, int nby2,
                                                                   int splits2) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int n1 = nbx1 * nby1, n2 = nbx2 * nby2 * splits2, nwg = n1 + n2;
  const int wg = blockIdx.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
  int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  if (t < n1) {
    g8b_unit<false, Epi>(p1, e1, t % nbx1, t / nbx1, 0, smem);
  } else {
    t -= n1;
    const int bx = t % nbx2, rest = t / nbx2;
    g8b_unit<true, Epi>(p2, e2, bx, rest % nby2, rest / nby2, smem);
  }
}

}  // namespace dprhot
