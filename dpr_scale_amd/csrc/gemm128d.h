// gemm128d.h -- the 128 x 128 x 64 tile of gemm_bf16.h with its operands staged by LDS-DMA (round 6).
//
// Why: the SQ counters of the 128 x 128 engine (profiles/r06_pmc_engine128.txt: dprhot_dq / dprhot_dc at 8192^2 x 768 and 4096 x 65536
// x 768) say it is ISSUE-bound, not bandwidth-bound: 57-60 M VALU instructions per launch against 13-18 M for the library's
// MT192x128x64 kernel at the same flops, 16-20 M LDS instructions against 9-10 M, MFMA duty 0.31-0.34 -- the global -> VGPR ->
// ds_write staging (addresses, four 16-byte loads and four 16-byte LDS stores per thread and K step, the registers that keep two
// K steps in flight) is what the waves spend their issue slots on.  global_load_lds moves a 1-KiB piece per wave instruction straight
// into LDS: no staging registers, no ds_write pass, one address add per piece.
//
// Same tile, same wave grid (2 x 2 waves of 64 x 64), same LDS images as gemm_bf16.h -- k-major [128 rows][64 k] with the 16-byte chunk
// c of row r at c ^ ((r >> 1) & 7), mn-major [64 k][128 mn] with the 32-byte column group cg of row k at cg ^ mswz(k) -- so load_frag's
// address arithmetic and every epilogue of gemm_bf16.h (begin / settle / finish on f32x4 acc[4][4]) are reused unchanged.  The DMA
// writes lane-linearly, so the swizzle is applied on the SOURCE address (guide 5.4 rule 21).  Two LDS buffers (64 KiB: two workgroups
// per CU), per K step:   wait for my pieces of step t (vmcnt(0)) | barrier | DMA step t + 1 into the other buffer | fragments + MFMAs
// of step t.  The barrier also tells that every wave is done reading the other buffer (step t - 1).  Fragment reads are inline asm:
// hipcc would answer a plain LDS read next to an LDS-DMA in flight with s_waitcnt vmcnt(0) and drain the step it was meant to overlap.
//
// Restrictions (the launcher checks them and falls back to gemm_bf16_kernel): bf16 operands, every K range a whole number of 64-deep
// steps (a DMA cannot zero the tail of a step; the one exception, dQ's context axis, is in g1_tile), operands addressable with 32-bit
// element offsets.
#pragma once
#include "gemm256.h"

namespace dprhot {

constexpr int G1_B = 128, G1_BK = 64, G1_IMG = G1_B * G1_BK;  // elements of one operand image (16 KiB)
constexpr size_t g1_lds_bytes = (size_t)4 * G1_IMG * 2;       // 2 buffers x {A, B}

__device__ const uint4 g1_zero16 = {0u, 0u, 0u, 0u};  // what the rows of a partial K step beyond K are read from

__device__ __forceinline__ unsigned g1_lds_addr(const void* p) {
  typedef __attribute__((address_space(3))) int lds_int;
  return (unsigned)(uintptr_t)(lds_int*)p;
}

// fragment of 16 rows r0.. and the 32-deep k slice kk of an image (gemm_bf16.h load_frag, as asm; settled by the caller's lgkmcnt(0))
template <bool KM>
__device__ __forceinline__ void g1_frag(bf16x8& out, bf16x4& lo, bf16x4& hi, const uint16_t* T, int r0, int kk, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if constexpr (KM) {
    const int row = r0 + i;
    const unsigned addr = g1_lds_addr(T + row * G1_BK + (((kk * 4 + g) ^ ((row >> 1) & 7)) << 3));
    asm volatile("ds_read_b128 %0, %1" : "=&v"(out) : "v"(addr));
  } else {
    const int k = kk * 32 + g * 8 + (i >> 2);
    const unsigned addr = g1_lds_addr(T + k * G1_B + (((r0 >> 4) ^ mswz(k)) << 4) + (i & 3) * 4);
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024" : "=&v"(lo), "=&v"(hi) : "v"(addr));  // rows k, k + 4
  }
}

// one tile (bx, by) of K slice bz; nbx = column tiles per row block (what the statistics epilogues index their per-tile arrays with)
template <bool A_KM, bool B_KM, class Epi>
__device__ __forceinline__ void g1_tile(const GemmArgs& p, const Epi& epi, int bx, int by, int bz, int nbx, uint16_t* smem) {
  constexpr int TM = 4, TN = 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = by * G1_B, n0 = bx * G1_B;
  const int kbeg = bz * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
  // Whole 64-deep steps (checked by the launcher) -- except with B mn-major (its K index is a row), where the last step of the last slice
  // may be partial: the missing rows of B are read from 16 zero bytes, the missing chunks (k-major A: K a multiple of 8) or rows (mn-major
  // A) of A from the step before -- any finite values: they meet zeros.  A DMA cannot zero-fill, but it can be pointed at zeros.
  constexpr bool KTAIL = !B_KM;  // (B mn-major: dQ = G x C over the contexts, dC = G^T x Q over the query rows)
  const int tail = KTAIL ? ((kend - kbeg) & (G1_BK - 1)) : 0;
  const int nt = (kend - kbeg + (KTAIL ? G1_BK - 1 : 0)) / G1_BK;

  // per-lane source offsets (elements) of this wave's four pieces per operand image; piece j of wave w is LDS bytes [(w * 4 + j) KiB, +1 KiB)
  unsigned oa[4], ob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pc = wave * 4 + j;
    if constexpr (A_KM) {  // 8 rows x 8 chunks: position (lane & 7) of row r holds chunk (lane & 7) ^ ((r >> 1) & 7)
      const int r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      oa[j] = (unsigned)min(m0 + r, p.M - 1) * (unsigned)p.lda + (unsigned)(c * 8);
    } else {  // 4 k rows x 16 half-groups: slot (pos >> 1) of row k holds group (pos >> 1) ^ mswz(k), half pos & 1
      const int k = pc * 4 + (lane >> 4), pos = lane & 15, mc = (((pos >> 1) ^ mswz(k)) << 1) | (pos & 1);
      oa[j] = (unsigned)k * (unsigned)p.lda + (unsigned)min(m0 + mc * 8, p.M - 8);
    }
    if constexpr (B_KM) {
      const int r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
      ob[j] = (unsigned)min(n0 + r, p.N - 1) * (unsigned)p.ldb + (unsigned)(c * 8);
    } else {
      const int k = pc * 4 + (lane >> 4), pos = lane & 15, mc = (((pos >> 1) ^ mswz(k)) << 1) | (pos & 1);
      ob[j] = (unsigned)k * (unsigned)p.ldb + (unsigned)min(n0 + mc * 8, p.N - 8);
    }
  }
  const size_t astep = A_KM ? (size_t)G1_BK : (size_t)G1_BK * p.lda;  // elements per K step
  const size_t bstep = B_KM ? (size_t)G1_BK : (size_t)G1_BK * p.ldb;
  const uint16_t* Ag = p.A + (A_KM ? (size_t)kbeg : (size_t)kbeg * p.lda);
  const uint16_t* Bg = p.B + (B_KM ? (size_t)kbeg : (size_t)kbeg * p.ldb);
  auto stage = [&](int t, int buf) {
    uint16_t* As = smem + buf * 2 * G1_IMG;
    uint16_t* Bs = As + G1_IMG;
    const uint16_t* a = Ag + (size_t)t * astep;
    const uint16_t* b = Bg + (size_t)t * bstep;
    if constexpr (KTAIL) {
      if (tail != 0 && t == nt - 1) {  // the partial step (at most once per launch and workgroup)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pc = wave * 4 + j;
          const uint16_t* src = a + oa[j];
          if constexpr (A_KM) {
            const int r = pc * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            if (c * 8 >= tail) src -= G1_BK;
          } else {
            if (pc * 4 + (lane >> 4) >= tail) src -= astep;
          }
          __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)src, (g2_lds_ptr*)(As + (wave * 4 + j) * 512), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = (wave * 4 + j) * 4 + (lane >> 4);
          const uint16_t* src = k < tail ? b + ob[j] : reinterpret_cast<const uint16_t*>(&g1_zero16);
          __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)src, (g2_lds_ptr*)(Bs + (wave * 4 + j) * 512), 16, 0, 0);
        }
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(a + oa[j]), (g2_lds_ptr*)(As + (wave * 4 + j) * 512), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(b + ob[j]), (g2_lds_ptr*)(Bs + (wave * 4 + j) * 512), 16, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the epilogue's own global reads first (consumed by settle() behind the first wait below), then step 0
  const TileCtx ctx{m0, n0, wm, wn, lane, tid, bx, nbx, bz, reinterpret_cast<float*>(smem)};
  const auto eraw = epi.template begin<G1_B, G1_B, 2, 2, TM, TN>(ctx);
  if (nt > 0) stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const auto est = epi.template settle<G1_B, G1_B, 2, 2, TM, TN>(ctx, eraw);

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of step t
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // everybody's pieces of step t; everybody is done reading the other buffer (step t - 1)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nt) stage(t + 1, buf ^ 1);
    const uint16_t* Ac = smem + buf * 2 * G1_IMG;
    const uint16_t* Bc = Ac + G1_IMG;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[TM], bfr[TN];
      bf16x4 alo[TM], ahi[TM], blo[TN], bhi[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) g1_frag<A_KM>(af[a], alo[a], ahi[a], Ac, wm * 64 + a * 16, kk, lane);
#pragma unroll
      for (int b = 0; b < TN; ++b) g1_frag<B_KM>(bfr[b], blo[b], bhi[b], Bc, wn * 64 + b * 16, kk, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);  // register-only MFMAs may not be hoisted above the wait (guide 5.4 rule 18)
      if constexpr (!A_KM) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          af[a][0] = alo[a][0]; af[a][1] = alo[a][1]; af[a][2] = alo[a][2]; af[a][3] = alo[a][3];
          af[a][4] = ahi[a][0]; af[a][5] = ahi[a][1]; af[a][6] = ahi[a][2]; af[a][7] = ahi[a][3];
        }
      }
      if constexpr (!B_KM) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          bfr[b][0] = blo[b][0]; bfr[b][1] = blo[b][1]; bfr[b][2] = blo[b][2]; bfr[b][3] = blo[b][3];
          bfr[b][4] = bhi[b][0]; bfr[b][5] = bhi[b][1]; bfr[b][6] = bhi[b][2]; bfr[b][7] = bhi[b][3];
        }
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the epilogue reuses the tile memory as scratch
  epi.template finish<G1_B, G1_B, 2, 2, TM, TN>(acc, ctx, est);
}

// Tile order.  Workgroup L (x fastest) runs on XCD L % 8; the nx column tiles of one row block (by, bz) read the SAME A block, and
// in the plain order they are neighbours in L, i.e. spread over all eight XCDs at the same time: every XCD's L2 fetches that block
// for itself (dC = G^T Q at 1024 x 65536 x 768: the 134 MB of G went through the fabric six times, 805 MB for a 120 us GEMM).
// Remapped so that XCD x takes row blocks x, x + 8, ... and walks their nx column tiles one after the other: the block is fetched
// once per XCD that needs it.  (Whole groups of eight row blocks only; the remainder keeps the plain order.)  L: linear index inside a
// grid of nx x ngr tiles -- a constant offset of L (the pair launch's second problem) only renames the XCDs.
__device__ __forceinline__ void g1_remap(int L, int nx, int ngr, int& bx, int& g) {
  const int whole = (ngr >> 3) << 3;
  if (L < nx * whole) {
    const int xcd = L & 7, slot = L >> 3;
    g = xcd + ((slot / nx) << 3);
    bx = slot % nx;
  } else {
    g = L / nx;
    bx = L - g * nx;
  }
}

template <bool A_KM, bool B_KM, class Epi, bool REMAP = true>
__global__ __launch_bounds__(256, 2) void gemm128d_kernel(GemmArgs p, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  // (REMAP = false, the similarity GEMM of a few hundred query rows against many contexts: the plain order already puts the column
  //  tile bx of every row block on XCD bx % 8 -- each block of C is fetched by one XCD, the small Q by all of them)
  // (Four LDS buffers with three K steps in flight instead of two, for grids of at most one workgroup per CU, were measured in round 6
  //  and changed nothing -- 256 x 8192 x 768 forward 22.9 / 23.0 us: a lone workgroup's K step is bound by its own fragment reads and
  //  MFMAs taking turns, not by the load latency; scratch/negative/README.md.)
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if constexpr (REMAP) {
    int g;
    g1_remap((bz * (int)gridDim.y + by) * (int)gridDim.x + bx, gridDim.x, gridDim.y * gridDim.z, bx, g);
    by = g % (int)gridDim.y;
    bz = g / (int)gridDim.y;
  }
  g1_tile<A_KM, B_KM>(p, epi, bx, by, bz, (int)gridDim.x, smem);
}

// The backward pair in ONE launch on this tile: [dQ tiles x K slices (problem 2: A k-major, B mn-major) | dC tiles (problem 1: A and B
// mn-major, one K range)] -- what gemm_pair_kernel is for the register-staged tiles.  Both halves in the XCD-aware order.
template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm128d_pair_kernel(GemmArgs p1, Epi e1, int nbx1, int nby1, GemmArgs p2, Epi e2, int nbx2, int nby2,
                                                               int splits2) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  // the dQ units FIRST: where they are the longer ones (few query rows against a long context axis: K slices several times K = B) they
  // must not be what the launch ends on
  const int L = blockIdx.x;
  const int n2 = nbx2 * nby2 * splits2;
  int bx, g;
  if (L < n2) {
    g1_remap(L, nbx2, nby2 * splits2, bx, g);
    g1_tile<true, false>(p2, e2, bx, g % nby2, g / nby2, nbx2, smem);
  } else {
    g1_remap(L - n2, nbx1, nby1, bx, g);
    g1_tile<false, false>(p1, e1, bx, g, 0, nbx1, smem);
  }
}

}  // namespace dprhot
