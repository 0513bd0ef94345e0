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
// Geometry: 256 threads = 4 wave64 as WM x WN; block tile BM x BN, K step 64; v_mfma_f32_16x16x32_bf16.
// Staging is global -> VGPR -> LDS with the loads for tile t+1 issued before the MFMAs of tile t and the
// LDS writes after them (two LDS buffers, one barrier per K step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dprhot {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;        // K step (bf16 elements)
constexpr int KPAD = 8;       // k-major LDS row = 64 + 8 bf16 = 144 B (row-to-row bank rotation)
constexpr int MNPAD = 16;     // mn-major LDS row = R + 16 bf16

struct GemmArgs {
  const uint16_t* A;
  const uint16_t* B;
  int M, N, K;
  int lda, ldb;  // leading dimensions in elements
  int kchunk;    // K range per blockIdx.z (multiple of BK); == K when not split
};

template <int R, bool KMAJOR>
struct TileGeom {
  static constexpr int kRowStride = KMAJOR ? (BK + KPAD) : (R + MNPAD);  // elements
  static constexpr int kRows = KMAJOR ? R : BK;
  static constexpr int kElems = kRows * kRowStride;
  static constexpr int kChunks = R * BK / 8;  // 16-byte chunks in the tile
  static constexpr int kIters = (kChunks + 255) / 256;
};

template <int R, bool KMAJOR>
struct StageRegs {
  uint4 v[TileGeom<R, KMAJOR>::kIters];
};

// global -> registers: each thread fetches kIters 16-byte chunks of the tile (zero outside the matrix)
template <int R, bool KMAJOR>
__device__ __forceinline__ void stage_load(StageRegs<R, KMAJOR>& regs, const uint16_t* __restrict__ P, int ld, int r0,
                                           int rdim, int k0, int kend, int tid) {
  using G = TileGeom<R, KMAJOR>;
#pragma unroll
  for (int it = 0; it < G::kIters; ++it) {
    const int c = tid + it * 256;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (G::kChunks % 256 == 0 || c < G::kChunks) {
      if constexpr (KMAJOR) {
        const int row = c >> 3, kc = c & 7;
        const int gr = r0 + row, gk = k0 + kc * 8;
        if (gr < rdim && gk < kend) v = *reinterpret_cast<const uint4*>(P + (size_t)gr * ld + gk);
      } else {
        constexpr int CPR = R / 8;  // chunks per k-row
        const int krow = c / CPR, mc = c % CPR;
        const int gk = k0 + krow, gr = r0 + mc * 8;
        if (gk < kend && gr < rdim) v = *reinterpret_cast<const uint4*>(P + (size_t)gk * ld + gr);
      }
    }
    regs.v[it] = v;
  }
}

// registers -> LDS tile
template <int R, bool KMAJOR>
__device__ __forceinline__ void stage_store(const StageRegs<R, KMAJOR>& regs, uint16_t* T, int tid) {
  using G = TileGeom<R, KMAJOR>;
#pragma unroll
  for (int it = 0; it < G::kIters; ++it) {
    const int c = tid + it * 256;
    if (G::kChunks % 256 == 0 || c < G::kChunks) {
      if constexpr (KMAJOR) {
        const int row = c >> 3, kc = c & 7;
        *reinterpret_cast<uint4*>(T + row * G::kRowStride + kc * 8) = regs.v[it];
      } else {
        constexpr int CPR = R / 8;
        const int krow = c / CPR, mc = c % CPR;
        *reinterpret_cast<uint4*>(T + krow * G::kRowStride + mc * 8) = regs.v[it];
      }
    }
  }
}

// MFMA operand fragment for the 16 rows r0..r0+15 and the 32-wide k slice kk of an LDS tile.
// Lane l = (g = l >> 4, i = l & 15) receives row r0 + i, k = kk*32 + g*8 + {0..7}.
template <int R, bool KMAJOR, bool USE_TR>
__device__ __forceinline__ bf16x8 load_frag(const uint16_t* T, int r0, int kk, int lane) {
  using G = TileGeom<R, KMAJOR>;
  const int i = lane & 15, g = lane >> 4;
  if constexpr (KMAJOR) {
    return *reinterpret_cast<const bf16x8*>(T + (r0 + i) * G::kRowStride + kk * 32 + g * 8);
  } else if constexpr (USE_TR) {
    // ds_read_b64_tr_b16: within each 16-lane group, source lane s supplies 4 contiguous bf16 =
    // row (s >> 2), columns 4*(s & 3).. of a 4 x 16 block; result lane i receives column i of that block.
    const uint16_t* p = T + (kk * 32 + g * 8 + (i >> 2)) * G::kRowStride + r0 + (i & 3) * 4;
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p + 4 * G::kRowStride));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  } else {
    // plain 16-bit gathers (slow; kept as the cross-check of the transpose read)
    bf16x8 r;
    const uint16_t* p = T + (kk * 32 + g * 8) * G::kRowStride + r0 + i;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)p[j * G::kRowStride];
    return r;
  }
}

// Epilogue functors receive one accumulator fragment: rows m..m+3 (m % 4 == 0), one column n.
// They are responsible for bounds (m + r < M, n < N).

template <int BM, int BN, int WM, int WN, bool A_KMAJOR, bool B_KMAJOR, bool USE_TR, class Epi>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p, Epi epi) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  using GA = TileGeom<BM, A_KMAJOR>;
  using GB = TileGeom<BN, B_KMAJOR>;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  uint16_t* const As0 = smem;
  uint16_t* const Bs0 = smem + 2 * GA::kElems;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);
  const int nt = (kend - kbeg + BK - 1) / BK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  StageRegs<BM, A_KMAJOR> ra;
  StageRegs<BN, B_KMAJOR> rb;
  if (nt > 0) {
    stage_load<BM, A_KMAJOR>(ra, p.A, p.lda, m0, p.M, kbeg, kend, tid);
    stage_load<BN, B_KMAJOR>(rb, p.B, p.ldb, n0, p.N, kbeg, kend, tid);
    stage_store<BM, A_KMAJOR>(ra, As0, tid);
    stage_store<BN, B_KMAJOR>(rb, Bs0, tid);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1 < nt);
    const uint16_t* Ac = As0 + cur * GA::kElems;
    const uint16_t* Bc = Bs0 + cur * GB::kElems;
    if (more) {
      stage_load<BM, A_KMAJOR>(ra, p.A, p.lda, m0, p.M, kbeg + (t + 1) * BK, kend, tid);
      stage_load<BN, B_KMAJOR>(rb, p.B, p.ldb, n0, p.N, kbeg + (t + 1) * BK, kend, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) af[a] = load_frag<BM, A_KMAJOR, USE_TR>(Ac, wm * (BM / WM) + a * 16, kk, lane);
#pragma unroll
      for (int b = 0; b < TN; ++b) bfr[b] = load_frag<BN, B_KMAJOR, USE_TR>(Bc, wn * (BN / WN) + b * 16, kk, lane);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
    if (more) {
      stage_store<BM, A_KMAJOR>(ra, As0 + (cur ^ 1) * GA::kElems, tid);
      stage_store<BN, B_KMAJOR>(rb, Bs0 + (cur ^ 1) * GB::kElems, tid);
    }
    __syncthreads();
  }

  // C/D layout of v_mfma_f32_16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int m = m0 + wm * (BM / WM) + a * 16 + g * 4;
      const int n = n0 + wn * (BN / WN) + b * 16 + i;
      epi(acc[a][b], m, n, (int)blockIdx.z);
    }
}

template <int BM, int BN, bool A_KMAJOR, bool B_KMAJOR>
constexpr size_t gemm_lds_bytes() {
  return 2 * (size_t)(TileGeom<BM, A_KMAJOR>::kElems + TileGeom<BN, B_KMAJOR>::kElems) * sizeof(uint16_t);
}

// ---- epilogues ------------------------------------------------------------------------------------------

// sim_score epilogue: * inv_T, masked columns -> -inf (dpr_task.py:104,211)
struct EpiSim {
  float* S;
  const uint8_t* colmask;
  int M, N;
  float inv_T;
  __device__ __forceinline__ void operator()(const f32x4& v, int m, int n, int) const {
    if (n >= N) return;
    const bool masked = colmask != nullptr && colmask[n] != 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (m + r < M) S[(size_t)(m + r) * N + n] = masked ? -INFINITY : v[r] * inv_T;
  }
};

// fp32 store with a scale that may live on the device (autograd grad_output); z selects a split-K slab
struct EpiScaleF32 {
  float* out;  // [splits][M][N]
  int M, N;
  float h_scale;
  const float* d_scale;
  __device__ __forceinline__ void operator()(const f32x4& v, int m, int n, int z) const {
    if (n >= N) return;
    const float s = h_scale * (d_scale ? *d_scale : 1.0f);
    float* o = out + (size_t)z * M * N;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (m + r < M) o[(size_t)(m + r) * N + n] = v[r] * s;
  }
};

}  // namespace dprhot
