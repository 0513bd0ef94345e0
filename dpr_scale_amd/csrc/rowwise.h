// rowwise.h -- HBM-bound row kernels of the in-batch contrastive path (gfx950, wave64).
//   softmax_ce_kernel : nn.CrossEntropyLoss (dpr_task.py:46,212) + its backward into dScores, one launch
//   rank_kernel       : compute_rank_metrics (dpr_task.py:235-246) as a count, no sort
//   topk_stream_kernel: torch.topk of run_retrieval_pytorch.py:149-150 and its shard re-merge (:272-277), streaming
//   cast / reduce helpers
// All global accesses are 16-byte vectors, lanes consecutive (1 KiB per wave instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dprhot {

typedef float rw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rw_bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of two bf16 (RNE): v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  const rw_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rw_bf16x2));
}

template <int N>
__device__ __forceinline__ float dprhot_dpp_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
// reductions over the 16 lanes of a DPP row, all lanes receive the result (VALU only, no ds_bpermute round trips)
__device__ __forceinline__ float dprhot_row16_max(float v) {
  v = fmaxf(v, dprhot_dpp_ror<8>(v));
  v = fmaxf(v, dprhot_dpp_ror<4>(v));
  v = fmaxf(v, dprhot_dpp_ror<2>(v));
  return fmaxf(v, dprhot_dpp_ror<1>(v));
}
__device__ __forceinline__ float dprhot_row16_sum(float v) {
  v += dprhot_dpp_ror<8>(v);
  v += dprhot_dpp_ror<4>(v);
  v += dprhot_dpp_ror<2>(v);
  return v + dprhot_dpp_ror<1>(v);
}

// Whole-wave reductions, every lane receives the result.  VALU only: four DPP row rotations inside each row of 16 lanes, then
// v_permlane16_swap / v_permlane32_swap (with D = S = x they leave x[l] and x[l ^ 16] resp. x[l ^ 32] side by side in every lane)
// across the four rows.  (The __shfl_xor butterfly these replace lowers to six ds_bpermute round trips of ~120 cycles each.)
__device__ __forceinline__ float wave_max(float v) {
  v = dprhot_row16_max(v);
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dprhot_row16_sum(v);
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// (max, sum of exp relative to max) pairs combine associatively; -inf max carries sum 0.
__device__ __forceinline__ void ms_combine(float& m, float& s, float m2, float s2) {
  const float M = fmaxf(m, m2);
  if (M == -INFINITY) { m = M; s = 0.f; return; }
  s = s * __expf(m - M) + s2 * __expf(m2 - M);
  m = M;
}

// ----------------------------------------------------------------------------------------------------
// fp32 -> bf16
// ----------------------------------------------------------------------------------------------------
// One workgroup per contiguous tile of 256 x 2 groups of four values (16 bytes in, 8 out per thread and group): the access order
// that took the gradient-hook legs from 0.6-0.7 to 0.75-0.94 of the HBM rate (gradcomm.h) -- this kernel was a persistent grid-stride
// loop over groups of eight values, capped at 2048 workgroups.
constexpr int CAST_UT = 2;
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n4) {
  const size_t base = (size_t)blockIdx.x * (256 * CAST_UT) + threadIdx.x;
  float4 v[CAST_UT];
#pragma unroll
  for (int u = 0; u < CAST_UT; ++u)
    if (base + u * 256 < n4) v[u] = reinterpret_cast<const float4*>(src)[base + u * 256];
#pragma unroll
  for (int u = 0; u < CAST_UT; ++u)
    if (base + u * 256 < n4) reinterpret_cast<uint2*>(dst)[base + u * 256] = make_uint2(pk_bf16(v[u].x, v[u].y), pk_bf16(v[u].z, v[u].w));
}

// one launch for both producer-side casts of a step (query rows + this rank's context rows)
__global__ __launch_bounds__(256) void cast2_bf16_kernel(const float* __restrict__ a, uint16_t* __restrict__ ao, size_t a8,
                                                         const float* __restrict__ b, uint16_t* __restrict__ bo, size_t b8) {
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < a8 + b8; c += (size_t)gridDim.x * blockDim.x) {
    const float* src = c < a8 ? a : b;
    uint16_t* dst = c < a8 ? ao : bo;
    const size_t k = c < a8 ? c : c - a8;
    const float4 x = reinterpret_cast<const float4*>(src)[2 * k];
    const float4 y = reinterpret_cast<const float4*>(src)[2 * k + 1];
    uint4 o;
    o.x = pk_bf16(x.x, x.y);
    o.y = pk_bf16(x.z, x.w);
    o.z = pk_bf16(y.x, y.y);
    o.w = pk_bf16(y.z, y.w);
    reinterpret_cast<uint4*>(dst)[k] = o;
  }
}

// ----------------------------------------------------------------------------------------------------
// Multi-rank producer side: ONE buffer travels over xGMI per step.  This rank's context rows (fp32 -> bf16) are
// followed by a few extra rows whose bytes carry the dummy-context mask, so that a single all-gather moves both
// (the reference issues four, dpr_task.py:169-176).  After the gather the extra rows simply are additional,
// always-masked columns of C; unpack_mask_kernel builds the column mask of the gathered matrix.
//   send [rows_c, d] bf16 : rows [0, n_ctx) = contexts ; rows [n_ctx, rows_c) = mask bytes (then zeros)
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_ctx_kernel(const float* __restrict__ c, const uint8_t* __restrict__ mask, int n_ctx, int d,
                                                       int rows_c, uint16_t* __restrict__ send) {
  const size_t n8 = (size_t)rows_c * d / 8, ctx8 = (size_t)n_ctx * d / 8;
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n8; k += (size_t)gridDim.x * blockDim.x) {
    uint4 o;
    if (k < ctx8) {
      const float4 x = reinterpret_cast<const float4*>(c)[2 * k];
      const float4 y = reinterpret_cast<const float4*>(c)[2 * k + 1];
      o = make_uint4(pk_bf16(x.x, x.y), pk_bf16(x.z, x.w), pk_bf16(y.x, y.y), pk_bf16(y.z, y.w));
    } else {
      const size_t b0 = (k - ctx8) * 16;  // byte position inside the mask region
      uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const size_t b = b0 + e;
        const uint32_t v = (mask != nullptr && b < (size_t)n_ctx) ? (mask[b] != 0) : 0u;
        w[e >> 2] |= v << (8 * (e & 3));
      }
      o = make_uint4(w[0], w[1], w[2], w[3]);
    }
    reinterpret_cast<uint4*>(send)[k] = o;
  }
}

__global__ __launch_bounds__(256) void unpack_mask_kernel(const uint16_t* __restrict__ gathered, int W, int n_ctx, int d, int rows_c,
                                                          uint8_t* __restrict__ colmask) {
  const int total = W * rows_c;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < total; n += gridDim.x * blockDim.x) {
    const int r = n / rows_c, local = n - r * rows_c;
    uint8_t m = 1;  // the extra rows are not contexts: always masked
    if (local < n_ctx) m = reinterpret_cast<const uint8_t*>(gathered)[((size_t)r * rows_c + n_ctx) * d * 2 + local];
    colmask[n] = m;
  }
}

// ----------------------------------------------------------------------------------------------------
// Second (and last) kernel of the fused training forward.  The sim kernel has left, per row, one
// (max, sum-exp) pair per column tile and the gold logit; this kernel turns them into the row logsumexp
// (a handful of values per row), then makes ONE streaming pass over S: G = (exp(S - lse) - onehot) * scale
// in bf16.  Fully parallel over 8-column chunks -- no row-sized serial dependency.
// The loss numerator is accumulated in 2^-24 fixed point with integer atomics (order-independent, hence
// deterministic, and no fences needed: the data travels in the atomics themselves); the last workgroup
// to arrive converts it to the float the caller reads.
// Algorithmic HBM bytes: 4 (S read) + 2 (G write) per score.
// ----------------------------------------------------------------------------------------------------
struct GFinalArgs {
  const float* S;
  int B, Nc;
  const int64_t* y;
  int64_t y_offset;
  float grad_scale;
  const float* part_m;
  const float* part_s;
  int nt;  // partials per row
  const float* gold;
  float* row_loss;
  float* row_lse;
  uint16_t* G;
  unsigned long long* acc;  // [0] fixed-point loss sum, [1] arrival ticket (zeroed by the sim kernel)
  float* loss_sum;          // out: loss_scale * sum_i row_loss[i]
  float loss_scale = 1.0f;  // 1: the numerator; 1/Nq: the mean of dpr_task.py:212 leaves the kernel ready
};

constexpr double kLossFix = 1048576.0;   // 2^20: 48 bits of 2^-20 fixed point hold sums up to 2.7e8 at 1e-6 resolution
constexpr double kLossMaxBlock = 67108864.0;  // 2^26: a single workgroup's sum beyond this is published as +inf

// Loss numerator across workgroups: ONE 64-bit integer atomic carries the arrival ticket (bits 48..63) and the 2^-20
// fixed-point sum (bits 0..47): order-independent, hence deterministic, and a single round trip.  A NaN / inf / out-of-range
// workgroup sum cannot travel in fixed point: it sets sticky bits in acc[1] (bit 0 NaN, bit 1 +inf, bit 2 -inf) BEFORE its
// ticket, and the last arriver publishes NaN / inf exactly as nn.CrossEntropyLoss would (diverged embeddings, a masked gold
// column) instead of a finite-looking number.  acc[0..1] are zeroed by the sim launch.
__device__ __forceinline__ void loss_ticket_add(unsigned long long* acc, double tot, unsigned nblocks, float* loss_sum, float loss_scale) {
  long long fx = 0;
  const bool nan = !(tot == tot), big = fabs(tot) >= kLossMaxBlock;  // inf counts as big
  if (nan || big) {
    atomicOr(&acc[1], nan ? 1ull : (tot > 0 ? 2ull : 4ull));
    __threadfence();  // rare path: the flag must be visible before this workgroup's ticket
  } else {
    fx = __double2ll_rn(tot * kLossFix);
    if (fx < 0) fx = 0;  // row losses are >= 0 up to rounding
  }
  const unsigned long long old = atomicAdd(&acc[0], (1ull << 48) | (unsigned long long)fx);
  if ((old >> 48) == (unsigned long long)nblocks - 1) {
    const unsigned long long flags = atomicOr(&acc[1], 0ull);
    float out = (float)((double)((old & ((1ull << 48) - 1)) + (unsigned long long)fx) / kLossFix);
    if (flags & 1ull) out = NAN;
    else if ((flags & 6ull) == 6ull) out = NAN;  // +inf and -inf
    else if (flags & 2ull) out = INFINITY;
    else if (flags & 4ull) out = -INFINITY;
    loss_sum[0] = out * loss_scale;
  }
}
constexpr int kGfMaxPairs = 2048;         // (max, sum) pairs one workgroup stages in LDS

// rows per workgroup / x-blocks per row for a given row length and CPT chunks per thread (host and device agree)
__host__ __device__ inline void gfinal_geometry(int Nc, int cpt, int* rpb, int* xblocks) {
  const int cpr = Nc >> 3, per_block = 256 * cpt;
  if (cpr >= per_block) {
    *rpb = 1;
    *xblocks = (cpr + per_block - 1) / per_block;
  } else {
    const int r = per_block / cpr;
    *rpb = r > 256 ? 256 : r;
    *xblocks = 1;
  }
}
// "small" problems: workgroup (0,0) reduces the loss of ALL rows itself (no atomics, no second launch)
__host__ __device__ inline bool gfinal_small(int B, int nt) { return B <= 256 && (long)B * nt <= kGfMaxPairs; }

// CPT = 8-column chunks per thread: 1 for the latency-bound training shapes (a wave's instruction stream is what
// a microsecond-scale launch pays for, so the work is spread as thin as possible), 8 for bandwidth-bound sizes.
template <int CPT>
__global__ __launch_bounds__(256) void gfinal_kernel(GFinalArgs p) {
  const int cpr = p.Nc >> 3;  // 8-column chunks per row
  int rpb, xblocks;
  gfinal_geometry(p.Nc, CPT, &rpb, &xblocks);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.y * rpb;
  __shared__ float s_lse[256];
  __shared__ float s_gold[256];
  __shared__ int s_y[256];
  __shared__ float s_part[4];
  __shared__ float s_pm[kGfMaxPairs], s_ps[kGfMaxPairs];
  // This launch is latency-bound (a few dependent trips to memory), so every global read it needs is issued up
  // front: first the logits themselves (their addresses depend on nothing computed here), then the per-tile
  // statistics, gold logits and labels into LDS.
  const int c_lo = blockIdx.x * 256 * CPT;
  const int span = xblocks > 1 ? min(256 * CPT, cpr - c_lo) : cpr;  // chunks of each row this block covers
  const int rows_here = min(rpb, p.B - row0);
  const int total = rows_here * span;
  float4 va[CPT], vb[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int idx = tid + k * 256;
    if (idx < total && p.G != nullptr) {
      const int lr = idx / span;
      const float* src = p.S + (size_t)(row0 + lr) * p.Nc + (size_t)(c_lo + idx - lr * span) * 8;
      va[k] = *reinterpret_cast<const float4*>(src);
      vb[k] = *reinterpret_cast<const float4*>(src + 4);
    }
  }
  const bool small = gfinal_small(p.B, p.nt);
  const bool owner = small && blockIdx.x == 0 && blockIdx.y == 0;  // reduces every row's loss
  const int st_row0 = owner ? 0 : row0;                            // rows whose statistics this block stages
  const int st_rows = owner ? p.B : rows_here;
  const int npairs = st_rows * p.nt;
  for (int i = tid; i < npairs; i += 256) {
    s_pm[i] = p.part_m[(size_t)st_row0 * p.nt + i];
    s_ps[i] = p.part_s[(size_t)st_row0 * p.nt + i];
  }
  if (tid < st_rows) {
    s_gold[tid] = p.gold[st_row0 + tid];
    s_y[tid] = (int)(p.y[st_row0 + tid] + p.y_offset);
  }
  __syncthreads();
  // Row logsumexp out of LDS in two short phases (cross-lane shuffles are ds_bpermute round trips, ~120 cycles
  // each and serialised; plain LDS traffic is cheaper here): tpr threads per row fold nt/tpr pairs each, then one
  // thread per row folds the tpr partial pairs.
  if (p.nt <= 8) {  // a handful of pairs per row: one thread per row, no second phase
    if (tid < st_rows) {
      float m = -INFINITY, sm = 0.f;
      for (int t = 0; t < p.nt; ++t) ms_combine(m, sm, s_pm[tid * p.nt + t], s_ps[tid * p.nt + t]);
      s_lse[tid] = m + logf(sm);
    }
  } else {
    __shared__ float s2_m[256], s2_s[256];
    int tpr = 16;
    while (tpr > 1 && st_rows * tpr > 256) tpr >>= 1;
    const int lr = tid / tpr, sub = tid - lr * tpr;
    if (lr < st_rows) {
      float m = -INFINITY, sm = 0.f;
      for (int t = sub; t < p.nt; t += tpr) ms_combine(m, sm, s_pm[lr * p.nt + t], s_ps[lr * p.nt + t]);
      s2_m[tid] = m;
      s2_s[tid] = sm;
    }
    __syncthreads();
    if (tid < st_rows) {
      float m = s2_m[tid * tpr], sm = s2_s[tid * tpr];
      for (int k = 1; k < tpr; ++k) ms_combine(m, sm, s2_m[tid * tpr + k], s2_s[tid * tpr + k]);
      s_lse[tid] = m + logf(sm);
    }
  }
  __syncthreads();
  const int base = row0 - st_row0;  // index of this block's first row in the staged arrays
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int idx = tid + k * 256;
    if (idx < total) {
      const int lr = idx / span;
      const int chunk = c_lo + idx - lr * span;
      const int row = row0 + lr;
      const float lse = s_lse[base + lr];
      const int yi = s_y[base + lr];
      if (chunk == 0) {
        if (p.row_lse) p.row_lse[row] = lse;
        if (p.row_loss) p.row_loss[row] = lse - s_gold[base + lr];
      }
      if (p.G != nullptr) {
        const int j = chunk * 8;
        const float v[8] = {va[k].x, va[k].y, va[k].z, va[k].w, vb[k].x, vb[k].y, vb[k].z, vb[k].w};
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float pr = __expf(v[e] - lse);  // exp(-inf) == 0 at masked columns
          if (j + e == yi) pr -= 1.0f;
          g[e] = pr * p.grad_scale;
        }
        *reinterpret_cast<uint4*>(p.G + (size_t)row * p.Nc + j) =
            make_uint4(pk_bf16(g[0], g[1]), pk_bf16(g[2], g[3]), pk_bf16(g[4], g[5]), pk_bf16(g[6], g[7]));
      }
    }
  }
  // loss numerator.  small: the owner block sums every row (plain store).  Otherwise each row-block adds its rows
  // with ONE fixed-point integer atomic (order-independent => deterministic) and the last arriver publishes.
  if (small ? owner : (blockIdx.x == 0)) {
    const float mine = tid < st_rows ? s_lse[tid] - s_gold[tid] : 0.f;  // st_rows <= 256
    const float bl = wave_sum(mine);
    if (lane == 0) s_part[wave] = bl;
    __syncthreads();
    if (tid == 0) {
      const double tot = (double)s_part[0] + (double)s_part[1] + (double)s_part[2] + (double)s_part[3];
      if (small || gridDim.y == 1) {
        p.loss_sum[0] = (float)tot * p.loss_scale;
      } else {
        loss_ticket_add(p.acc, tot, gridDim.y, p.loss_sum, p.loss_scale);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// Short-row forward (Nc <= 4096, the BASELINE training shapes): the sim GEMM is split over K into `splits` slabs
// of partial logits (more, thinner workgroups: these launches are bound by per-CU load latency, not by bandwidth
// or MFMA) and THIS kernel does the whole row softmax with the row held in registers: sums the slabs, row
// max / sum (DPP over 16 lanes, then LDS across the row's lane groups), logsumexp, loss, dScores -- one pass.
// tpr (power of two, 16..1024) threads share a row; a 1024-thread workgroup holds 1024 / tpr rows.
// Loss numerator across workgroups: ONE 64-bit atomic carries both the arrival ticket (bits 48..63) and the
// 2^-24 fixed-point sum (bits 0..47): order-independent, hence deterministic, and a single round trip.
// ----------------------------------------------------------------------------------------------------
struct GShortArgs {
  const float* slabs;  // [splits][B][Nc] partial logits (mask and 1/T already applied: -inf at masked columns)
  int splits;
  size_t slab_stride;
  int B, Nc;
  const int64_t* y;
  int64_t y_offset;
  float grad_scale;
  float* S_out;  // optional [B][Nc]
  float* row_loss;
  float* row_lse;
  uint16_t* G;
  unsigned long long* acc;  // acc[0]: packed ticket|sum, zeroed by the sim kernel
  float* loss_sum;
  int tpr;
  float loss_scale = 1.0f;
};

constexpr unsigned long long kTicketOne = 1ull << 48;
constexpr unsigned long long kSumMask = kTicketOne - 1;

template <int CPT>
__global__ __launch_bounds__(1024) void gfinal_short_kernel(GShortArgs p) {
  const int tid = threadIdx.x, tpr = p.tpr, rpb = blockDim.x / tpr;
  const int lr = tid / tpr, j = tid - lr * tpr;
  const int row = blockIdx.x * rpb + lr;
  const int cpr = p.Nc >> 3;
  const bool active = row < p.B;
  __shared__ float s_red[64], s_red2[64], s_gold[64], s_rl[64];
  float v[CPT][8];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int chunk = j + k * tpr;
    if (active && chunk < cpr) {
      const float* src = p.slabs + (size_t)row * p.Nc + (size_t)chunk * 8;
      float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      int z = 1;
      for (; z + 7 < p.splits; z += 8) {  // many slabs (vocabulary-wide vectors: up to 32): eight slabs' loads in flight, added in slab order
        float4 a2[8], b2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a2[u] = *reinterpret_cast<const float4*>(src + (size_t)(z + u) * p.slab_stride);
          b2[u] = *reinterpret_cast<const float4*>(src + (size_t)(z + u) * p.slab_stride + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a.x += a2[u].x; a.y += a2[u].y; a.z += a2[u].z; a.w += a2[u].w;
          b.x += b2[u].x; b.y += b2[u].y; b.z += b2[u].z; b.w += b2[u].w;
        }
      }
      for (; z < p.splits; ++z) {
        const float4 a2 = *reinterpret_cast<const float4*>(src + (size_t)z * p.slab_stride);
        const float4 b2 = *reinterpret_cast<const float4*>(src + (size_t)z * p.slab_stride + 4);
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
        b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      }
      v[k][0] = a.x; v[k][1] = a.y; v[k][2] = a.z; v[k][3] = a.w;
      v[k][4] = b.x; v[k][5] = b.y; v[k][6] = b.z; v[k][7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = -INFINITY;
    }
  }
  const int yi = active ? (int)(p.y[row] + p.y_offset) : -1;
  // row max
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < CPT; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, v[k][e]);
  m = dprhot_row16_max(m);
  const int grp = tid >> 4, gpr = tpr >> 4;  // 16-lane groups; groups per row
  if (tpr > 16) {
    if ((tid & 15) == 0) s_red[grp] = m;
    __syncthreads();
    m = s_red[lr * gpr];
    for (int g2 = 1; g2 < gpr; ++g2) m = fmaxf(m, s_red[lr * gpr + g2]);
  }
  // row sum of exp, gold logit
  float sm = 0.f;
  if (m != -INFINITY) {
#pragma unroll
    for (int k = 0; k < CPT; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += __expf(v[k][e] - m);
  }
  sm = dprhot_row16_sum(sm);
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c0 = (j + k * tpr) * 8;
    if (yi >= c0 && yi < c0 + 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (yi == c0 + e) s_gold[lr] = v[k][e];
    }
  }
  if (tpr > 16) {
    if ((tid & 15) == 0) s_red2[grp] = sm;
    __syncthreads();
    sm = s_red2[lr * gpr];
    for (int g2 = 1; g2 < gpr; ++g2) sm += s_red2[lr * gpr + g2];
  } else {
    __syncthreads();  // s_gold
  }
  const float lse = m + logf(sm);
  if (active && j == 0) {
    const float l = lse - s_gold[lr];
    s_rl[lr] = l;
    if (p.row_lse) p.row_lse[row] = lse;
    if (p.row_loss) p.row_loss[row] = l;
  } else if (j == 0) {
    s_rl[lr] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int chunk = j + k * tpr;
    if (active && chunk < cpr) {
      const int c0 = chunk * 8;
      if (p.S_out != nullptr) {
        float* dst = p.S_out + (size_t)row * p.Nc + c0;
        *reinterpret_cast<float4*>(dst) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[k][4], v[k][5], v[k][6], v[k][7]);
      }
      if (p.G != nullptr) {
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float pr = __expf(v[k][e] - lse);
          if (c0 + e == yi) pr -= 1.0f;
          g[e] = pr * p.grad_scale;
        }
        *reinterpret_cast<uint4*>(p.G + (size_t)row * p.Nc + c0) =
            make_uint4(pk_bf16(g[0], g[1]), pk_bf16(g[2], g[3]), pk_bf16(g[4], g[5]), pk_bf16(g[6], g[7]));
      }
    }
  }
  __syncthreads();  // s_rl
  if (tid == 0) {
    double tot = 0.0;
    for (int r = 0; r < rpb; ++r) tot += (double)s_rl[r];
    if (gridDim.x == 1) {
      p.loss_sum[0] = (float)tot * p.loss_scale;
    } else {
      loss_ticket_add(p.acc, tot, gridDim.x, p.loss_sum, p.loss_scale);
    }
  }
}

// ----------------------------------------------------------------------------------------------------
// Row softmax cross-entropy + dScores.  TPR threads cooperate on one row (64 = one wave, 256 = the block);
// a block handles 256 / TPR rows.  Pass 1 streams the row once (online max/sum); pass 2 re-reads it (the
// row is L2-resident: <= 256 KiB) and emits G in bf16.  Algorithmic HBM bytes: 4 (S read) + 2 (G write)
// per score.
// ----------------------------------------------------------------------------------------------------
struct SoftmaxArgs {
  const float* S;
  int B, Nc;
  const int64_t* y;
  int64_t y_offset;
  float grad_scale;
  const int64_t* row_win_start;  // optional [B]
  int win_len;
  float* row_loss;
  float* row_lse;
  uint16_t* G;
};

template <int TPR>
__global__ __launch_bounds__(256) void softmax_ce_kernel(SoftmaxArgs p) {
  constexpr int RPB = 256 / TPR;
  const int tid = threadIdx.x;
  const int sub = tid / TPR, t = tid % TPR;
  const int row = blockIdx.x * RPB + sub;
  __shared__ float sm_m[4], sm_s[4];
  const bool active = row < p.B;
  const float* Srow = p.S + (size_t)(active ? row : 0) * p.Nc;
  int lo = 0, hi = p.Nc;
  if (p.row_win_start != nullptr && active) {
    lo = (int)(p.row_win_start[row] + p.y_offset);
    hi = lo + p.win_len;
  }
  const bool windowed = p.row_win_start != nullptr;

  float m = -INFINITY, s = 0.f;
  if (active) {
    for (int j = t * 4; j < p.Nc; j += TPR * 4) {
      float4 v = *reinterpret_cast<const float4*>(Srow + j);
      if (windowed) {
        if (j + 0 < lo || j + 0 >= hi) v.x = -INFINITY;
        if (j + 1 < lo || j + 1 >= hi) v.y = -INFINITY;
        if (j + 2 < lo || j + 2 >= hi) v.z = -INFINITY;
        if (j + 3 < lo || j + 3 >= hi) v.w = -INFINITY;
      }
      const float lm = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      if (lm > m) {  // rescale only when the running max moves
        s *= __expf(m - lm);  // m == -inf -> s == 0 * 0
        m = lm;
      }
      if (m != -INFINITY) s += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
    }
  }
  // combine across the TPR threads of the row
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    ms_combine(m, s, m2, s2);
  }
  if constexpr (TPR == 256) {
    const int w = tid >> 6;
    if ((tid & 63) == 0) { sm_m[w] = m; sm_s[w] = s; }
    __syncthreads();
    m = sm_m[0]; s = sm_s[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) ms_combine(m, s, sm_m[k], sm_s[k]);
  }
  if (!active) return;
  const float lse = m + logf(s);
  const int64_t yi = p.y[row] + p.y_offset;
  if (t == 0) {
    if (p.row_lse) p.row_lse[row] = lse;
    if (p.row_loss) p.row_loss[row] = lse - Srow[yi];
  }
  if (p.G == nullptr) return;
  uint16_t* Grow = p.G + (size_t)row * p.Nc;
  const float gs = p.grad_scale;
  for (int j = t * 8; j < p.Nc; j += TPR * 8) {
    const float4 a = *reinterpret_cast<const float4*>(Srow + j);
    const float4 b = *reinterpret_cast<const float4*>(Srow + j + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float g[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int col = j + e;
      float pr = (windowed && (col < lo || col >= hi)) ? 0.f : __expf(v[e] - lse);  // exp(-inf) == 0
      if (col == yi) pr -= 1.0f;
      g[e] = pr * gs;
    }
    *reinterpret_cast<uint4*>(Grow + j) = make_uint4(pk_bf16(g[0], g[1]), pk_bf16(g[2], g[3]), pk_bf16(g[4], g[5]), pk_bf16(g[6], g[7]));
  }
}

// ----------------------------------------------------------------------------------------------------
// out[0] = scale * sum(x[0..n))   single workgroup, fixed order -> deterministic
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_sum_body(const float* __restrict__ x, int n, float scale, float* out) {
  __shared__ float sm[4];
  float a = 0.f;
  int i = threadIdx.x;
  for (; i + 7 * 256 < n; i += 8 * 256) {  // eight loads in flight, added in index order (8192 row losses: 8.2 -> ~3 us)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += v[u];
  }
  for (; i < n; i += 256) a += x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sm[0] + sm[1] + sm[2] + sm[3]) * scale;
}
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ x, int n, float scale, float* out) {
  reduce_sum_body(x, n, scale, out);
}

// ----------------------------------------------------------------------------------------------------
// split-K combine: out[i] = scale * sum_z part[z][i]      (n4 = elements / 4)
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ part, int splits, size_t n4, float h_scale, const float* d_scale,
                                                   float* __restrict__ out, unsigned nblocks) {
  const float sc = h_scale * (d_scale ? *d_scale : 1.0f);
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < n4; c += (size_t)nblocks * blockDim.x) {
    const float4* src = reinterpret_cast<const float4*>(part) + c;
    float4 a = src[0];
    int z = 1;
    for (; z + 8 <= splits; z += 8) {  // eight slabs in flight, added in slab order (the sum is the one the plain loop forms)
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = src[(size_t)(z + u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
    }
    for (; z < splits; ++z) {
      const float4 b = src[(size_t)z * n4];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
    reinterpret_cast<float4*>(out)[c] = a;
  }
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, size_t n4, float h_scale,
                                                            const float* d_scale, float* __restrict__ out) {
  splitk_reduce_body(part, splits, n4, h_scale, d_scale, out, gridDim.x);
}
// The same launch with one more workgroup, which sums the row losses the forward left (reduce_sum_kernel's arithmetic): the one-call
// training step then has no loss launch of its own between its forward and its backward (round 6: 4.6 us of a 82 us step at 1024 x 8192).
__global__ __launch_bounds__(256) void splitk_reduce_loss_kernel(const float* __restrict__ part, int splits, size_t n4, float h_scale,
                                                                 const float* d_scale, float* __restrict__ out, const float* __restrict__ lx,
                                                                 int ln, float lscale, float* lout) {
  if (blockIdx.x == gridDim.x - 1) {
    reduce_sum_body(lx, ln, lscale, lout);
    return;
  }
  splitk_reduce_body(part, splits, n4, h_scale, d_scale, out, gridDim.x - 1);
}

// ----------------------------------------------------------------------------------------------------
// rank of the gold column: 1 + #{S > gold} + #{S == gold, j < y}      one workgroup per row
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rank_kernel(const float* __restrict__ S, int rows, int cols, const int64_t* __restrict__ y,
                                                   int64_t y_offset, int64_t* __restrict__ rank) {
  const int row = blockIdx.x;
  const float* Srow = S + (size_t)row * cols;
  const int yi = (int)(y[row] + y_offset);
  const float gold = Srow[yi];
  int cnt = 0;
  const int c4 = cols & ~3;
  for (int j = threadIdx.x * 4; j < c4; j += 1024) {
    const float4 v = *reinterpret_cast<const float4*>(Srow + j);
    cnt += (v.x > gold) + (v.y > gold) + (v.z > gold) + (v.w > gold);
    cnt += (v.x == gold && j + 0 < yi) + (v.y == gold && j + 1 < yi) + (v.z == gold && j + 2 < yi) + (v.w == gold && j + 3 < yi);
  }
  for (int j = c4 + threadIdx.x; j < cols; j += 256) {
    const float v = Srow[j];
    cnt += (v > gold) + (v == gold && j < yi);
  }
  __shared__ int sm[4];
  cnt = wave_sum_i(cnt);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) rank[row] = 1 + (int64_t)(sm[0] + sm[1] + sm[2] + sm[3]);
}

// ----------------------------------------------------------------------------------------------------
// Pairwise scores (citadel_task.py:137-146, `sim_score(..., pairwise=True)`): query b against its OWN M contexts only,
//   S[b][j] = sum_k q[b][k] * c[b*M + j][k]        (fp32 in, fp32 accumulate; masked pairs -> -inf)
// and its backward  dq[b] = sum_j g[b][j] * c[b*M + j],  dc[b*M + j] = g[b][j] * q[b].
// Router vectors are vocabulary-wide (d = 30522): these are pure HBM streams (8 bytes per multiply-add), no GEMM.
// Rows are only guaranteed 8-byte aligned (d even) -> float2 accesses; odd d takes the scalar tail path throughout.
// ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pairwise_fwd_kernel(const float* __restrict__ q, const float* __restrict__ c,
                                                           const uint8_t* __restrict__ mask, int B, int M, int d, float* __restrict__ S) {
  const int pair = blockIdx.x, b = pair / M;
  const float* qr = q + (size_t)b * d;
  const float* cr = c + (size_t)pair * d;
  float acc = 0.f;
  if ((d & 1) == 0) {
    const int d2 = d >> 1;
    for (int k = threadIdx.x; k < d2; k += 256) {
      const float2 a = reinterpret_cast<const float2*>(qr)[k], e = reinterpret_cast<const float2*>(cr)[k];
      acc = fmaf(a.x, e.x, acc);
      acc = fmaf(a.y, e.y, acc);
    }
  } else {
    for (int k = threadIdx.x; k < d; k += 256) acc = fmaf(qr[k], cr[k], acc);
  }
  __shared__ float sm[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) S[pair] = (mask != nullptr && mask[pair] != 0) ? -INFINITY : (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// grid (ceil(d / 512), B): thread t owns columns 2 t, 2 t + 1 of its 512-column slab for all M contexts of query b
__global__ __launch_bounds__(256) void pairwise_bwd_kernel(const float* __restrict__ g, const float* __restrict__ q,
                                                           const float* __restrict__ c, int B, int M, int d, float* __restrict__ dq,
                                                           float* __restrict__ dc) {
  const int b = blockIdx.y;
  const int k0 = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (k0 >= d) return;
  const bool two = k0 + 1 < d, vec = (d & 1) == 0;
  const float* qr = q + (size_t)b * d;
  float q0 = qr[k0], q1 = two ? qr[k0 + 1] : 0.f;
  float a0 = 0.f, a1 = 0.f;
  for (int j = 0; j < M; ++j) {
    const float gj = g[(size_t)b * M + j];  // 0 at masked pairs (the caller zeroes them: -inf carries no gradient)
    const size_t row = ((size_t)b * M + j) * d;
    float c0, c1 = 0.f;
    if (vec) {
      const float2 v = *reinterpret_cast<const float2*>(c + row + k0);
      c0 = v.x; c1 = v.y;
    } else {
      c0 = c[row + k0];
      if (two) c1 = c[row + k0 + 1];
    }
    if (gj != 0.f) {  // a masked pair takes no part at all: inf / NaN in a masked (dummy) context row must not reach dq as 0 * inf
      a0 = fmaf(gj, c0, a0);
      a1 = fmaf(gj, c1, a1);
    }
    if (dc != nullptr) {
      if (vec) *reinterpret_cast<float2*>(dc + row + k0) = make_float2(gj * q0, gj * q1);
      else { dc[row + k0] = gj * q0; if (two) dc[row + k0 + 1] = gj * q1; }
    }
  }
  if (dq != nullptr) {
    dq[(size_t)b * d + k0] = a0;
    if (two) dq[(size_t)b * d + k0 + 1] = a1;
  }
}

// ----------------------------------------------------------------------------------------------------
// Streaming top-k per row (torch.topk of run_retrieval_pytorch.py:149-150,156-157, and its shard re-merge
// :272-277), in the total order (score desc, column asc).  State = the k best so far, sorted, in HBM
// ([rows][k] values + int64 columns); one workgroup per row folds one chunk of columns into it:
//   scan the chunk in windows of 4096 / 8192 columns (16 / 32 values per thread in registers, the next window's loads in flight,
//   every row starting at a different window), append the values that beat the current k-th entry to an LDS buffer, and merge
//   buffer + state when enough are waiting -- the threshold only ever rises, so after the first windows almost nothing is
//   appended and the kernel is a pure stream over the scores (268 MB in 62 us = 4.3 TB/s at 1024 x 65536).
//   k <= 256 (12 KB of LDS, 4 rows per CU): merges by COUNTING (the state is sorted, so an entry's new position is a count:
//   no sort, two barriers); an empty state first takes the k-th best of the 256 per-thread maxima of its first window as a
//   bound (a valid lower bound of the row's k-th best: k elements are at or ahead of it), so ~3 % of that window qualifies
//   instead of all of it.  k <= 1024 (48 KB): bitonic sort of buffer + state (up to 4096 slots); an empty state is started by
//   radix select (see the kernel).
//   A candidate list of <= 256 entries (what the GEMM's filter epilogue leaves of a warm chunk) is merged by counting straight
//   from / to HBM.
// Exact and deterministic (the buffer order is arbitrary, the sort is by the total order).  k <= 1024.
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool tk_before(float v1, long long j1, float v2, long long j2) {  // (v1,j1) ranks ahead of (v2,j2)
  return v1 > v2 || (v1 == v2 && j1 < j2);
}

constexpr int TK_KMAX = 1024;        // run_retrieval_pytorch.py takes any --topk; dragon/README recipes use 1000
// two sizes of the kernel: sort size P (LDS: 12 bytes per slot), state slots KM, candidate slots P - KM.  k <= 256 runs on
// 12 KB of LDS (every row of a 1024-query batch resident at once), k <= 1024 on 48 KB (three rows per CU)
constexpr int TK_KSMALL = 256;
constexpr int TK_KWIDE = 4096;       // the wide instantiation: 8192 sort slots (96 KB of LDS, one row per CU)
constexpr size_t tk_lds_bytes(int P) { return (size_t)P * (sizeof(float) + sizeof(long long)); }

struct TopkArgs {
  const float* S;  // [rows][ld]
  int rows, cols;
  long long ld;
  long long col_offset;  // global column index of S[:, 0]
  int k;
  float* vals;     // [rows][k]
  int64_t* idx;    // [rows][k]
  int first;       // 1: start from an empty state
  // candidate-list mode (EpiFilter output): S[row][0..cnt[row]) are scores whose columns are cand_j[row][..];
  // cnt[row] is reset to 0 for the next chunk
  const int* cand_j;
  int* cnt;
};

// number of ranks the candidate (cv, ci) pushes the entry (v, id) down: ties in the value are rare, the 64-bit id compare sits behind
// a branch that almost never runs
__device__ __forceinline__ int tk_ahead(float cv, long long ci, float v, long long id) {
  int a = cv > v ? 1 : 0;
  if (__builtin_expect(cv == v, 0)) a = ci < id ? 1 : 0;
  return a;
}

// state [0,k) (sorted best-first) + candidates [k, k+cnt) (any order) -> the k best, sorted, in [0,k); k + cnt <= NS * 256.
// By counting, no sort: a state entry moves down by the number of candidates ahead of it, a candidate lands at (state entries
// ahead of it, by bisection) + (candidates ahead of it).  NS entries per thread, cnt broadcast reads, two barriers.
template <int NS>
__device__ __forceinline__ void tk_merge_count(float* sv, long long* si, int k, int cnt, int tid) {
  float v[NS];
  long long id[NS];
  int r[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int e = tid + s * 256;
    const bool on = e < k + cnt;
    v[s] = on ? sv[e] : -INFINITY;
    id[s] = on ? si[e] : 0x7fffffffffffffffLL;
    int rr = e;
    if (on && e >= k) {
      int lo = 0, hi = k;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tk_before(sv[mid], si[mid], v[s], id[s])) lo = mid + 1; else hi = mid;
      }
      rr = lo;
    }
    r[s] = on ? rr : 0x40000000;
  }
#pragma unroll 4
  for (int j = k; j < k + cnt; ++j) {
    const float cv = sv[j];
    const long long ci = si[j];
#pragma unroll
    for (int s = 0; s < NS; ++s) r[s] += tk_ahead(cv, ci, v[s], id[s]);
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (r[s] < k) { sv[r[s]] = v[s]; si[r[s]] = id[s]; }
  __syncthreads();
}

// bitonic network over P slots (a power of two, already padded), best-first; ends with a barrier.
// A stage of stride <= 64 keeps every wave inside its own blocks of 128 slots (64 consecutive pair indices t cover 128 consecutive
// slots), so consecutive stages of that kind need no workgroup barrier between them -- LDS operations of one wave execute in
// order; only the compiler has to be told.  For P = 1024 that leaves 6 + 2 barriers of 55 (k = 1000 retrieval: each merge launch is
// a sort of ~1024 candidates per row).
__device__ __forceinline__ void tk_bitonic(float* sv, long long* si, int P, int tid) {
  bool wave_local = false;  // the stage before this one was synchronised inside the waves only
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const bool wide = stride > 64;
      if (wide && wave_local) __syncthreads();
      for (int t = tid; t < P / 2; t += 256) {
        const int lo = (t / stride) * 2 * stride + (t % stride), hi = lo + stride;
        const bool up = (lo & size) == 0;  // this block sorts best-first
        const float a = sv[lo], b = sv[hi];
        const long long ia = si[lo], ib = si[hi];
        const bool swap = up ? tk_before(b, ib, a, ia) : tk_before(a, ia, b, ib);
        if (swap) { sv[lo] = b; sv[hi] = a; si[lo] = ib; si[hi] = ia; }
      }
      if (wide) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      wave_local = !wide;
    }
  }
  __syncthreads();
}

template <bool COUNTING>  // the 12 KB kernel (k <= 256, <= 1024 slots) merges by counting; the 48 KB one sorts
__device__ __forceinline__ void tk_flush(float* sv, long long* si, int k, int cnt, int tid) {
  if (COUNTING) {
    if (k + cnt <= 256) return tk_merge_count<1>(sv, si, k, cnt, tid);
    if (k + cnt <= 512) return tk_merge_count<2>(sv, si, k, cnt, tid);
    return tk_merge_count<4>(sv, si, k, cnt, tid);
  }
  // bitonic sort of everything (padded to a power of two P) best-first
  int P = 64;
  while (P < k + cnt) P <<= 1;
  for (int i = k + cnt + tid; i < P; i += 256) { sv[i] = -INFINITY; si[i] = 0x7fffffffffffffffLL; }
  __syncthreads();
  tk_bitonic(sv, si, P, tid);
}

#ifdef TK_TIMING  // scratch/tk_probe.hip: per-workgroup time (10 ns ticks) spent in each part of the kernel, thread 0's view
__device__ unsigned long long g_tk_tm[4096 * 8];
#define TK_T0() unsigned long long tk_last_ = wall_clock64(), tk_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TK_T(i) do { const unsigned long long t_ = wall_clock64(); tk_acc_[i] += t_ - tk_last_; tk_last_ = t_; } while (0)
#define TK_TN(i) (tk_acc_[i] += 1)
#define TK_TEND() do { if (tid == 0) for (int i_ = 0; i_ < 8; ++i_) g_tk_tm[(size_t)blockIdx.x * 8 + i_] = tk_acc_[i_]; } while (0)
#else
#define TK_T0()
#define TK_T(i)
#define TK_TN(i)
#define TK_TEND()
#endif

template <int TK_P, int TK_KM, int TK_WIN>
__global__ __launch_bounds__(256, TK_P <= 1024 ? 4 : 2) void topk_stream_kernel(TopkArgs p) {
  constexpr int TK_CAP = TK_P - TK_KM;
  constexpr int TK_FOLD_AT = TK_CAP - 512 < 384 ? TK_CAP - 512 : 384;  // a fold appends up to 512 values
  // candidate lists up to TK_NSORT entries are merged by counting, longer ones streamed through the buffer
  constexpr int TK_NSORT = TK_P <= 1024 ? 256 : 2048;
  static_assert(TK_KM + TK_NSORT <= TK_P && TK_KM + 257 <= TK_P, "scratch areas behind the state");
  // 12 bytes per slot: 12 / 48 KB for the two sizes of round 1-3, 96 KB for the wide one (k <= 4096: run_retrieval_pytorch.py:149
  // accepts any --topk) -- dynamic LDS, one allocation (tk_lds_bytes)
  extern __shared__ __attribute__((aligned(16))) unsigned char tk_smem[];
  float* const sv = reinterpret_cast<float*>(tk_smem);
  long long* const si = reinterpret_cast<long long*>(tk_smem + (size_t)TK_P * sizeof(float));
  __shared__ int s_cnt, s_win2[2];  // window counters alternate: the reset of one never races the adds into the other
  const int row = blockIdx.x, tid = threadIdx.x, k = p.k;
  const float* Srow = p.S + (size_t)row * p.ld;
  const int* Jrow = p.cand_j != nullptr ? p.cand_j + (size_t)row * p.ld : nullptr;
  const int ncols = p.cnt != nullptr ? p.cnt[row] : p.cols;
  if (p.cnt != nullptr && ncols == 0) return;  // (uniform) nothing qualified for this row in this chunk
  for (int i = tid; i < k; i += 256) {
    const long long j = p.first ? -1 : (long long)p.idx[(size_t)row * k + i];
    sv[i] = (p.first || j < 0) ? -INFINITY : p.vals[(size_t)row * k + i];
    si[i] = j < 0 ? 0x7fffffffffffffffLL : j;
  }
  if (tid == 0) { s_cnt = 0; s_win2[0] = 0; s_win2[1] = 0; }
  if (p.cnt != nullptr && ncols <= TK_NSORT) {
    // A warm chunk leaves a row a short candidate list (~k * chunk / columns seen): no pass over a score matrix, no sort of the state.
    for (int i = tid; i < ncols; i += 256) {
      sv[k + i] = Srow[i];
      si[k + i] = p.col_offset + Jrow[i];
    }
    if constexpr (TK_P <= 1024) {
      // k <= 256, <= 256 candidates: every entry's final position is a count (tk_merge_count): two barriers
      __syncthreads();
      if (tid == 0) p.cnt[row] = 0;
      const int n = k + ncols;
      if (n <= 256) tk_merge_count<1>(sv, si, k, ncols, tid);
      else tk_merge_count<2>(sv, si, k, ncols, tid);
      for (int i = tid; i < k; i += 256) {
        p.vals[(size_t)row * k + i] = sv[i];
        p.idx[(size_t)row * k + i] = si[i] == 0x7fffffffffffffffLL ? -1 : (int64_t)si[i];
      }
    } else {
      // k up to 1024, up to 2048 candidates: counting would cost (k + m) * m compares.  Sort the candidates alone (bitonic over the
      // next power of two), then merge two sorted lists by bisection: a candidate lands at (its index) + (state entries ahead of
      // it), a state entry at (its index) + (candidates ahead of it); results go straight to HBM.
      int Pm = 64;
      while (Pm < ncols) Pm <<= 1;
      for (int i = ncols + tid; i < Pm; i += 256) { sv[k + i] = -INFINITY; si[k + i] = 0x7fffffffffffffffLL; }
      __syncthreads();
      if (tid == 0) p.cnt[row] = 0;
      float* cv = sv + k;
      long long* ci = si + k;
      tk_bitonic(cv, ci, Pm, tid);
      for (int e = tid; e < k + ncols; e += 256) {
        const bool is_state = e < k;
        const float v = sv[e];
        const long long id = si[e];
        const float* ov = is_state ? cv : sv;       // the OTHER sorted list
        const long long* oi = is_state ? ci : si;
        int lo = 0, hi = is_state ? ncols : k;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (tk_before(ov[mid], oi[mid], v, id)) lo = mid + 1; else hi = mid;
        }
        const int r = (is_state ? e : e - k) + lo;
        if (r < k) {
          p.vals[(size_t)row * k + r] = v;
          p.idx[(size_t)row * k + r] = id == 0x7fffffffffffffffLL ? -1 : (int64_t)id;
        }
      }
    }
    return;
  }
  TK_T0();
  __syncthreads();
  if (p.cnt != nullptr && tid == 0) p.cnt[row] = 0;
  const bool vec = (p.ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.S) & 15) == 0);
  if constexpr (TK_P > 1024) {
    // Large k from an empty state (--topk 1000 of the DRAGON recipes): the buffer scheme below would sort 2048-4096 slots a dozen
    // times before its threshold is worth anything (1.9 ms for 1024 x 65536, k = 1000).  Radix select instead: histogram the row by
    // the top byte of an order-preserving key, find the bin the k-th best falls into, refine byte by byte until "everything ahead of
    // the bin + the bin" fits a 2048-slot sort, collect exactly those, sort once.  Membership is decided on values only (equal
    // floats have equal keys, -0 == +0), the tie rule on ids by the sort: exact.  2-3 streaming passes + 1 collecting pass.
    if (p.first && p.cnt == nullptr && ncols >= 4096) {
      int* hist = reinterpret_cast<int*>(si);  // 256 bins + 3 control words (the state is empty: its slots are free)
      auto key_of = [](float x) -> unsigned {
        if (x == 0.f) return 0x80000000u;
        if (x != x) return 0u;  // (NaN never reaches a histogram or the collecting pass: it never qualifies, as in the streaming scheme)
        const unsigned u = __float_as_uint(x);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      };
      auto for_each = [&](auto&& f) {
        for (int j0 = tid * 4; j0 < ncols; j0 += 4096) {
          float x[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = j0 + u * 1024;
            if (vec && j + 3 < ncols) {
              const float4 y = *reinterpret_cast<const float4*>(Srow + j);
              x[u][0] = y.x; x[u][1] = y.y; x[u][2] = y.z; x[u][3] = y.w;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) x[u][e] = (j + e < ncols) ? Srow[j + e] : 0.f;
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int j = j0 + u * 1024 + e;
              if (j < ncols) f(x[u][e], j);
            }
        }
      };
      unsigned prefix = 0;
      int need = k, shift = 24, total = 0;
      bool ok = false;
      for (int pass = 0; pass < 4; ++pass, shift -= 8) {
        hist[tid] = 0;
        if (tid == 0) hist[256] = -1;
        __syncthreads();
        for_each([&](float x, int) {
          if (x != x) return;  // NaN never qualifies
          const unsigned key = key_of(x);
          if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        });
        __syncthreads();
        int above = 0;
        for (int b = tid + 1; b < 256; ++b) above += hist[b];
        const int mine = hist[tid];
        if (above < need && need <= above + mine) { hist[256] = tid; hist[257] = need - above; hist[258] = mine; }
        __syncthreads();
        if (hist[256] < 0) break;  // fewer than k values that are not NaN: the streaming scheme leaves the rest of the state empty
        prefix = (prefix << 8) | (unsigned)hist[256];
        need = hist[257];
        total = (k - need) + hist[258];  // strictly ahead of the bin + the bin itself
        __syncthreads();
        // (the wide kernel's k alone exceeds 2048: it collects as soon as the candidates fit its 8192 slots)
        if (total <= (TK_P <= 4096 ? 2048 : TK_P) || (pass == 3 && total <= TK_P)) { ok = true; break; }
      }
      if (ok) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for_each([&](float x, int j) {
          if (x == x && (key_of(x) >> shift) >= prefix) {
            const int pos = atomicAdd(&s_cnt, 1);
            sv[pos] = x;
            si[pos] = p.col_offset + j;
          }
        });
        __syncthreads();
        tk_flush<false>(sv, si, 0, s_cnt, tid);  // == total; bitonic, best-first
        for (int i = tid; i < k; i += 256) {
          p.vals[(size_t)row * k + i] = sv[i];
          p.idx[(size_t)row * k + i] = si[i] == 0x7fffffffffffffffLL ? -1 : (int64_t)si[i];
        }
        return;
      }
      // more exact ties at the k-th value than the buffer holds: the streaming scheme below handles that; restore the empty state
      for (int i = tid; i < k; i += 256) { sv[i] = -INFINITY; si[i] = 0x7fffffffffffffffLL; }
      __syncthreads();
    }
  }
  float tv = sv[k - 1];
  long long ti = si[k - 1];
  // incl: (tv, ti) is an element of THIS window that is not in the state yet (the bound of an empty state, below): it qualifies too
  bool incl = false;
  // merging costs (k + m) * m compares for m waiting candidates: merge early and often (the threshold rises sooner, too)
  const int flush_at = TK_P <= 1024 ? max(k, 64) : min(TK_CAP / 2, max(256, 2 * k));
  int wpar = 0;
  // Rows are a power-of-two stride apart (ld * 4 bytes): workgroups walking their rows in step would all be on the same few HBM
  // channels at any moment.  Every row starts at a different window and wraps around (the result does not depend on the order).
  const int nwin = (ncols + TK_WIN * 1024 - 1) / (TK_WIN * 1024), nfull = ncols / (TK_WIN * 1024);
  const int win0 = nfull > 0 ? (int)(((unsigned)row * 7u) % (unsigned)nfull) : 0;
  // the next window's loads are in flight while this one is examined (a window is a dependent trip to HBM otherwise)
  float vn[TK_WIN][4];
  int cn[TK_WIN][4];
  auto load_window = [&](int itx) {
    const int b = ((win0 + itx) % nwin) * (TK_WIN * 1024);
#pragma unroll
    for (int w = 0; w < TK_WIN; ++w) {
      const int j = b + w * 1024 + tid * 4;
      if (vec && j + 3 < ncols) {
        const float4 x = *reinterpret_cast<const float4*>(Srow + j);
        vn[w][0] = x.x; vn[w][1] = x.y; vn[w][2] = x.z; vn[w][3] = x.w;
        if (Jrow != nullptr) {
          const int4 y = *reinterpret_cast<const int4*>(Jrow + j);
          cn[w][0] = y.x; cn[w][1] = y.y; cn[w][2] = y.z; cn[w][3] = y.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vn[w][e] = (j + e < ncols) ? Srow[j + e] : NAN;  // NaN never qualifies
          if (Jrow != nullptr) cn[w][e] = (j + e < ncols) ? Jrow[j + e] : 0;
        }
      }
    }
  };
  if (nwin > 0) load_window(0);
  for (int it = 0; it < nwin; ++it, wpar ^= 1) {
    const int base = ((win0 + it) % nwin) * (TK_WIN * 1024);
    float v[TK_WIN][4];
    int cj[TK_WIN][4];  // column of each value (implicit for a score matrix, loaded for a candidate list)
    int mine = 0;
#pragma unroll
    for (int w = 0; w < TK_WIN; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[w][e] = vn[w][e];
        cj[w][e] = Jrow != nullptr ? cn[w][e] : base + w * 1024 + tid * 4 + e;
      }
    if (it + 1 < nwin) load_window(it + 1);
    if (p.first && it == 0 && p.cnt == nullptr && k <= 256 && base + 1024 <= ncols) {
      // Empty state: everything would qualify and the first window alone would cost a 2048-slot sort per 1024 values.  A valid
      // bound is cheap: the 256 per-thread maxima are 256 distinct elements, so the k-th best of them has k elements at or ahead
      // of it -- the row's k-th best cannot rank behind it.  Only values at or ahead of the bound enter the buffer (~1.5 % of a
      // window of random scores for k = 100).
      float bv = -INFINITY;
      long long bj = 0x7fffffffffffffffLL;
#pragma unroll
      for (int w = 0; w < TK_WIN; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const long long gj = p.col_offset + cj[w][e];
          if (tk_before(v[w][e], gj, bv, bj)) { bv = v[w][e]; bj = gj; }
        }
      float* mv = sv + TK_KM;
      long long* mj = si + TK_KM;
      mv[tid] = bv;
      mj[tid] = bj;
      if (tid == 0) { mv[256] = -INFINITY; mj[256] = 0x7fffffffffffffffLL; }
      __syncthreads();
      int r = 0;
#pragma unroll 8
      for (int t = 0; t < 256; ++t) r += tk_ahead(mv[t], mj[t], bv, bj);
      if (r == k - 1) { mv[256] = bv; mj[256] = bj; }  // ranks among the maxima are distinct unless sentinels tie (then none
      __syncthreads();                                  // may match: the bound stays the empty state's, everything qualifies)
      if (mj[256] != 0x7fffffffffffffffLL || mv[256] > -INFINITY) { tv = mv[256]; ti = mj[256]; incl = true; }
    }
    // qualifiers of this window, one bit per held value (a tie in the value is rare: the id compare sits behind a branch; NaN
    // never qualifies)
    unsigned qm = 0;
#pragma unroll
    for (int w = 0; w < TK_WIN; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = v[w][e];
        bool q = x > tv;
        if (__builtin_expect(x == tv, 0)) {
          const long long gj = p.col_offset + cj[w][e];
          q = gj < ti || (incl && gj == ti);
        }
        qm |= (q ? 1u : 0u) << (w * 4 + e);
      }
    mine = __popc(qm);
    TK_T(1);  // loads landed, qualifiers counted (window 0: + the bound)
    if (mine) atomicAdd(&s_win2[wpar], mine);
    __syncthreads();
    const int win = s_win2[wpar], cnt0 = s_cnt;
    if (tid == 0) s_win2[wpar ^ 1] = 0;  // the OTHER counter (next window's): nobody touches it before the next barrier pair
    __syncthreads();
    TK_T(2);  // barrier pair
    if (win == 0) continue;  // (uniform) the common case once the threshold has risen
    if (cnt0 + win <= TK_CAP) {
      if (qm) {  // one slot reservation per thread, then plain stores
        int pos = k + atomicAdd(&s_cnt, mine);
#pragma unroll
        for (int w = 0; w < TK_WIN; ++w)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (qm & (1u << (w * 4 + e))) {
              sv[pos] = v[w][e];
              si[pos] = p.col_offset + cj[w][e];
              ++pos;
            }
      }
      __syncthreads();
      TK_T(3);  // append
      if (s_cnt > flush_at) {  // uniform
        tk_flush<(TK_P <= 1024)>(sv, si, k, s_cnt, tid);
        if (tid == 0) s_cnt = 0;
        tv = sv[k - 1];
        ti = si[k - 1];
        incl = false;
        __syncthreads();
        TK_T(4);  // flush
        TK_TN(6);
      }
    } else {
      TK_TN(7);
      // too many qualifiers for the buffer (in practice the first window of an empty state, where everything qualifies):
      // fold 512 values at a time and sort as soon as TK_FOLD_AT candidates are waiting.  The bitonic sort costs P log^2 P LDS
      // operations (28 us at P = 2048 with four workgroups sharing a CU's LDS, a third of that at P = 1024), and after the
      // very first one the threshold already rejects most of what follows.
#pragma unroll
      for (int w = 0; w < TK_WIN; ++w) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (s_cnt > TK_FOLD_AT) {  // uniform (read after a barrier)
            tk_flush<(TK_P <= 1024)>(sv, si, k, s_cnt, tid);
            if (tid == 0) s_cnt = 0;
            tv = sv[k - 1];
            ti = si[k - 1];
            incl = false;
            __syncthreads();
          }
#pragma unroll
          for (int e = 2 * h; e < 2 * h + 2; ++e) {
            const long long gj = p.col_offset + cj[w][e];
            // (qm was taken against the window's opening threshold: re-test, the folds raise it)
            if ((qm & (1u << (w * 4 + e))) && (tk_before(v[w][e], gj, tv, ti) || (incl && v[w][e] == tv && gj == ti))) {
              const int pos = atomicAdd(&s_cnt, 1);
              sv[k + pos] = v[w][e];
              si[k + pos] = gj;
            }
          }
          __syncthreads();
        }
      }
    }
  }
  TK_T(5);  // (slow path, if taken)
  if (s_cnt > 0) tk_flush<(TK_P <= 1024)>(sv, si, k, s_cnt, tid);
  __syncthreads();
  TK_T(4);
  for (int i = tid; i < k; i += 256) {
    p.vals[(size_t)row * k + i] = sv[i];
    p.idx[(size_t)row * k + i] = si[i] == 0x7fffffffffffffffLL ? -1 : (int64_t)si[i];
  }
  TK_T(0);
  TK_TEND();
}

}  // namespace dprhot
