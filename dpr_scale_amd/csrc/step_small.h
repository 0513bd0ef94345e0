// step_small.h -- the second (and last) launch of a whole training step at the latency-bound BASELINE shapes
// (B <= 32 queries per rank, Nc <= 1152 gathered contexts, d % 16 == 0: cfg1, cfg2, cfg4 per rank).
//
// dpr_task.py:209-212 (softmax cross-entropy) and the autograd backward of :98-105 (dQ = G C, dC = G^T Q) in ONE
// kernel.  At these sizes every launch costs a kernel boundary plus one dependent trip to memory (~3.7 us), whatever it
// computes, so the step is the number of dependent launches.  The row softmax needs complete rows and the backward GEMMs
// need complete columns of G, which is why they were two launches; here each workgroup owns 64 columns of d, recomputes
// the (tiny: <= 32 x 512) softmax from the partial-logit slabs of the sim launch -- 128 KiB of L2 reads per workgroup at
// cfg2 -- keeps G in LDS and feeds both GEMMs from that one image:
//     dQ[:, n0:n0+64]      = G   x C[:, n0:n0+64]     A = G row-major = k-major fragments (ds_read_b128)
//     dC_part[:, n0:n0+64] = G^T x Q[:, n0:n0+64]     A = the SAME image read through ds_read_b64_tr_b16
// d/16 workgroups of 1024 threads (48 at d = 768), no atomics, fixed summation orders (bit-reproducible); workgroup 0
// also writes loss / logsumexp / G / logits.  What in-kernel stamps showed for a 256-thread version (cfg2, 5.4 us inside
// the kernel): 1.7 us pulling the 128 KiB of slabs + operand tiles through one CU's L1 (64 B/clk), 1.8 us of softmax
// ALU at 32 scores per thread, 0.9 + 0.7 us in two latency-bound MFMA loops -- hence 1024 threads (8 scores each), the
// dQ contraction split over 8 wave pairs and one 16-row block of dC per wave.
#pragma once
#include "gemm_bf16.h"
#include "rowwise.h"

namespace dprhot {

constexpr int SS_ROWS = 32;     // query rows of one row block (rows beyond B are zero)
constexpr int SS_MAXB = 64;     // query rows of the launch: two row blocks at most
constexpr int SS_MAXNC = 1152;  // G image + C tile + dQ partials must fit the 160 KiB of LDS
// TW = columns of d per workgroup (16 in the library); the C / Q tile images have row stride TW + 8 elements

struct StepSmallArgs {
  const float* slabs;  // [splits][B][Nc] partial logits (mask and 1/T applied: -inf at masked columns)
  int splits;
  size_t slab_stride;
  int B, Nc, d;
  const int64_t* y;
  int64_t y_offset;
  float grad_scale;
  const uint16_t* Qb;  // [B][d]  bf16 (written by the sim launch)
  const uint16_t* Cb;  // [Nc][d] bf16
  float h_scale;
  const float* d_scale;
  float* dQ;           // [B][d]
  float* dC;           // [Nc][d]
  float* S_out;        // optional [B][Nc]
  float* row_loss;     // optional [B]
  float* row_lse;      // optional [B]
  float* loss_sum;     // [1]
  uint16_t* G;         // optional [B][Nc]
  int stamp_period, stamp_row;  // > 0: dC[m][0] = loss numerator for m % stamp_period == stamp_row (EpiScaleF32)
  float loss_scale = 1.0f;
};

inline size_t step_small_lds(int Nc, int TW) {
  const int ncp = (Nc + 31) / 32 * 32, ts = TW + 8;
  return (size_t)SS_ROWS * (ncp + 8) * 2 + (size_t)ncp * ts * 2 + (size_t)SS_ROWS * ts * 2 + SS_MAXB * sizeof(float) +
         (size_t)8 * SS_ROWS * TW * sizeof(float);  // + the dQ partial sums of the 8 K slices
}

template <int CTRL>
__device__ __forceinline__ float ss_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// reductions over 8 consecutive lanes (aligned): xor 1, xor 2 inside the quad, then the two quads of the half row
__device__ __forceinline__ float ss_max8(float v) {
  v = fmaxf(v, ss_dpp<0xB1>(v));   // quad_perm [1,0,3,2]
  v = fmaxf(v, ss_dpp<0x4E>(v));   // quad_perm [2,3,0,1]
  return fmaxf(v, ss_dpp<0x141>(v));  // row_half_mirror
}
__device__ __forceinline__ float ss_sum8(float v) {
  v += ss_dpp<0xB1>(v);
  v += ss_dpp<0x4E>(v);
  return v + ss_dpp<0x141>(v);
}

__device__ __forceinline__ bf16x8 ss_tr_frag(const uint16_t* T, int stride, int k0, int c0, int lane) {
  // fragment of an [k][col] image for 16 columns c0.. and the 32 k values k0..: see load_frag (gemm_bf16.h)
  const int i = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const uint16_t* p = T + (k0 + g * 8 + (i >> 2)) * stride + c0 + (i & 3) * 4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4*)(p + 4 * stride));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// CPT: 8-value chunks of a row per thread (32 threads per row): Nc <= 256 * CPT.  TW: columns of d per workgroup.
// NS: partial-logit slabs read (>= p.splits; the sim launch writes 1 slab above 512 columns, up to 4 below).
// NRB: row blocks of 32 (1: B <= 32, the body runs once, straight-line as before; 2: B <= 64, unrolled twice).
template <int CPT, int TW, int NS, int NRB = 1>
__global__ __launch_bounds__(1024) void step_small_kernel(StepSmallArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t ss_smem[];
  constexpr int TS = TW + 8;          // tile image row stride (elements)
  constexpr int TC = TW / 8;          // 16-byte chunks per tile row
  constexpr int CU = (256 * CPT * TC + 1023) / 1024;  // C-tile chunks per thread
  constexpr int NF = TW / 16;         // 16-column fragments of the tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workgroup -> column tile: linear ids are dealt round-robin to the 8 XCDs; give each XCD a CONTIGUOUS run of tiles, so
  // that the 4 neighbouring 16-column tiles that share every 128-byte line of C / Q rows meet in one L2 (PMC: 3.5 MB of
  // HBM traffic per launch at cfg2 with the identity mapping against 1.5 MB algorithmic)
  const int nwg = gridDim.x;
  const int tile = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;
  const int n0 = tile * TW;
  const int Nc = p.Nc, cpr = Nc >> 3;
  const int ncp = (Nc + 31) / 32 * 32, gs = ncp + 8;
  uint16_t* const Gs = ss_smem;                       // [32][gs]   G, row-major
  uint16_t* const Cs = Gs + SS_ROWS * gs;             // [ncp][TS]  C[:, n0:n0+TW]
  uint16_t* const Qs = Cs + ncp * TS;                 // [32][TS]   Q[:, n0:n0+TW]
  float* const s_rl = reinterpret_cast<float*>(Qs + SS_ROWS * TS);  // [SS_MAXB] row losses of all row blocks
  float* const red = s_rl + SS_MAXB;                  // [8][32][TW] dQ partial sums

  // Row blocks of 32 (round 3: B <= 64 -- the per-GPU batch of the DRAGON / NQ recipes -- takes two turns through the body below with
  // the C tile resident and the dC accumulators kept in registers; a third launch plus a G round trip through HBM cost more than
  // the second softmax: 5 + ~2.5 us against ~11).  The first block's loads are all issued back to back with the C tile's.
  const int lrow = tid >> 5, tr = tid & 31;
  const float dsc = p.d_scale ? *p.d_scale : 1.0f;
  const bool lead = tile == 0;
  const float sc = p.h_scale * dsc;
  const int i = lane & 15, g = lane >> 4;
  f32x4 dcacc[CPT][NF];
#pragma unroll
  for (int it = 0; it < CPT; ++it)
#pragma unroll
    for (int b = 0; b < NF; ++b) dcacc[it][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
  // ---- all global reads of the block, back to back (nothing is used before the last one is issued) ----
  DPRHOT_TM(8);
  const int row = rb * SS_ROWS + lrow;
  const bool active = row < p.B;
  const int64_t yraw = active ? p.y[row] : (int64_t)-1;
  uint4 qreg = make_uint4(0u, 0u, 0u, 0u);
  if (active && tr < TC) qreg = *reinterpret_cast<const uint4*>(p.Qb + (size_t)row * p.d + n0 + tr * 8);
  uint4 creg[CU];  // the C tile: loaded and parked in LDS by the first block only
  if (rb == 0) {
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      const int q = tid + u * 1024, j = q / TC, cc = q % TC;
      creg[u] = make_uint4(0u, 0u, 0u, 0u);
      if (j < Nc) creg[u] = *reinterpret_cast<const uint4*>(p.Cb + (size_t)j * p.d + n0 + cc * 8);
    }
  }
  // partial-logit slabs (at most NS): every load is issued unconditionally on a valid address (absent slabs re-read
  // slab 0 and are dropped by a select at the add) -- a loop over p.splits would wait for one slab before asking for
  // the next: dependent trips to L2 instead of one
  float4 sa[CPT][NS], sb[CPT][NS];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int chunk = tr + k * 32;
    const bool ok = active && chunk < cpr;
    const float* src = p.slabs + (ok ? (size_t)row * Nc + (size_t)chunk * 8 : (size_t)0);
#pragma unroll
    for (int z = 0; z < NS; ++z) {
      const float* sz = src + (z < p.splits ? (size_t)z * p.slab_stride : (size_t)0);
      sa[k][z] = *reinterpret_cast<const float4*>(sz);
      sb[k][z] = *reinterpret_cast<const float4*>(sz + 4);
    }
  }

  // ---- operand tiles -> LDS ----
  DPRHOT_TM(9);
  if (rb > 0) __syncthreads();  // the previous block's readers of Gs / Qs / red are done
  if (tr < TC) *reinterpret_cast<uint4*>(Qs + lrow * TS + tr * 8) = qreg;
  if (rb == 0) {
#pragma unroll
    for (int u = 0; u < CU; ++u) {
      const int q = tid + u * 1024, j = q / TC, cc = q % TC;
      if (j < ncp) *reinterpret_cast<uint4*>(Cs + j * TS + cc * 8) = creg[u];
    }
  }

  // ---- row softmax (32 lanes per row: DPP over 16, one lane exchange across the two halves), loss, G ----
  DPRHOT_TM(10);
  float v[CPT][8];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    float4 a = sa[k][0], b = sb[k][0];
#pragma unroll
    for (int z = 1; z < NS; ++z) {
      const bool on = z < p.splits;
      a.x += on ? sa[k][z].x : 0.f; a.y += on ? sa[k][z].y : 0.f; a.z += on ? sa[k][z].z : 0.f; a.w += on ? sa[k][z].w : 0.f;
      b.x += on ? sb[k][z].x : 0.f; b.y += on ? sb[k][z].y : 0.f; b.z += on ? sb[k][z].z : 0.f; b.w += on ? sb[k][z].w : 0.f;
    }
    const bool ok = active && (tr + k * 32) < cpr;
    v[k][0] = ok ? a.x : -INFINITY; v[k][1] = ok ? a.y : -INFINITY; v[k][2] = ok ? a.z : -INFINITY; v[k][3] = ok ? a.w : -INFINITY;
    v[k][4] = ok ? b.x : -INFINITY; v[k][5] = ok ? b.y : -INFINITY; v[k][6] = ok ? b.z : -INFINITY; v[k][7] = ok ? b.w : -INFINITY;
  }
  DPRHOT_TM(11);
  const int yi = active ? (int)(yraw + p.y_offset) : -1;
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < CPT; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, v[k][e]);
  m = dprhot_row16_max(m);
  m = fmaxf(m, __shfl_xor(m, 16));
  float sm = 0.f, gold = 0.f;  // exactly one lane of the row holds the gold column
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c0 = (tr + k * 32) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (yi == c0 + e) gold = v[k][e];
  }
  // every workgroup repeats this softmax, so its ALU time is on the critical path of the whole launch: ONE exponential
  // per score -- e = exp(v - max) feeds the row sum and, scaled by 1 / sum, the probabilities
  float ex[CPT][8];
  const bool dead = m == -INFINITY;  // a row with every column masked (the reference yields NaN there as well)
#pragma unroll
  for (int k = 0; k < CPT; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ex[k][e] = dead ? 0.f : __expf(v[k][e] - m);
      sm += ex[k][e];
    }
  sm = dprhot_row16_sum(sm);
  gold = dprhot_row16_sum(gold);
  sm += __shfl_xor(sm, 16);
  gold += __shfl_xor(gold, 16);
  const float lse = m + logf(sm);
  const float inv_sm = 1.0f / sm;  // sm = 0 (dead row): inf * 0 = NaN, like exp(v - lse) with lse = NaN
  if (tr == 0) {
    const float l = active ? lse - gold : 0.f;
    s_rl[rb * SS_ROWS + lrow] = l;
    if (lead && active) {
      if (p.row_lse) p.row_lse[row] = lse;
      if (p.row_loss) p.row_loss[row] = l;
    }
  }
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int chunk = tr + k * 32;
    if (chunk * 8 < ncp) {
      uint4 gv = make_uint4(0u, 0u, 0u, 0u);
      if (active && chunk < cpr) {
        const int c0 = chunk * 8;
        float gg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float pr = ex[k][e] * inv_sm;
          if (c0 + e == yi) pr -= 1.0f;
          gg[e] = pr * p.grad_scale;
        }
        gv = make_uint4(pk_bf16(gg[0], gg[1]), pk_bf16(gg[2], gg[3]), pk_bf16(gg[4], gg[5]), pk_bf16(gg[6], gg[7]));
        if (lead) {
          if (p.G != nullptr) *reinterpret_cast<uint4*>(p.G + (size_t)row * Nc + c0) = gv;
          if (p.S_out != nullptr) {
            float* dst = p.S_out + (size_t)row * Nc + c0;
            *reinterpret_cast<float4*>(dst) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(v[k][4], v[k][5], v[k][6], v[k][7]);
          }
        }
      }
      *reinterpret_cast<uint4*>(Gs + lrow * gs + chunk * 8) = gv;
    }
  }
  DPRHOT_TM(12);
  __syncthreads();
  DPRHOT_TM(13);
  if constexpr (NRB == 1) {  // one block: the loss is complete here, summed under the GEMMs
    if (lead && tid == 0) {
      double tot = 0.0;
      for (int r = 0; r < p.B; ++r) tot += (double)s_rl[r];
      p.loss_sum[0] = (float)tot * p.loss_scale;
      s_rl[0] = (float)tot * p.loss_scale;  // (row losses are no longer needed) for the stamp below, read after the next barrier
    }
  }

  // ---- dQ[rows of the block, n0:n0+TW] = G[32, Nc] x C[Nc, TW]: wave w -> rows (w & 1) * 16.., K slice w >> 1 of 8; partial sums
  //      through LDS, added in slice order ----
  {
    const int wm = wave & 1, ks = wave >> 1;
    f32x4 acc[NF];
#pragma unroll
    for (int b = 0; b < NF; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < CPT; ++t) {  // ncp / 32 <= 8 * CPT K steps of 32
      const int kk = ks + t * 8;
      if (kk * 32 < ncp) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Gs + (wm * 16 + i) * gs + kk * 32 + g * 8);
#pragma unroll
        for (int b = 0; b < NF; ++b) {
          const bf16x8 bfr = ss_tr_frag(Cs, TS, kk * 32, b * 16, lane);
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[b], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NF; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(ks * SS_ROWS + wm * 16 + g * 4 + r) * TW + b * 16 + i] = acc[b][r];
  }
  __syncthreads();
  // ---- dC_part[0:Nc, n0:n0+TW] += G^T[Nc, 32] x Q[32, TW]: one 16-row block of contexts per wave and round ----
  DPRHOT_TM(14);
  {
    bf16x8 bq[NF];
#pragma unroll
    for (int b = 0; b < NF; ++b) bq[b] = ss_tr_frag(Qs, TS, 0, b * 16, lane);
    if constexpr (NRB == 1) {  // one block: every 16-row block of contexts is stored as soon as it is multiplied
      const bool stamp = lead && p.stamp_period > 0;  // column 0 of d belongs to workgroup 0
      float* out = p.dC + (size_t)(wave * 16 + g * 4) * p.d + n0 + i;
      const size_t step = (size_t)256 * p.d;
#pragma unroll
      for (int it = 0; it < CPT; ++it) {  // Nc <= 256 * CPT rows, 256 per round of the sixteen waves
        const int j0 = wave * 16 + it * 256;
        if (j0 < Nc) {
          const bf16x8 af = ss_tr_frag(Gs, gs, 0, j0, lane);  // A(m = context j0 + i, k = query row) = G[k][m]
          f32x4 acc[NF];
#pragma unroll
          for (int b = 0; b < NF; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bq[b], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          float* o = out;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (j0 + g * 4 + r < Nc) {
#pragma unroll
              for (int b = 0; b < NF; ++b) {
                float vv = acc[b][r] * sc;
                if (b == 0 && i == 0 && stamp && (j0 + g * 4 + r) % p.stamp_period == p.stamp_row) vv = s_rl[0];
                o[b * 16] = vv;
              }
            }
            o += p.d;
          }
        }
        out += step;
      }
    } else {
#pragma unroll
      for (int it = 0; it < CPT; ++it) {
        const int j0 = wave * 16 + it * 256;
        if (j0 < Nc) {
          const bf16x8 af = ss_tr_frag(Gs, gs, 0, j0, lane);
#pragma unroll
          for (int b = 0; b < NF; ++b) dcacc[it][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bq[b], dcacc[it][b], 0, 0, 0);
        }
      }
    }
  }
  // dQ of the block: add the 8 K slices in order
  for (int e = tid; e < SS_ROWS * TW; e += 1024) {
    const int r = e / TW, ccol = e - r * TW;
    if (rb * SS_ROWS + r < p.B) {
      float s = red[e];
#pragma unroll
      for (int k = 1; k < 8; ++k) s += red[k * SS_ROWS * TW + e];
      p.dQ[(size_t)(rb * SS_ROWS + r) * p.d + n0 + ccol] = s * sc;
    }
  }
  }  // row blocks

  if constexpr (NRB > 1) {
  __syncthreads();  // every block's row losses are in s_rl (and red is free)
  if (lead && tid == 0) {
    double tot = 0.0;
    for (int r = 0; r < p.B; ++r) tot += (double)s_rl[r];
    p.loss_sum[0] = (float)tot * p.loss_scale;
    red[0] = (float)tot * p.loss_scale;  // for the stamp below
  }
  {
    const bool stamp = lead && p.stamp_period > 0;  // column 0 of d belongs to workgroup 0
    float lsum = 0.f;
    if (stamp) {  // workgroup-uniform
      __syncthreads();
      lsum = red[0];
    }
    float* out = p.dC + (size_t)(wave * 16 + g * 4) * p.d + n0 + i;
    const size_t step = (size_t)256 * p.d;
#pragma unroll
    for (int it = 0; it < CPT; ++it) {
      const int j0 = wave * 16 + it * 256;
      if (j0 < Nc) {
        float* o = out;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (j0 + g * 4 + r < Nc) {
#pragma unroll
            for (int b = 0; b < NF; ++b) {
              float vv = dcacc[it][b][r] * sc;
              if (b == 0 && i == 0 && stamp && (j0 + g * 4 + r) % p.stamp_period == p.stamp_row) vv = lsum;
              o[b * 16] = vv;
            }
          }
          o += p.d;
        }
      }
      out += step;
    }
  }
  }  // NRB > 1
  DPRHOT_TM(15);
}

}  // namespace dprhot
