// skinny.h -- the training step when a rank holds FEW query rows against MANY gathered contexts (B <= 128, Nc in the
// thousands: BASELINE cfg3 per rank = 128 x 8256 x 768).  dpr_task.py:197-212 and its autograd backward, four launches:
//
//   sk_sim_kernel   S = (q x C^T) / T with the column mask (dpr_task.py:98-105,:211), fp32 logits + one logsumexp per
//                   (row, 128-column tile) + the gold logit                                     [reads C once]
//   sk_g_kernel     row logsumexp from the tile values, loss, G = (exp(S - lse) - onehot) * scale in bf16
//   sk_bwd_kernel   dC_part = G^T x Q (128 x 128 units) and split-K partial sums of dQ = G x C (128 x 64 units over a
//                   slice of contexts) side by side in one launch                               [writes dC, reads C once]
//   sk_dq_reduce_kernel adds the dQ partials.
// (A variant that recomputed G inside the dC units -- no sk_g launch, no G round trip -- measured slower: six d tiles per
//  context tile repeat 16 K exponentials and 33 KB of statistics reads each, 12 us per workgroup against 6.)
//
// Why not the gemm_bf16.h engine: at M = 128 every unit of work is a short K loop whose cost is the latency of its
// dependent loads; a CU needs >= 128 KB of loads in flight to keep its 64 B/clk L2 port busy (Little's law at ~1 us),
// which register staging cannot hold.  Here every unit puts its WHOLE operand footprint in flight at once: LDS-DMA
// (global_load_lds_dwordx4) into a ring that covers the unit's K range, fp32 operands (q, the logits) through registers
// issued before the DMAs.  Waits are counted vmcnt(N) + raw s_barrier, so the tail of the ring stays in flight.
// Epilogues go through LDS so that every global store is 16 bytes per lane, whole 128-byte lines per row.
#pragma once
#include "gemm_bf16.h"
#include "gemm256.h"
#include "rowwise.h"
#include "step_small.h"

namespace dprhot {

constexpr int SK_THREADS = 256;
constexpr int SK_ROWS = 32;    // query rows of a sim unit
constexpr int SK_COLS = 128;   // context columns of a dC unit (and of the wide sim unit)
constexpr int SK_SCOLS = 64;   // narrow sim unit: chosen when the wide units would leave most CUs idle (sk_plan in dprhot.hip)
constexpr int SK_KC = 64;      // k per ring slot of the sim kernel
constexpr int SK_SLOTS = 4;
constexpr int SK_MAXB = 128;
constexpr int SK_DN = 128;     // d columns of a dC unit
constexpr int SK_QN = 64;      // d columns of a dQ unit
#ifndef SK_QSLOTS_EXP
#define SK_QSLOTS_EXP 3
#endif
constexpr int SK_QSLOTS = SK_QSLOTS_EXP;   // (3) ring slots (64 contexts each) of a dQ unit (72 KiB: two workgroups per CU)
constexpr int SK_MAXG = 16;    // groups of 4 statistics tiles per half row: 128 tiles (Nc <= 16384 at 128 columns per tile; doubling it
                               // for the narrow sim unit costs the loss workgroup of sk_g_kernel 1.5 us: 5.8 -> 7.3 us for the launch)

// bijective XCD-contiguous renumbering: consecutive results run on ONE XCD (workgroup w runs on XCD w % 8)
__device__ __forceinline__ int sk_xcd_order(int wg, int nwg) {
  const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
}

template <int N>
__device__ __forceinline__ void sk_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n * PER vector-memory operations of this wave are outstanding (n in 0..7)
template <int PER>
__device__ __forceinline__ void sk_wait_younger(int n) {
  switch (n) {
    case 0: sk_wait_vm<0>(); break;
    case 1: sk_wait_vm<PER>(); break;
    case 2: sk_wait_vm<2 * PER>(); break;
    case 3: sk_wait_vm<3 * PER>(); break;
    case 4: sk_wait_vm<4 * PER>(); break;
    case 5: sk_wait_vm<5 * PER>(); break;
    case 6: sk_wait_vm<6 * PER>(); break;
    default: sk_wait_vm<7 * PER>(); break;
  }
}
// wait until at most n vector-memory operations of this wave are outstanding (n a constant after unrolling, 0..15)
__device__ __forceinline__ void sk_wait_vm_n(int n) {
  switch (n) {
    case 0: sk_wait_vm<0>(); break;
    case 1: sk_wait_vm<1>(); break;
    case 2: sk_wait_vm<2>(); break;
    case 3: sk_wait_vm<3>(); break;
    case 4: sk_wait_vm<4>(); break;
    case 5: sk_wait_vm<5>(); break;
    case 6: sk_wait_vm<6>(); break;
    case 7: sk_wait_vm<7>(); break;
    case 8: sk_wait_vm<8>(); break;
    case 9: sk_wait_vm<9>(); break;
    case 10: sk_wait_vm<10>(); break;
    case 11: sk_wait_vm<11>(); break;
    case 12: sk_wait_vm<12>(); break;
    case 13: sk_wait_vm<13>(); break;
    case 14: sk_wait_vm<14>(); break;
    default: sk_wait_vm<15>(); break;
  }
}
__device__ __forceinline__ void sk_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// sim + statistics
// ---------------------------------------------------------------------------------------------------------------------
struct SkSimArgs {
  const float* q;        // [B][d] fp32, or nullptr when Qin holds the bf16 rows already
  const uint16_t* Qin;   // [B][d] bf16 (q == nullptr)
  const uint16_t* C;     // [Nc][d] bf16
  uint16_t* Qb;          // bf16 image of q for the backward (q != nullptr), or nullptr
  int B, Nc, d;
  const int64_t* y;      // [B]
  int64_t y_offset;
  const uint8_t* colmask;  // [Nc] or nullptr
  float inv_T;
  float* S;              // [B][Nc]
  float* tile_lse;       // [ceil(nt/4)][B][4]: logsumexp of row i over the columns of statistics tile t at [t >> 2][i][t & 3]
  float* gold;           // [B]
  const uint8_t* packed;   // packed multi-rank layout (EpiSim::mask_at), or nullptr
  int p_rows_c, p_n_ctx, p_row_bytes;
  // packed layout with n_ctx a multiple of the unit width: column tile t = (rank t / tpr, tile t % tpr of that rank's n_ctx real
  // rows) -- the header rows between the ranks' blocks are never multiplied (their logits are -inf by definition), and cfg3 per
  // rank is 4 x 64 = 256 units, one per CU, instead of 260 (4 CUs with two units each set the launch's duration: 9.8 vs 6.9 us).
  int tiles_per_rank;      // 0: tile t = columns [t * COLS, ...)
  // Fused-dScores plan (round 4, sk_bwdf_kernel below): P [B][Nc] bf16 = exp(S - tile_lse) -- every tile's OWN softmax, the gold
  // column stored as 0 (its gradient term is added in fp32 by the backward units) -- instead of the fp32 logits (S == nullptr then).
  // What the backward needs of the row softmax is then one factor per (row, tile), exp(tile_lse - row_lse): no launch in between
  // has to see whole rows.
  uint16_t* P = nullptr;
  unsigned* zero_me = nullptr;  // a counter of the NEXT launch of the step (sk_bwdf_kernel's finishing role): zeroed here, one launch ahead
  float* zero_dq = nullptr;     // round 6 (option sk_dq_atomic): dQ [B][d], zero-filled HERE -- this launch runs first and its stores are idle
  int zero_n4 = 0;              //   at the start -- because the NEXT launch's dQ units ADD their slabs into it (no finishing launch)
};
__device__ __forceinline__ void sk_zero_fill(const SkSimArgs& p, int nthreads) {
  if (p.zero_dq == nullptr) return;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int e = (int)blockIdx.x * nthreads + (int)threadIdx.x; e < p.zero_n4; e += (int)gridDim.x * nthreads) reinterpret_cast<f32x4*>(p.zero_dq)[e] = z;
}

constexpr int SK_ASTAGE = 2 * SK_ROWS * SK_KC;  // elements: two [32 rows][64 k] images of the q rows
inline size_t sk_sim_lds() { return (size_t)SK_ASTAGE * 2 + (size_t)SK_SLOTS * SK_COLS * SK_KC * 2; }  // = 8 slots x 64 columns

// A_F32: q is fp32 (rounded to bf16 right after the loads land; the ct == 0 units also write the bf16 rows to Qb).
// NCH = d / 64 (compile-time: the q rows of the unit live in registers, one 16-byte bf16 chunk per thread and k chunk).
// LDS = 72 KiB -> two workgroups per CU: all units of cfg3 per rank (4 x 64 = 256 with tiles_per_rank, else 260) are resident at once, and one
// workgroup's MFMA / epilogue overlaps the other's loads.
// COLS = columns of a unit (128 with SLOTS = 4 ring slots, or 64 with 8): every wave multiplies COLS / 4 of them.
// NW = 8 (512 threads; 128-column units of the training path): two wave rows of 16 query rows each, the q rows' k chunks dealt to the
// two thread halves, 16 threads per row in the statistics phase -- half the instruction stream per wave where one wave per SIMD had
// nothing to hide its latencies behind.
template <int NCH, bool A_F32, int COLS = SK_COLS, int SLOTS = SK_SLOTS, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void sk_sim_kernel(SkSimArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sk_smem[];
  constexpr int D = NCH * SK_KC;
  constexpr int NT = NW * 64;
  constexpr int WR = NW / 4;  // wave rows (1: a wave multiplies all 32 query rows; 2: 16 each)
  constexpr int NA = 2 / WR;  // 16-row blocks per wave
  constexpr int NQ = NCH / WR;  // k chunks of the q rows a thread holds (chunk i * WR + tid / 256)
  static_assert(NCH % WR == 0, "k chunks per thread half");
  constexpr int IPC = COLS * SK_KC * 2 / 1024 / NW;  // DMA instructions per wave and ring slot
  constexpr int CPW = COLS / 4, NB = CPW / 16;       // columns per wave, 16-column MFMA blocks per wave
  uint16_t* const Ast = sk_smem;                      // 2 x [32][64]: 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
  uint16_t* const ring = sk_smem + SK_ASTAGE;         // SLOTS x [COLS n][64 k], same swizzle
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave & 3, wr = wave >> 2, kh = tid >> 8;
  const int nrb = (p.B + SK_ROWS - 1) / SK_ROWS;
  const int unit = sk_xcd_order(blockIdx.x, gridDim.x);
  const int rb = unit % nrb, ct = unit / nrb;
  const int m0 = rb * SK_ROWS;
  const int n0 = p.tiles_per_rank > 0 ? (ct / p.tiles_per_rank) * p.p_rows_c + (ct % p.tiles_per_rank) * COLS : ct * COLS;
  if (p.zero_me != nullptr && blockIdx.x == 0 && tid == 0) *p.zero_me = 0u;
  sk_zero_fill(p, NT);
  DPRHOT_TMB(0, 0);

  // ---- every global read of the unit's first phase, back to back: the q rows (registers: thread t holds the 8 values
  //      k = kc * 64 + (t & 7) * 8 .. of row t >> 3 for every k chunk kc), the first ring slots, mask bytes, labels
  const int arow = (tid & 255) >> 3, ac8 = tid & 7;
  uint4 areg[NQ][A_F32 ? 2 : 1];
  {
    const int gr = min(m0 + arow, p.B - 1);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int kc = i * WR + kh;
      if constexpr (A_F32) {
        const float* src = p.q + (size_t)gr * D + kc * SK_KC + ac8 * 8;
        areg[i][0] = *reinterpret_cast<const uint4*>(src);
        areg[i][1] = *reinterpret_cast<const uint4*>(src + 4);
      } else {
        areg[i][0] = *reinterpret_cast<const uint4*>(p.Qin + (size_t)gr * D + kc * SK_KC + ac8 * 8);
      }
    }
  }
  const int i16 = lane & 15, g4 = lane >> 4;
  constexpr int TPR = NT / SK_ROWS;            // statistics / store phase: threads per row (8 | 16)
  const int srow = tid / TPR, sseg = tid % TPR;

  unsigned cof[IPC];  // element offset of this lane's source chunk inside a k chunk, per DMA instruction (8 rows of 128 bytes)
#pragma unroll
  for (int j = 0; j < IPC; ++j) {
    const int row = (wave * IPC + j) * 8 + (lane >> 3);
    cof[j] = (unsigned)min(n0 + row, p.Nc - 1) * (unsigned)D + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
  auto issue = [&](int kc, int slot) {
    uint16_t* dst = ring + slot * (COLS * SK_KC);
#pragma unroll
    for (int j = 0; j < IPC; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.C + cof[j] + kc * SK_KC), (g2_lds_ptr*)(dst + (wave * IPC + j) * 512), 16, 0, 0);
  };
  // SLOTS - 1 chunks here; chunk c + SLOTS - 1 is issued right behind chunk c's barrier, into the slot chunk c - 1 was read from
  // (at that barrier every wave has finished reading it): one barrier per chunk instead of two
#pragma unroll
  for (int c = 0; c < SLOTS - 1; ++c)
    if (c < NCH) issue(c, c);
  // mask bytes and labels: raw loads only (a select on a loaded value here would park an s_waitcnt in front of everything
  // that follows); they are turned into what the epilogue needs there
  // (branch-free: one unconditional byte load per column from a valid address)
  const bool have_mask = p.packed != nullptr || p.colmask != nullptr;
  const uint8_t* const mbase = p.packed != nullptr ? p.packed : (p.colmask != nullptr ? p.colmask : reinterpret_cast<const uint8_t*>(p.y));
  uint8_t mraw[NB];
  bool mover[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = min(n0 + wc * CPW + nb * 16 + i16, p.Nc - 1);
    const int rc = p.packed != nullptr ? p.p_rows_c : 1;
    const int r = n / rc, j = n - r * rc;
    mover[nb] = p.packed != nullptr && j >= p.p_n_ctx;  // mask / padding rows of the packed buffer: always masked
    const size_t off = p.packed != nullptr ? (size_t)(r * rc + p.p_n_ctx) * p.p_row_bytes + min(j, p.p_n_ctx - 1)
                                           : (p.colmask != nullptr ? (size_t)n : (size_t)0);
    mraw[nb] = mbase[off];
  }
  // labels are column indices (< 2^31): the low word of the int64 is all that is needed
  const int yraw = reinterpret_cast<const int*>(p.y)[2 * min(m0 + srow, p.B - 1)];

  DPRHOT_TMB(0, 1);
  // ---- q rows -> bf16, kept in registers (and written to Qb); one k chunk at a time goes to LDS inside the K loop
  uint4 abf[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    if constexpr (A_F32) {
      const uint4 a = areg[i][0], b = areg[i][1];
      abf[i] = make_uint4(pack_bf16_rne(a.x, a.y), pack_bf16_rne(a.z, a.w), pack_bf16_rne(b.x, b.y), pack_bf16_rne(b.z, b.w));
      if (p.Qb != nullptr && ct == 0 && m0 + arow < p.B)
        *reinterpret_cast<uint4*>(p.Qb + (size_t)(m0 + arow) * D + (i * WR + kh) * SK_KC + ac8 * 8) = abf[i];
    } else {
      abf[i] = areg[i][0];
    }
  }
  DPRHOT_TMB(0, 2);
  f32x4 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    uint16_t* const Ac = Ast + (c & 1) * (SK_ROWS * SK_KC);
    if (WR == 1 || (c % WR) == kh) {  // (the thread half that holds this k chunk)
      // asm store: hipcc orders a plain LDS store behind every LDS-DMA in flight with s_waitcnt vmcnt(0) (it cannot tell the
      // q image from the ring slots), which would drain the ring at every chunk; sk_barrier() below waits lgkmcnt(0)
      typedef __attribute__((address_space(3))) uint16_t lds_u16;
      const unsigned addr = (unsigned)(uintptr_t)(lds_u16*)(Ac + arow * SK_KC + ((ac8 ^ ((arow >> 1) & 7)) << 3));
      typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
      const sk_u32x4 val = {abf[c / WR].x, abf[c / WR].y, abf[c / WR].z, abf[c / WR].w};
      asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" ::"v"(addr), "v"(val) : "memory");
    }
    sk_wait_younger<IPC>(min(c + SLOTS - 2, NCH - 1) - c);  // this wave's share of chunk c has landed
    sk_barrier();                                                // ... everybody's has; the q chunk is in place
    if (c + SLOTS - 1 < NCH) issue(c + SLOTS - 1, (c + SLOTS - 1) % SLOTS);
    const uint16_t* Bs = ring + (c % SLOTS) * (COLS * SK_KC);
#pragma unroll
    for (int kk = 0; kk < SK_KC / 32; ++kk) {
      bf16x8 af[NA], bf[NB];
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int row = (wr * NA + a) * 16 + i16;
        af[a] = *reinterpret_cast<const bf16x8*>(Ac + row * SK_KC + (((kk * 4 + g4) ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int row = wc * CPW + b * 16 + i16;
        bf[b] = *reinterpret_cast<const bf16x8*>(Bs + row * SK_KC + (((kk * 4 + g4) ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
  sk_barrier();  // the ring is free: its start becomes the fp32 logit tile [32][COLS + 4]
  DPRHOT_TMB(0, 3);

  // ---- epilogue: mask, 1/T -> LDS tile -> per-row tile logsumexp, gold logit, coalesced fp32 store
  constexpr int TS = COLS + 4;
  constexpr int QD = COLS / 4 / TPR;  // float4 runs per thread in the statistics / store phase
  float* const T = reinterpret_cast<float*>(ring);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = wc * CPW + b * 16 + i16;
    const bool masked = (n0 + col >= p.Nc) || mover[b] || (have_mask && mraw[b] != 0);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[((wr * NA + a) * 16 + g4 * 4 + r) * TS + col] = masked ? -INFINITY : acc[a][b][r] * p.inv_T;
  }
  sk_barrier();
  {
    const int row = m0 + srow;
    const int yi = yraw + (int)p.y_offset - n0;  // gold column relative to the tile
    float4 v[QD];
    float mx = -INFINITY;
#pragma unroll
    for (int qd = 0; qd < QD; ++qd) {
      v[qd] = *reinterpret_cast<const float4*>(T + srow * TS + (qd * TPR + sseg) * 4);
      mx = fmaxf(mx, fmaxf(fmaxf(v[qd].x, v[qd].y), fmaxf(v[qd].z, v[qd].w)));
    }
    mx = ss_max8(mx);
    if constexpr (TPR == 16) mx = fmaxf(mx, ss_dpp<0x140>(mx));  // row_mirror: the other eight lanes of the row
    float sm = 0.f;
    float4 e[QD];  // exp(v - max): summed here, stored (scaled) as the tile softmax below -- one exponential per score
    if (mx != -INFINITY) {
#pragma unroll
      for (int qd = 0; qd < QD; ++qd) {
        e[qd] = make_float4(__expf(v[qd].x - mx), __expf(v[qd].y - mx), __expf(v[qd].z - mx), __expf(v[qd].w - mx));
        sm += (e[qd].x + e[qd].y) + (e[qd].z + e[qd].w);
      }
    } else {
#pragma unroll
      for (int qd = 0; qd < QD; ++qd) e[qd] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    sm = ss_sum8(sm);
    if constexpr (TPR == 16) sm += ss_dpp<0x140>(sm);
    if (row < p.B) {
      if (sseg == 0) p.tile_lse[((size_t)(ct >> 2) * p.B + row) * 4 + (ct & 3)] = mx == -INFINITY ? -INFINITY : mx + logf(sm);
      const float inv = sm > 0.f ? 1.0f / sm : 0.f;  // (a tile whose columns are all masked: every probability is 0)
#pragma unroll
      for (int qd = 0; qd < QD; ++qd) {
        const int col = (qd * TPR + sseg) * 4;
        if (p.S != nullptr && n0 + col < p.Nc) *reinterpret_cast<float4*>(p.S + (size_t)row * p.Nc + n0 + col) = v[qd];
        if (yi >= col && yi < col + 4) p.gold[row] = yi == col ? v[qd].x : (yi == col + 1 ? v[qd].y : (yi == col + 2 ? v[qd].z : v[qd].w));
        if (p.P != nullptr && n0 + col < p.Nc) {
          // exp(-inf - mx) == 0 at masked columns; the gold column leaves as 0 (sk_bwdf_kernel adds its term in fp32)
          const float e0 = yi == col ? 0.f : e[qd].x * inv, e1 = yi == col + 1 ? 0.f : e[qd].y * inv;
          const float e2 = yi == col + 2 ? 0.f : e[qd].z * inv, e3 = yi == col + 3 ? 0.f : e[qd].w * inv;
          *reinterpret_cast<uint2*>(p.P + (size_t)row * p.Nc + n0 + col) = make_uint2(pk_bf16(e0, e1), pk_bf16(e2, e3));
        }
      }
    }
  }
  if (p.tiles_per_rank > 0 && ct % p.tiles_per_rank == p.tiles_per_rank - 1) {
    // the header rows behind this rank's real rows: masked columns of the logit matrix
    const int hdr = p.p_rows_c - p.p_n_ctx, h0 = (ct / p.tiles_per_rank) * p.p_rows_c + p.p_n_ctx;
    for (int i = tid; i < SK_ROWS * hdr; i += NT) {
      const int row = m0 + i / hdr;
      if (row < p.B) {
        if (p.S != nullptr) p.S[(size_t)row * p.Nc + h0 + i % hdr] = -INFINITY;
        if (p.P != nullptr) p.P[(size_t)row * p.Nc + h0 + i % hdr] = 0;
      }
    }
  }
  DPRHOT_TMB(0, 4);
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same sim unit with PRIVATE rings -- no barrier inside the K loop.
// scratch/ingest2 (profiles/r05_ingest2.txt) settled what the K loop of sk_sim_kernel is bound by: NOT the texture path's
// acceptance of LDS-DMA pieces -- the same 196 KiB C tile, the same 1-KiB pieces, the same one-barrier-per-chunk ring arrive
// in 1.7 us (48 B/clk per CU) when nothing is multiplied, against 3.5 us for the kernel's twelve chunks -- and not something
// register streaming would fix (MFMA-fragment-shaped global_load_dwordx4, 16 rows x 64 B per instruction: 20 B/clk; whole
// 128-byte lines to registers: 64 B/clk, but those are not fragments).  What is left is the loop's own chain: every chunk is
// wait -> workgroup barrier -> six fragment reads -> four MFMAs, eight waves in lock step, ~700 clocks per 16-KiB chunk with one
// workgroup per CU and nothing to run beside it.  A C row is used by exactly one 16-column MFMA block, so here every wave owns
// its 16 columns outright: it DMAs them (2 KiB per k chunk, two pieces) into its own ring, waits on its own vmcnt, reads its own
// fragments -- no other wave ever touches them, so no barrier orders them.  The q block (32 rows x d, rounded to bf16) is shared by
// all eight waves: it is written to LDS ONCE, whole, in the prologue (d = 768: 48 KiB) behind the only barrier of the loop.  Waves
// drift apart freely; two of them per SIMD cover each other's waits.  LDS = d / 64 x 4 KiB + 8 x SLOTS x 2 KiB (768: 144 KiB with
// six chunks in flight per wave): one workgroup per CU, so the launch takes this kernel only where the grid fits the chip once
// (cfg3 per rank: 256 units).  Outputs are those of sk_sim_kernel<NCH, true, 128, 4, 8> bit for bit (same k order per element).
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned sk_u32x4p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sk_ldsr128(sk_u32x4p& r, const void* lds_ptr) {  // asm: a plain ds_read next to LDS-DMAs in flight gets vmcnt(0) from hipcc
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  const unsigned addr = (unsigned)(uintptr_t)(lds_u32*)const_cast<void*>(lds_ptr);
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
}
__device__ __forceinline__ bf16x8 sk_as_frag(const sk_u32x4p& r) {
  union { sk_u32x4p u; bf16x8 f; } c;
  c.u = r;
  return c.f;
}
__device__ __forceinline__ f32x4 sk_ld16_early(const void* ptr) {  // asm loads: invisible to hipcc's wait counters, waited for by hand
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
__device__ __forceinline__ unsigned sk_ld4u_early(const void* ptr) {
  unsigned r;
  asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
__device__ __forceinline__ unsigned sk_ld1_early(const void* ptr) {
  unsigned r;
  asm volatile("global_load_ubyte %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
__device__ __forceinline__ void sk_lds_w128(void* lds_ptr, uint4 v) {
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  const unsigned addr = (unsigned)(uintptr_t)(lds_u32*)lds_ptr;
  const sk_u32x4p val = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" ::"v"(addr), "v"(val) : "memory");
}
__device__ __forceinline__ void sk_lds_w64(void* lds_ptr, uint2 v) {
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  typedef unsigned sk_u32x2p __attribute__((ext_vector_type(2)));
  const unsigned addr = (unsigned)(uintptr_t)(lds_u32*)lds_ptr;
  const sk_u32x2p val = {v.x, v.y};
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(val) : "memory");
}
inline size_t sk_simp_lds(int nch, int slots) { return (size_t)nch * SK_ROWS * SK_KC * 2 + (size_t)8 * slots * 16 * SK_KC * 2; }

template <int NCH, int SLOTS>
__global__ __launch_bounds__(512, 2) void sk_simp_kernel(SkSimArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sk_smem[];
  constexpr int D = NCH * SK_KC, NT = 512, COLS = SK_COLS;
  constexpr int QIMG = SK_ROWS * SK_KC;  // elements of one k chunk of the q block  [32 rows][64 k], 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
  constexpr int WSLOT = 16 * SK_KC;      // elements of one ring slot of a wave     [16 n][64 k], same swizzle
  constexpr int NQ = NCH / 2;            // k chunks of the q rows a thread holds (chunk 2 i + tid / 256)
  constexpr int NS = SLOTS < NCH ? SLOTS : NCH;
  static_assert(NCH % 2 == 0, "k chunks per thread half");
  uint16_t* const Qs = sk_smem;
  uint16_t* const ring = sk_smem + NCH * QIMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = tid >> 8;
  const int nrb = (p.B + SK_ROWS - 1) / SK_ROWS;
  const int unit = sk_xcd_order(blockIdx.x, gridDim.x);
  const int rb = unit % nrb, ct = unit / nrb;
  const int m0 = rb * SK_ROWS;
  const int n0 = p.tiles_per_rank > 0 ? (ct / p.tiles_per_rank) * p.p_rows_c + (ct % p.tiles_per_rank) * COLS : ct * COLS;
  if (p.zero_me != nullptr && blockIdx.x == 0 && tid == 0) *p.zero_me = 0u;
  sk_zero_fill(p, NT);
  DPRHOT_TMB(0, 0);

  // ---- every global read of the prologue back to back: the q rows (fp32; OLDEST, so that a counted wait releases them while the
  //      chunks behind them are still landing), the first NS chunks of this wave's columns, mask byte and label.  The q rows, the
  //      mask byte and the label are asm loads counted by hand: hipcc answers the first use of an ordinary load next to an LDS-DMA
  //      with s_waitcnt vmcnt(0), which held the whole prologue (q AND six chunks, 194 KiB per CU) in front of the first MFMA.
  // One load instruction = 8 rows x 128 contiguous bytes (whole lines): lane l takes floats [h * 32 + (l & 7) * 4, + 4) of k chunk kc of
  // row l >> 3, h = 0, 1.  (sk_sim_kernel's form -- 8 consecutive floats per lane as two loads 16 bytes apart -- makes every instruction
  // touch half of each of 16 lines: the access shape scratch/ingest2 measured at 20 B/clk per CU against 64 for whole lines, and the
  // q block is a third of this unit's bytes.)
  const int arow = (tid & 255) >> 3, ac8 = tid & 7;
  f32x4 areg[NQ][2];
  {
    const int gr = min(m0 + arow, p.B - 1);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const float* src = p.q + (size_t)gr * D + (i * 2 + kh) * SK_KC + ac8 * 4;
      areg[i][0] = sk_ld16_early(src);
      areg[i][1] = sk_ld16_early(src + 32);
    }
  }
  const int i16 = lane & 15, g4 = lane >> 4;
  unsigned cof[2];  // element offset of this lane's source chunk inside a k chunk, per DMA instruction (8 rows of 128 bytes)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wave * 16 + j * 8 + (lane >> 3);
    cof[j] = (unsigned)min(n0 + row, p.Nc - 1) * (unsigned)D + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
  uint16_t* const myring = ring + wave * (SLOTS * WSLOT);
  auto issue = [&](int kc, int slot) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.C + cof[j] + kc * SK_KC), (g2_lds_ptr*)(myring + slot * WSLOT + j * 512), 16, 0, 0);
  };
#pragma unroll
  for (int c = 0; c < NS; ++c) issue(c, c);
  const bool have_mask = p.packed != nullptr || p.colmask != nullptr;
  const uint8_t* const mbase = p.packed != nullptr ? p.packed : (p.colmask != nullptr ? p.colmask : reinterpret_cast<const uint8_t*>(p.y));
  unsigned mraw;
  bool mover;
  {
    const int n = min(n0 + wave * 16 + i16, p.Nc - 1);
    const int rc = p.packed != nullptr ? p.p_rows_c : 1;
    const int r = n / rc, j = n - r * rc;
    mover = p.packed != nullptr && j >= p.p_n_ctx;
    const size_t off = p.packed != nullptr ? (size_t)(r * rc + p.p_n_ctx) * p.p_row_bytes + min(j, p.p_n_ctx - 1)
                                           : (p.colmask != nullptr ? (size_t)n : (size_t)0);
    mraw = sk_ld1_early(mbase + off);
  }
  constexpr int TPR = NT / SK_ROWS;  // 16 threads per row in the statistics phase
  const int srow = tid / TPR, sseg = tid % TPR;
  unsigned yraw = sk_ld4u_early(reinterpret_cast<const int*>(p.y) + 2 * min(m0 + srow, p.B - 1));
  constexpr int NEARLY = 2;  // the two loads behind the DMAs
  DPRHOT_TMB(0, 1);

  // ---- q rows -> bf16 -> the whole q block into LDS (and to Qb for the backward), one barrier
  sk_wait_vm<2 * NS + NEARLY>();  // the q rows have landed (in-order return); the chunks stay in flight
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(areg[i][0]), "+v"(areg[i][1]));
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int kc = i * 2 + kh;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 a = areg[i][h];
      const uint2 v = make_uint2(pack_bf16_rne(__float_as_uint(a[0]), __float_as_uint(a[1])), pack_bf16_rne(__float_as_uint(a[2]), __float_as_uint(a[3])));
      // elements k = h * 32 + ac8 * 4 .. + 4 of the row: half (ac8 & 1) of 16-byte chunk h * 4 + (ac8 >> 1)
      const int ch = h * 4 + (ac8 >> 1);
      if (p.Qb != nullptr && ct == 0 && m0 + arow < p.B) *reinterpret_cast<uint2*>(p.Qb + (size_t)(m0 + arow) * D + kc * SK_KC + h * 32 + ac8 * 4) = v;
      sk_lds_w64(Qs + kc * QIMG + arow * SK_KC + ((ch ^ ((arow >> 1) & 7)) << 3) + (ac8 & 1) * 4, v);  // (asm: a plain LDS store next to the DMAs would drain them)
    }
  }
  sk_barrier();  // lgkmcnt(0) + raw s_barrier: the q block is in place, nobody's chunks were waited for
  DPRHOT_TMB(0, 2);

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    // this wave's chunk c has landed: what may still be outstanding are the chunks behind it -- prologue chunks c + 1 .. NS - 1 and the two
    // early loads (c < NS), refills NS .. min(c - 1 + SLOTS, NCH - 1) issued by iterations 0 .. c - 1 -- two operations per chunk.  (The Qb
    // stores of the ct == 0 units sit between the prologue and the refills: not counted, so those units wait for a little more.)
    {
      const int last = (c - 1 + SLOTS < NCH - 1) ? c - 1 + SLOTS : NCH - 1;  // last chunk issued so far (c >= 1), or NS - 1
      const int issued_last = c == 0 ? NS - 1 : (last > NS - 1 ? last : NS - 1);
      sk_wait_vm_n(2 * (issued_last - c) + (c < NS ? NEARLY : 0));  // (a constant after unrolling)
    }
    const uint16_t* const Bs = myring + (c % SLOTS) * WSLOT;
    const uint16_t* const Ac = Qs + c * QIMG;
    sk_u32x4p bq[2], aq[2][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      sk_ldsr128(bq[kk], Bs + i16 * SK_KC + (((kk * 4 + g4) ^ ((i16 >> 1) & 7)) << 3));
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = a * 16 + i16;
        sk_ldsr128(aq[kk][a], Ac + row * SK_KC + (((kk * 4 + g4) ^ ((row >> 1) & 7)) << 3));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (c + SLOTS < NCH) issue(c + SLOTS, c % SLOTS);  // the slot just read is this wave's alone: refill it at once
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sk_as_frag(aq[kk][a]), sk_as_frag(bq[kk]), acc[a], 0, 0, 0);
  }
  sk_wait_vm<0>();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" : "+v"(mraw), "+v"(yraw));
  __syncthreads();  // every wave is done with its ring: the ring's start becomes the fp32 logit tile [32][COLS + 4]
  DPRHOT_TMB(0, 3);

  // ---- epilogue: mask, 1/T -> LDS tile -> per-row tile logsumexp, gold logit, tile softmax / logits out in 16-byte stores
  constexpr int TS = COLS + 4;
  float* const T = reinterpret_cast<float*>(ring);
  {
    const int col = wave * 16 + i16;
    const bool masked = (n0 + col >= p.Nc) || mover || (have_mask && mraw != 0);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(a * 16 + g4 * 4 + r) * TS + col] = masked ? -INFINITY : acc[a][r] * p.inv_T;
  }
  __syncthreads();
  {
    const int row = m0 + srow;
    const int yi = (int)yraw + (int)p.y_offset - n0;  // gold column relative to the tile
    const int col = sseg * 8;                    // this thread's eight consecutive columns
    const float4 v0 = *reinterpret_cast<const float4*>(T + srow * TS + col), v1 = *reinterpret_cast<const float4*>(T + srow * TS + col + 4);
    float mx = fmaxf(fmaxf(fmaxf(v0.x, v0.y), fmaxf(v0.z, v0.w)), fmaxf(fmaxf(v1.x, v1.y), fmaxf(v1.z, v1.w)));
    mx = ss_max8(mx);
    mx = fmaxf(mx, ss_dpp<0x140>(mx));  // row_mirror: the other eight lanes of the row
    float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float sm = 0.f;
    if (mx != -INFINITY) {
      e[0] = __expf(v0.x - mx); e[1] = __expf(v0.y - mx); e[2] = __expf(v0.z - mx); e[3] = __expf(v0.w - mx);
      e[4] = __expf(v1.x - mx); e[5] = __expf(v1.y - mx); e[6] = __expf(v1.z - mx); e[7] = __expf(v1.w - mx);
      // (the order of sk_sim_kernel's sums: the two float4 of a thread there are columns sseg * 4 and 64 + sseg * 4 -- a different
      //  association of the same 128 terms; tile_lse may differ from that kernel's in the last bit)
      sm = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
    }
    sm = ss_sum8(sm);
    sm += ss_dpp<0x140>(sm);
    if (row < p.B) {
      if (sseg == 0) p.tile_lse[((size_t)(ct >> 2) * p.B + row) * 4 + (ct & 3)] = mx == -INFINITY ? -INFINITY : mx + logf(sm);
      const float inv = sm > 0.f ? 1.0f / sm : 0.f;
      if (yi >= col && yi < col + 8) {
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        float gsel = vv[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) gsel = yi == col + u ? vv[u] : gsel;
        p.gold[row] = gsel;
      }
      if (n0 + col < p.Nc) {  // (Nc % 8 == 0 and n0 % 8 == 0: the eight columns are in or out together)
        if (p.S != nullptr) {
          *reinterpret_cast<float4*>(p.S + (size_t)row * p.Nc + n0 + col) = v0;
          *reinterpret_cast<float4*>(p.S + (size_t)row * p.Nc + n0 + col + 4) = v1;
        }
        if (p.P != nullptr) {
          float pe[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) pe[u] = yi == col + u ? 0.f : e[u] * inv;  // the gold column leaves as 0 (its term is added in fp32 by the backward)
          *reinterpret_cast<uint4*>(p.P + (size_t)row * p.Nc + n0 + col) =
              make_uint4(pk_bf16(pe[0], pe[1]), pk_bf16(pe[2], pe[3]), pk_bf16(pe[4], pe[5]), pk_bf16(pe[6], pe[7]));
        }
      }
    }
  }
  if (p.tiles_per_rank > 0 && ct % p.tiles_per_rank == p.tiles_per_rank - 1) {
    const int hdr = p.p_rows_c - p.p_n_ctx, h0 = (ct / p.tiles_per_rank) * p.p_rows_c + p.p_n_ctx;
    for (int i = tid; i < SK_ROWS * hdr; i += NT) {
      const int row = m0 + i / hdr;
      if (row < p.B) {
        if (p.S != nullptr) p.S[(size_t)row * p.Nc + h0 + i % hdr] = -INFINITY;
        if (p.P != nullptr) p.P[(size_t)row * p.Nc + h0 + i % hdr] = 0;
      }
    }
  }
  DPRHOT_TMB(0, 4);
}

// ---------------------------------------------------------------------------------------------------------------------
// row logsumexp, loss, G = (softmax - onehot) * scale   (dpr_task.py:212 and its backward into the scores)
// ---------------------------------------------------------------------------------------------------------------------
struct SkGArgs {
  const float* S;         // [B][Nc]
  const float* tile_lse;  // [ceil(nt/4)][B][4]
  const float* gold;      // [B]
  int nt;
  int B, Nc;
  const int64_t* y;
  int64_t y_offset;
  float grad_scale;
  uint16_t* G;            // [B][Nc] bf16
  float* row_loss;        // [B] (required: the backward launch forms the loss from it)
  float* row_lse;         // optional
  int parts;              // workgroups per row
};

// logsumexp of one row from its tile values: 64 lanes of a wave, lane l takes groups l, l + 64 (nt <= 512)
// (tv: the lane's masked tile values, tiles 4 (lane + 64 u) .. + 3 -- callers that need more than the logsumexp of them)
__device__ __forceinline__ float sk_row_lse_tv(const float* tile_lse, int nt, int B, int row, int lane, float4 (&tv)[2]) {
  const int ng = (nt + 3) >> 2;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int gq = lane + 64 * u;
    tv[u] = *reinterpret_cast<const float4*>(tile_lse + ((size_t)(gq < ng ? gq : 0) * B + row) * 4);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = (lane + 64 * u) * 4;
    tv[u].x = t + 0 < nt ? tv[u].x : -INFINITY;
    tv[u].y = t + 1 < nt ? tv[u].y : -INFINITY;
    tv[u].z = t + 2 < nt ? tv[u].z : -INFINITY;
    tv[u].w = t + 3 < nt ? tv[u].w : -INFINITY;
    mx = fmaxf(mx, fmaxf(fmaxf(tv[u].x, tv[u].y), fmaxf(tv[u].z, tv[u].w)));
  }
  mx = wave_max(mx);
  float sm = 0.f;
  if (mx != -INFINITY) {
#pragma unroll
    for (int u = 0; u < 2; ++u) sm += __expf(tv[u].x - mx) + __expf(tv[u].y - mx) + __expf(tv[u].z - mx) + __expf(tv[u].w - mx);
  }
  sm = wave_sum(sm);
  return mx + logf(sm);  // every lane holds the same bits (butterfly reductions)
}
__device__ __forceinline__ float sk_row_lse(const float* tile_lse, int nt, int B, int row, int lane) {
  float4 tv[2];
  return sk_row_lse_tv(tile_lse, nt, B, row, lane, tv);
}

// grid = B * parts workgroups: workgroup (row, part) turns its share of the row's logits into G.  Every wave derives the row
// logsumexp itself (<= 2 KB of tile values); part 0 publishes the row's logsumexp and loss.  The loss SUM over the rows is formed
// by the backward launch (sk_bwd_kernel: 512 bytes of row losses, fixed order) -- the extra workgroup that used to fold all
// B x nt tile values a second time was this launch's tail.
__global__ __launch_bounds__(SK_THREADS) void sk_g_kernel(SkGArgs p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int row = blockIdx.x / p.parts, part = blockIdx.x - row * p.parts;
  const int cpr = p.Nc >> 3;                       // 8-column chunks per row
  const int per = (cpr + p.parts - 1) / p.parts;   // chunks of this part
  const int c_lo = part * per, c_hi = min(cpr, c_lo + per);
  constexpr int CPT = 4;                           // chunks per thread kept in flight (per <= 1024)
  float4 va[CPT], vb[CPT];
  const float* Srow = p.S + (size_t)row * p.Nc;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int ch = c_lo + tid + k * SK_THREADS;
    const float* src = Srow + (size_t)(ch < c_hi ? ch : c_lo) * 8;
    va[k] = *reinterpret_cast<const float4*>(src);
    vb[k] = *reinterpret_cast<const float4*>(src + 4);
  }
  const int yi = reinterpret_cast<const int*>(p.y)[2 * row] + (int)p.y_offset;
  const float gold = p.gold[row];
  const float lse = sk_row_lse(p.tile_lse, p.nt, p.B, row, lane);
  if (part == 0 && tid == 0) {
    p.row_loss[row] = lse - gold;  // (always a real buffer: the caller's or the workspace's -- the backward launch sums it)
    if (p.row_lse) p.row_lse[row] = lse;
  }
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int ch = c_lo + tid + k * SK_THREADS;
    if (ch < c_hi) {
      const int j = ch * 8;
      const float v[8] = {va[k].x, va[k].y, va[k].z, va[k].w, vb[k].x, vb[k].y, vb[k].z, vb[k].w};
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float pr = __expf(v[e] - lse);  // exp(-inf) == 0 at masked columns
        if (j + e == yi) pr -= 1.0f;
        g[e] = pr * p.grad_scale;
      }
      *reinterpret_cast<uint4*>(p.G + (size_t)row * p.Nc + j) = make_uint4(pk_bf16(g[0], g[1]), pk_bf16(g[2], g[3]), pk_bf16(g[4], g[5]), pk_bf16(g[6], g[7]));
    }
  }
}

// loss numerator = sum of the B <= 128 row losses in a fixed order (every wave that needs it computes the same bits)
__device__ __forceinline__ float sk_loss_sum(const float* row_loss, int B, int lane) {
  const float a = (lane < B ? row_loss[lane] : 0.f) + (lane + 64 < B ? row_loss[lane + 64] : 0.f);
  return wave_sum(a);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward GEMMs, one launch: [dQ partial-sum units | dC units]
// ---------------------------------------------------------------------------------------------------------------------
struct SkBwdArgs {
  const uint16_t* G;      // [B][Nc] bf16
  const uint16_t* Qb;     // [B][d]
  const uint16_t* C;      // [Nc][d]
  int B, Nc, d;
  float h_scale;
  const float* d_scale;
  void* dC;               // [Nc][d] fp32, or bf16 when dc_bf16 (the wire format of the reduce-scatter, written here)
  const float* row_loss;  // [B] (sk_g_kernel), or nullptr: backward only -- the forward call formed the loss, no stamp
  float* loss_sum;        // [1] out: loss_scale * sum of the row losses (written by dC unit 0; row_loss != nullptr)
  float loss_scale;
  int dc_bf16;
  int stamp_period, stamp_row;  // > 0 (fp32 dC only): dC[m][0] = loss_sum where m % stamp_period == stamp_row (EpiScaleF32)
  int ksteps;             // 64-context steps per dQ slice
  int nslices;
  float* part;            // [nslices][B][d] dQ partial sums (nslices > 1)
  float* dQ;              // [B][d]: nslices == 1 -- the one slice IS the sum, scaled and stored here
  int ndq_pad;
  int nt_store = 0;       // 1: dC (and an unsplit dQ) leave with non-temporal stores: nobody re-reads them inside the step, and written normally
                          // they displace the operands in L2 / the Infinity Cache (cfg3 per rank 31.1 -> 29.9 us, router width 116 -> 99.5)            // dQ units rounded up to a multiple of 8 (keeps workgroup % 8 == XCD for the dC units)
};

constexpr int SK_DC_TS = SK_DN + 4;  // fp32 output tile row stride in LDS
constexpr int SK_QA = SK_MAXB * 64;  // elements of the G part of a dQ slot  [128 rows][64 k]
constexpr int SK_QB = 64 * SK_QN;    // elements of the C part of a dQ slot  [64 k][64 n]
inline size_t sk_bwd_lds() {
  const size_t dc = (size_t)SK_COLS * SK_DC_TS * sizeof(float);  // dC: epilogue tile (>= G image + Q image)
  const size_t dq = (size_t)SK_QSLOTS * (SK_QA + SK_QB) * 2;      // ring; its epilogue tile [128][SK_QN + 4] fp32 is the dC one's size
  const size_t dqe = (size_t)SK_MAXB * (SK_QN + 4) * sizeof(float);
  const size_t m = dc > dq ? dc : dq;
  return m > dqe ? m : dqe;
}

// dC unit = 128 contexts x 128 columns of d: dC[ct*128.., dt*128..] = G^T x Q over the B rows (B % 32 == 0).
// Both operands are "mn-major" (contraction index = row of the matrix in HBM): DMA with the source-side swizzle of
// gemm_bf16.h (32-byte group cg of row k at cg ^ mswz(k)), fragments through ds_read_b64_tr_b16.
__device__ __forceinline__ void sk_dc_unit(const SkBwdArgs& p, int unit, uint16_t* sk_smem) {
  constexpr size_t kImg = (size_t)SK_MAXB * SK_COLS;  // elements of one image
  uint16_t* const Gs = sk_smem;                 // [128 k = query row][128 m = context]
  uint16_t* const Qs = sk_smem + kImg;          // [128 k = query row][128 n = d column]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ndt = (p.d + SK_DN - 1) / SK_DN;  // d % 64 == 0: the last tile may be half a tile (router width 30528 = 238.5 x 128)
  // consecutive units run on one XCD (sk_xcd_order): let them share the tile of the LARGER operand -- the G tile (walk the d tiles)
  // when there are more contexts than vector components, the Q tile (walk the context tiles) at router width, where Q is 7.8 MB and
  // every tile of it was fetched once per context tile (PMC: 128 MB fetched per launch for 70 MB of operands)
  const int nctile = (p.Nc + SK_COLS - 1) / SK_COLS;
  const bool ct_fast = p.Nc < p.d;
  const int dt = ct_fast ? unit / nctile : unit % ndt, ct = ct_fast ? unit % nctile : unit / ndt;
  const int n0 = ct * SK_COLS, c0 = dt * SK_DN;
  const int kmax = p.B;  // contraction length
  DPRHOT_TMB(1, 0);
  {
    // one DMA instruction = 4 rows of 256 bytes; lane l: row + (l >> 4), position l & 15
    const int nins = kmax / 4 / 4;  // per wave and image
    for (int j = 0; j < nins; ++j) {
      const int krow = (wave * nins + j) * 4 + (lane >> 4), pos = lane & 15;
      const int col = ((((pos >> 1) ^ mswz(krow))) << 4) + (pos & 1) * 8;
      // contexts beyond Nc (ragged last tile) only feed output rows that are never stored: any valid address will do
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.G + (size_t)krow * p.Nc + min(n0 + col, p.Nc - 8)),
                                       (g2_lds_ptr*)(Gs + (wave * nins + j) * 4 * SK_COLS), 16, 0, 0);
      // (columns beyond d in a ragged last tile feed output columns that are never stored)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.Qb + (size_t)krow * p.d + min(c0 + col, p.d - 8)),
                                       (g2_lds_ptr*)(Qs + (wave * nins + j) * 4 * SK_DN), 16, 0, 0);
    }
  }
  const float sc = p.h_scale * (p.d_scale ? *p.d_scale : 1.0f);
  const bool stamp = dt == 0 && p.stamp_period > 0 && !p.dc_bf16 && p.row_loss != nullptr;
  float lsum = 0.f;
  if ((stamp || unit == 0) && p.row_loss != nullptr) {
    lsum = sk_loss_sum(p.row_loss, p.B, lane) * p.loss_scale;
    if (unit == 0 && tid == 0) p.loss_sum[0] = lsum;
  }
  DPRHOT_TMB(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  DPRHOT_TMB(1, 2);

  // 4 waves as 2 x 2, 64 contexts x 64 d columns each
  const int wm = wave >> 1, wn = wave & 1;
  const int i16 = lane & 15, g4 = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < kmax / 32; ++kk) {
    bf16x8 af[4], bf[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) af[a] = load_frag<128, 64, false, true>(Gs, wm * 64 + a * 16, kk, lane);
#pragma unroll
    for (int b = 0; b < 4; ++b) bf[b] = load_frag<128, 64, false, true>(Qs, wn * 64 + b * 16, kk, lane);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
  }
  __syncthreads();  // the images are dead: the fp32 tile takes their place
  DPRHOT_TMB(1, 3);
  float* const T = reinterpret_cast<float*>(sk_smem);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(wm * 64 + a * 16 + g4 * 4 + r) * SK_DC_TS + wn * 64 + b * 16 + i16] = acc[a][b][r] * sc;
  __syncthreads();
  DPRHOT_TMB(1, 4);
  if (p.dc_bf16) {  // 8 values per lane: whole 256-byte rows of bf16
    uint16_t* const out = static_cast<uint16_t*>(p.dC);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int e = tid + it * SK_THREADS, row = e >> 4, c8 = e & 15;
      const int m = n0 + row;
      if (m < p.Nc && c0 + c8 * 8 < p.d) {
        const float4 a = *reinterpret_cast<const float4*>(T + row * SK_DC_TS + c8 * 8);
        const float4 b = *reinterpret_cast<const float4*>(T + row * SK_DC_TS + c8 * 8 + 4);
        typedef unsigned sk_u4 __attribute__((ext_vector_type(4)));
        const sk_u4 w = {pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w)};
        if (p.nt_store) __builtin_nontemporal_store(w, reinterpret_cast<sk_u4*>(out + (size_t)m * p.d + c0 + c8 * 8));
        else *reinterpret_cast<sk_u4*>(out + (size_t)m * p.d + c0 + c8 * 8) = w;
      }
    }
  } else {
    float* const out = static_cast<float*>(p.dC);
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int e = tid + it * SK_THREADS, row = e >> 5, cq = e & 31;
      const int m = n0 + row;
      if (m < p.Nc && c0 + cq * 4 < p.d) {
        float4 v = *reinterpret_cast<const float4*>(T + row * SK_DC_TS + cq * 4);
        if (stamp && cq == 0 && m % p.stamp_period == p.stamp_row) v.x = lsum;
        if (p.nt_store) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(out + (size_t)m * p.d + c0 + cq * 4));
        else *reinterpret_cast<float4*>(out + (size_t)m * p.d + c0 + cq * 4) = v;
      }
    }
  }
  DPRHOT_TMB(1, 5);
}

// (Round 3 measured the alternative -- 128 columns of d per unit, two 32 KiB ring slots, twice the slices: per-workgroup stamps
//  K loop 7 steps x 0.9 us (only ONE step's DMA in flight behind the one being multiplied: a step waits a whole DMA latency),
//  epilogue 2.7 us (64 KiB of partial sums leave one CU at ~24 GB/s), launch 14.3 us against 13.4 for this form.)
// 64-wide mn-major image [k][64]: two k rows share a 256-byte bank row; the 32 lanes of a transpose read served together
// touch rows k0..k0+3 and k0+8..k0+11 of one 32-byte column group, so group cg of row k sits at slot cg ^ sk_swz64(k)
__device__ __forceinline__ int sk_swz64(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }

// dQ unit = (slice of contexts, 64 columns of d), all B rows: partial sum of dQ = G x C over the slice
__device__ __forceinline__ void sk_dq_unit(const SkBwdArgs& p, int unit, uint16_t* sk_smem) {
  constexpr int IA = SK_QA * 2 / 1024 / 4;  // DMA instructions per wave and slot: G part (4)
  constexpr int IB = SK_QB * 2 / 1024 / 4;  //                                      C part (2)
  constexpr int SLOT = SK_QA + SK_QB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ndt = p.d / SK_QN;
  const int dt = unit % ndt, ks = unit / ndt;
  const int c0 = dt * SK_QN;
  const int nk = (p.Nc + 63) / 64;
  const int s0 = ks * p.ksteps;
  const int ns = min(p.ksteps, nk - s0);  // steps of this unit (>= 1 by construction of the grid)

  DPRHOT_TMB(2, 0);
  // per-lane source coordinates of this wave's DMA instructions (one instruction = 1 KiB = 8 rows of 128 bytes)
  unsigned arow[IA], bcol[IB];
  int akin[IA], bk[IB];
#pragma unroll
  for (int j = 0; j < IA; ++j) {  // G part: k-major rows, 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
    const int row = (wave * IA + j) * 8 + (lane >> 3);
    arow[j] = (unsigned)min(row, p.B - 1) * (unsigned)p.Nc;
    akin[j] = ((lane & 7) ^ ((row >> 1) & 7)) << 3;
  }
#pragma unroll
  for (int j = 0; j < IB; ++j) {  // C part: k rows of 64 d columns, 32-byte group cg of row k at cg ^ sk_swz64(k)
    const int krow = (wave * IB + j) * 8 + (lane >> 3), pos = lane & 7;
    bk[j] = krow;
    bcol[j] = (unsigned)(c0 + ((((pos >> 1) ^ sk_swz64(krow)) << 4) + (pos & 1) * 8));
  }
  auto issue = [&](int s, int slot) {
    uint16_t* As = sk_smem + slot * SLOT;
    uint16_t* Bs = As + SK_QA;
    const int k0 = (s0 + s) * 64;
#pragma unroll
    for (int j = 0; j < IA; ++j)  // k beyond Nc (ragged last step): the address stays inside the row; zeroed in LDS below
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.G + arow[j] + min(k0 + akin[j], p.Nc - 8)), (g2_lds_ptr*)(As + (wave * IA + j) * 512), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IB; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.C + (unsigned)min(k0 + bk[j], p.Nc - 1) * (unsigned)p.d + bcol[j]),
                                       (g2_lds_ptr*)(Bs + (wave * IB + j) * 512), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < SK_QSLOTS; ++s)
    if (s < ns) issue(s, s);

  DPRHOT_TMB(2, 1);
  // 4 waves: 32 rows x 64 columns each
  const int i16 = lane & 15, g4 = lane >> 4;
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  for (int s = 0; s < ns; ++s) {
    sk_wait_younger<IA + IB>(min(s + SK_QSLOTS - 1, ns - 1) - s);
    const int slot = s % SK_QSLOTS;
    uint16_t* As = sk_smem + slot * SLOT;
    const uint16_t* Bs = As + SK_QA;
    const int kvalid = p.Nc - (s0 + s) * 64;  // < 64 only in the ragged last step of the last slice
    if (kvalid < 64) {
      sk_barrier();  // everybody's DMA of this slot has landed before anyone overwrites part of it
      // zero the k columns beyond Nc of the G part: 128 rows x 8 chunks
      for (int e = tid; e < SK_MAXB * 8; e += SK_THREADS) {
        const int row = e >> 3, ch = e & 7;
        if (ch * 8 >= kvalid) *reinterpret_cast<uint4*>(As + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3)) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    sk_barrier();
    // Both k slices of the step are read before the first wait: one LDS round trip per step instead of two (the step is a latency
    // chain: barrier -> reads -> MFMAs -> barrier -> DMA issue).
    // The transpose reads go through inline asm: hipcc parks an s_waitcnt vmcnt(0) in front of the builtin form whenever an
    // LDS-DMA is in flight (it cannot tell the slots apart), which would drain the whole ring at every step.  An asm read is
    // invisible to its counters, hence the explicit lgkmcnt(0) + sched_barrier behind the block (guide section 5.7, form iii).
    bf16x8 af[2][2];
    bf16x4 lo[2][4], hi[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = wave * 32 + a * 16 + i16;
        af[kk][a] = *reinterpret_cast<const bf16x8*>(As + row * 64 + (((kk * 4 + g4) ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int k = kk * 32 + g4 * 8 + (i16 >> 2);
        const unsigned addr = (unsigned)(uintptr_t)(lds_bf16x4*)(Bs + k * SK_QN + ((b ^ sk_swz64(k)) << 4) + (i16 & 3) * 4);
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512" : "=&v"(lo[kk][b]), "=&v"(hi[kk][b]) : "v"(addr));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 bf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        bf16x8 r;
        r[0] = lo[kk][b][0]; r[1] = lo[kk][b][1]; r[2] = lo[kk][b][2]; r[3] = lo[kk][b][3];
        r[4] = hi[kk][b][0]; r[5] = hi[kk][b][1]; r[6] = hi[kk][b][2]; r[7] = hi[kk][b][3];
        bf[b] = r;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][a], bf[b], acc[a][b], 0, 0, 0);
    }
    if (s + SK_QSLOTS < ns) {
      sk_barrier();
      issue(s + SK_QSLOTS, slot);
    }
  }
  sk_barrier();
  DPRHOT_TMB(2, 2);
  // ---- partial tile [128][64] fp32 through LDS -> 16-byte stores, 256 bytes per row
  constexpr int TS = SK_QN + 4;
  float* const T = reinterpret_cast<float*>(sk_smem);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(wave * 32 + a * 16 + g4 * 4 + r) * TS + b * 16 + i16] = acc[a][b][r];
  sk_barrier();
  const bool fin = p.nslices == 1;  // the one slice covers every context: this tile IS dQ's
  float* out = fin ? p.dQ : p.part + (size_t)ks * p.B * p.d;
  const float sc = fin ? p.h_scale * (p.d_scale ? *p.d_scale : 1.0f) : 1.0f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int e = tid + it * SK_THREADS, row = e >> 4, cq = e & 15;
    if (row < p.B) {
      float4 v = *reinterpret_cast<const float4*>(T + row * TS + cq * 4);
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      if (p.nt_store) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(out + (size_t)row * p.d + c0 + cq * 4));
      else *reinterpret_cast<float4*>(out + (size_t)row * p.d + c0 + cq * 4) = v;
    }
  }
  DPRHOT_TMB(2, 3);
}

// Workgroups [0, ndq_pad) are the dQ units (longer: they start first), the rest the dC units; inside each family consecutive
// unit numbers run on one XCD (units that share a G slice / a G tile meet in one L2).
__global__ __launch_bounds__(SK_THREADS, 2) void sk_bwd_kernel(SkBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sk_smem[];
  const int b = blockIdx.x;
  const int ndq = p.nslices * (p.d / SK_QN);
  if (b < p.ndq_pad) {
    if (b >= ndq) return;  // padding
    sk_dq_unit(p, sk_xcd_order(b, ndq), sk_smem);
  } else {
    sk_dc_unit(p, sk_xcd_order(b - p.ndq_pad, (int)gridDim.x - p.ndq_pad), sk_smem);
  }
}

// dQ = scale * sum of the partial slabs.  64 outputs (float4) per workgroup, the slabs dealt to 4 thread groups whose loads are
// all in flight at once; fixed summation order (bit-reproducible).  nslices <= 64.
__global__ __launch_bounds__(256) void sk_dq_reduce_kernel(const float* __restrict__ part, int nslices, size_t n4, float h_scale,
                                                           const float* d_scale, float* __restrict__ out) {
  __shared__ float4 s_acc[4][64];
  const int tid = threadIdx.x, o = tid & 63, zg = tid >> 6;
  const size_t idx = (size_t)blockIdx.x * 64 + o;
  const bool ok = idx < n4;
  float4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int z = zg + 4 * u;
    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && z < nslices) v[u] = reinterpret_cast<const float4*>(part)[(size_t)z * n4 + idx];
  }
  float4 a = v[0];
#pragma unroll
  for (int u = 1; u < 16; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  s_acc[zg][o] = a;
  __syncthreads();
  if (zg == 0 && ok) {
    const float sc = h_scale * (d_scale ? *d_scale : 1.0f);
    const float4 b = s_acc[1][o], c = s_acc[2][o], e = s_acc[3][o];
    a.x = ((a.x + b.x) + (c.x + e.x)) * sc;
    a.y = ((a.y + b.y) + (c.y + e.y)) * sc;
    a.z = ((a.z + b.z) + (c.z + e.z)) * sc;
    a.w = ((a.w + b.w) + (c.w + e.w)) * sc;
    reinterpret_cast<float4*>(out)[idx] = a;
  }
}


// =====================================================================================================================
// Round 4: the step WITHOUT the dScores launch  (sk_sim_kernel with P  ->  sk_bwdf_kernel  [-> sk_dq_reduce_kernel])
// =====================================================================================================================
// G = (softmax(S) - onehot) * scale needs every row's logsumexp -- a dependency on ALL column tiles of the sim launch, which is why
// sk_g_kernel sat between the sim and the backward launches (4.6 of cfg3-per-rank's 29.7 us, 0.18 of the HBM rate, plus a 4 MB fp32
// logit round trip and a 2 MB G round trip).  Factor it instead:
//     softmax_ij = exp(S_ij - lse_i) = exp(S_ij - lse_it) * exp(lse_it - lse_i) =: P_ij * f_it          (t = the 128-column tile of j)
// P is local to a sim unit (its own tile's softmax: written by the sim epilogue in bf16 -- the precision G had) and f_it is ONE
// number per (row, tile) that every backward unit derives from the nt tile values of its rows (33 KB of L2 reads, 32 exponentials
// per thread).  Then
//     dQ_i = sum_t f_it * (sum_{j in t} P_ij C_j)  + g_i * C[y_i]       a dQ unit multiplies into a scratch accumulator per 64-context
//                                                                       step and adds f * scratch to its sum (32 FMAs per step)
//     dC_j = sum_i P_ij (f_it q_i)                 + [j == y_i] g_i q_i  a dC unit scales its Q image's rows in LDS by f_it once
// with g_i = (exp(S_gold,i - lse_i) - 1) * scale in fp32: the gold column is 0 in P and its term is added from the bf16 operand row
// in the epilogue -- exact where the old path rounded G_gold to bf16, and immune to f_it underflowing when the gold tile lies far
// below the row maximum.  Units work in TILE space (tile t = the sim's statistics tile: with the packed layout's remapping the
// header rows between the ranks' blocks belong to no tile; the last tile of a rank writes their zero gradient rows and the stamp).
//
// Loads that must not drain the LDS-DMA queue (tile statistics, labels, gold logits, the gold rows) are inline-asm loads counted by
// hand: hipcc answers any ordinary load next to an LDS-DMA with s_waitcnt vmcnt(0) (guide section 5.7; visible in sk_sim_kernel's
// prologue), which would serialise "statistics -> logsumexp -> factors" behind the operand DMAs instead of under them.
struct SkBwdFArgs {
  const uint16_t* P;      // [B][Nc] bf16 tile-local softmax (sk_sim_kernel), gold column 0
  const uint16_t* Qb;     // [B][d]
  const uint16_t* C;      // [Nc][d]
  int B, Nc, d;
  const float* tile_lse;  // [ceil(nt/4)][B][4]
  const float* gold;      // [B] gold logits
  const int64_t* y;
  int64_t y_offset;
  int nt;                 // statistics tiles of 128 columns
  int tiles_per_rank, p_rows_c, p_n_ctx;  // remapped tiles (SkSimArgs::tiles_per_rank), else 0: tile t = columns [128 t, ...)
  float grad_scale;       // 1 / (Nq T): d loss / d S
  float h_scale;
  const float* d_scale;
  void* dC;               // [Nc][d] fp32, or bf16 (dc_bf16)
  int dc_bf16;
  int stamp_period, stamp_row;
  float* loss_sum;        // [1] out: loss_scale * sum of the row losses (dC unit 0)
  float loss_scale;
  float* row_loss;        // [B] out or nullptr
  float* row_lse;         // [B] out or nullptr
  int ksteps, nslices;    // dQ: 64-context steps per slice (tile space), slices
  float* part;            // [nslices][B][d] slice-normalised slabs (finished by sk_dq_finish_kernel)
  float* dQ;              // (unused: the finishing launch writes dQ)
  int ndq_pad;
  int nt_store;
  int dbg = 0;            // TIMING EXPERIMENTS ONLY (option sk_dbg): 1 dC units leave at once, 2 dQ units leave at once, 
  // Round 5: the finishing role INSIDE this launch (sk_fin_unit): the dQ units publish their slabs with write-through stores and count
  // themselves in; the last workgroups of the grid wait for that count and fold the slabs.  nullptr: sk_dq_finish_kernel does it.
  unsigned* tail_cnt = nullptr;  // zeroed by the sim launch (SkSimArgs::zero_me)
  int nfin = 0;                  // finishing workgroups (rows / (threads / 256))
  int reg_scale = 0;             // 1 (round 5): the dC units scale the Q FRAGMENTS by f in registers (no scale pass over the LDS image, one barrier less)
  int tail_fence = 0;            // 1: ordinary slab stores + ONE release fence per dQ unit, acquire fence + ordinary loads in the finishing role (A/B of the publish form)
  // Round 6 (option sk_dq_atomic): NO finishing launch and no slabs.  A dQ unit derives the rows' logsumexp itself -- at its END, from
  // loads issued under its last step, not in front of its loop where rounds 4's forms lost -- scales its slab by exp(m_is - lse_i) *
  // scale, adds the gold term g_i C[y_i] of the rows whose gold column lies in its slice, and ADDS the tile into dQ with
  // global_atomic_add_f32 (dQ was zero-filled by the sim launch: SkSimArgs::zero_dq).  The order of the (<= 16) additions per element
  // depends on the run: dQ is then reproducible to rounding, not to the bit (the reference's cuBLAS split-K is no different).
  int dq_atomic = 0;
};

constexpr int SK_FT = 8;                            // statistics tiles a dQ unit may touch ((ksteps + 1) / 2 + 1 <= SK_FT: sk_fused_ok)
constexpr int SK_FX = (SK_FT * 128 + 2 * 128) * 4;  // bytes of a dQ unit's table behind its ring: w[SK_FT][128] (+ 1 KiB spare)
constexpr int SK_FDC_TABLES = 6 * 128 * 4 + 16;  // fi, gv, rl, ym, cnt [128] each + dupf (sk_dc_unit_f)
inline size_t sk_bwdf_lds() {  // the larger of the two unit kinds' needs
  const size_t dq = (size_t)SK_QSLOTS * (SK_QA + SK_QB) * 2 + SK_FX, dc = (size_t)SK_COLS * (SK_DN + 4) * 4 + SK_FDC_TABLES;
  return dq > dc ? dq : dc;
}
constexpr int SK_FDC_TAB = SK_COLS * SK_DC_TS * 4;  // byte offset of a dC unit's tables (behind its fp32 epilogue tile)

__device__ __forceinline__ f32x4 sk_ld16_hidden(const void* ptr) {
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
__device__ __forceinline__ float sk_ld4_hidden(const void* ptr) {
  float r;
  asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
typedef unsigned sk_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sk_u32x2 sk_ld8_hidden(const void* ptr) {
  sk_u32x2 r;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
// write-through (sc1) 16-byte store: leaves this XCD's L2 at once -- the payload of an in-launch hand-off (guide 6 G16, R1 form);
// the statement ends with s_nop 1 (guide 5.7 item 1: hipcc may otherwise overwrite the data registers before the store has read them)
__device__ __forceinline__ void sk_st16_sc1(void* ptr, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}
// the matching load: served by L2 / memory, never by this CU's L1 (asm: counted by hand like the other hidden loads)
__device__ __forceinline__ f32x4 sk_ld16_sc1(const void* ptr) {
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
// LDS stores next to LDS-DMAs in flight: asm, for the reason given at sk_sim_kernel's q chunk (hipcc would drain the queue first)
__device__ __forceinline__ void sk_lds_st32(void* lds_ptr, unsigned v) {
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  const unsigned addr = (unsigned)(uintptr_t)(lds_u32*)lds_ptr;
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ float sk_bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float sk_bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// first column of statistics tile t
__device__ __forceinline__ int sk_tile_col(const SkBwdFArgs& p, int t) {
  return p.tiles_per_rank > 0 ? (t / p.tiles_per_rank) * p.p_rows_c + (t % p.tiles_per_rank) * SK_COLS : t * SK_COLS;
}
// 64-context step (tile space) that holds global column n, or -1 (a header row of the packed layout, or beyond Nc)
__device__ __forceinline__ int sk_step_of_col(const SkBwdFArgs& p, int n) {
  if (n < 0 || n >= p.Nc) return -1;
  if (p.tiles_per_rank == 0) return n >> 6;
  const int r = n / p.p_rows_c, j = n - r * p.p_rows_c;
  return j < p.p_n_ctx ? ((r * p.tiles_per_rank + (j >> 7)) << 1) + ((j >> 6) & 1) : -1;
}

// The statistics of ONE row per group of PH threads (row = tid / PH, part = tid % PH: groups part, part + PH, ... of four tile values;
// PH = 2 in the four-wave units, 4 in the eight-wave ones).  issue() before the unit's DMAs, finish() after the counted wait: the
// row's logsumexp, identical bits in every lane of the group.
template <int NG, int PH = 2>
struct SkRowStats {
  f32x4 sv[NG];
  __device__ __forceinline__ void issue(const float* tile_lse, int nt, int B, int prow, int ph) {
    const int ng = (nt + 3) >> 2;
#pragma unroll
    for (int u = 0; u < NG; ++u) sv[u] = sk_ld16_hidden(tile_lse + ((size_t)min(ph + PH * u, ng - 1) * B + prow) * 4);
  }
  // (guide section 5.7 item 3: makes the loaded registers opaque HERE, behind the caller's counted wait -- no consumer is scheduled above it)
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int u = 0; u < NG; ++u) asm volatile("" : "+v"(sv[u]));
  }
  __device__ __forceinline__ float finish(int nt, int ph) const {
    float mx = -INFINITY;
    float v[NG][4];
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int t = (ph + PH * u) * 4;  // (groups beyond the last re-read the last group: t >= nt there)
      v[u][0] = t + 0 < nt ? sv[u][0] : -INFINITY;
      v[u][1] = t + 1 < nt ? sv[u][1] : -INFINITY;
      v[u][2] = t + 2 < nt ? sv[u][2] : -INFINITY;
      v[u][3] = t + 3 < nt ? sv[u][3] : -INFINITY;
      mx = fmaxf(mx, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
    }
    mx = fmaxf(mx, ss_dpp<0xB1>(mx));  // the other lanes of the row (lane ^ 1, then lane ^ 2)
    if constexpr (PH == 4) mx = fmaxf(mx, ss_dpp<0x4E>(mx));
    float sm = 0.f;
    if (mx != -INFINITY) {
#pragma unroll
      for (int u = 0; u < NG; ++u) sm += (__expf(v[u][0] - mx) + __expf(v[u][1] - mx)) + (__expf(v[u][2] - mx) + __expf(v[u][3] - mx));
    }
    // (a + b == b + a: every lane of the group holds the same bits)
    sm += ss_dpp<0xB1>(sm);
    if constexpr (PH == 4) sm += ss_dpp<0x4E>(sm);
    return mx + logf(sm);
  }
};

// dQ unit (fused): (slice of 64-context steps in tile space) x (64 columns of d), all B rows.
// The unit does NOT need the row logsumexp.  Its slab is normalised to the slice's own reference m = max of the slice's tile values:
//     slab_i = sum_{t in slice} exp(lse_it - m_i) * (sum_{j in t} P_ij C_j)
// (one 4-byte load and one exponential per thread and tile), and the finishing launch -- which exists anyway to add the slabs up --
// multiplies slab s by exp(m_is - lse_i) * scale and adds the gold term g_i C[y_i] (sk_dq_finish_kernel: 65 tile values per ROW there,
// not per unit).  Measured forms of this unit that derived the row logsumexp themselves (per-workgroup stamps, dQ units alone on the
// chip): 15.5-16.1 us against 10.8 for sk_dq_unit -- the statistics round trip and 32 exponentials per thread sit in front of, or
// inside, a latency-chained loop whichever way they are scheduled (before the first step: prologue 3.5 us; under step 0's MFMAs: the
// ring runs dry behind them), and the launch came out exactly as long as the two launches it replaced (29.4 us per step both ways).
// The factor FMAs of step s run under the MFMAs of step s + 1 (two scratch accumulators), and a slot is refilled as soon as every
// wave holds its fragments.
template <int NG, int NW = 4, bool ATOMIC = false>
__device__ __forceinline__ void sk_dq_unit_f(const SkBwdFArgs& p, int unit, uint16_t* sk_smem) {
  constexpr int NT = NW * 64;  // threads
  constexpr int PH = NW / 2;   // threads per row of the weight table
  constexpr int NB = 16 / NW;  // 16-column groups of a wave: four waves = 4 row groups x 64 columns, eight = 4 row groups x 2 halves of 32
  constexpr int IA = SK_QA * 2 / 1024 / NW;  // 4 | 2
  constexpr int IB = SK_QB * 2 / 1024 / NW;  // 2 | 1
  constexpr int PER = IA + IB;
  constexpr int SLOT = SK_QA + SK_QB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = NW == 4 ? wave : wave >> 1, wc = NW == 4 ? 0 : wave & 1;
  const int ndt = p.d / SK_QN;
  const int dt = unit % ndt, ks = unit / ndt;
  const int c0 = dt * SK_QN;
  const int nk = p.tiles_per_rank > 0 ? 2 * p.nt : (p.Nc + 63) / 64;
  const int s0 = ks * p.ksteps;
  const int ns = min(p.ksteps, nk - s0);
  const int t0 = s0 >> 1;
  float* const out = p.part + (size_t)ks * p.B * p.d;
  // publish (finishing role in this launch): the slab went out with write-through stores; every wave drains its own, the barrier
  // collects the waves, ONE relaxed device-scope add counts the unit in (guide 5 "in-launch split-K reduction", the sc1 form: no
  // L2 write-back fence -- this XCD's L2 is full of the dC units' dirty lines)
  auto publish = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (p.tail_fence) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (guide 6 G16 pitfall 12: the wait behind buffer_wbl2 restated where hipcc cannot drop it)
      }
      __hip_atomic_fetch_add(p.tail_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if (ns <= 0 && ATOMIC) return;  // (an empty slice adds nothing)
  if (ns <= 0) {  // a remapped tiling with fewer steps than the plan's slices cover: this slice is empty, its slab is zero
    for (int e = tid; e < p.B * (SK_QN / 4); e += NT) {
      float* const dst = out + (size_t)(e >> 4) * p.d + c0 + (e & 15) * 4;
      if (p.tail_cnt != nullptr && !p.tail_fence) sk_st16_sc1(dst, f32x4{0.f, 0.f, 0.f, 0.f});
      else *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.tail_cnt != nullptr) publish();
    return;
  }
  float* const fs = reinterpret_cast<float*>(sk_smem + SK_QSLOTS * SLOT);  // [SK_FT][128]: weight of (local tile, row)
  DPRHOT_TMB(2, 0);

  // ---- hidden loads first (older than every DMA of this wave): the tile values of this slice, row tid / PH, tiles t0 + tid % PH + PH u
  const int prow = min(tid / PH, p.B - 1), ph = tid % PH;
  float lt[SK_FT / PH];
#pragma unroll
  for (int u = 0; u < SK_FT / PH; ++u) {
    const int t = min(t0 + ph + PH * u, p.nt - 1);
    lt[u] = sk_ld4_hidden(p.tile_lse + ((size_t)(t >> 2) * p.B + prow) * 4 + (t & 3));
  }

  // (sk_dq_atomic: the row's gold logit and label, also ahead of every DMA; its tile values are issued under the LAST step)
  // (ATOMIC is a template parameter, not a run-time switch: the statistics registers of the epilogue below must not enter the register
  //  allocation of the slab form -- as a run-time branch they made every sk_bwdf_kernel spill, and a spill's scratch traffic inside
  //  the loop breaks its hand-counted vmcnt waits: wrong tiles, caught by tests/test_fused_dscores.py)
  float gl_a = 0.f, yf_a = 0.f, m_keep = -INFINITY;
  if constexpr (ATOMIC) {
    gl_a = sk_ld4_hidden(p.gold + prow);
    yf_a = sk_ld4_hidden(reinterpret_cast<const int*>(p.y) + 2 * prow);
  }

  // ---- ring DMAs: per-lane source coordinates (one instruction = 1 KiB = 8 rows of 128 bytes)
  unsigned arow[IA];
  int akin[IA], bkr[IB];
  unsigned bcol[IB];
#pragma unroll
  for (int j = 0; j < IA; ++j) {
    const int row = (wave * IA + j) * 8 + (lane >> 3);
    arow[j] = (unsigned)min(row, p.B - 1) * (unsigned)p.Nc;
    akin[j] = ((lane & 7) ^ ((row >> 1) & 7)) << 3;
  }
#pragma unroll
  for (int j = 0; j < IB; ++j) {
    const int krow = (wave * IB + j) * 8 + (lane >> 3), pos = lane & 7;
    bkr[j] = krow;
    bcol[j] = (unsigned)(c0 + ((((pos >> 1) ^ sk_swz64(krow)) << 4) + (pos & 1) * 8));
  }
  // first context of the NEXT step to be issued, kept incrementally (a division per refill sat in the loop's critical path):
  // + 64 per step, + the header rows where a rank's tiles end (remapped tiles only)
  const int spr = 2 * p.tiles_per_rank;  // steps per rank
  int nxt_col, nxt_left;                  // column of the next step to issue; steps left in its rank (remapped tiles)
  {
    const int gs = s0;
    nxt_col = sk_tile_col(p, gs >> 1) + (gs & 1) * 64;
    nxt_left = spr > 0 ? spr - gs % spr : 0x7fffffff;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // (provably uniform: the DMA destinations then reach m0 without a v_readfirstlane each)
  auto issue = [&](int s, int slot) {  // (steps are issued in order: s is the next one)
    uint16_t* As = sk_smem + slot * SLOT;
    uint16_t* Bs = As + SK_QA;
    const int k0 = nxt_col;
    nxt_col += 64;
    if (--nxt_left == 0) { nxt_col += p.p_rows_c - p.p_n_ctx; nxt_left = spr; }
#pragma unroll
    for (int j = 0; j < IA; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.P + arow[j] + min(k0 + akin[j], p.Nc - 8)), (g2_lds_ptr*)(As + (wave_u * IA + j) * 512), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < IB; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.C + (unsigned)min(k0 + bkr[j], p.Nc - 1) * (unsigned)p.d + bcol[j]),
                                       (g2_lds_ptr*)(Bs + (wave_u * IB + j) * 512), 16, 0, 0);
  };
  // Two steps are issued here, and step s + 2 inside step s, behind its barrier and its fragment reads: at that barrier every wave has
  // finished READING slot (s - 1) % 3 -- the slot step s + 2 goes to -- so the refill needs no barrier of its own (one barrier per step
  // instead of two; the stamps showed the wait for a slot at 0.11 of 0.73 us per step: one step less of look-ahead is affordable --
  // and a ring of TWO slots runs the units alone on the chip just as fast: the third slot pays only next to the dC units, by 1.1 us
  // per launch)
#pragma unroll
  for (int s = 0; s < SK_QSLOTS - 1; ++s)
    if (s < ns) issue(s, s);
  // (Measured and not kept: warming this XCD's L2 for the later steps -- one lane per 128-byte line of their operands into a register
  //  nobody reads, issued right behind the first three slots' DMAs, counted in the waits of steps 0..2.  The loop did not get faster
  //  (10.3 against 10.2 us by the stamps: the step is not waiting for far memory) and the prologue grew from 1.4 to 3.5 us.)
  DPRHOT_TMB(2, 1);

  const int i16 = lane & 15, g4 = lane >> 4;
  f32x4 acc[2][NB], tA[2][NB], tB[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = tA[a][b] = tB[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  auto add_scaled = [&](int s, const f32x4 (&tm)[2][NB]) {  // sum += w(row, tile of step s) * (P x C of step s)
    const int tl = ((s0 + s) >> 1) - t0;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const f32x4 fr = *reinterpret_cast<const f32x4*>(fs + tl * 128 + wr * 32 + a * 16 + g4 * 4);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] = fmaf(fr[r], tm[a][b][r], acc[a][b][r]);
    }
  };
#if defined(DPRHOT_TIMING) && DPRHOT_TIMING >= 2
  unsigned long long tk_[5] = {0, 0, 0, 0, 0}, tl_ = wall_clock64();  // thread 0's view of a step: wait | barrier | LDS reads | barrier + refill | MFMA issue + FMAs
#define SK_LT(i) do { const unsigned long long t_ = wall_clock64(); tk_[i] += t_ - tl_; tl_ = t_; } while (0)
#else
#define SK_LT(i) do {} while (0)
#endif
  // one step: FIRST = step 0 (forms the weight table); cur receives this step's product, prev holds the previous step's
  auto body = [&](auto first_tag, int s, f32x4 (&cur)[2][NB], const f32x4 (&prev)[2][NB]) {
    constexpr bool FIRST = decltype(first_tag)::value;
    SK_LT(4);
    sk_wait_younger<PER>(min(s + SK_QSLOTS - 2, ns - 1) - s);  // slot s has landed (and, in step 0, the tile values issued ahead of it)
    SK_LT(0);
    const int slot = s % SK_QSLOTS;
    uint16_t* As = sk_smem + slot * SLOT;
    const uint16_t* Bs = As + SK_QA;
    const int kvalid = p.tiles_per_rank > 0 ? 64 : p.Nc - (s0 + s) * 64;  // < 64 only in the ragged last step of an unmapped layout
    if (kvalid < 64) {
      sk_barrier();
      for (int e = tid; e < SK_MAXB * 8; e += NT) {
        const int row = e >> 3, ch = e & 7;
        if (ch * 8 >= kvalid) *reinterpret_cast<uint4*>(As + row * 64 + ((ch ^ ((row >> 1) & 7)) << 3)) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    sk_barrier();  // (step 1: also publishes the table step 0 wrote)
    SK_LT(1);
    bf16x8 af[2][2];
    bf16x4 lo[2][NB], hi[2][NB];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = wr * 32 + a * 16 + i16;
        af[kk][a] = *reinterpret_cast<const bf16x8*>(As + row * 64 + (((kk * 4 + g4) ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int k = kk * 32 + g4 * 8 + (i16 >> 2);
        const unsigned addr = (unsigned)(uintptr_t)(lds_bf16x4*)(Bs + k * SK_QN + (((wc * NB + b) ^ sk_swz64(k)) << 4) + (i16 & 3) * 4);
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512" : "=&v"(lo[kk][b]), "=&v"(hi[kk][b]) : "v"(addr));
      }
    }
    f32x4 frp[2];  // the PREVIOUS step's weights (its FMAs ride in this step's MFMA gaps)
    if constexpr (!FIRST) {
      const int tlp = ((s0 + s - 1) >> 1) - t0;
#pragma unroll
      for (int a = 0; a < 2; ++a) frp[a] = *reinterpret_cast<const f32x4*>(fs + tlp * 128 + wr * 32 + a * 16 + g4 * 4);
    }
    // The refill of step s + 2 goes into the slot read in step s - 1 (free since this step's barrier), and it is issued HERE, behind
    // the fragment reads: six (three) LDS-DMA pieces per wave take the texture path 0.2-0.3 us to accept, which the LDS reads'
    // latency now overlaps (stamps: "refill + fragment reads" 0.41 -> 0.33 us per step, the loop 10.4 -> 9.8 us).
    __builtin_amdgcn_sched_barrier(0);
    if (s + SK_QSLOTS - 1 < ns) issue(s + SK_QSLOTS - 1, (s + SK_QSLOTS - 1) % SK_QSLOTS);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    SK_LT(2);
    // (Measured and not kept: the refill issued unconditionally -- clamped addresses -- INSIDE the MFMA block, one LDS-DMA piece per
    //  16 / 6 MFMAs through sched_group_barrier(0x010): the scheduler clumped five of the six pieces behind the eleventh MFMA, the
    //  stamps moved 0.1 us from "wait for the slot" into the MFMA block, and the launch took 14.0 against 13.8 us.)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 bf[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        bf16x8 r;
        r[0] = lo[kk][b][0]; r[1] = lo[kk][b][1]; r[2] = lo[kk][b][2]; r[3] = lo[kk][b][3];
        r[4] = hi[kk][b][0]; r[5] = hi[kk][b][1]; r[6] = hi[kk][b][2]; r[7] = hi[kk][b][3];
        bf[b] = r;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
          cur[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][a], bf[b], kk == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : cur[a][b], 0, 0, 0);
      if constexpr (!FIRST) {
        // half of the previous step's factor FMAs per k slice: sum += w * (P x C of step s - 1)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[kk][b][r] = fmaf(frp[kk][r], prev[kk][b][r], acc[kk][b][r]);
      }
    }
    if constexpr (!FIRST) {
      // one MFMA, one FMA pair: the vector ALU work sits in the matrix pipe's issue gaps instead of behind the last MFMA
#pragma unroll
      for (int i = 0; i < 4 * NB; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
      // (the sums are made opaque HERE: otherwise the FMAs are sunk past the refill below, where nothing covers them)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(acc[a][b]));
    }
    SK_LT(3);
    if constexpr (FIRST) {
      // under the MFMAs: the slice's reference per row and the weights of its tiles (published by the next barrier of this workgroup)
#pragma unroll
      for (int u = 0; u < SK_FT / PH; ++u) asm volatile("" : "+v"(lt[u]));
      const int tlast = min((s0 + ns - 1) >> 1, p.nt - 1) - t0;  // last local tile of this slice
      float m = -INFINITY;
#pragma unroll
      for (int u = 0; u < SK_FT / PH; ++u) m = fmaxf(m, ph + PH * u <= tlast ? lt[u] : -INFINITY);
      m = fmaxf(m, ss_dpp<0xB1>(m));  // the other tiles of the row (lane ^ 1, then lane ^ 2)
      if constexpr (PH == 4) m = fmaxf(m, ss_dpp<0x4E>(m));
      if constexpr (ATOMIC) m_keep = m;
      const int row = tid / PH;
      if (row < p.B) {
#pragma unroll
        for (int u = 0; u < SK_FT / PH; ++u) {
          const int tl = ph + PH * u;
          // a tile (or a whole slice) without an unmasked column weighs 0
          const float w = (tl <= tlast && lt[u] != -INFINITY) ? __expf(lt[u] - m) : 0.f;
          sk_lds_st32(fs + tl * 128 + row, __float_as_uint(w));
        }
      }
    }
  };
  body(std::true_type{}, 0, tA, tB);
  for (int s = 1; s < ns; s += 2) {
    body(std::false_type{}, s, tB, tA);
    if (s + 1 < ns) body(std::false_type{}, s + 1, tA, tB);
  }
  sk_barrier();  // every wave is done with the ring (and, when the slice is one step long, the table is published here)
  // (ATOMIC: ALL tile values of the row -- issued here, behind the unit's last counted wait, so that they land under the last factor
  //  FMAs and the staging of the tile; issued inside the loop they would keep their registers allocated across it)
  SkRowStats<ATOMIC ? NG * 2 / PH : 1, PH> st_a;
  if constexpr (ATOMIC) st_a.issue(p.tile_lse, p.nt, p.B, prow, ph);
  if ((ns - 1) & 1) add_scaled(ns - 1, tB);
  else add_scaled(ns - 1, tA);
  DPRHOT_TMB(2, 2);
#if defined(DPRHOT_TIMING) && DPRHOT_TIMING >= 2
  SK_LT(4);
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    // slots 3.. of this workgroup's record: accumulated ticks of the five parts of a step (the unit's end stamp moves to slot 7)
    g_dprhot_tmb[(2 * 4096 + blockIdx.x) * 8 + 4] = tk_[0];
    g_dprhot_tmb[(2 * 4096 + blockIdx.x) * 8 + 5] = tk_[1];
    g_dprhot_tmb[(2 * 4096 + blockIdx.x) * 8 + 6] = tk_[2];
    g_dprhot_tmb[(3 * 4096 + blockIdx.x) * 8 + 0] = tk_[3];
    g_dprhot_tmb[(3 * 4096 + blockIdx.x) * 8 + 1] = tk_[4];
    g_dprhot_tmb[(3 * 4096 + blockIdx.x) * 8 + 2] = (unsigned long long)ns;
  }
#endif
  // ---- slab tile [128][64] fp32 through LDS -> 16-byte stores, 256 bytes per row
  constexpr int TS = SK_QN + 4;
  float* const T = reinterpret_cast<float*>(sk_smem);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(wr * 32 + a * 16 + g4 * 4 + r) * TS + (wc * NB + b) * 16 + i16] = acc[a][b][r];
  if constexpr (ATOMIC) {
    // ---- no slab: the tile is scaled to the row softmax HERE and added into dQ.  Per row (the PH threads that hold its statistics):
    //      e_i = exp(m_is - lse_i) * scale (0 for a slice without an unmasked column), and -- when the row's gold column lies in this
    //      slice -- g_i = (exp(S_gold,i - lse_i) - 1) * scale with its column, the term sk_dq_finish_kernel adds once per row.
    //      Table e [128] | g [128] | gold column or -1 [128] in the dead ring, behind the staged tile (NOT in the weight table's place:
    //      a slower wave may still be reading its weights in add_scaled above).
    float* const fs = reinterpret_cast<float*>(sk_smem) + 10240;  // byte 40960: the tile ends at 128 * 68 * 4 = 34816, the ring at 73728
    static_assert(SK_MAXB * (SK_QN + 4) * 4 <= 40960 && 40960 + 3 * 128 * 4 <= SK_QSLOTS * (SK_QA + SK_QB) * 2, "sk_dq_atomic table");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    st_a.pin();
    asm volatile("" : "+v"(gl_a), "+v"(yf_a));
    const float lse = st_a.finish(p.nt, ph);
    const float sc = p.h_scale * (p.d_scale ? *p.d_scale : 1.0f) * p.grad_scale;
    if (ph == 0 && tid / PH < p.B) {
      const int row = tid / PH;
      const int yg = __float_as_int(yf_a) + (int)p.y_offset;
      const int gstep = sk_step_of_col(p, yg);
      const bool mine = gstep >= s0 && gstep < s0 + ns;
      fs[row] = m_keep == -INFINITY ? 0.f : __expf(m_keep - lse) * sc;
      fs[128 + row] = mine ? (__expf(gl_a - lse) - 1.0f) * sc : 0.f;
      reinterpret_cast<int*>(fs)[256 + row] = mine ? yg : -1;
    }
    sk_barrier();  // the staged tile and the table
    float* const dq = p.dQ + c0;
    const int rot = (ks * SK_MAXB) / max(p.nslices, 1);
#pragma unroll 4
    for (int it = 0; it < SK_MAXB * SK_QN / NT; ++it) {
      // one wave = one row: 256 contiguous bytes per atomic instruction; the slices of a dQ tile finish together, so each starts at
      // its own row (rotation by slice): at any moment the units adding into one tile are in different lines
      const int e = tid + it * NT, row = ((e >> 6) + rot) & (SK_MAXB - 1), col = e & 63;
      if (row < p.B) {
        float v = T[row * TS + col] * fs[row];
        const int yg = reinterpret_cast<const int*>(fs)[256 + row];
        if (yg >= 0) v = fmaf(fs[128 + row], __uint_as_float((unsigned)p.C[(size_t)yg * p.d + c0 + col] << 16), v);  // (wave-uniform branch)
        (void)__hip_atomic_fetch_add(dq + (size_t)row * p.d + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_atomic_add_f32, no return
      }
    }
    DPRHOT_TMB(2, 3);
    return;
  }
  sk_barrier();
#pragma unroll
  for (int it = 0; it < SK_MAXB * 16 / NT; ++it) {
    const int e = tid + it * NT, row = e >> 4, cq = e & 15;
    if (row < p.B) {
      const float4 v = *reinterpret_cast<const float4*>(T + row * TS + cq * 4);
      if (p.tail_cnt != nullptr && !p.tail_fence) sk_st16_sc1(out + (size_t)row * p.d + c0 + cq * 4, f32x4{v.x, v.y, v.z, v.w});
      else if (p.nt_store && !(p.dbg & 16)) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(out + (size_t)row * p.d + c0 + cq * 4));
      else *reinterpret_cast<float4*>(out + (size_t)row * p.d + c0 + cq * 4) = v;
    }
  }
  if (p.tail_cnt != nullptr) publish();
  DPRHOT_TMB(2, 3);
}

// dC unit (fused): statistics tile t (128 contexts) x 128 columns of d.
// The thread pair that derives row i's logsumexp also owns row i of the Q image: it scales the row by f in place and KEEPS the
// unscaled values; after the GEMM it adds g_i * q_i to the output row of its gold column (one row of the fp32 tile per gold column:
// no lookup tables, no second trip to Q).  Rows that share a gold column (cnt > 1: rare) are added one after the other instead.
template <int NG, int NW = 4>
__device__ __forceinline__ void sk_dc_unit_f(const SkBwdFArgs& p, int unit, uint16_t* sk_smem) {
  constexpr int NT = NW * 64;   // threads
  constexpr int PH = NW / 2;    // threads per query row (statistics, the kept part of the Q row)
  constexpr int WN = NW / 2;    // waves across the 128 d columns (2 x 64 | 4 x 32), two wave rows of 64 contexts either way
  constexpr int NBF = 8 / WN;   // 16-column groups of a wave
  constexpr int RC = 16 / PH;   // 16-byte chunks of a Q row kept per thread
  constexpr size_t kImg = (size_t)SK_MAXB * SK_COLS;
  uint16_t* const Gs = sk_smem;         // [128 k = query row][128 m = context]: P
  uint16_t* const Qs = sk_smem + kImg;  // [128 k = query row][128 n = d column]: Q, rows scaled by f in place
  char* const tab = reinterpret_cast<char*>(sk_smem) + SK_FDC_TAB;
  float* const fi = reinterpret_cast<float*>(tab);  // [128] f of (row, this tile)
  float* const gv = fi + 128;                       // [128] g of the row
  float* const rl = gv + 128;                       // [128] row loss
  int* const ym = reinterpret_cast<int*>(rl + 128);  // [128] gold column relative to the tile, or -1
  int* const cnt = ym + 128;                         // [128] rows whose gold column is context m of this tile
  int* const dupf = cnt + 128;                       // [1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ndt = (p.d + SK_DN - 1) / SK_DN;
  const int dt = unit % ndt, t = unit / ndt;
  const int n0 = sk_tile_col(p, t), c0 = dt * SK_DN;
  const int kmax = p.B;
  const int nvalid = p.tiles_per_rank > 0 ? SK_COLS : min(SK_COLS, p.Nc - n0);  // contexts of this tile

  DPRHOT_TMB(1, 0);
  if (tid < SK_COLS) cnt[tid] = 0;  // (no DMA in flight yet: a plain store)
  if (tid == 0) dupf[0] = 0;
  const int prow = min(tid / PH, p.B - 1), ph = tid % PH;
  SkRowStats<NG * 2 / PH, PH> st;
  st.issue(p.tile_lse, p.nt, p.B, prow, ph);
  float lt = sk_ld4_hidden(p.tile_lse + ((size_t)(t >> 2) * p.B + prow) * 4 + (t & 3));
  float gl = sk_ld4_hidden(p.gold + prow);
  float yf = sk_ld4_hidden(reinterpret_cast<const int*>(p.y) + 2 * prow);
  const int nins = kmax / 4 / NW;  // per wave and image (B % 32 == 0: 2, 4, 6 or 8 | 1..4)
  for (int j = 0; j < nins; ++j) {
    const int krow = (wave * nins + j) * 4 + (lane >> 4), pos = lane & 15;
    const int col = ((((pos >> 1) ^ mswz(krow))) << 4) + (pos & 1) * 8;
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.P + (size_t)krow * p.Nc + min(n0 + col, p.Nc - 8)),
                                     (g2_lds_ptr*)(Gs + (wave * nins + j) * 4 * SK_COLS), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.Qb + (size_t)krow * p.d + min(c0 + col, p.d - 8)),
                                     (g2_lds_ptr*)(Qs + (wave * nins + j) * 4 * SK_DN), 16, 0, 0);
  }
  DPRHOT_TMB(1, 1);
  if constexpr (NW == 4) sk_wait_younger<4>(nins >> 1);  // 2 * nins DMAs behind the hidden loads
  else sk_wait_younger<2>(nins);
  __builtin_amdgcn_sched_barrier(0);
  st.pin();
  asm volatile("" : "+v"(lt), "+v"(gl), "+v"(yf));
  const float lse = st.finish(p.nt, ph);
  const float f = __expf(lt - lse) * p.grad_scale;
  const int yrel = __float_as_int(yf) + (int)p.y_offset - n0;  // gold column relative to this tile
  const int row = tid / PH;
  const bool mine = row < p.B && yrel >= 0 && yrel < nvalid;
  const float g = (__expf(gl - lse) - 1.0f) * p.grad_scale;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the images have landed; cnt[] is initialised
  DPRHOT_TMB(1, 2);
  // Tables first; the rows that have a gold column in this tile keep their unscaled half row (the gold term needs it).
  // (A pass in which every thread scaled its own half row read the image at a 128-byte lane stride: 8-way bank conflicts on every
  //  b128 access, 2.1 us per unit by the stamps.)
  uint4 raw[RC];
  {
    const uint4* base = reinterpret_cast<const uint4*>(Qs + min(row, kmax - 1) * SK_DN + ph * (SK_DN / PH));
    if (mine) {
#pragma unroll
      for (int j = 0; j < RC; ++j) raw[j] = base[j];
    }
    if (ph == 0) {
      if (row < p.B) {
        fi[row] = f;
        gv[row] = g;
        rl[row] = lse - gl;
        ym[row] = mine ? yrel : -1;
        if (mine) atomicAdd(&cnt[yrel], 1);
        if (unit == 0) {
          if (p.row_loss) p.row_loss[row] = lse - gl;
          if (p.row_lse) p.row_lse[row] = lse;
        }
      } else {
        fi[row] = 0.f;
        rl[row] = 0.f;
        ym[row] = -1;
      }
    }
  }
  __syncthreads();
  const bool rs = p.reg_scale != 0;
  if (!rs) {
    // Q rows x f in place (both bf16): 16-byte chunk c of the image belongs to row c >> 4 whatever the column swizzle; consecutive
    // lanes take consecutive chunks
    uint4* const img = reinterpret_cast<uint4*>(Qs);
#pragma unroll
    for (int j = 0; j < SK_MAXB * 16 / NT; ++j) {
      const int c = tid + j * NT;
      const float fr = fi[c >> 4];
      uint4 w = img[c];
      w.x = pk_bf16(sk_bf_lo(w.x) * fr, sk_bf_hi(w.x) * fr);
      w.y = pk_bf16(sk_bf_lo(w.y) * fr, sk_bf_hi(w.y) * fr);
      w.z = pk_bf16(sk_bf_lo(w.z) * fr, sk_bf_hi(w.z) * fr);
      w.w = pk_bf16(sk_bf_lo(w.w) * fr, sk_bf_hi(w.w) * fr);
      img[c] = w;
    }
    __syncthreads();
  }
  const bool solo = mine && cnt[yrel] == 1;
  if (mine && !solo) dupf[0] = 1;  // two rows with one gold column (read behind the next barrier)
  // loss numerator (every unit that stamps it, and unit 0 which publishes it): fixed order
  const float sc = p.h_scale * (p.d_scale ? *p.d_scale : 1.0f);
  const bool last_of_rank = p.tiles_per_rank > 0 && t % p.tiles_per_rank == p.tiles_per_rank - 1;
  const bool stamp = dt == 0 && p.stamp_period > 0 && !p.dc_bf16;
  float lsum = 0.f;
  if (stamp || unit == 0) {
    lsum = sk_loss_sum(rl, SK_MAXB, lane) * p.loss_scale;
    if (unit == 0 && tid == 0) p.loss_sum[0] = lsum;
  }
  DPRHOT_TMB(1, 3);
  const int wm = wave / WN, wn = wave % WN;
  const int i16 = lane & 15, g4 = lane >> 4;
  f32x4 acc[4][NBF];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NBF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kk = 0; kk < kmax / 32; ++kk) {
    bf16x8 af[4], bf[NBF];
#pragma unroll
    for (int a = 0; a < 4; ++a) af[a] = load_frag<128, 64, false, true>(Gs, wm * 64 + a * 16, kk, lane);
#pragma unroll
    for (int b = 0; b < NBF; ++b) bf[b] = load_frag<128, 64, false, true>(Qs, wn * (SK_DN / WN) + b * 16, kk, lane);
    if (rs) {
      // the fragment's element j is query row k = kk * 32 + g4 * 8 + j (load_frag): times f of that row, rounded to bf16 exactly as the
      // scale pass rounds (same product, same RNE) -- the image stays as the DMA left it, no pass over it, no barrier behind one
      const f32x4 f0 = *reinterpret_cast<const f32x4*>(fi + kk * 32 + g4 * 8), f1 = *reinterpret_cast<const f32x4*>(fi + kk * 32 + g4 * 8 + 4);
#pragma unroll
      for (int b = 0; b < NBF; ++b) {
        union { bf16x8 v; unsigned w[4]; } u;
        u.v = bf[b];
        u.w[0] = pk_bf16(sk_bf_lo(u.w[0]) * f0[0], sk_bf_hi(u.w[0]) * f0[1]);
        u.w[1] = pk_bf16(sk_bf_lo(u.w[1]) * f0[2], sk_bf_hi(u.w[1]) * f0[3]);
        u.w[2] = pk_bf16(sk_bf_lo(u.w[2]) * f1[0], sk_bf_hi(u.w[2]) * f1[1]);
        u.w[3] = pk_bf16(sk_bf_lo(u.w[3]) * f1[2], sk_bf_hi(u.w[3]) * f1[3]);
        bf[b] = u.v;
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NBF; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
  }
  __syncthreads();  // the images are dead: the fp32 tile takes their place (the tables lie behind it)
  DPRHOT_TMB(1, 4);
  float* const T = reinterpret_cast<float*>(sk_smem);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < NBF; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(wm * 64 + a * 16 + g4 * 4 + r) * SK_DC_TS + wn * (SK_DN / WN) + b * 16 + i16] = acc[a][b][r];
  __syncthreads();
  if (solo) {  // the only row with this gold column: g * q added to that output row in fp32, this thread's 128 / PH columns
    float* const trow0 = T + yrel * SK_DC_TS;
#pragma unroll
    for (int j = 0; j < RC; ++j) {
      // 16-byte chunk c = RC ph + j of image row `row` holds the columns of 32-byte group ((c / 2) ^ mswz(row)), half c & 1
      const int c = ph * RC + j;
      float* const trow = trow0 + (((c >> 1) ^ mswz(row)) << 4) + (c & 1) * 8;
      float4 a = *reinterpret_cast<const float4*>(trow), b = *reinterpret_cast<const float4*>(trow + 4);
      a.x = fmaf(g, sk_bf_lo(raw[j].x), a.x); a.y = fmaf(g, sk_bf_hi(raw[j].x), a.y);
      a.z = fmaf(g, sk_bf_lo(raw[j].y), a.z); a.w = fmaf(g, sk_bf_hi(raw[j].y), a.w);
      b.x = fmaf(g, sk_bf_lo(raw[j].z), b.x); b.y = fmaf(g, sk_bf_hi(raw[j].z), b.y);
      b.z = fmaf(g, sk_bf_lo(raw[j].w), b.z); b.w = fmaf(g, sk_bf_hi(raw[j].w), b.w);
      *reinterpret_cast<float4*>(trow) = a;
      *reinterpret_cast<float4*>(trow + 4) = b;
    }
  }
  if (dupf[0] != 0) {  // rows that share a gold column: one after the other, ascending (bit-reproducible)
    for (int i = 0; i < p.B; ++i) {
      const int m = ym[i];
      if (m >= 0 && cnt[m] > 1) {  // (uniform: every thread reads the same table entries)
        if (tid < SK_DN && c0 + tid < p.d) T[m * SK_DC_TS + tid] = fmaf(gv[i], sk_bf_lo((unsigned)p.Qb[(size_t)i * p.d + c0 + tid]), T[m * SK_DC_TS + tid]);
        __syncthreads();
      }
    }
  }
  __syncthreads();
  DPRHOT_TMB(1, 5);
  if (p.dc_bf16) {  // 8 values per lane: whole 256-byte rows of bf16
    uint16_t* const out = static_cast<uint16_t*>(p.dC);
#pragma unroll
    for (int it = 0; it < SK_COLS * 16 / NT; ++it) {
      const int e = tid + it * NT, mrow = e >> 4, c8 = e & 15;
      const int m = n0 + mrow;
      if (mrow < nvalid && c0 + c8 * 8 < p.d) {
        const float4 a = *reinterpret_cast<const float4*>(T + mrow * SK_DC_TS + c8 * 8);
        const float4 b = *reinterpret_cast<const float4*>(T + mrow * SK_DC_TS + c8 * 8 + 4);
        typedef unsigned sk_u4 __attribute__((ext_vector_type(4)));
        const sk_u4 w = {pk_bf16(a.x * sc, a.y * sc), pk_bf16(a.z * sc, a.w * sc), pk_bf16(b.x * sc, b.y * sc), pk_bf16(b.z * sc, b.w * sc)};
        if (p.nt_store) __builtin_nontemporal_store(w, reinterpret_cast<sk_u4*>(out + (size_t)m * p.d + c0 + c8 * 8));
        else *reinterpret_cast<sk_u4*>(out + (size_t)m * p.d + c0 + c8 * 8) = w;
      }
    }
  } else {
    float* const out = static_cast<float*>(p.dC);
#pragma unroll
    for (int it = 0; it < SK_COLS * 32 / NT; ++it) {
      const int e = tid + it * NT, mrow = e >> 5, cq = e & 31;
      const int m = n0 + mrow;
      if (mrow < nvalid && c0 + cq * 4 < p.d) {
        float4 v = *reinterpret_cast<const float4*>(T + mrow * SK_DC_TS + cq * 4);
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        if (stamp && cq == 0 && m % p.stamp_period == p.stamp_row) v.x = lsum;
        if (p.nt_store) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(out + (size_t)m * p.d + c0 + cq * 4));
        else *reinterpret_cast<float4*>(out + (size_t)m * p.d + c0 + cq * 4) = v;
      }
    }
  }
  if (last_of_rank) {
    // the header rows behind this rank's real rows belong to no tile: zero gradient rows, and the loss stamp of the packed step
    const int hdr = p.p_rows_c - p.p_n_ctx, h0 = (t / p.tiles_per_rank) * p.p_rows_c + p.p_n_ctx;
    for (int e = tid; e < hdr * 32; e += NT) {
      const int m = h0 + (e >> 5), cq = e & 31;
      if (c0 + cq * 4 < p.d) {
        if (p.dc_bf16) {
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.dC) + (size_t)m * p.d + c0 + cq * 4) = make_uint2(0u, 0u);
        } else {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (stamp && cq == 0 && m % p.stamp_period == p.stamp_row) v.x = lsum;
          *reinterpret_cast<float4*>(static_cast<float*>(p.dC) + (size_t)m * p.d + c0 + cq * 4) = v;
        }
      }
    }
  }
}

// Finishing launch of the fused plan: dQ = scale * ( sum_s exp(m_is - lse_i) * slab_s  +  g_i * C[y_i] ), slabs in slice order
// (bit-reproducible).  One workgroup per (row, 128 float4 of the row): the row's logsumexp from its nt tile values (every wave
// derives it: 2 KB, no barrier), the slices' references m_is from the same values -- exactly the maxima the dQ units used -- and the
// gold row of C.  The slab loads are in flight before any of that is consumed.
struct SkFinArgs {
  const float* part;      // [nslices][B][d] slice-normalised slabs (sk_dq_unit_f)
  int nslices, ksteps, nk;
  int B, d;
  const float* tile_lse;
  int nt;
  const float* gold;
  const int64_t* y;
  int64_t y_offset;
  const uint16_t* C;      // [Nc][d]
  float grad_scale, h_scale;
  const float* d_scale;
  float* dQ;
  int parts;              // workgroups per row
  int plain = 0;          // 1: the slabs are plain partial sums of softmax x C (sk_bwdp_kernel): every slice's factor is grad_scale
};

__global__ __launch_bounds__(256) void sk_dq_finish_kernel(SkFinArgs p) {
  __shared__ float s_e[64];
  __shared__ float4 s_t[128];  // the row's tile values (nt <= 512)
  const int tid = threadIdx.x, lane = tid & 63;
  const int row = blockIdx.x / p.parts, part = blockIdx.x - row * p.parts;
  const int nq = p.d >> 2, q4 = part * (int)blockDim.x + tid;  // (one workgroup per row up to d = 1024: a whole number of waves)
  const bool ok = q4 < nq;
  const size_t slab4 = (size_t)p.B * nq, o4 = (size_t)row * nq + (ok ? q4 : 0);
  const float4* const part4 = reinterpret_cast<const float4*>(p.part);
  constexpr int CH = 16;
  float4 v[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) v[u] = u < p.nslices ? part4[(size_t)u * slab4 + o4] : make_float4(0.f, 0.f, 0.f, 0.f);  // (uniform branch)
  const int yg = reinterpret_cast<const int*>(p.y)[2 * row] + (int)p.y_offset;
  const uint2 cg = *reinterpret_cast<const uint2*>(p.C + (size_t)yg * p.d + (ok ? q4 : 0) * 4);
  const float gl = p.gold[row];
  // The slices' references come from the tile values the logsumexp loads anyway, through LDS: a loop of dependent 4-byte global
  // loads per slice thread (up to 8 L2 round trips in front of the barrier everybody waits at) was most of this launch (4.8 us).
  float4 tv[2];
  const float lse = sk_row_lse_tv(p.tile_lse, p.nt, p.B, row, lane, tv);
  if (tid < 64) {
    s_t[lane] = tv[0];
    s_t[lane + 64] = tv[1];
  }
  __syncthreads();
  if (tid < p.nslices) {
    const int s0 = tid * p.ksteps, ns = min(p.ksteps, p.nk - s0);
    float m = -INFINITY;
    if (ns > 0) {
      const int tlo = s0 >> 1, thi = min((s0 + ns - 1) >> 1, p.nt - 1);
      const float* const st = reinterpret_cast<const float*>(s_t);
      for (int t = tlo; t <= thi; ++t) m = fmaxf(m, st[t]);
    }
    s_e[tid] = p.plain ? p.grad_scale : (m == -INFINITY ? 0.f : __expf(m - lse) * p.grad_scale);
  }
  __syncthreads();
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = 0; base < p.nslices; base += CH) {
    if (base > 0) {
#pragma unroll
      for (int u = 0; u < CH; ++u) v[u] = base + u < p.nslices ? part4[(size_t)(base + u) * slab4 + o4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const float e = base + u < p.nslices ? s_e[base + u] : 0.f;
      a.x = fmaf(e, v[u].x, a.x); a.y = fmaf(e, v[u].y, a.y); a.z = fmaf(e, v[u].z, a.z); a.w = fmaf(e, v[u].w, a.w);
    }
  }
  if (ok) {
    const float g = (__expf(gl - lse) - 1.0f) * p.grad_scale;
    const float sc = p.h_scale * (p.d_scale ? *p.d_scale : 1.0f);
    a.x = fmaf(g, sk_bf_lo(cg.x), a.x) * sc;
    a.y = fmaf(g, sk_bf_hi(cg.x), a.y) * sc;
    a.z = fmaf(g, sk_bf_lo(cg.y), a.z) * sc;
    a.w = fmaf(g, sk_bf_hi(cg.y), a.w) * sc;
    reinterpret_cast<float4*>(p.dQ)[(size_t)row * nq + q4] = a;
  }
}

// Finishing role inside the backward launch (round 5; option sk_tail): what sk_dq_finish_kernel does, by the LAST workgroups of the
// grid, behind a count of the dQ units that have published their slabs.  Why inside: as a launch of its own the fold cost 4.6-4.8 us
// plus a 0.5 us gap for 4.9 MB -- one cold start (kernel arguments, first wave, first data ~2 us) behind a boundary that also waits
// for the write-back of the dC units' 25 MB -- on the step's critical path sim -> dQ units -> finish.  Here the finishing workgroups
// are dispatched as soon as the first units of the launch leave (they are the grid's last blocks: every dQ unit has been handed a CU
// before any of them is -- in-order dispatch, the argument of scratch/negative/chain.h), thread 0 polls the count with relaxed
// device-scope loads, and the slabs arrive by sc1 loads from where the dQ units' write-through stores put them: no L2 write-back
// fence, no invalidate.  A wait beyond 20 ms -- a bug, not a load condition -- poisons dQ with NaN.
// One row per 256 threads (d <= 1024), NT / 256 rows per workgroup; LDS: the dynamic array's first bytes (never a second
// __shared__ object next to LDS-DMA code: guide 5, trap 4a).
template <int NW>
__device__ __forceinline__ void sk_fin_unit(const SkBwdFArgs& p, int f, uint16_t* sk_smem) {
  constexpr int NT = NW * 64, RPW = NT / 256;
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 8, t = tid & 255;
  const int row = min(f * RPW + g, p.B - 1);
  const bool live = f * RPW + g < p.B;
  const int nq = p.d >> 2;
  const bool ok = live && t < nq;
  const int q4 = t < nq ? t : 0;
  float* const s_e = reinterpret_cast<float*>(sk_smem) + g * (64 + 512);  // [64] slice factors | [128] float4 tile values of the row
  float4* const s_t = reinterpret_cast<float4*>(s_e + 64);
  const int ndq = p.nslices * (p.d / SK_QN);
  const int nk = p.tiles_per_rank > 0 ? 2 * p.nt : (p.Nc + 63) / 64;
  DPRHOT_TMB(3, 0);
  // everything that does not depend on the slabs first: labels, gold rows, tile values (the sim launch wrote them: a launch ago)
  const int yg = reinterpret_cast<const int*>(p.y)[2 * row] + (int)p.y_offset;
  const uint2 cg = *reinterpret_cast<const uint2*>(p.C + (size_t)yg * p.d + q4 * 4);
  const float gl = p.gold[row];
  float4 tv[2];
  const float lse = sk_row_lse_tv(p.tile_lse, p.nt, p.B, row, lane, tv);
  if ((tid & 255) < 64) {
    s_t[lane] = tv[0];
    s_t[lane + 64] = tv[1];
  }
  __syncthreads();
  if (t < p.nslices) {
    const int s0 = t * p.ksteps, ns = min(p.ksteps, nk - s0);
    float m = -INFINITY;
    if (ns > 0) {
      const int tlo = s0 >> 1, thi = min((s0 + ns - 1) >> 1, p.nt - 1);
      const float* const st = reinterpret_cast<const float*>(s_t);
      for (int tt = tlo; tt <= thi; ++tt) m = fmaxf(m, st[tt]);
    }
    s_e[t] = m == -INFINITY ? 0.f : __expf(m - lse) * p.grad_scale;
  }
  DPRHOT_TMB(3, 1);
  // ---- the wait: every dQ unit has counted itself in
  bool poisoned = false;
  if (tid == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(p.tail_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ndq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 2000000ull) { poisoned = true; break; }  // 20 ms of the 100 MHz clock
    }
    reinterpret_cast<int*>(s_e)[63] = poisoned ? 1 : 0;  // (nslices <= 62 is asserted by the host when this role is on)
    if (p.tail_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  poisoned = reinterpret_cast<const int*>(reinterpret_cast<const float*>(sk_smem))[63] != 0;
  DPRHOT_TMB(3, 2);
  const size_t slab4 = (size_t)p.B * nq, o4 = (size_t)row * nq + q4;
  const float* const base = p.part + o4 * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int CH = 10;
  for (int b0 = 0; b0 < p.nslices; b0 += CH) {
    f32x4 v[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const float* const src = base + (size_t)min(b0 + u, p.nslices - 1) * slab4 * 4;
      v[u] = p.tail_fence ? sk_ld16_hidden(src) : sk_ld16_sc1(src);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < CH; ++u) asm volatile("" : "+v"(v[u]));
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const float e = b0 + u < p.nslices ? s_e[b0 + u] : 0.f;
      a.x = fmaf(e, v[u][0], a.x); a.y = fmaf(e, v[u][1], a.y); a.z = fmaf(e, v[u][2], a.z); a.w = fmaf(e, v[u][3], a.w);
    }
  }
  DPRHOT_TMB(3, 3);
  if (ok) {
    const float gq = (__expf(gl - lse) - 1.0f) * p.grad_scale;
    const float sc = p.h_scale * (p.d_scale ? *p.d_scale : 1.0f);
    a.x = fmaf(gq, sk_bf_lo(cg.x), a.x) * sc;
    a.y = fmaf(gq, sk_bf_hi(cg.x), a.y) * sc;
    a.z = fmaf(gq, sk_bf_lo(cg.y), a.z) * sc;
    a.w = fmaf(gq, sk_bf_hi(cg.y), a.w) * sc;
    if (poisoned) a = make_float4(NAN, NAN, NAN, NAN);
    reinterpret_cast<float4*>(p.dQ)[(size_t)row * nq + q4] = a;
  }
  DPRHOT_TMB(3, 4);
}

// NW = 4: 256 threads.  NW = 8: 512 threads, the same LDS, half the output tile per wave (option sk_w8).  The stamps showed a dQ
// unit's step bound by ONE wave's instruction stream (wait, barrier, 20 LDS reads, 16 MFMAs in a dependent chain: 0.73 us, of which
// 0.11 waiting for the slot), and two dQ workgroups sharing a CU running at the speed of one: the CU has issue room for twice the waves.
template <int NG, int NW, bool ATOMIC = false>
__global__ __launch_bounds__(NW * 64, NW / 2) void sk_bwdf_kernel(SkBwdFArgs p) {  // (second argument: waves per SIMD, two workgroups per CU either way)
  extern __shared__ __attribute__((aligned(16))) uint16_t sk_smem[];
  const int b = blockIdx.x;
  const int ndq = p.nslices * (p.d / SK_QN);
  if (b < p.ndq_pad) {
    if (b >= ndq || (p.dbg & 2)) return;  // padding
    sk_dq_unit_f<NG, NW, ATOMIC>(p, sk_xcd_order(b, ndq), sk_smem);
  } else if (b < (int)gridDim.x - p.nfin) {
    if (p.dbg & 1) return;
    sk_dc_unit_f<NG, NW>(p, sk_xcd_order(b - p.ndq_pad, (int)gridDim.x - p.nfin - p.ndq_pad), sk_smem);
  } else {
    sk_fin_unit<NW>(p, b - ((int)gridDim.x - p.nfin), sk_smem);
  }
}


// =====================================================================================================================
// Round 4, second form: the backward launch with ONE kind of unit  (sk_sim_kernel with P  ->  sk_bwdp_kernel  ->  sk_dq_finish_kernel)
// =====================================================================================================================
// With B <= 128 a statistics tile of P -- [B rows][128 contexts] -- is complete in BOTH directions the backward needs it: its columns
// are all the scores of 128 contexts (dC of those contexts, given the Q columns), its rows are whole K ranges of the dQ product (a
// partial dQ, given the same contexts' C rows).  sk_bwdf_kernel pulls every P tile through the texture path 6 + 12 times (one dC unit
// per 128 d columns, one dQ unit per 64) and its dQ units are latency chains of 13 steps; here a unit is
//     (slice of up to SK_PT consecutive statistics tiles) x (64 columns of d)
// and per tile it loads P [128][128] and C [128][64] ONCE (48 KiB), scales the rows of P by f_it = exp(lse_it - lse_i) in LDS -- the
// row softmax in bf16, what the dScores launch used to write -- and multiplies twice: dQ += P' C (accumulated over the slice's tiles
// in registers) and dC_t = P'^T Q (stored per tile; Q [128][64] is loaded once per unit, unscaled: the gold term g_i q_i is added to
// the staged fp32 tile from it).  The slabs are then plain partial sums: the finishing launch adds them and the gold rows of C.
// One workgroup of eight waves per CU (152 KiB of LDS: both operands double-buffered, the next tile's DMAs fly under this tile's
// MFMAs), 16 slices x d / 64 units (192 at d = 768).  Texture-path bytes per launch at cfg3 per rank: 71 MB against 91.
// LDS stores next to the DMAs in flight are asm (sk_lds_st*): hipcc would drain the queue in front of each plain one.
constexpr int SK_PT = 8;                            // statistics tiles a unit may cover (its factor table)
constexpr int SK_P_IMG = SK_MAXB * SK_COLS;         // elements of a P tile image   [128 rows][128 contexts], 32-byte groups at g ^ mswz(row)
constexpr int SK_P_CIMG = SK_COLS * SK_QN;          // elements of a C / Q image    [128 k][64 columns], 32-byte groups at g ^ sk_swz64(k)
constexpr int SK_P_TS = SK_QN + 4;                  // row stride of the fp32 staging tile
constexpr int SK_P_OFF_C = 2 * SK_P_IMG * 2;        // byte offsets: two P images | two C images | Q | staging tile | tables
constexpr int SK_P_OFF_Q = SK_P_OFF_C + 2 * SK_P_CIMG * 2;
constexpr int SK_P_OFF_T = SK_P_OFF_Q + SK_P_CIMG * 2;
constexpr int SK_P_OFF_TAB = SK_P_OFF_T + SK_COLS * SK_P_TS * 4;
constexpr int SK_P_TABLES = SK_PT * 128 * 4 + 4 * 128 * 4 + 16;  // f [SK_PT][128] | g, row loss, gold column, count [128] | flag
inline size_t sk_bwdp_lds() { return (size_t)SK_P_OFF_TAB + SK_P_TABLES; }

__device__ __forceinline__ void sk_lds_st128(void* lds_ptr, uint4 v) {
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  const unsigned addr = (unsigned)(uintptr_t)(lds_u32*)lds_ptr;
  typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
  const sk_u32x4 val = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" ::"v"(addr), "v"(val) : "memory");
}
__device__ __forceinline__ void sk_lds_add32(void* lds_ptr, unsigned v) {
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  const unsigned addr = (unsigned)(uintptr_t)(lds_u32*)lds_ptr;
  asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// LDS READS next to the DMAs in flight are asm as well: hipcc orders a plain ds_read behind every LDS-DMA that may have written its
// address -- s_waitcnt vmcnt(0) in front of each (seen in this kernel's first build: the next tile's DMAs were drained six times
// per step).  The results are valid behind SK_LGKM0() only (the compiler does not know the asm is a load).
__device__ __forceinline__ unsigned sk_lds_addr(const void* lds_ptr) {
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  return (unsigned)(uintptr_t)(lds_u32*)const_cast<void*>(lds_ptr);
}
typedef unsigned sk_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sk_lds_ld128_async(sk_u32x4_t& r, const void* lds_ptr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(sk_lds_addr(lds_ptr)) : "memory");
}
__device__ __forceinline__ void sk_lds_ld32_async(unsigned& r, const void* lds_ptr) {
  asm volatile("ds_read_b32 %0, %1" : "=v"(r) : "v"(sk_lds_addr(lds_ptr)) : "memory");
}
#define SK_LGKM0()                                        \
  do {                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
// transposed 8-byte read pair (rows k, k + 4 of a k-major image): the two halves of an MFMA operand fragment; OFF = byte distance of 4 rows
template <int OFF>
__device__ __forceinline__ void sk_tr_pair_async(bf16x4& lo, bf16x4& hi, const uint16_t* lds_ptr) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(sk_lds_addr(lds_ptr)), "n"(OFF) : "memory");
}
__device__ __forceinline__ bf16x8 sk_join(const bf16x4& lo, const bf16x4& hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// SkBwdFArgs as for sk_bwdf_kernel, with ksteps = statistics tiles per unit and nslices = slices (slabs).
template <int NG>
__global__ __launch_bounds__(512, 2) void sk_bwdp_kernel(SkBwdFArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t sk_smem[];
  char* const smem = reinterpret_cast<char*>(sk_smem);
  uint16_t* const Pimg = sk_smem;
  uint16_t* const Cimg = reinterpret_cast<uint16_t*>(smem + SK_P_OFF_C);
  uint16_t* const Qs = reinterpret_cast<uint16_t*>(smem + SK_P_OFF_Q);
  float* const T = reinterpret_cast<float*>(smem + SK_P_OFF_T);
  float* const ftab = reinterpret_cast<float*>(smem + SK_P_OFF_TAB);  // [SK_PT][128]: exp(lse_it - lse_i), 0 for rows >= B
  float* const gv = ftab + SK_PT * 128;                               // g of the row
  float* const rl = gv + 128;                                         // row loss
  int* const ym = reinterpret_cast<int*>(rl + 128);                   // gold column (global), or -1
  int* const cnt = ym + 128;                                          // rows whose gold column is context m of the current tile
  int* const dupf = cnt + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, g4 = lane >> 4;
  const int ndt = p.d / SK_QN;
  const int unit = sk_xcd_order(blockIdx.x, gridDim.x);  // (the d tiles of a slice -- one P tile sequence -- on one XCD)
  const int dt = unit % ndt, sl = unit / ndt, c0 = dt * SK_QN;
  const int t_lo = sl * p.ksteps, ntl = min(p.ksteps, p.nt - t_lo);
  DPRHOT_TMB(1, 0);

  // ---- hidden loads first: the row's tile values (four threads per row), this slice's tile values, gold logit, label
  const int prow = min(tid >> 2, p.B - 1), ph = tid & 3;
  SkRowStats<NG / 2, 4> st;
  st.issue(p.tile_lse, p.nt, p.B, prow, ph);
  float lt[SK_PT / 4];
#pragma unroll
  for (int u = 0; u < SK_PT / 4; ++u) {
    const int t = min(t_lo + ph + 4 * u, p.nt - 1);
    lt[u] = sk_ld4_hidden(p.tile_lse + ((size_t)(t >> 2) * p.B + prow) * 4 + (t & 3));
  }
  float gl = sk_ld4_hidden(p.gold + prow);
  float yf = sk_ld4_hidden(reinterpret_cast<const int*>(p.y) + 2 * prow);

  // ---- DMA coordinates: P image pieces of 4 rows x 256 bytes (4 per wave), C / Q image pieces of 8 rows x 128 bytes (2 per wave)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  unsigned p_row[4];
  int p_col[4], c_row[2];
  unsigned c_col[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 4 + (lane >> 4), pos = lane & 15;
    p_row[j] = (unsigned)min(row, p.B - 1) * (unsigned)p.Nc;
    p_col[j] = (((pos >> 1) ^ mswz(row)) << 4) + (pos & 1) * 8;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int krow = (wave * 2 + j) * 8 + (lane >> 3), pos = lane & 7;
    c_row[j] = krow;
    c_col[j] = (unsigned)(c0 + ((((pos >> 1) ^ sk_swz64(krow)) << 4) + (pos & 1) * 8));
  }
  auto issue_tile = [&](int t, int buf) {
    const int n0 = sk_tile_col(p, t);
    uint16_t* const Pd = Pimg + buf * SK_P_IMG;
    uint16_t* const Cd = Cimg + buf * SK_P_CIMG;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.P + p_row[j] + min(n0 + p_col[j], p.Nc - 8)), (g2_lds_ptr*)(Pd + (wave_u * 4 + j) * 512), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.C + (unsigned)min(n0 + c_row[j], p.Nc - 1) * (unsigned)p.d + c_col[j]),
                                       (g2_lds_ptr*)(Cd + (wave_u * 2 + j) * 512), 16, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < 2; ++j)
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(p.Qb + (unsigned)min(c_row[j], p.B - 1) * (unsigned)p.d + c_col[j]),
                                     (g2_lds_ptr*)(Qs + (wave_u * 2 + j) * 512), 16, 0, 0);
  issue_tile(t_lo, 0);
  DPRHOT_TMB(1, 1);
  sk_wait_vm<8>();  // the hidden loads have landed (eight DMAs of this wave are younger)
  __builtin_amdgcn_sched_barrier(0);
  st.pin();
#pragma unroll
  for (int u = 0; u < SK_PT / 4; ++u) asm volatile("" : "+v"(lt[u]));
  asm volatile("" : "+v"(gl), "+v"(yf));
  const float lse = st.finish(p.nt, ph);
  const int row = tid >> 2;
  {
#pragma unroll
    for (int u = 0; u < SK_PT / 4; ++u) {
      const int tl = ph + 4 * u;
      const float w = (row < p.B && tl < ntl && lt[u] != -INFINITY) ? __expf(lt[u] - lse) : 0.f;
      sk_lds_st32(ftab + tl * 128 + row, __float_as_uint(w));
    }
    if (ph == 0) {
      const bool real = row < p.B;
      const float g = (__expf(gl - lse) - 1.0f) * p.grad_scale;
      sk_lds_st32(gv + row, __float_as_uint(real ? g : 0.f));
      sk_lds_st32(rl + row, __float_as_uint(real ? lse - gl : 0.f));
      sk_lds_st32(ym + row, (unsigned)(real ? __float_as_int(yf) + (int)p.y_offset : -1));
      if (unit == 0 && real) {
        if (p.row_loss) p.row_loss[row] = lse - gl;
        if (p.row_lse) p.row_lse[row] = lse;
      }
    }
    if (tid == 0) sk_lds_st32(dupf, 0u);
  }
  DPRHOT_TMB(1, 2);
  const float sc = p.h_scale * (p.d_scale ? *p.d_scale : 1.0f);
  const bool stamp = dt == 0 && p.stamp_period > 0 && !p.dc_bf16;
  float lsum = 0.f;

  const int wr = wave >> 1, wc = wave & 1;  // both products: four groups of 32 output rows x two of 32 columns
  f32x4 dq[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) dq[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the staged dC tile of the tile before (rows n0p .. n0p + nvp): gold rows that share a column, then the stores -- issued at the
  // START of the next tile's step, so that the wait for that tile's operands does not wait for these stores' acknowledgements
  auto flush_tile = [&](int tp) {
    const int n0p = sk_tile_col(p, tp);
    const int nvp = p.tiles_per_rank > 0 ? SK_COLS : min(SK_COLS, p.Nc - n0p);
    unsigned dupv;
    sk_lds_ld32_async(dupv, dupf);
    SK_LGKM0();
    if (dupv != 0) {  // rows that share a gold column: one after the other, ascending (bit-reproducible); rare
      for (int i = 0; i < p.B; ++i) {
        const int m = ym[i] - n0p;
        if (m >= 0 && m < nvp && cnt[m] > 1) {  // (uniform: every thread reads the same table entries)
          if (tid < SK_QN) {
            const float qv = sk_bf_lo((unsigned)Qs[i * SK_QN + (((tid >> 4) ^ sk_swz64(i)) << 4) + (tid & 15)]);
            sk_lds_st32(T + m * SK_P_TS + tid, __float_as_uint(fmaf(gv[i], qv, T[m * SK_P_TS + tid])));
          }
          sk_barrier();
        }
      }
      if (tid == 0) sk_lds_st32(dupf, 0u);
      sk_barrier();
    }
    if (p.dc_bf16) {
      uint16_t* const out = static_cast<uint16_t*>(p.dC);
      sk_u32x4_t ta[SK_COLS * 8 / 512], tb[SK_COLS * 8 / 512];
#pragma unroll
      for (int it = 0; it < SK_COLS * 8 / 512; ++it) {
        const int e = tid + it * 512, mrow = e >> 3, c8 = e & 7;
        sk_lds_ld128_async(ta[it], T + mrow * SK_P_TS + c8 * 8);
        sk_lds_ld128_async(tb[it], T + mrow * SK_P_TS + c8 * 8 + 4);
      }
      SK_LGKM0();
#pragma unroll
      for (int it = 0; it < SK_COLS * 8 / 512; ++it) {
        const int e = tid + it * 512, mrow = e >> 3, c8 = e & 7;
        if (mrow < nvp) {
          const float4 a = make_float4(__uint_as_float(ta[it][0]), __uint_as_float(ta[it][1]), __uint_as_float(ta[it][2]), __uint_as_float(ta[it][3]));
          const float4 b = make_float4(__uint_as_float(tb[it][0]), __uint_as_float(tb[it][1]), __uint_as_float(tb[it][2]), __uint_as_float(tb[it][3]));
          typedef unsigned sk_u4 __attribute__((ext_vector_type(4)));
          const sk_u4 w = {pk_bf16(a.x * sc, a.y * sc), pk_bf16(a.z * sc, a.w * sc), pk_bf16(b.x * sc, b.y * sc), pk_bf16(b.z * sc, b.w * sc)};
          sk_u4* const dst = reinterpret_cast<sk_u4*>(out + (size_t)(n0p + mrow) * p.d + c0 + c8 * 8);
          if (p.nt_store) __builtin_nontemporal_store(w, dst);
          else *dst = w;
        }
      }
    } else {
      float* const out = static_cast<float*>(p.dC);
      sk_u32x4_t tv[SK_COLS * 16 / 512];
#pragma unroll
      for (int it = 0; it < SK_COLS * 16 / 512; ++it) {
        const int e = tid + it * 512, mrow = e >> 4, cq = e & 15;
        sk_lds_ld128_async(tv[it], T + mrow * SK_P_TS + cq * 4);
      }
      SK_LGKM0();
#pragma unroll
      for (int it = 0; it < SK_COLS * 16 / 512; ++it) {
        const int e = tid + it * 512, mrow = e >> 4, cq = e & 15;
        const int m = n0p + mrow;
        if (mrow < nvp) {
          float4 v = make_float4(__uint_as_float(tv[it][0]), __uint_as_float(tv[it][1]), __uint_as_float(tv[it][2]), __uint_as_float(tv[it][3]));
          v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
          if (stamp && cq == 0 && m % p.stamp_period == p.stamp_row) v.x = lsum;
          f32x4* const dst = reinterpret_cast<f32x4*>(out + (size_t)m * p.d + c0 + cq * 4);
          if (p.nt_store) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, dst);
          else *dst = f32x4{v.x, v.y, v.z, v.w};
        }
      }
    }
    if (p.tiles_per_rank > 0 && tp % p.tiles_per_rank == p.tiles_per_rank - 1) {
      // the header rows behind this rank's real rows belong to no tile: zero gradient rows, and the loss stamp of the packed step
      const int hdr = p.p_rows_c - p.p_n_ctx, h0 = (tp / p.tiles_per_rank) * p.p_rows_c + p.p_n_ctx;
      for (int e = tid; e < hdr * 16; e += 512) {
        const int m = h0 + (e >> 4), cq = e & 15;
        if (p.dc_bf16) {
          *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.dC) + (size_t)m * p.d + c0 + cq * 4) = make_uint2(0u, 0u);
        } else {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (stamp && cq == 0 && m % p.stamp_period == p.stamp_row) v.x = lsum;
          *reinterpret_cast<float4*>(static_cast<float*>(p.dC) + (size_t)m * p.d + c0 + cq * 4) = v;
        }
      }
    }
  };

#if defined(DPRHOT_TIMING) && DPRHOT_TIMING >= 2
  unsigned long long pk_[7] = {0, 0, 0, 0, 0, 0, 0}, pl_ = wall_clock64();  // thread 0's view of a tile step
#define SK_PLT(i) do { const unsigned long long t_ = wall_clock64(); pk_[i] += t_ - pl_; pl_ = t_; } while (0)
#else
#define SK_PLT(i) do {} while (0)
#endif
  for (int ti = 0; ti < ntl; ++ti) {
    const int t = t_lo + ti, buf = ti & 1;
    const int n0 = sk_tile_col(p, t);
    const int nvalid = p.tiles_per_rank > 0 ? SK_COLS : min(SK_COLS, p.Nc - n0);
    uint16_t* const Pb = Pimg + buf * SK_P_IMG;
    const uint16_t* const Cb = Cimg + buf * SK_P_CIMG;
    SK_PLT(6);
    sk_wait_vm<0>();  // this tile's operands have landed (and the stores issued in front of them are acknowledged)
    SK_PLT(0);
    sk_barrier();     // ... everybody's have; the tile before is multiplied, staged and patched
    SK_PLT(1);
    // (No DMA is in flight between here and the refill below, but hipcc cannot know: it put s_waitcnt vmcnt(0) in front of every
    //  plain LDS read of the store loop -- each store waited for the one before it.  Every LDS access of the loop is asm.)
    unsigned ymr_u, gr_u;
    sk_lds_ld32_async(ymr_u, ym + row);
    sk_lds_ld32_async(gr_u, gv + row);
    if (ti == 0) {
      if (stamp || unit == 0) {  // loss numerator, fixed order (every wave computes the same bits)
        lsum = sk_loss_sum(rl, SK_MAXB, lane) * p.loss_scale;
        if (unit == 0 && tid == 0) p.loss_sum[0] = lsum;
      }
    } else {
      flush_tile(t - 1);
    }
    SK_PLT(2);
    // ---- P rows x f in place (bf16 both ways; columns beyond the matrix become 0); the gold-column counts start at 0
    {
      uint4* const img = reinterpret_cast<uint4*>(Pb);
      constexpr int NJ = SK_P_IMG / 8 / 512;
      sk_u32x4_t pw[NJ];
      unsigned fru[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = tid + j * 512;
        sk_lds_ld128_async(pw[j], img + c);
        sk_lds_ld32_async(fru[j], ftab + ti * 128 + (c >> 4));
      }
      SK_LGKM0();
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = tid + j * 512, r = c >> 4, slot = c & 15;
        const int ctx0 = (((((slot >> 1) ^ mswz(r)) << 1) | (slot & 1))) << 3;  // first context of this 16-byte chunk
        const float fr = __uint_as_float(fru[j]);
        uint4 w;
        w.x = pk_bf16(sk_bf_lo(pw[j][0]) * fr, sk_bf_hi(pw[j][0]) * fr);
        w.y = pk_bf16(sk_bf_lo(pw[j][1]) * fr, sk_bf_hi(pw[j][1]) * fr);
        w.z = pk_bf16(sk_bf_lo(pw[j][2]) * fr, sk_bf_hi(pw[j][2]) * fr);
        w.w = pk_bf16(sk_bf_lo(pw[j][3]) * fr, sk_bf_hi(pw[j][3]) * fr);
        if (ctx0 >= nvalid) w = make_uint4(0u, 0u, 0u, 0u);  // (whatever the clamped source held: not even 0 x inf)
        sk_lds_st128(img + c, w);
      }
      if (tid < SK_COLS) sk_lds_st32(cnt + tid, 0u);
    }
    const int ymr = (int)ymr_u;  // (valid since the SK_LGKM0 above)
    const float gr = __uint_as_float(gr_u);
    sk_barrier();
    SK_PLT(3);
    if (ti + 1 < ntl) issue_tile(t + 1, buf ^ 1);  // (the other buffers: last read before this step's first barrier)
    // ---- from here to the end of the step every LDS access is asm (the DMAs above are in flight)
    const int yrel = ymr - n0;  // this row's gold column relative to the tile (rows >= B: far negative)
    const bool mine = yrel >= 0 && yrel < nvalid;
    if (mine && ph == 0) sk_lds_add32(cnt + yrel, 1u);
    // ---- dQ += P' C  and  dC_t = P'^T Q
    f32x4 dc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) dc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < SK_COLS / 32; ++kk) {
      sk_u32x4_t aqr[2];
      bf16x4 acl[2], ach[2], bql[2], bqh[2], bcl[2], bch[2];
      const int k = kk * 32 + g4 * 8 + (i16 >> 2);  // (transposed reads: rows k and k + 4 of a k-major image)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int r = wr * 32 + a * 16 + i16, kc = kk * 4 + g4;  // 16-byte chunk kc of row r: contexts 8 kc ..
        sk_lds_ld128_async(aqr[a], Pb + r * SK_COLS + ((((kc >> 1) ^ mswz(r)) << 1 | (kc & 1)) << 3));
        // P as [k = query row][m = context]: the 16 contexts of block 2 wr + a
        sk_tr_pair_async<4 * SK_COLS * 2>(acl[a], ach[a], Pb + k * SK_COLS + (((wr * 2 + a) ^ mswz(k)) << 4) + (i16 & 3) * 4);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int off = (((wc * 2 + b) ^ sk_swz64(k)) << 4) + (i16 & 3) * 4;
        sk_tr_pair_async<4 * SK_QN * 2>(bql[b], bqh[b], Cb + k * SK_QN + off);
        sk_tr_pair_async<4 * SK_QN * 2>(bcl[b], bch[b], Qs + k * SK_QN + off);
      }
      SK_LGKM0();
      bf16x8 aq[2], ac[2], bq[2], bc[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        aq[a] = *reinterpret_cast<const bf16x8*>(&aqr[a]);
        ac[a] = sk_join(acl[a], ach[a]);
        bq[a] = sk_join(bql[a], bqh[a]);
        bc[a] = sk_join(bcl[a], bch[a]);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          dq[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[a], bq[b], dq[a][b], 0, 0, 0);
          dc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[a], bc[b], dc[a][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sk_lds_st32(T + (wr * 32 + a * 16 + g4 * 4 + r) * SK_P_TS + (wc * 2 + b) * 16 + i16, __float_as_uint(dc[a][b][r] * p.grad_scale));
    SK_PLT(4);
    sk_barrier();
    SK_PLT(5);
    // ---- the gold term: g_i q_i added to the staged row of the row's gold column (fp32; four threads per row, 16 columns each)
    {
      unsigned cn = 0;
      sk_u32x4_t qv[2], tv[4];
      float* const trow = T + (mine ? yrel : 0) * SK_P_TS + ph * 16;
      if (mine) {
        sk_lds_ld32_async(cn, cnt + yrel);
        const uint16_t* const qsrc = Qs + row * SK_QN + ((ph ^ sk_swz64(row)) << 4);
        sk_lds_ld128_async(qv[0], qsrc);
        sk_lds_ld128_async(qv[1], qsrc + 8);
#pragma unroll
        for (int h = 0; h < 4; ++h) sk_lds_ld128_async(tv[h], trow + h * 4);
      }
      SK_LGKM0();
      if (mine) {
        if (cn == 1) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            sk_u32x4_t a = tv[2 * h], b = tv[2 * h + 1];
            const sk_u32x4_t q = qv[h];
            a[0] = __float_as_uint(fmaf(gr, sk_bf_lo(q[0]), __uint_as_float(a[0]))); a[1] = __float_as_uint(fmaf(gr, sk_bf_hi(q[0]), __uint_as_float(a[1])));
            a[2] = __float_as_uint(fmaf(gr, sk_bf_lo(q[1]), __uint_as_float(a[2]))); a[3] = __float_as_uint(fmaf(gr, sk_bf_hi(q[1]), __uint_as_float(a[3])));
            b[0] = __float_as_uint(fmaf(gr, sk_bf_lo(q[2]), __uint_as_float(b[0]))); b[1] = __float_as_uint(fmaf(gr, sk_bf_hi(q[2]), __uint_as_float(b[1])));
            b[2] = __float_as_uint(fmaf(gr, sk_bf_lo(q[3]), __uint_as_float(b[2]))); b[3] = __float_as_uint(fmaf(gr, sk_bf_hi(q[3]), __uint_as_float(b[3])));
            sk_lds_st128(trow + h * 8, make_uint4(a[0], a[1], a[2], a[3]));
            sk_lds_st128(trow + h * 8 + 4, make_uint4(b[0], b[1], b[2], b[3]));
          }
        } else if (ph == 0) {
          sk_lds_st32(dupf, 1u);  // (read behind the next barrier: flush_tile)
        }
      }
    }
  }
  DPRHOT_TMB(1, 3);
#if defined(DPRHOT_TIMING) && DPRHOT_TIMING >= 2
  SK_PLT(6);
  if (threadIdx.x == 0 && blockIdx.x < 4096) {  // (the records of unit kind 2 are free in this launch)
    for (int i = 0; i < 7; ++i) g_dprhot_tmb[(2 * 4096 + blockIdx.x) * 8 + i] = pk_[i];
    g_dprhot_tmb[(2 * 4096 + blockIdx.x) * 8 + 7] = (unsigned long long)ntl;
  }
#endif
  sk_wait_vm<0>();
  sk_barrier();
  flush_tile(t_lo + ntl - 1);  // (its first statement is a plain LDS read: see the loop's first barrier)
  DPRHOT_TMB(1, 4);
  sk_barrier();  // the staging tile is free: the slab goes through it
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) sk_lds_st32(T + (wr * 32 + a * 16 + g4 * 4 + r) * SK_P_TS + (wc * 2 + b) * 16 + i16, __float_as_uint(dq[a][b][r]));
  sk_barrier();
  {
    float* const out = p.part + (size_t)sl * p.B * p.d;
    sk_u32x4_t tv[SK_MAXB * 16 / 512];
#pragma unroll
    for (int it = 0; it < SK_MAXB * 16 / 512; ++it) {
      const int e = tid + it * 512;
      sk_lds_ld128_async(tv[it], T + (e >> 4) * SK_P_TS + (e & 15) * 4);
    }
    SK_LGKM0();
#pragma unroll
    for (int it = 0; it < SK_MAXB * 16 / 512; ++it) {
      const int e = tid + it * 512, r = e >> 4, cq = e & 15;
      if (r < p.B) {
        const f32x4 v = {__uint_as_float(tv[it][0]), __uint_as_float(tv[it][1]), __uint_as_float(tv[it][2]), __uint_as_float(tv[it][3])};
        f32x4* const dst = reinterpret_cast<f32x4*>(out + (size_t)r * p.d + c0 + cq * 4);
        if (p.nt_store) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
    }
  }
  DPRHOT_TMB(1, 5);
}

}  // namespace dprhot
