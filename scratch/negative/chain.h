// chain.h -- the latency-bound training step (BASELINE cfg2: 32 x 256 x 768; dpr_task.py:197-212 and its backward) in ONE launch.
//
// step_small.h needs the complete logit rows before it can form one dScore, and the logits are a reduction over d that no single
// workgroup can afford (every workgroup would read all of q and c): two dependent phases.  As two launches the phases cost two
// host-side submissions per step -- and the step's eager issue loop (one C call per step, what a binding does) is host-bound on
// slower hosts: 11.1 us per step next to 9.3 us of device time.  Here the phases are two ROLES of one grid:
//
//   workgroups [0, n_sim_pad)   sim units: 32 query rows x 32 contexts x HALF of d each.  fp32 rows in, bf16 (RNE) images to LDS
//                               and to Qb / Cb, one 16x16x32 MFMA k step per wave, the waves' partial tiles summed in wave order,
//                               1/T and the column mask applied, one fp32 slab of partial logits per half of d.  Then: every
//                               thread's stores made visible device-wide (release fence), one atomic add on the slot's counter.
//   workgroups [n_sim_pad, ..)  the column tiles of step_small.h, unchanged, behind a wait: thread 0 polls the counter until all
//                               sim units have signalled, acquire fence, then softmax-CE, dScores, dQ and dC as before.
//
// No deadlock by construction: sim units never wait; the hardware dispatches a queue's workgroups in index order (per XCD), so every
// sim unit of this launch has been handed a CU before the first waiting workgroup of its XCD is; a waiting workgroup only ever
// waits for workgroups that are running or next in line.  (A co-resident grid barrier has no such order and deadlocks when two
// processes share the GPU; a cooperative launch costs 37 us: DESIGN.md section 5.)  A wait that exceeds 10 ms -- a bug, not a load
// condition -- poisons the slot: the loss is published as NaN from then on.
//
// The counter lives in a library-owned slot (64 bytes of a per-device pool, one slot per workspace pointer: dprhot.hip), zero between
// launches: the last finishing tile workgroup resets it.  Nothing is asked of the caller's workspace.
#pragma once
#include "step_small.h"

namespace dprhot {

struct ChainArgs {
  const float* q;          // [B][d] fp32
  const float* c;          // [Nc][d] fp32
  uint16_t* Qb;            // [B][d] bf16 out
  uint16_t* Cb;            // [Nc][d] bf16 out
  int B, Nc, d;
  const uint8_t* colmask;  // [Nc] or nullptr
  float inv_T;
  float* slabs;            // [2][B][Nc] partial logits (what StepSmallArgs::slabs reads)
  size_t slab_stride;
  int n_sim;               // 2 * ceil(Nc / 32) sim units ...
  int n_sim_pad;           // ... rounded up to a multiple of 8 (the tile workgroups keep index % 8 == XCD)
  unsigned* slot;          // [0] sim units done, [1] tile workgroups done, [2] poison
};

constexpr int CH_ROWS = 32, CH_COLS = 32;
constexpr int CH_PS = 36;  // row stride (floats) of a wave's partial tile
inline size_t chain_sim_lds(int d) {
  const int kh = d / 2;
  return (size_t)2 * CH_ROWS * (kh + 8) * 2 + (size_t)16 * CH_ROWS * CH_PS * sizeof(float);
}

// d % 64 == 0, d <= 1024: a half of d is at most 16 MFMA k steps (one per wave)
__device__ __forceinline__ void chain_sim_unit(const ChainArgs& p, const int unit, uint16_t* const smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = unit >> 1, half = unit & 1;
  const int kh = p.d >> 1, k0 = half * kh, n0 = ct * CH_COLS;
  const int stride = kh + 8;                       // image row stride (elements)
  uint16_t* const As = smem;                       // [32][stride]  q rows, this half of d
  uint16_t* const Bs = As + CH_ROWS * stride;      // [32][stride]  context rows
  float* const P = reinterpret_cast<float*>(Bs + CH_ROWS * stride);  // [16][32][CH_PS] partial tiles

  // ---- fp32 rows -> bf16 images (LDS and Qb / Cb): 4 values per thread and piece, all loads issued before the first use
  const int cpr = kh >> 2;                         // float4 pieces per row
  const int total = CH_ROWS * cpr;                 // <= 4096: at most 4 pieces per thread and operand
  uint4 av[4], bv[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + it * 1024;
    av[it] = make_uint4(0u, 0u, 0u, 0u);
    bv[it] = make_uint4(0u, 0u, 0u, 0u);
    if (idx < total) {
      const int row = idx / cpr, c4 = idx - row * cpr;
      av[it] = *reinterpret_cast<const uint4*>(p.q + (size_t)min(row, p.B - 1) * p.d + k0 + c4 * 4);
      bv[it] = *reinterpret_cast<const uint4*>(p.c + (size_t)min(n0 + row, p.Nc - 1) * p.d + k0 + c4 * 4);
    }
  }
  uint8_t mraw = 0;
  const int mrow = tid >> 5, mcol = tid & 31;      // this thread's element of the 32 x 32 output
  if (p.colmask != nullptr) mraw = p.colmask[min(n0 + mcol, p.Nc - 1)];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + it * 1024;
    if (idx < total) {
      const int row = idx / cpr, c4 = idx - row * cpr;
      const uint2 a = make_uint2(pack_bf16_rne(av[it].x, av[it].y), pack_bf16_rne(av[it].z, av[it].w));
      const uint2 b = make_uint2(pack_bf16_rne(bv[it].x, bv[it].y), pack_bf16_rne(bv[it].z, bv[it].w));
      *reinterpret_cast<uint2*>(As + row * stride + c4 * 4) = a;
      *reinterpret_cast<uint2*>(Bs + row * stride + c4 * 4) = b;
      if (ct == 0 && row < p.B) *reinterpret_cast<uint2*>(p.Qb + (size_t)row * p.d + k0 + c4 * 4) = a;
      if (n0 + row < p.Nc) *reinterpret_cast<uint2*>(p.Cb + (size_t)(n0 + row) * p.d + k0 + c4 * 4) = b;
    }
  }
  __syncthreads();

  // ---- wave w: k step w of this half (32 deep), the whole 32 x 32 tile
  const int i16 = lane & 15, g4 = lane >> 4;
  const int ksteps = kh >> 5;
  if (wave < ksteps) {
    bf16x8 af[2], bf[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const bf16x8*>(As + (a * 16 + i16) * stride + wave * 32 + g4 * 8);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const bf16x8*>(Bs + (b * 16 + i16) * stride + wave * 32 + g4 * 8);
    float* const T = P + wave * (CH_ROWS * CH_PS);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(a * 16 + g4 * 4 + r) * CH_PS + b * 16 + i16] = acc[r];
      }
  }
  __syncthreads();

  // ---- the k steps' partial tiles in wave order, 1/T, mask -> this half's slab
  float s = P[mrow * CH_PS + mcol];
  for (int w = 1; w < ksteps; ++w) s += P[w * (CH_ROWS * CH_PS) + mrow * CH_PS + mcol];
  const bool masked = p.colmask != nullptr && mraw != 0;
  if (mrow < p.B && n0 + mcol < p.Nc) p.slabs[(size_t)half * p.slab_stride + (size_t)mrow * p.Nc + n0 + mcol] = masked ? -INFINITY : s * p.inv_T;
}

template <int CPT>
__global__ __launch_bounds__(1024) void step_chain_kernel(ChainArgs c, StepSmallArgs p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t ss_smem[];
  const int b = blockIdx.x;
  if (b < c.n_sim_pad) {
    if (b >= c.n_sim) return;  // padding
    chain_sim_unit(c, b, ss_smem);
    // Every wave waits until its own stores have been taken by this XCD's L2 (vmcnt counts stores), the barrier collects the waves,
    // and ONE release by thread 0 writes the L2's dirty lines back to where the other XCDs read from before the counter moves.
    // (A release fence per wave -- __threadfence() in all 1024 threads -- is sixteen L2 write-backs per unit: the launch took 29 us.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&c.slot[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(&c.slot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)c.n_sim) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 1000000ull) {  // 10 ms of the 100 MHz clock: not a load condition
        __hip_atomic_store(&c.slot[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one invalidate for the workgroup: L1 is the CU's, L2 the XCD's; the barrier hands it on
  }
  __syncthreads();
  const int ntile = (int)gridDim.x - c.n_sim_pad;
  step_small_body<CPT, 16, 2, 1>(p, b - c.n_sim_pad, ntile, ss_smem);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(&c.slot[1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)ntile - 1) {  // the last one out: every workgroup is past its wait -- the slot is zero again for the next launch
      if (__hip_atomic_load(&c.slot[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) p.loss_sum[0] = NAN;  // (sticky)
      __hip_atomic_store(&c.slot[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&c.slot[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace dprhot
