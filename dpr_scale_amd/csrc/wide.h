// wide.h -- the sim launch for vocabulary-wide vectors (CITADEL router loss, dpr_scale/task/citadel_task.py:249-262: d = 30522):
// few rows, few columns, a contraction tens of thousands long, operands arriving as fp32 encoder outputs.  The step is HBM-bound
// (57 flop/byte at B = 128): what matters is how many bytes of the two fp32 matrices a CU keeps in flight.  gemm_bf16.h stages fp32
// through registers (global -> VGPR -> v_cvt_pk_bf16_f32 -> LDS): the bytes in flight are bounded by the staging registers and the
// launch ran at 3.3 TB/s whatever the tile.  Here the fp32 tiles go global -> LDS by DMA (global_load_lds_dwordx4: no registers, four
// 32 KiB ring slots per workgroup = 96 KiB in flight behind the slot being multiplied) and are rounded to bf16 on the way from LDS
// into the MFMA fragment (two ds_read_b128 + four v_cvt_pk_bf16_f32 per fragment; the matrix pipe is idle most of the time anyway).
//   tile 128 x 128, K step = 32 fp32 (128-byte rows: the geometry of the bf16 k-major image, same c ^ ((r >> 1) & 7) swizzle,
//   applied on the DMA's source side), split-K slabs of partial logits as in the short-row plan (summed by gfinal_short_kernel),
//   bf16 images of q / c for the backward written once per element from the same LDS tiles -- by FOUR EXTRA WAVES: the in-order
//   vmcnt of a wave counts its stores with its loads, so a wave that both waits for DMA slots and stores 64 KB-strided 32-byte
//   pieces (~24 GB/s per CU) waits for its own stores at every slot (measured: fill 46 us + stores 13 us = 59 us, the sum); the
//   copying waves never wait on vmcnt, the fetching waves never store inside the loop.
#pragma once
#include "gemm_bf16.h"
#include "gemm256.h"
#include "skinny.h"

namespace dprhot {

constexpr int WD_B = 128;      // tile rows and columns (64 x 64 tiles with 256-byte runs -- 64 fp32 of k per slot, 8 slabs -- were measured and
                               // lost: 69 us alone against 45, q then crosses the L2 -> CU path sixteen times)
constexpr int WD_KS = 32;      // fp32 k values per ring slot and operand row (128 bytes)
constexpr int WD_SLOTS = 4;
constexpr int WD_THREADS = 512;  // waves 0-3 fetch and multiply, waves 4-7 write the bf16 images
constexpr int WD_TILE = WD_B * WD_KS;  // floats of one operand tile of a slot
constexpr size_t wd_lds_bytes = (size_t)WD_SLOTS * 2 * WD_TILE * sizeof(float);  // 128 KiB

struct WideSimArgs {
  const float* q;    // [B][d] fp32
  const float* c;    // [Nc][d] fp32
  uint16_t* Qb;      // [B][d] bf16 out (written by the tiles of column 0)
  uint16_t* Cb;      // [Nc][d] bf16 out (written by the tiles of row 0)
  int B, Nc, d;
  const uint8_t* colmask;  // [Nc] or nullptr
  float inv_T;
  float* slabs;      // split bz stores its partial logits [B][Nc] at slabs + bz * slab_stride
  size_t slab_stride;
  int kchunk;        // k values per split (a multiple of 32)
  unsigned long long* zero_words;  // cleared by workgroup (0,0,0) for the softmax launch's ticket
  int n_zero;
  int no_copy;       // timing experiment only (option wide_nocopy): skip the bf16 copy-out
  int nbx, nby;      // column / row tiles (the grid is 1-D: nbx * nby * splits workgroups)
};

// two 16-byte chunks of fp32 -> one bf16x8 MFMA fragment (RNE)
__device__ __forceinline__ bf16x8 wd_frag(const float* T, int row, int g4) {
  const int sw = (row >> 1) & 7;
  const float4 lo = *reinterpret_cast<const float4*>(T + row * WD_KS + (((2 * g4) ^ sw) << 2));
  const float4 hi = *reinterpret_cast<const float4*>(T + row * WD_KS + (((2 * g4 + 1) ^ sw) << 2));
  const uint4 w = make_uint4(cvt_pk_bf16(lo.x, lo.y), cvt_pk_bf16(lo.z, lo.w), cvt_pk_bf16(hi.x, hi.y), cvt_pk_bf16(hi.z, hi.w));
  return __builtin_bit_cast(bf16x8, w);
}

__device__ __forceinline__ void wd_body(const WideSimArgs& p, float* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i16 = lane & 15, g4 = lane >> 4;
  // XCD-aware unit order (workgroup w runs on XCD w % 8): the nbx column tiles that share one K chunk of q are consecutive units and
  // therefore meet in ONE L2 -- as a plain 3-D grid they sat on eight different XCDs and q crossed the fabric eight times
  // (250 MB of fills for 141 MB of operands: the launch ran at the fabric's 6 TB/s, not at HBM's)
  const int nbx = p.nbx, nby = p.nby;
  const int unit = sk_xcd_order(blockIdx.x, gridDim.x);
  const int bx = unit % nbx, by = (unit / nbx) % nby, bz = unit / (nbx * nby);
  const int m0 = by * WD_B, n0 = bx * WD_B;
  const int kbeg = bz * p.kchunk, kend = min(p.d, kbeg + p.kchunk);
  const int ns = (kend - kbeg) / WD_KS;

  // per-lane byte offsets of this wave's four DMA instructions per operand (one instruction = 8 rows of 128 bytes); LDS position
  // (row r, 16-byte slot lane & 7) receives source chunk (lane & 7) ^ ((r >> 1) & 7)
  unsigned oa[4], ob[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 8 + (lane >> 3);
    const unsigned ch = (unsigned)(((lane & 7) ^ ((r >> 1) & 7)) * 16);
    oa[j] = (unsigned)min(m0 + r, p.B - 1) * (unsigned)p.d * 4u + ch;
    ob[j] = (unsigned)min(n0 + r, p.Nc - 1) * (unsigned)p.d * 4u + ch;
  }
  auto issue = [&](int s, int slot) {
    float* A = smem + slot * (2 * WD_TILE);
    float* Bm = A + WD_TILE;
    const size_t kb = (size_t)(kbeg + s * WD_KS) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(reinterpret_cast<const char*>(p.q) + kb + oa[j]), (g2_lds_ptr*)(A + (wave * 4 + j) * 256), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(reinterpret_cast<const char*>(p.c) + kb + ob[j]), (g2_lds_ptr*)(Bm + (wave * 4 + j) * 256), 16, 0, 0);
  };
  if (wave < 4) {
#pragma unroll
    for (int s = 0; s < WD_SLOTS; ++s)
      if (s < ns) issue(s, s);
  }
  if (p.zero_words != nullptr && bx == 0 && by == 0 && bz == 0 && tid < p.n_zero) p.zero_words[tid] = 0ull;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool copier = wave >= 4;
  // copy-out (waves 4-7): thread t owns 16 consecutive k of row t >> 1 (four chunks): 32 contiguous bytes of bf16 per thread and step
  const int ct = tid - 256, crow = ct >> 1, chalf = ct & 1;
  for (int s = 0; s < ns; ++s) {
    if (!copier) sk_wait_younger<8>(min(s + WD_SLOTS - 1, ns - 1) - s);  // this wave's share of slot s has landed (DMAs only: exact)
    sk_barrier();                                                        // ... everybody's has
    const float* A = smem + (s % WD_SLOTS) * (2 * WD_TILE);
    const float* Bm = A + WD_TILE;
    if (!copier) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = wd_frag(A, wm * 64 + a * 16 + i16, g4);
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = wd_frag(Bm, wn * 64 + b * 16 + i16, g4);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
    } else {
      // K step sg of q is copied by the tile column sg % nbx, of c by the tile row sg % nby: every element once, spread evenly
      const int sg = (kbeg / WD_KS) + s;
      const bool copy_a = !p.no_copy && sg % nbx == bx, copy_b = !p.no_copy && sg % nby == by;
      const int k0 = kbeg + s * WD_KS + chalf * 16;
      const int sw = (crow >> 1) & 7;
      auto copy = [&](const float* T, uint16_t* dst, int row, int rows) {
        uint32_t w[8];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const float4 v = *reinterpret_cast<const float4*>(T + crow * WD_KS + (((chalf * 4 + ch) ^ sw) << 2));
          w[2 * ch] = cvt_pk_bf16(v.x, v.y);
          w[2 * ch + 1] = cvt_pk_bf16(v.z, v.w);
        }
        if (row < rows) {
          uint4* o = reinterpret_cast<uint4*>(dst + (size_t)row * p.d + k0);
          o[0] = make_uint4(w[0], w[1], w[2], w[3]);
          o[1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
      };
      if (copy_a) copy(A, p.Qb, m0 + crow, p.B);
      if (copy_b) copy(Bm, p.Cb, n0 + crow, p.Nc);
    }
    if (s + WD_SLOTS < ns) {
      sk_barrier();  // every wave is done reading this slot
      if (!copier) issue(s + WD_SLOTS, s % WD_SLOTS);
    }
  }
  sk_barrier();  // the ring is free: per-wave patches for the stores

  // ---- epilogue: * 1/T, masked columns -inf, per-wave 16 x 68 fp32 patch -> 16-byte stores (256 contiguous bytes per row)
  if (wave >= 4) return;
  constexpr int TS = 64 + 4;
  float* const T = smem + wave * (16 * TS);
  float* const out = p.slabs + (size_t)bz * p.slab_stride;
  const int nw0 = n0 + wn * 64;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(g4 * 4 + r) * TS + b * 16 + i16] = acc[a][b][r] * p.inv_T;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const int mb = m0 + wm * 64 + a * 16;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int e = lane + it * 64, row = e >> 4, cq = e & 15;
      const int m = mb + row, n = nw0 + cq * 4;
      float4 v = *reinterpret_cast<const float4*>(T + row * TS + cq * 4);
      if (m < p.B && n < p.Nc) {  // Nc % 8 == 0: a run of four columns is inside or outside as a whole
        if (p.colmask != nullptr) {
          const uint32_t mk = *reinterpret_cast<const uint32_t*>(p.colmask + n);
          if (mk & 0x000000ffu) v.x = -INFINITY;
          if (mk & 0x0000ff00u) v.y = -INFINITY;
          if (mk & 0x00ff0000u) v.z = -INFINITY;
          if (mk & 0xff000000u) v.w = -INFINITY;
        }
        *reinterpret_cast<float4*>(out + (size_t)m * p.Nc + n) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

__global__ __launch_bounds__(WD_THREADS, 1) void wide_sim_kernel(WideSimArgs p) {
  extern __shared__ __attribute__((aligned(16))) float wd_smem[];
  wd_body(p, wd_smem);
}

}  // namespace dprhot
