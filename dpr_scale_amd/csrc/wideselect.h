// wideselect.h -- streaming top-k per row for k BEYOND the LDS-resident kernels of rowwise.h (k > 4096): torch.topk of
// run_retrieval_pytorch.py:149-150 with any --topk (:69), and its shard re-merge (:272-277), in the frozen total order
// (score desc, id asc).  The state -- the k best so far, sorted -- lives in HBM; folding one chunk of scores into it is
//   wsel_select_kernel   radix select over state + chunk: four passes over an order-preserving 32-bit key (one byte per pass, a
//                        256-bin histogram in LDS per row), exact: the key T of the k-th best, how many rank strictly ahead of it,
//                        how many entries of key T are still needed
//                        -- and, when only some of the entries AT T are taken, the same select over their ids (the total order
//                        continues with the id): the id of the last one taken
//   wsel_collect_kernel  one more pass: everything ahead of T, and the entries at T up to that id, go out in arbitrary order
//   wsel_sort_kernel     bitonic sort of each block of 4096 collected entries in LDS (total order)
//   wsel_merge_kernel    log2(blocks) passes over HBM: two sorted runs -> one, every entry finds its place by bisection in the other run
// No score is sorted that does not end up in the result; nothing here depends on k fitting anywhere but HBM.  Exact and deterministic.
// Ties are exact at any multiplicity (a corpus of identical passages selects by id alone).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowwise.h"

namespace dprhot {

constexpr int WSEL_THREADS = 256;
constexpr int WSEL_BLOCK = 4096;    // entries sorted per workgroup in LDS (12 bytes each: 48 KiB)
constexpr int WSEL_REC = 8;         // words per row record: T, n_gt, need, n_eq, kprime, err, id_cut (lo, hi)

// ascending with the value; +0 and -0 share a key (they tie as floats: v1 > v2 is false both ways)
__device__ __forceinline__ unsigned wsel_key(float v) {
  unsigned u = __float_as_uint(v);
  if (u == 0x80000000u) u = 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct WselArgs {
  const float* S;        // [rows][ld] chunk scores
  int rows, cols;
  long long ld, col_offset;
  int k;
  const float* vals;     // [rows][k] state (sorted best-first; unfilled slots: idx < 0), read unless first
  const int64_t* idx;
  int first;
  unsigned* rec;         // [rows][WSEL_REC]
  float* out_v;          // [rows][k] collected entries (unsorted), then sort / merge buffers
  int64_t* out_i;
};

// candidate e of a row: e < nstate -> state slot e, else chunk column e - nstate; returns false for an unfilled state slot
__device__ __forceinline__ bool wsel_cand(const WselArgs& p, int row, int nstate, int e, float& v, long long& id) {
  if (e < nstate) {
    id = p.idx[(size_t)row * p.k + e];
    v = p.vals[(size_t)row * p.k + e];
    return id >= 0;
  }
  v = p.S[(size_t)row * p.ld + (e - nstate)];
  id = p.col_offset + (e - nstate);
  return true;
}

__global__ __launch_bounds__(WSEL_THREADS) void wsel_select_kernel(WselArgs p) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need, s_ngt, s_neq, s_total;
  const int row = blockIdx.x, tid = threadIdx.x;
  const int nstate = p.first ? 0 : p.k, n = nstate + p.cols;
  unsigned prefix = 0, need = 0, ngt = 0;
  for (int pass = 0; pass < 4; ++pass) {
    hist[tid] = 0;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    for (int e = tid; e < n; e += WSEL_THREADS) {
      float v;
      long long id;
      if (!wsel_cand(p, row, nstate, e, v, id)) continue;
      const unsigned key = wsel_key(v);
      if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      if (pass == 0) {
        unsigned total = 0;
        for (int b = 0; b < 256; ++b) total += hist[b];
        s_total = total;
        need = total < (unsigned)p.k ? total : (unsigned)p.k;
      }
      unsigned cum = 0;
      int b = 255;
      for (; b > 0; --b) {  // from the best bin down: the bin in which the need-th best lies
        if (cum + hist[b] >= need) break;
        cum += hist[b];
      }
      s_ngt = ngt + cum;
      s_need = need - cum;
      s_neq = hist[b];
      s_prefix = prefix | ((unsigned)b << shift);
    }
    __syncthreads();
    prefix = s_prefix;
    need = s_need;
    ngt = s_ngt;
    __syncthreads();
  }
  // ---- ties AT the k-th key of which only some are taken: the total order continues with the id (ascending), so the same radix
  //      select runs over the ids of the entries at T -- first one pass for the highest byte any of them uses, then one pass per byte
  //      from there down: the id of the need-th smallest (ids are distinct: exact)
  const unsigned T = prefix, neq = s_neq;
  unsigned long long id_cut = 0x7fffffffffffffffull;
  if (need < neq) {
    __shared__ unsigned long long s_or, s_idp;
    if (tid == 0) s_or = 0ull;
    __syncthreads();
    unsigned long long mine = 0ull;
    for (int e = tid; e < n; e += WSEL_THREADS) {
      float v;
      long long id;
      if (wsel_cand(p, row, nstate, e, v, id) && wsel_key(v) == T) mine |= (unsigned long long)id;
    }
    atomicOr(&s_or, mine);
    __syncthreads();
    int top = 7;
    while (top > 0 && ((s_or >> (8 * top)) & 255ull) == 0ull) --top;
    unsigned long long idp = 0ull;  // the bytes above `top` are zero in every tie's id
    unsigned want = need;
    for (int byte = top; byte >= 0; --byte) {
      hist[tid] = 0;
      __syncthreads();
      const int sh = 8 * byte;
      for (int e = tid; e < n; e += WSEL_THREADS) {
        float v;
        long long id;
        if (!wsel_cand(p, row, nstate, e, v, id) || wsel_key(v) != T) continue;
        const unsigned long long u = (unsigned long long)id;
        if (byte == 7 || (u >> (sh + 8)) == (idp >> (sh + 8))) atomicAdd(&hist[(unsigned)((u >> sh) & 255ull)], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned cum = 0;
        int b = 0;
        for (; b < 255; ++b) {  // from the SMALLEST byte up
          if (cum + hist[b] >= want) break;
          cum += hist[b];
        }
        s_need = want - cum;
        s_idp = idp | ((unsigned long long)b << sh);
      }
      __syncthreads();
      want = s_need;
      idp = s_idp;
      __syncthreads();
    }
    id_cut = idp;  // ties with id <= id_cut are taken: exactly `need` of them
  }
  if (tid == 0) {
    unsigned* r = p.rec + (size_t)row * WSEL_REC;
    const unsigned kprime = s_total < (unsigned)p.k ? s_total : (unsigned)p.k;
    r[0] = T;                       // key of the kprime-th best
    r[1] = kprime == 0 ? 0u : ngt;  // entries strictly ahead of T
    r[2] = kprime == 0 ? 0u : need; // entries AT T taken (>= 1 when kprime > 0)
    r[3] = neq;                     // entries AT T in all
    r[4] = kprime;
    r[5] = 0;                       // error word (none defined any more: kept for the ABI's contract)
    r[6] = (unsigned)(id_cut & 0xffffffffull);
    r[7] = (unsigned)(id_cut >> 32);
  }
}

__global__ __launch_bounds__(WSEL_THREADS) void wsel_collect_kernel(WselArgs p) {
  __shared__ unsigned s_gt, s_eq;
  const int row = blockIdx.x, tid = threadIdx.x;
  const int nstate = p.first ? 0 : p.k, n = nstate + p.cols;
  const unsigned* r = p.rec + (size_t)row * WSEL_REC;
  const unsigned T = r[0], ngt = r[1], kprime = r[4];
  const long long id_cut = (long long)(((unsigned long long)r[7] << 32) | r[6]);
  float* ov = p.out_v + (size_t)row * p.k;
  int64_t* oi = p.out_i + (size_t)row * p.k;
  if (tid == 0) { s_gt = 0; s_eq = 0; }
  __syncthreads();
  if (kprime > 0) {
    for (int e = tid; e < n; e += WSEL_THREADS) {
      float v;
      long long id;
      if (!wsel_cand(p, row, nstate, e, v, id)) continue;
      const unsigned key = wsel_key(v);
      if (key > T) {
        const unsigned pos = atomicAdd(&s_gt, 1u);  // (< ngt by construction)
        ov[pos] = v;
        oi[pos] = id;
      } else if (key == T && id <= id_cut) {
        const unsigned q = atomicAdd(&s_eq, 1u);    // (< need by construction)
        ov[ngt + q] = v;
        oi[ngt + q] = id;
      }
    }
  }
  for (int e = kprime + tid; e < p.k; e += WSEL_THREADS) {  // fewer candidates than k: the tail stays unfilled
    ov[e] = -INFINITY;
    oi[e] = -1;
  }
}

// sorts block blockIdx.y of row blockIdx.x in place (total order; unfilled slots -- idx < 0 -- last)
__global__ __launch_bounds__(WSEL_THREADS) void wsel_sort_kernel(float* v_io, int64_t* i_io, int k) {
  extern __shared__ __attribute__((aligned(16))) char wsel_smem[];
  long long* si = reinterpret_cast<long long*>(wsel_smem);
  float* sv = reinterpret_cast<float*>(wsel_smem + (size_t)WSEL_BLOCK * sizeof(long long));
  const int row = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x;
  const int base = blk * WSEL_BLOCK, cnt = min(WSEL_BLOCK, k - base);
  float* gv = v_io + (size_t)row * k + base;
  int64_t* gi = i_io + (size_t)row * k + base;
  for (int e = tid; e < WSEL_BLOCK; e += WSEL_THREADS) {
    const bool on = e < cnt && gi[e] >= 0;
    sv[e] = on ? gv[e] : -INFINITY;
    si[e] = on ? (long long)gi[e] : 0x7fffffffffffffffLL;
  }
  __syncthreads();
  for (int size = 2; size <= WSEL_BLOCK; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < WSEL_BLOCK / 2; t += WSEL_THREADS) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;  // this pair's run is sorted best-first
        const float v1 = sv[lo], v2 = sv[hi];
        const long long j1 = si[lo], j2 = si[hi];
        const bool swap = up ? tk_before(v2, j2, v1, j1) : tk_before(v1, j1, v2, j2);
        if (swap) {
          sv[lo] = v2; sv[hi] = v1;
          si[lo] = j2; si[hi] = j1;
        }
      }
      __syncthreads();
    }
  }
  for (int e = tid; e < cnt; e += WSEL_THREADS) {
    const bool on = si[e] != 0x7fffffffffffffffLL;
    gv[e] = on ? sv[e] : -INFINITY;
    gi[e] = on ? (int64_t)si[e] : -1;
  }
}

// one merge pass: runs of L sorted entries -> runs of 2 L.  An entry's place = its place in its own run + the entries of the
// partner run that rank ahead of it (bisection; the order is total, ids are distinct, unfilled slots rank last among themselves
// by position)
// final != nullptr: this pass writes the caller's state; rows whose record carries an error keep their old state (rec word 5)
__global__ __launch_bounds__(WSEL_THREADS) void wsel_merge_kernel(const float* sv, const int64_t* si, float* dv, int64_t* di, int k, int L,
                                                                  const unsigned* final_rec) {
  const int row = blockIdx.y;
  const int g = blockIdx.x * WSEL_THREADS + threadIdx.x;
  if (g >= k) return;
  if (final_rec != nullptr && final_rec[(size_t)row * WSEL_REC + 5] != 0) return;
  const float* v = sv + (size_t)row * k;
  const int64_t* id = si + (size_t)row * k;
  const int run = g / L, pos = g - run * L;
  const int o0 = (run ^ 1) * L, o1 = min(k, o0 + L);  // partner run (empty beyond k)
  const float mv = v[g];
  const int64_t mraw = id[g];
  const long long mi = mraw >= 0 ? (long long)mraw : 0x7fffffffffffffffLL;
  int lo = 0, hi = o1 > o0 ? o1 - o0 : 0;
  const bool left = (run & 1) == 0;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const float ov = v[o0 + mid];
    const int64_t oraw = id[o0 + mid];
    const long long oi = oraw >= 0 ? (long long)oraw : 0x7fffffffffffffffLL;
    // does the partner's entry rank ahead of mine?  (equal only between two unfilled slots: the left run's go first)
    const bool ahead = tk_before(ov, oi, mv, mi) || (!left && ov == mv && oi == mi);
    if (ahead) lo = mid + 1; else hi = mid;
  }
  const int outp = (run >> 1) * 2 * L + pos + lo;
  dv[(size_t)row * k + outp] = mv;
  di[(size_t)row * k + outp] = mraw;
}

}  // namespace dprhot
