// dprhot.hip -- C ABI (include/dprhot.h) over the gfx950 kernels in gemm_bf16.h / rowwise.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC dprhot.hip -o libdprhot.so
#include "../../include/dprhot.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "gemm_bf16.h"
#include "rowwise.h"

using namespace dprhot;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(DPRHOT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define REQUIRE(cond, ...) \
  do {                     \
    if (!(cond)) return fail(DPRHOT_E_INVALID, __VA_ARGS__); \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

bool use_tr() {  // DPRHOT_NO_TR=1 swaps the LDS transpose read for plain 16-bit gathers (cross-check)
  static const bool v = []() {
    const char* e = getenv("DPRHOT_NO_TR");
    return !(e && e[0] == '1');
  }();
  return v;
}

int force_tile() {  // DPRHOT_TILE=0..3 pins the tile config (tuning / tests)
  static const int v = []() {
    const char* e = getenv("DPRHOT_TILE");
    return e ? atoi(e) : -1;
  }();
  return v;
}

constexpr int kNumCU = 256;

// tile configurations, largest first
struct TileCfg { int bm, bn; };
constexpr TileCfg kTiles[4] = {{128, 128}, {64, 128}, {64, 64}, {32, 64}};

// Largest tile (with BM no larger than M needs) that still yields >= want workgroups; else the one with most.
int pick_tile(int M, int N, int splits_hint, int want) {
  if (force_tile() >= 0 && force_tile() < 4) return force_tile();
  const int bm_cap = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
  int best = -1;
  for (int t = 0; t < 4; ++t) {
    if (kTiles[t].bm > bm_cap) continue;
    if (best < 0) best = t;
    const long wgs = (long)cdiv(M, kTiles[t].bm) * cdiv(N, kTiles[t].bn) * splits_hint;
    if (wgs >= want) return t;
    best = t;  // keeps shrinking: smallest admissible tile has the most workgroups
  }
  return best;
}

template <int BM, int BN, int WM, int WN, bool AK, bool BKM, bool TR, class Epi>
int launch_one(const GemmArgs& a, const Epi& epi, int splits, hipStream_t st) {
  auto kern = gemm_bf16_kernel<BM, BN, WM, WN, AK, BKM, TR, Epi>;
  constexpr size_t lds = gemm_lds_bytes<BM, BN, AK, BKM>();
  static bool attr_done = false;  // benign race: idempotent
  if (lds > 48 * 1024 && !attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, epi);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

template <bool AK, bool BKM, class Epi>
int launch_gemm(int tile, const GemmArgs& a, const Epi& epi, int splits, hipStream_t st) {
  constexpr bool needs_tr = !(AK && BKM);
  const bool tr = needs_tr ? use_tr() : false;
#define DPRHOT_TILE_CASE(T, BM, BN)                                                           \
  case T:                                                                                     \
    if constexpr (needs_tr) {                                                                 \
      return tr ? launch_one<BM, BN, 2, 2, AK, BKM, true, Epi>(a, epi, splits, st)            \
                : launch_one<BM, BN, 2, 2, AK, BKM, false, Epi>(a, epi, splits, st);          \
    } else {                                                                                  \
      return launch_one<BM, BN, 2, 2, AK, BKM, false, Epi>(a, epi, splits, st);               \
    }
  switch (tile) {
    DPRHOT_TILE_CASE(0, 128, 128)
    DPRHOT_TILE_CASE(1, 64, 128)
    DPRHOT_TILE_CASE(2, 64, 64)
    DPRHOT_TILE_CASE(3, 32, 64)
  }
#undef DPRHOT_TILE_CASE
  return fail(DPRHOT_E_UNSUPPORTED, "bad tile id %d", tile);
}

int check_shape(int B, int Nc, int d) {
  REQUIRE(B > 0 && Nc > 0 && d > 0, "non-positive shape B=%d Nc=%d d=%d", B, Nc, d);
  REQUIRE(d % 8 == 0, "d=%d must be a multiple of 8 (16-byte rows)", d);
  REQUIRE(Nc % 8 == 0, "Nc=%d must be a multiple of 8 (16-byte rows; pad with masked columns)", Nc);
  return DPRHOT_OK;
}

// split-K plan of dQ = G x C  (M = B, N = d, K = Nc)
struct DqPlan { int tile, splits, kchunk; };
DqPlan dq_plan(int B, int Nc, int d) {
  DqPlan p;
  p.tile = pick_tile(B, d, 8, 2 * kNumCU);
  const int tiles = cdiv(B, kTiles[p.tile].bm) * cdiv(d, kTiles[p.tile].bn);
  int splits = cdiv(2 * kNumCU, tiles);
  const int ksteps = cdiv(Nc, BK);
  if (splits > ksteps / 2) splits = ksteps / 2;  // at least two K steps per split
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  p.kchunk = cdiv(ksteps, splits) * BK;
  p.splits = cdiv(Nc, p.kchunk);
  return p;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

int dprhot_version(void) { return DPRHOT_VERSION; }
const char* dprhot_last_error(void) { return g_err; }

int dprhot_workspace_bytes(int B, int Nc, int d, size_t* h_out) {
  REQUIRE(h_out != nullptr, "h_out is NULL");
  if (int rc = check_shape(B, Nc, d)) return rc;
  const DqPlan p = dq_plan(B, Nc, d);
  const size_t dq_part = align256((size_t)p.splits * B * d * sizeof(float));
  const size_t logits = align256((size_t)B * Nc * sizeof(float));  // inbatch_fwd with S_out == NULL
  *h_out = (dq_part > logits ? dq_part : logits) + 256;
  return DPRHOT_OK;
}

int dprhot_cast_bf16(const float* src, dprhot_bf16* dst, size_t n, void* stream) {
  REQUIRE(src && dst, "NULL pointer");
  REQUIRE(n % 8 == 0, "n=%zu must be a multiple of 8", n);
  REQUIRE(aligned16(src) && aligned16(dst), "pointers must be 16-byte aligned");
  if (n == 0) return DPRHOT_OK;
  const size_t n8 = n / 8;
  const int blocks = (int)((n8 + 255) / 256 > 2048 ? 2048 : (n8 + 255) / 256);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n8);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_sim_fwd(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const uint8_t* colmask, float inv_T,
                   float* S, void* stream) {
  REQUIRE(Q && C && S, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(Q) && aligned16(C) && aligned16(S), "pointers must be 16-byte aligned");
  GemmArgs a{Q, C, B, Nc, d, d, d, cdiv(d, BK) * BK};
  EpiSim epi{S, colmask, B, Nc, inv_T};
  const int tile = pick_tile(B, Nc, 1, 2 * kNumCU);
  return launch_gemm<true, true>(tile, a, epi, 1, (hipStream_t)stream);
}

int dprhot_softmax_ce_fwd_bwd(const float* S, int B, int Nc, const int64_t* y, int64_t y_offset, float grad_scale,
                              const int64_t* row_win_start, int win_len, float* row_loss, float* row_lse, dprhot_bf16* G, void* stream) {
  REQUIRE(S && y, "NULL pointer");
  REQUIRE(B > 0 && Nc > 0 && Nc % 8 == 0, "bad shape B=%d Nc=%d (Nc %% 8 == 0)", B, Nc);
  REQUIRE(aligned16(S) && (G == nullptr || aligned16(G)), "pointers must be 16-byte aligned");
  REQUIRE(row_win_start == nullptr || win_len > 0, "win_len must be > 0 with row_win_start");
  SoftmaxArgs p{S, B, Nc, y, y_offset, grad_scale, row_win_start, win_len, row_loss, row_lse, G};
  if (Nc <= 4096) {
    hipLaunchKernelGGL(softmax_ce_kernel<64>, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(softmax_ce_kernel<256>, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
  }
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_reduce_sum(const float* x, int n, float scale, float* out, void* stream) {
  REQUIRE(x && out && n > 0, "bad argument");
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, scale, out);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_dq(const dprhot_bf16* G, const dprhot_bf16* C, int B, int Nc, int d, float h_scale, const float* d_scale, float* dQ,
              void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(G && C && dQ, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(G) && aligned16(C) && aligned16(dQ), "pointers must be 16-byte aligned");
  const DqPlan p = dq_plan(B, Nc, d);
  GemmArgs a{G, C, B, d, Nc, Nc, d, p.kchunk};
  if (p.splits == 1) {
    EpiScaleF32 epi{dQ, B, d, h_scale, d_scale};
    return launch_gemm<true, false>(p.tile, a, epi, 1, (hipStream_t)stream);
  }
  const size_t need = (size_t)p.splits * B * d * sizeof(float);
  if (workspace == nullptr || workspace_bytes < need)
    return fail(DPRHOT_E_WORKSPACE, "dq needs %zu workspace bytes, got %zu", need, workspace_bytes);
  REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  EpiScaleF32 epi{static_cast<float*>(workspace), B, d, 1.0f, nullptr};
  if (int rc = launch_gemm<true, false>(p.tile, a, epi, p.splits, (hipStream_t)stream)) return rc;
  const size_t n4 = (size_t)B * d / 4;
  const int blocks = (int)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace),
                     p.splits, n4, h_scale, d_scale, dQ);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_dc(const dprhot_bf16* G, const dprhot_bf16* Q, int B, int Nc, int d, float h_scale, const float* d_scale,
              float* dC_part, void* stream) {
  REQUIRE(G && Q && dC_part, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(G) && aligned16(Q) && aligned16(dC_part), "pointers must be 16-byte aligned");
  // A(m = ctx column, k = query row) = G[k][m]  (mn-major, lda = Nc);  B(k, n) = Q[k][n]  (mn-major, ldb = d)
  GemmArgs a{G, Q, Nc, d, B, Nc, d, cdiv(B, BK) * BK};
  EpiScaleF32 epi{dC_part, Nc, d, h_scale, d_scale};
  const int tile = pick_tile(Nc, d, 1, 2 * kNumCU);
  return launch_gemm<false, false>(tile, a, epi, 1, (hipStream_t)stream);
}

int dprhot_rank_of_gold(const float* S, int rows, int cols, const int64_t* y, int64_t y_offset, int64_t* rank, void* stream) {
  REQUIRE(S && y && rank, "NULL pointer");
  REQUIRE(rows > 0 && cols > 0, "bad shape rows=%d cols=%d", rows, cols);
  REQUIRE(cols % 4 == 0 ? aligned16(S) : true, "S must be 16-byte aligned");
  if (cols % 4 != 0) return fail(DPRHOT_E_UNSUPPORTED, "cols=%d must be a multiple of 4", cols);
  hipLaunchKernelGGL(rank_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, rows, cols, y, y_offset, rank);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_topk(const float* S, int rows, int cols, int k, float* values, int64_t* indices, void* stream) {
  REQUIRE(S && values && indices, "NULL pointer");
  REQUIRE(rows > 0 && cols > 0 && k > 0 && k <= 128 && k <= cols, "bad shape rows=%d cols=%d k=%d", rows, cols, k);
  hipLaunchKernelGGL(topk_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, rows, cols, k, values, indices);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_inbatch_fwd(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                       const uint8_t* colmask, float inv_T, float grad_scale, float* S_out, float* row_loss, float* row_lse, float* loss_sum,
                       dprhot_bf16* G, void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(Q && C && y && row_loss && loss_sum, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  float* S = S_out;
  if (S == nullptr) {
    // logits live in the caller's workspace for the duration of the call
    const size_t need = (size_t)B * Nc * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need)
      return fail(DPRHOT_E_WORKSPACE, "inbatch_fwd needs %zu workspace bytes (or S_out), got %zu", need, workspace_bytes);
    S = static_cast<float*>(workspace);
  }
  if (int rc = dprhot_sim_fwd(Q, B, C, Nc, d, colmask, inv_T, S, stream)) return rc;
  if (int rc = dprhot_softmax_ce_fwd_bwd(S, B, Nc, y, y_offset, grad_scale, nullptr, 0, row_loss, row_lse, G, stream)) return rc;
  return dprhot_reduce_sum(row_loss, B, 1.0f, loss_sum, stream);
}

int dprhot_inbatch_bwd(const dprhot_bf16* G, const dprhot_bf16* Q, const dprhot_bf16* C, int B, int Nc, int d, float h_scale,
                       const float* d_scale, float* dQ, float* dC_part, void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(G && Q && C, "NULL pointer");
  if (dC_part != nullptr)
    if (int rc = dprhot_dc(G, Q, B, Nc, d, h_scale, d_scale, dC_part, stream)) return rc;
  if (dQ != nullptr)
    if (int rc = dprhot_dq(G, C, B, Nc, d, h_scale, d_scale, dQ, workspace, workspace_bytes, stream)) return rc;
  return DPRHOT_OK;
}

}  // extern "C"
