// dprhot.hip -- C ABI (include/dprhot.h) over the gfx950 kernels in gemm_bf16.h / gemm256.h / step_small.h / rowwise.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC dprhot.hip -o libdprhot.so
//
// Launch structure of one training step on one rank:
//   dprhot_inbatch_step_f32   2 launches at the BASELINE shapes: sim GEMM (fp32 in, partial-logit slabs) -> one kernel
//                             for softmax-CE, dScores, dQ and dC (step_small.h); elsewhere = the two calls below
//   dprhot_inbatch_fwd(_f32)  2 launches: sim GEMM (+mask, 1/T, softmax statistics or K-split slabs) -> G + loss
//   dprhot_inbatch_bwd        1 launch: dC_part = G^T Q and dQ = G C side by side (+1 when dQ is split over Nc)
//   few rows x many contexts (B <= 128, Nc >= 2048: cfg3 per rank): the four launches of skinny.h (sk_plan)
// Plans (tile, split-K, kernel family) are pure functions of the shape: pick_tile / fwd_plan / dq_plan / big_ok /
// big_bwd_ok / small_step_ok below.
#include "../../include/dprhot.h"

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>

#include <type_traits>
#include "gemm_bf16.h"
#include "gemm256.h"
#include "gemm8p.h"
#include "gemm8pb.h"
#include "gemm128d.h"
#include "rowwise.h"
#include "step_small.h"
#include "skinny.h"
#include "gradcomm.h"
#include "wide.h"
#include "wideselect.h"

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

// Multi-rank packed layout (dprhot_inbatch_step_packed_f32): while set, every sim epilogue built on this thread reads the
// column mask from the gathered buffer and the dC epilogues stamp the loss numerator (see EpiSim / EpiScaleF32).
struct PackedSpec {
  const uint8_t* base = nullptr;
  int rows_c = 1, n_ctx = 0, row_bytes = 0;
  const float* stamp_src = nullptr;
};
thread_local PackedSpec g_packed;
struct PackedScope {
  explicit PackedScope(const PackedSpec& p) { g_packed = p; }
  ~PackedScope() { g_packed = PackedSpec{}; }
};
template <class E>
E with_packed_mask(E e) {
  if (g_packed.base != nullptr) {
    e.packed = g_packed.base;
    e.p_rows_c = g_packed.rows_c;
    e.p_n_ctx = g_packed.n_ctx;
    e.p_row_bytes = g_packed.row_bytes;
  }
  return e;
}
template <class E>
E with_loss_stamp(E e) {
  if (g_packed.stamp_src != nullptr) {
    e.stamp_src = g_packed.stamp_src;
    e.stamp_period = g_packed.rows_c;
    e.stamp_row = g_packed.n_ctx;
  }
  return e;
}

// loss_sum[0] = g_loss_scale * sum of the row losses, for every launch built on this thread while a LossScaleScope is alive
// (dprhot_train_step_*: the mean of dpr_task.py:212 leaves the kernel ready, no separate division launch)
thread_local float g_loss_scale = 1.0f;
struct LossScaleScope {
  explicit LossScaleScope(float s) { g_loss_scale = s; }
  ~LossScaleScope() { g_loss_scale = 1.0f; }
};
// The one-call training step may carry the sum of the row losses into the launch that combines its dQ slabs (splitk_reduce_loss_kernel):
// `armed` while its forward runs, `pending` from the forward's skipped launch until launch_dq (or the step's own fall-back) consumes it.
struct LossDefer { bool armed = false, pending = false; const float* src = nullptr; int n = 0; float scale = 1.f; float* out = nullptr; };
thread_local LossDefer g_loss_defer;

// dC_part as bf16 (the reduce-scatter's wire format written by the dC epilogue): set by dprhot_train_step_* for the plans that have
// such an epilogue (the skinny step); every other plan refuses while it is set
thread_local bool g_dc_bf16 = false;
struct DcKindScope {
  explicit DcKindScope(bool bf16) { g_dc_bf16 = bf16; }
  ~DcKindScope() { g_dc_bf16 = false; }
};

// split-K partial sums of dQ left in a caller buffer (dprhot_train_step_*: the operator's backward launch adds them up): while set,
// the skinny step writes its slabs there and skips its reduction launch
thread_local float* g_dq_part = nullptr;
struct DqPartScope {
  explicit DqPartScope(float* p) { g_dq_part = p; }
  ~DqPartScope() { g_dq_part = nullptr; }
};

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(DPRHOT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// "dynamic LDS size raised for this kernel" is a per-device fact (one code object per device): one bit per device ordinal
struct AttrOnce {
  unsigned long long devs = 0;
  static int cur() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
  }
  bool operator!() const { return !((devs >> cur()) & 1ull); }
  void operator=(bool) { devs |= 1ull << cur(); }  // benign race: idempotent
};

#define REQUIRE(cond, ...) \
  do {                     \
    if (!(cond)) return fail(DPRHOT_E_INVALID, __VA_ARGS__); \
  } while (0)

// ---- explicit process-wide options (dprhot_set_option): test and A/B switches of the plans.  Production never sets one; they replace
// the environment switches the library used to cache on first use (hidden configuration behind an ABI that advertises none).
enum OptId { OPT_TILE, OPT_NO_TR, OPT_UNFUSED_BWD, OPT_BIG_MIN, OPT_NO_NL, OPT_NO_BIG_BWD, OPT_NO_SKINNY, OPT_NO_SMALL_STEP, OPT_NO_SHORT,
             OPT_SK_COLS, OPT_SEARCH_UNFUSED, OPT_NO_8PB, OPT_NO_WIDE, OPT_WIDE_NOCOPY, OPT_NO_8P_STORE, OPT_NO_WIDE_BWD, OPT_NT_STORES, OPT_SK_DQ_SLICES, OPT_SK_FUSED, OPT_SK_DBG, OPT_SK_W8, OPT_SK_PAIR, OPT_SK_SIM_W8, OPT_SK_SIM_PRIV, OPT_SK_TAIL, OPT_SK_DC_REGSCALE, OPT_G8_ONE_TILE, OPT_NL_P16, OPT_SK_DQ_ATOMIC, OPT_NL_MIN, OPT_G128_DMA, OPT_DC_ALONE_8P, OPT_DQ_ONE_ROUND, OPT_DQ_CAP_FEW, OPT_LOSS_WITH_DQ, OPT_NL128, OPT_NL128_BELOW, OPT_PAIR128, OPT_PAIR128_CAP, OPT_NL128_MIN_TILES, OPT_P16_STAGED, OPT_COUNT };
struct OptDef { OptId id; const char* name; int def; const char* what; };
constexpr OptDef kOptDefs[OPT_COUNT] = {
    {OPT_TILE, "tile", -1, "0..5 pins the tile of the single-GEMM launches (gemm_bf16.h), -1 = plan"},
    {OPT_NO_TR, "no_tr", 0, "1 swaps the LDS transpose read for plain 16-bit gathers (cross-check)"},
    {OPT_UNFUSED_BWD, "unfused_bwd", 0, "1 runs dC and dQ as two launches"},
    {OPT_BIG_MIN, "big_min", 256, "fewest 256x256 tiles for which the large-shape kernels are chosen (0 = never; tests lower it)"},
    {OPT_NO_NL, "no_nl", 0, "1 disables the no-logits forward"},
    {OPT_NO_BIG_BWD, "no_big_bwd", 0, "1 disables the 256x256 backward pair"},
    {OPT_NO_SKINNY, "no_skinny", 0, "1 disables the few-rows x many-contexts plan (skinny.h)"},
    {OPT_NO_SMALL_STEP, "no_small_step", 0, "1 disables the fused softmax+backward kernel of the latency-bound shapes (step_small.h)"},
    {OPT_NO_SHORT, "no_short", 0, "1 disables the short-row (K-split slab) forward"},
    {OPT_SK_COLS, "sk_cols", 0, "64 / 128 forces the sim unit width of the skinny plan, 0 = plan"},
    {OPT_SEARCH_UNFUSED, "search_unfused", 0, "1 materialises every chunk's scores in dprhot_search (no filter epilogue)"},
    {OPT_NO_8PB, "no_8pb", 0, "1 keeps the backward pair on gemm256.h instead of the phase-interleaved schedule"},
    {OPT_NO_WIDE, "no_wide", 0, "1 keeps vocabulary-wide fp32 operands on the register-staged sim kernel (wide.h off)"},
    {OPT_WIDE_NOCOPY, "wide_nocopy", 0, "TIMING EXPERIMENT ONLY: 1 drops the bf16 copy-out of wide.h (the backward then reads garbage)"},
    {OPT_NO_8P_STORE, "no_8p_store", 0, "1 keeps dprhot_sim_fwd's large shapes on the round-1 256 x 256 kernel (gemm256.h)"},
    {OPT_NO_WIDE_BWD, "no_wide_bwd", 0, "1 keeps the backward of vocabulary-wide vectors on the generic pair kernel (skinny.h units off)"},
    {OPT_NT_STORES, "nt_stores", 1, "0 writes dC_part of the few-rows plans (skinny.h units: cfg3 per rank, router width) with plain instead of non-temporal stores (A/B of the cache policy)"},
    {OPT_SK_DQ_SLICES, "sk_dq_slices", 0, "context slices of the few-rows plan's dQ units (0 = plan)"},
    {OPT_SK_FUSED, "sk_fused", 1, "few-rows plan without its dScores launch (G == NULL): 1 = where it measured faster (B x Nc >= 2^19), 2 = wherever the plan exists (tests), 0 = never"},
    {OPT_SK_DBG, "sk_dbg", 0, "TIMING EXPERIMENTS ONLY (fused few-rows backward; bits): 1 dC units leave at once, 2 dQ units leave at once, 4 (unused), 8 the plain slab sum in the finishing launch's place (wrong dQ), 16 the dQ slabs leave with ordinary instead of non-temporal stores"},
    {OPT_SK_W8, "sk_w8", 1, "fused few-rows backward with eight waves per workgroup (512 threads, half the output tile per wave: 13.0-13.5 against 13.9-14.4 us at cfg3 per rank); 0 = four"},
    {OPT_SK_PAIR, "sk_pair", 0, "fused few-rows backward with one kind of unit (sk_bwdp_kernel: a P tile is loaded once for both products: measured 21.2 against 13.5 us at cfg3 per rank); 0 = dQ units and dC units (sk_bwdf_kernel)"},
    {OPT_SK_SIM_W8, "sk_sim_w8", 1, "few-rows sim launch with eight waves per workgroup (128-column units, fp32 q): 0 = four"},
    {OPT_SK_SIM_PRIV, "sk_sim_priv", 1, "few-rows sim launch with wave-private rings and no barrier in the K loop (sk_simp_kernel; fp32 q, 128-column units, grids of at most one unit per CU): 0 = sk_sim_kernel"},
    {OPT_SK_TAIL, "sk_tail", 0, "fused few-rows backward: 1 = the dQ slabs are folded by the last workgroups of the backward launch itself, behind a count of the dQ units (write-through slab stores, sc1 loads; no sk_dq_finish launch), 2 = the same with ordinary stores + one release fence per dQ unit and an acquire fence in the finishing role; 0 = the finishing launch (measured: scratch/negative/README.md, round 5)"},
    {OPT_SK_DC_REGSCALE, "sk_dc_regscale", 0, "fused few-rows backward, dC units: 1 = the Q fragments are scaled by f in registers between the transpose read and the MFMA (no scale pass over the LDS image, one workgroup barrier less; measured: the pass's 0.75 us reappear in the MFMA loop, step 25.7-25.8 against 25.3-25.6 us), 0 = the scale pass of round 4"},
    {OPT_G8_ONE_TILE, "g8_one_tile", 0, "storing epilogues of the phase-interleaved 256 x 256 kernel (dScores pass, stored logits): 1 = one workgroup per tile instead of persistent workgroups (a finished workgroup's stores drain under its successor's prologue)"},
    {OPT_NL_P16, "nl_p16", 1, "no-logits forward with the dScores wanted: 1 = ONE pass of the GEMM (strip statistics + the tile's fp16 softmax numerators, Epi8StatsP, two-phase schedule) and a row kernel that rescales them into G in place; 0 = two GEMM passes (statistics, then the logits recomputed into G: Epi8G)"},
    {OPT_SK_DQ_ATOMIC, "sk_dq_atomic", 0, "fused few-rows backward: 1 = the dQ units scale their tiles to the row softmax themselves and ADD them into dQ (global_atomic_add_f32; dQ zero-filled by the sim launch): no slabs, no finishing launch -- dQ reproducible to rounding, not to the bit; 0 = slice-normalised slabs + sk_dq_finish_kernel (bit-reproducible)"},
    {OPT_NL_MIN, "nl_min", 128, "fewest 256x256 tiles from which the forward never stores the logits (round 6: 128 -- with the one-pass forward 1024 x 8192 x 768 steps in 81 instead of 92 us, 512 x 16384 in 116 instead of 126; at 64 tiles it is a wash, at 32 it loses); the smaller of this and big_min counts"},
    {OPT_G128_DMA, "g128_dma", 1, "128 x 128 x 64 tile of the GEMM engine with its operands staged by LDS-DMA (gemm128d.h) instead of global -> VGPR -> ds_write (gemm_bf16.h): 1 = wherever the launch qualifies (bf16 operands, whole 64-deep K steps, 32-bit offsets); 0 = never"},
    {OPT_DC_ALONE_8P, "dc_alone_8p", 1, "long context axis (512 <= B <= 2048, Nc >= 32 B): the dC tiles ALONE on the phase-interleaved 256 x 256 kernel (gemm8pb.h) in a launch of their own, in front of the launch with the dQ units, instead of dC on the 128 x 128 tile: 1 = where the row pitch of G is a multiple of 128 KiB (Nc = 65536: the 128-wide tiles' 256-byte pieces then alias in the memory system -- 180 us where 126 are expected -- and the 512-byte pieces of the 256-wide tile do not: backward 325 -> 275 us at 1024 x 65536, 557 -> 470 at 2048 x 65536; it LOSES at 32768 / 49152 columns: profiles/r06_dc_alone_ab.txt), 2 = always, 0 = never"},
    {OPT_DQ_ONE_ROUND, "dq_one_round", 1, "long context axis (1024 <= B <= 2048, Nc >= 32 B): 1 = the context slices of the dQ units are cut so that the units fill the 256 CUs once (up to 32 slices: 1536 x 65536 x 768 backward 419 -> 351 us, 2048 x 65536 497 -> 459, 1024 x 65536 294 -> 282; profiles/r06_dq_round_ab.txt), 0 = at most 16 slices"},
    {OPT_DQ_CAP_FEW, "dq_cap_few", 0, "256 x 256 backward pair under 512 query rows: the most context slices a dQ tile is cut into; 0 = the rule (16, or 32 where 16 slices already take more than one round of the 256 CUs), else that many"},
    {OPT_LOSS_WITH_DQ, "loss_with_dq", 1, "one-call training step on the no-logits forward, single rank: 1 = the sum of the row losses is formed by one more workgroup of the launch that combines the dQ slabs (same arithmetic as reduce_sum_kernel) instead of a launch of its own between forward and backward; 0 = its own launch"},
    {OPT_NL128, "nl128", 1, "training forward (dScores wanted, logits not) in ONE pass on the 128 x 128 LDS-DMA tile (EpiSimP: strip statistics + fp16 softmax numerators, then the row kernel of the 256 x 256 family) where the 256-wide tiles would leave most of the chip idle: 1 = on, 0 = the logits-storing forward (sim GEMM with fp32 S, streaming softmax) there"},
    {OPT_NL128_BELOW, "nl128_max_tiles", 512, "the 128-tile one-pass forward takes the shapes with at most this many 128 x 128 tiles (512 = one round of two workgroups per CU; beyond, where the 256 x 256 no-logits forward qualifies, that one runs: 1536 x 8192 x 768, 768 tiles, 48.6 us against ~37)"},
    {OPT_PAIR128, "pair128", 1, "backward pair (dC tiles next to split-K dQ tiles in one launch) on the 128 x 128 LDS-DMA tile (gemm128d_pair_kernel): 1 = where the rule pair128_use picks it, 2 = wherever the launch qualifies (A/B), 0 = never (the register-staged pair / the 256 x 256 pair)"},
    {OPT_PAIR128_CAP, "pair128_slices", 0, "K slices of a dQ tile in the 128 x 128 backward pair: 0 = the rule of pair128_plan, else that many (A/B)"},
    {OPT_NL128_MIN_TILES, "nl128_min_tiles", 32, "fewest 128 x 128 tiles for the 128-tile one-pass forward (32 against 128, step us: 512 x 2048 44.1 -> 40.7, 768 x 2048 49.7 -> 42.5, 1024 x 1024 47.3 -> 43.5, 256 x 4096 / 192 x 4096 / 256 x 2048 level)"},
    {OPT_P16_STAGED, "p16_staged", 1, "one-pass forward on the 256 x 256 kernel: the fp16 numerators leave through the per-wave LDS patch (64-byte pieces of 16 rows per store instruction instead of 32-byte pieces of 32 rows; bit-identical): 1 = except where the row pitch of G is a multiple of 128 KiB (forward us, arms alternating, profiles/r06_p16_staged_ab.txt: 8192^2 155.4 -> 147.8, 4096 x 8192 83.2 -> 79.3, 4096 x 16384 153 -> 146 and its step 390 -> 372; at 65536 columns the forward gains nothing and the step of 8192 x 65536 LOSES 100 us of 2880), 2 = always, 0 = never"},
};
constexpr bool opt_table_in_enum_order() {  // (round 6: a row added in the wrong place made two options answer to each other's names)
  for (int i = 0; i < OPT_COUNT; ++i)
    if (kOptDefs[i].id != (OptId)i) return false;
  return true;
}
static_assert(opt_table_in_enum_order(), "kOptDefs must list the options in the order of enum OptId");
long long g_opt_epoch = 0;  // bumped by every dprhot_set_option: host-side caches of plan facts key on it (dprhot_options_epoch)
int g_opt[OPT_COUNT];  // the defaults of the table above (one place: a default typed twice was typed wrong once)
const bool g_opt_defaults = [] {
  for (int i = 0; i < OPT_COUNT; ++i) g_opt[i] = kOptDefs[i].def;
  return true;
}();
inline int opt(OptId i) { return __atomic_load_n(&g_opt[i], __ATOMIC_RELAXED); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool use_tr() { return opt(OPT_NO_TR) == 0; }

int force_tile() { return opt(OPT_TILE); }

bool unfused_bwd() { return opt(OPT_UNFUSED_BWD) != 0; }

constexpr int kNumCU = 256;

// tile configurations {BM, BN, BK}
struct TileSpec { int bm, bn, bk; };
constexpr int kNumTiles = 7;  // 0-5: gemm_bf16.h (register staging); 6: gemm256.h (k-major x k-major, K % 64 == 0 only)
constexpr int kBigTile = 6;
constexpr TileSpec kTiles[kNumTiles] = {{128, 128, 64}, {64, 128, 64}, {64, 64, 64}, {32, 64, 64}, {32, 64, 256}, {32, 32, 256},
                                        {256, 256, 64}};

int big_min_wgs() { return opt(OPT_BIG_MIN); }
bool big_ok(int M, int N, int K) {
  const long wgs = (long)((M + 255) / 256) * ((N + 255) / 256);
  return M > 128 && K % 64 == 0 && K >= 128 && big_min_wgs() > 0 && wgs >= big_min_wgs();
}

// The phase-interleaved persistent 256x256 kernel (gemm8p.h) and the forward built on it that never stores the logits: same gate
// as the 256x256 tile, plus an even number of K steps and operands addressable with 32-bit byte offsets.
// (the storing / filtering uses of the same kernel -- dprhot_sim_fwd, dprhot_search -- keep the 256-tile gate they were measured at)
bool g8_ok(int M, int N, int K) {
  return !opt(OPT_NO_NL) && force_tile() < 0 && big_ok(M, N, K) && K % 128 == 0 && (double)M * K * 2 < 4.0e9 && (double)N * K * 2 < 4.0e9;
}
bool nl_ok(int M, int N, int K) {
  const long wgs = (long)((M + 255) / 256) * ((N + 255) / 256);
  const int need = big_min_wgs() < opt(OPT_NL_MIN) ? big_min_wgs() : opt(OPT_NL_MIN);
  const bool big = M > 128 && K % 64 == 0 && K >= 128 && big_min_wgs() > 0 && wgs >= need;  // (big_ok with this plan's own tile count)
  return !opt(OPT_NO_NL) && force_tile() < 0 && big && K % 128 == 0 && (double)M * K * 2 < 4.0e9 && (double)N * K * 2 < 4.0e9;
}

// One-pass training forward on the 128 x 128 LDS-DMA tile (EpiSimP + g8_lse_p2g_kernel): more than 128 rows, whole 64-deep K steps,
// at least 32 128-wide tiles (option nl128_min_tiles); where the 256 x 256 forward qualifies too, at most one round of them (option
// nl128_max_tiles) and fewer than 256 tiles of 256 x 256 -- where it does not (K a multiple of 64 but not of 128), every size; 32-bit
// element offsets.
bool nl128_ok(int M, int N, int K) {
  if (!opt(OPT_NL128) || opt(OPT_NO_NL) || !opt(OPT_G128_DMA) || !opt(OPT_NL_P16) || force_tile() >= 0) return false;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128), t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
  return M > 128 && K % 64 == 0 && K >= 64 && N % 8 == 0 && N >= 1024 && t128 >= opt(OPT_NL128_MIN_TILES) && (!nl_ok(M, N, K) || (t128 <= opt(OPT_NL128_BELOW) && t256 < 256)) &&
         (double)M * K < 4.0e9 && (double)N * K < 4.0e9 && (N + 63) / 64 <= 1024 * 64;
}

// Tile for D[M,N] with contraction length K: the largest tile (BM capped by M) that still yields `want`
// workgroups, else the smallest; small-M problems with a long K use the BK=256 variants (latency-bound:
// fewer, fatter K steps keep a whole K range in flight).
int pick_tile(int M, int N, int K, int splits_hint, int want) {
  if (force_tile() >= 0 && force_tile() < kBigTile) return force_tile();
  if (M <= 32) {
    if (K < 256) return 3;
    const long wg64 = (long)cdiv(N, 64) * splits_hint;
    return wg64 >= want / 2 ? 4 : 5;
  }
  const int first = M <= 64 ? 1 : 0;
  // (round 6: the 128 x 128 tile is the LDS-DMA kernel now and wins earlier -- 768 x 8192 x 768 stored-logits GEMM 30.0 -> 22.6 us with 384
  //  workgroups where the plan wanted 512 and took 64 x 128 tiles; a tie at 256 workgroups, a loss below: scratch/sim_tile_ab.py)
  if (first == 0 && opt(OPT_G128_DMA) != 0 && K % 64 == 0 &&
      (long)cdiv(M, kTiles[0].bm) * cdiv(N, kTiles[0].bn) * splits_hint * 8 >= (long)want * 5)
    return 0;
  for (int t = first; t <= 2; ++t) {
    const long wgs = (long)cdiv(M, kTiles[t].bm) * cdiv(N, kTiles[t].bn) * splits_hint;
    if (wgs >= want) return t;
  }
  return 2;
}

template <int BM, int BN, int BK, bool AK, bool BKM, bool TR, class Epi, bool AF = false, bool BF = false>
int launch_one(const GemmArgs& a, const Epi& epi, int splits, hipStream_t st) {
  auto kern = gemm_bf16_kernel<BM, BN, BK, 2, 2, AK, BKM, TR, Epi, AF, BF>;
  constexpr size_t lds = gemm_lds_bytes<BM, BN, BK, AK, BKM>();
  static AttrOnce attr_done;  // benign race: idempotent
  if (lds > 48 * 1024 && !attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), splits);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a, epi);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

// persistent (one workgroup per CU walking its tiles with the K pipeline running across tile boundaries) when the
// epilogue stores next to nothing; one workgroup per tile when it stores the whole tile: a finished workgroup's stores
// drain while its successor on the CU is already loading, a persistent one would have to wait for them
template <class Epi, bool PERSIST>
int launch_big(const GemmArgs& a, const Epi& epi, hipStream_t st) {
  auto kern = gemm256_kernel<Epi, PERSIST>;
  static AttrOnce attr_done;  // benign race: idempotent
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2_lds_total));
    attr_done = true;
  }
  const int nbx = cdiv(a.N, G2_B), nby = cdiv(a.M, G2_B);
  const int grid = (!PERSIST || nbx * nby < kNumCU) ? nbx * nby : kNumCU;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G2_THREADS), g2_lds_total, st, a, epi, nbx, nby);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

// SCHED: 4 = four phases of 8 MFMAs per K step, 2 = two phases of 16 (bit-identical; the dScores pass, whose store epilogue delays
// the first DMA waits of the next tile, is consistently ~4 % faster on the two-phase schedule: 111-114 vs 116-121 us at 8192^2 x 768)
template <class Epi, int SCHED = 4>
int launch_g8(const GemmArgs& a, const Epi& epi, hipStream_t st) {
  auto kern = gemm8p_kernel<Epi, 0, SCHED>;
  static AttrOnce attr_done;  // benign race: idempotent
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g8_lds_total));
    attr_done = true;
  }
  const int nbx = cdiv(a.N, G2_B), nby = cdiv(a.M, G2_B);
  const bool stores_tile = std::is_same<Epi, Epi8G>::value || std::is_same<Epi, Epi8Store>::value || std::is_same<Epi, Epi8StatsP>::value;
  const int grid = (nbx * nby < kNumCU || (stores_tile && opt(OPT_G8_ONE_TILE) != 0)) ? nbx * nby : kNumCU;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G2_THREADS), g8_lds_total, st, a, epi, nbx, nby);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}
// the inputs every gemm8p.h sim epilogue shares
Epi8Base g8_base(const dprhot_bf16* Q, int B, int Nc, const int64_t* y, int64_t y_offset, const uint8_t* colmask, float inv_T, float* gold) {
  EpiSim e{nullptr, colmask, B, Nc, inv_T, nullptr, nullptr, y, y_offset, gold, nullptr, 0};
  Epi8Base b;
  b.sim = with_packed_mask(e);
  b.dummy = Q;
  return b;
}

template <bool AK, bool BKM, class Epi>
int launch_gemm(int tile, const GemmArgs& a, const Epi& epi, int splits, hipStream_t st) {
  if constexpr (AK && BKM) {
    if (tile == kBigTile) {
      if (splits != 1 || a.K % 64 != 0) return fail(DPRHOT_E_UNSUPPORTED, "256x256 tile: no split-K, K %% 64 == 0");
      return launch_big<Epi, std::is_same<Epi, EpiFilter>::value>(a, epi, st);
    }
  }
  constexpr bool needs_tr = !(AK && BKM);
  const bool tr = needs_tr ? use_tr() : false;
  // (K: whole 64-deep steps -- or, with B mn-major, any K from 64 on (a multiple of 8 when A is k-major): the partial last step is read against zeros, g1_tile)
  if (tile == 0 && opt(OPT_G128_DMA) != 0 && (tr || !needs_tr) && (a.K % 64 == 0 || (!BKM && a.K >= 64 && (!AK || a.K % 8 == 0))) && a.kchunk % 64 == 0 && a.M >= 8 && a.N >= 8 &&
      (double)(AK ? a.M : a.K) * a.lda < 4.0e9 && (double)(BKM ? a.N : a.K) * a.ldb < 4.0e9) {  // (32-bit element offsets inside one K range)
    // the same tile with LDS-DMA staging (gemm128d.h): same images, same fragments, same epilogues
    auto kern = gemm128d_kernel<AK, BKM, Epi>;
    static AttrOnce attr_done;
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g1_lds_bytes));
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(cdiv(a.N, 128), cdiv(a.M, 128), splits), dim3(256), g1_lds_bytes, st, a, epi);
    HIP_TRY(hipGetLastError());
    return DPRHOT_OK;
  }
#define DPRHOT_TILE_CASE(T, BM, BN, BK_)                                                       \
  case T:                                                                                      \
    if constexpr (needs_tr) {                                                                  \
      return tr ? launch_one<BM, BN, BK_, AK, BKM, true, Epi>(a, epi, splits, st)              \
                : launch_one<BM, BN, BK_, AK, BKM, false, Epi>(a, epi, splits, st);            \
    } else {                                                                                   \
      return launch_one<BM, BN, BK_, AK, BKM, false, Epi>(a, epi, splits, st);                 \
    }
  switch (tile) {
    DPRHOT_TILE_CASE(0, 128, 128, 64)
    DPRHOT_TILE_CASE(1, 64, 128, 64)
    DPRHOT_TILE_CASE(2, 64, 64, 64)
    DPRHOT_TILE_CASE(3, 32, 64, 64)
    DPRHOT_TILE_CASE(4, 32, 64, 256)
    DPRHOT_TILE_CASE(5, 32, 32, 256)
  }
#undef DPRHOT_TILE_CASE
  return fail(DPRHOT_E_UNSUPPORTED, "bad tile id %d", tile);
}

// sim GEMM reading fp32 operands directly (AF: q is fp32, BF: c is fp32); bf16 copies go to a.Acopy / a.Bcopy
template <bool AF, bool BF>
int launch_sim_f32(int tile, const GemmArgs& a, const EpiSim& epi, int splits, hipStream_t st) {
  switch (tile) {
    case 0: return launch_one<128, 128, 64, true, true, false, EpiSim, AF, BF>(a, epi, splits, st);
    case 1: return launch_one<64, 128, 64, true, true, false, EpiSim, AF, BF>(a, epi, splits, st);
    case 2: return launch_one<64, 64, 64, true, true, false, EpiSim, AF, BF>(a, epi, splits, st);
    case 3: return launch_one<32, 64, 64, true, true, false, EpiSim, AF, BF>(a, epi, splits, st);
    case 4: return launch_one<32, 64, 256, true, true, false, EpiSim, AF, BF>(a, epi, splits, st);
    case 5: return launch_one<32, 32, 256, true, true, false, EpiSim, AF, BF>(a, epi, splits, st);
  }
  return fail(DPRHOT_E_UNSUPPORTED, "bad tile id %d", tile);
}

// ---- backward pair: dC (tile 0 or 2) next to dQ (tile 0, 2 or 4) in one launch --------------------------
template <class C1, class C2>
int launch_pair_one(const GemmArgs& a1, const EpiScaleF32& e1, const GemmArgs& a2, const EpiScaleF32& e2, int splits2,
                    hipStream_t st) {
  auto kern = gemm_pair_kernel<C1, C2, EpiScaleF32, EpiScaleF32>;
  constexpr size_t lds = C1::lds > C2::lds ? C1::lds : C2::lds;
  static AttrOnce attr_done;
  if (lds > 48 * 1024 && !attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  const int nbx1 = cdiv(a1.N, C1::BN), nby1 = cdiv(a1.M, C1::BM);
  const int nbx2 = cdiv(a2.N, C2::BN), nby2 = cdiv(a2.M, C2::BM);
  const long blocks = (long)nbx1 * nby1 + (long)nbx2 * nby2 * splits2;
  if (blocks > 0x7fffffffL) return fail(DPRHOT_E_UNSUPPORTED, "grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a1, e1, nbx1, nby1, a2, e2, nbx2, nby2);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

template <bool TR>
int launch_pair_tr(int t1, int t2, const GemmArgs& a1, const EpiScaleF32& e1, const GemmArgs& a2, const EpiScaleF32& e2,
                   int splits2, hipStream_t st) {
  using DC0 = GemmCfg<128, 128, 64, false, false, TR>;
  using DC2 = GemmCfg<64, 64, 64, false, false, TR>;
  using DQ0 = GemmCfg<128, 128, 64, true, false, TR>;
  using DQ2 = GemmCfg<64, 64, 64, true, false, TR>;
  using DQ4 = GemmCfg<32, 64, 256, true, false, TR>;
  if (t1 == 0 && t2 == 0) return launch_pair_one<DC0, DQ0>(a1, e1, a2, e2, splits2, st);
  if (t1 == 0 && t2 == 2) return launch_pair_one<DC0, DQ2>(a1, e1, a2, e2, splits2, st);
  if (t1 == 0 && t2 == 4) return launch_pair_one<DC0, DQ4>(a1, e1, a2, e2, splits2, st);
  if (t1 == 2 && t2 == 0) return launch_pair_one<DC2, DQ0>(a1, e1, a2, e2, splits2, st);
  if (t1 == 2 && t2 == 2) return launch_pair_one<DC2, DQ2>(a1, e1, a2, e2, splits2, st);
  if (t1 == 2 && t2 == 4) return launch_pair_one<DC2, DQ4>(a1, e1, a2, e2, splits2, st);
  return fail(DPRHOT_E_UNSUPPORTED, "no fused backward for tiles (%d,%d)", t1, t2);
}

int check_shape(int B, int Nc, int d) {
  REQUIRE(B > 0 && Nc > 0 && d > 0, "non-positive shape B=%d Nc=%d d=%d", B, Nc, d);
  REQUIRE(d % 8 == 0, "d=%d must be a multiple of 8 (16-byte rows)", d);
  REQUIRE(Nc % 8 == 0, "Nc=%d must be a multiple of 8 (16-byte rows; pad with masked columns)", Nc);
  return DPRHOT_OK;
}

// split-K plan of dQ = G x C  (M = B, N = d, K = Nc); tile restricted to what the fused pair kernel has
struct DqPlan { int tile, splits, kchunk; bool big; };
// the 256x256 LDS-DMA backward pair (gemm256.h): both contraction lengths multiples of 64, enough 256-row blocks to
// matter, operands addressable with 32-bit element offsets
bool big_bwd_ok(int B, int Nc, int d) {
  if (opt(OPT_NO_BIG_BWD) || force_tile() >= 0 || big_min_wgs() <= 0 || unfused_bwd()) return false;
  if (B % 64 != 0 || Nc % 64 != 0) return false;
  if ((double)B * Nc >= 4.0e9 || (double)Nc * d >= 4.0e9) return false;
  const long units = (long)((Nc + 255) / 256 + (B + 255) / 256) * ((d + 255) / 256);
  if (big_min_wgs() < 96) return units >= big_min_wgs();  // DPRHOT_BIG_MIN lowered: tests push small shapes through
  return B >= 256 && Nc >= 1024 && d >= 256 && units >= 96;
}
DqPlan dq_plan(int B, int Nc, int d) {
  DqPlan p;
  p.big = big_bwd_ok(B, Nc, d);
  if (p.big) {
    // dQ units as long as dC units (K = B each): split Nc into ~Nc/B slices, at most 16
    int splits = (Nc + B - 1) / B;
    // at most 16 slices (bounds the fp32 slab traffic) -- 32 under 512 rows when the launch takes more than one round of the chip anyway:
    // the dQ units are then as short as the dC units (K = B) instead of the long pole at the end of the grid.  Measured, step us
    // (profiles/r06_dq_cap_ab.txt): 256 x 32768 172.6 -> 146.6, 384 x 16384 116.5 -> 107.4, 448 x 16384 124.1 -> 113.5; where 16 slices
    // fit one round, 32 lose or change nothing (256 x 16384 86.0 -> 100.3, 256 x 8192 60.2 -> 60.9).
    int cap = 16;
    if (B < 512) {
      const int forced = opt(OPT_DQ_CAP_FEW);
      const long at16 = (long)cdiv(Nc, 256) * cdiv(d, 256) + (long)cdiv(B, 256) * cdiv(d, 256) * 16;
      cap = forced > 0 ? forced : (at16 > kNumCU ? 32 : 16);
    }
    if (splits > cap) splits = cap;
    // Long context axis (the dQ units run in a launch of their own: dprhot_inbatch_bwd's long_axis rule): ONE round of units on the
    // 256 CUs instead -- 1024 x 65536 x 768: 12 tiles x 21 slices = 252 units where 16 slices left a quarter of the chip idle;
    // 2048 x 65536: 24 x 10 = 240 units in one round where 24 x 16 = 384 took two (round 6)
    if (B >= 1024 && B <= 2048 && (long)Nc >= 32L * B && opt(OPT_DQ_ONE_ROUND) != 0) {
      const int tiles = cdiv(B, 256) * cdiv(d, 256);
      int one = kNumCU / tiles;
      if (one > 32) one = 32;
      if (one >= 1 && one < (Nc + B - 1) / B) splits = one;
    }
    if (splits < 1) splits = 1;
    p.tile = kBigTile;
    p.kchunk = cdiv(cdiv(Nc, 128), splits) * 128;  // an even number of 64-deep K steps per slice (gemm8pb.h)
    p.splits = cdiv(Nc, p.kchunk);
    return p;
  }
  if (B <= 32 && Nc >= 256) p.tile = 4;
  else if (B <= 256) p.tile = 2;
  else p.tile = 0;
  const TileSpec ts = kTiles[p.tile];
  const int tiles = cdiv(B, ts.bm) * cdiv(d, ts.bn);
  const int ksteps = cdiv(Nc, ts.bk);
  int splits = cdiv(kNumCU, tiles);
  const int min_steps = ts.bk >= 256 ? 1 : 2;
  if (splits > ksteps / min_steps) splits = ksteps / min_steps;
  if (splits > 16) splits = 16;  // bounds the fp32 slab traffic (splits * B * d * 4 bytes each way)
  if (splits < 1) splits = 1;
  p.kchunk = cdiv(ksteps, splits) * ts.bk;
  p.splits = cdiv(Nc, p.kchunk);
  return p;
}

// The backward pair on the 128 x 128 LDS-DMA tile (gemm128d_pair_kernel): dC over K = B (any count above 64: the last step may be partial,
// g1_tile), dQ over K = Nc in slices of whole steps (any multiple of 8 -- the packed layout of 2 or 4 ranks is rarely a multiple of 64),
// 32-bit element offsets, the LDS transpose read for the mn-major operands.
struct Pair128Plan { bool ok; int splits, kchunk; };
Pair128Plan pair128_plan(int B, int Nc, int d) {
  Pair128Plan p{};
  p.ok = opt(OPT_PAIR128) != 0 && opt(OPT_G128_DMA) != 0 && use_tr() && force_tile() < 0 && !unfused_bwd() && Nc % 8 == 0 &&
         d % 8 == 0 && B >= 64 && d >= 8 && (double)B * Nc < 4.0e9 && (double)Nc * d < 4.0e9;
  // K slices of a dQ tile.  The dQ units lead the grid and the (shorter) dC tiles fill in behind them, so the launch is as long as the
  // larger of one dQ unit and the chip's share of all K steps; every slice costs a slab of B x d fp32 written and read again.  Two
  // lower bounds, the larger wins, at most 16: (balance) a dQ unit no longer than twice the per-slot average of the launch's K steps
  // (512 slots: two workgroups per CU); (traffic is cheap) slabs up to a quarter of the dC bytes: Nc / (4 B).  Measured against 8 / 16 /
  // 32 slices (profiles/r06_pair128_ab.txt): 256 x 8192 22.2 / 22.7 / 26.9 us, 384 x 8192 24.6 / 28.9 / 29.4, 512 x 8192 28.5 / 33.1 /
  // 33.2, 448 x 16384 42.8 / 45.3 / 53.0, 256 x 32768 55.8 / 50.2 / 54.6.
  const double avg = (double)B * Nc * ((d + 127) / 128 * 128) / 268435456.0;  // K steps per slot: 2 B Nc d / (128 * 128 * 64) / 512
  int s = (int)((double)Nc / (128.0 * (avg < 1.0 ? 1.0 : avg)) + 0.999);
  if (s < Nc / (4 * B)) s = Nc / (4 * B);
  if (s > 16) s = 16;
  if (opt(OPT_PAIR128_CAP) > 0) s = opt(OPT_PAIR128_CAP);
  if (s > Nc / 256) s = Nc / 256;
  if (s < 1) s = 1;
  p.kchunk = cdiv(cdiv(Nc, 64), s) * 64;
  p.splits = cdiv(Nc, p.kchunk);
  return p;
}
// Where it runs (measured against the plan it replaces at 60 shapes, profiles/r06_pair128_ab.txt): under 1024 query rows everywhere, and up
// to B x Nc = 2^25 scores above that.  Beyond, the 256 x 256 kernels keep the backward: their tiles need half the operand bytes per flop,
// and at that size they fill the chip (8192^2: 191 against 200 us; 1024 x 65536: 258 / 295, the 128-wide tiles' pieces alias at a 128 KiB
// row pitch; 4096 x 32768: 410 / 458) -- although 4096 x 16384 (216 / 201) and 2048 x 49152 (393 / 359) would still gain.
// What it replaced: the register-staged pair under the 256 x 256 gate (2048 x 4096 x 768: 70 -> 39 us, 1024 x 4096 44 -> 27, 4096 x 4096
// 86 -> 62), the 256 x 256 pair of the few-rows shapes (256 x 8192 29.7 -> 21.0, 768 x 16384 76.8 -> 57.0, 256 x 32768 76.9 -> 52.5,
// 1024 x 16384 79.5 -> 70.3) and the two separate launches of the long axis under 512 rows (512 x 65536 184 -> 164, 128 x 32768 63 -> 37).
bool pair128_use(int B, int Nc, int d) {
  if (!pair128_plan(B, Nc, d).ok) return false;
  if (opt(OPT_PAIR128) == 2) return true;
  // (not under 2048 contexts -- 1024 with at most 512 rows: a dozen 128-wide tiles lose to the smaller register-staged ones there,
  //  128 x 128 5.8 against 3.8 us, 512 x 512 12.6 / 11.6, 1024 x 1024 19.8 / 18.9; 512 x 1024 14.7 / 17.2 and 512 x 2048 15.8 / 21.1 win)
  if (Nc < 1024 || (Nc < 2048 && B > 512)) return false;
  // (above the bound only the shapes the 256 x 256 pair does not take -- row or column counts that are no multiples of 64: they would run
  //  on the register-staged pair, 2000 x 32000 x 768: 448 against 262 us)
  if (B < 1024 || !big_bwd_ok(B, Nc, d)) return true;
  if ((double)B * Nc > 33554432.0) return false;
  // From 1024 rows on the 256 x 256 pair keeps the shapes whose units fill whole rounds of the 256 CUs: with d = 1024 (four column tiles)
  // 1024 x 8192, 2048 x 8192 and 4096 x 8192 are exactly 256 units and 2048 x 16384 is 512 -- 48.9 / 73.2 / 124.9 / 141.2 us against 55.7 /
  // 91.8 / 132.4 / 148.5 on the 128-wide pair -- where d = 768 leaves them at 192 / 384 units, three quarters of a round (tie, tie, tie,
  // 124 -> 111 for the 128-wide pair): profiles/r06_pair128_dim_ab.txt.
  const DqPlan p8 = dq_plan(B, Nc, d);
  const long u8 = ((long)cdiv(Nc, 256) + (long)cdiv(B, 256) * p8.splits) * cdiv(d, 256);
  const long rounds = (u8 + kNumCU - 1) / kNumCU;
  return u8 * 10 < rounds * kNumCU * 9;  // (fill below 0.9)
}

int dc_tile(int B, int Nc, int d) { return ((long)cdiv(Nc, 128) * cdiv(d, 128) >= kNumCU) ? 0 : 2; }

// Few query rows against many contexts (skinny.h): B <= 128 (a multiple of 32), d a multiple of 128 up to 1024, 512 (256 above 64 rows) <= Nc <= 16384
// (beyond that the per-unit recomputation of the row logsumexp from Nc / 128 tile values stops being cheap).  Lower bound, measured
// against the three-launch plan and the fused small step (scripts/bench_rankstep.py, round 3): 128 x 520 18.2 vs 22.0 us, 128 x 1032
// 18.5 vs 26.2, 96 x 776 17.8 vs 22.8, 64 x 776 17.1 vs 18.0, 64 x 1088 18.2 vs 20.1 (64 x 520: 17.0 both); with B = 32 the fused
// small step wins up to ~1000 columns (776: 16.2 vs 16.4; 904: 16.4 vs 16.5) and loses above (1056: 17.4 vs 16.6).
struct SkPlan { bool ok; int nt, nts, scols, nrb, ksteps, nslices; };
SkPlan sk_plan(int B, int Nc, int d) {
  const bool off = opt(OPT_NO_SKINNY) != 0, no_small = opt(OPT_NO_SMALL_STEP) != 0;  // (no_small lets this plan take the small-step shapes)
  const int min_nc = B > 64 ? 256 : 512;  // (128 x 264: 17.8 vs 21.1 us; 96 x 392: 17.5 vs 21.2; 64 x 392: 17.0 vs 16.8)
  const bool fused_small = (B <= SS_ROWS && Nc <= 1024) || (B <= SS_MAXB && Nc <= 256);  // shapes the fused small step keeps (small_step_ok)
  SkPlan p{};
  p.ok = !off && force_tile() < 0 && !unfused_bwd() && B <= SK_MAXB && B % 32 == 0 && d % 128 == 0 && d >= 128 && d <= 1024 && Nc >= min_nc &&
         Nc <= 16384 && (no_small || !fused_small);
  p.nt = cdiv(Nc, SK_COLS);
  {
    // sim unit width: 128 columns x 4 ring slots, or 64 columns x 8 slots (twice the units, two thirds of a unit's K range in flight).
    // The narrow unit pays when the wide ones would leave most of the chip idle (B = 32 x Nc = 2112, cfg2 gathered over 8 ranks: 17
    // wide units, step 18.8 -> 17.6 us); with every CU busy it loses (cfg3 per rank, 260 wide units: sim 12.4 vs 11.0 us, and the G
    // launch then folds twice the tile statistics).  DPRHOT_SK_COLS=64 / 128 forces one.
    const int forced = opt(OPT_SK_COLS);
    const bool few = cdiv(B, SK_ROWS) * cdiv(Nc, SK_COLS) < kNumCU / 2;
    const bool narrow = forced == 64 || (forced != 128 && few);
    p.scols = (narrow && cdiv(Nc, SK_SCOLS) <= 8 * SK_MAXG) ? SK_SCOLS : SK_COLS;
  }
  p.nts = cdiv(Nc, p.scols);  // statistics tiles = sim units per row block
  p.nrb = cdiv(B, SK_ROWS);
  if (!p.ok) return p;  // (also keeps d < 64 away from the division below: every entry point derives this plan for its workspace layout)
  const int nk = cdiv(Nc, 64), ndt = d / SK_QN;
  int ns = kNumCU / 2 / ndt;  // dQ units: (slice of contexts) x (64 columns of d); ~half a unit per CU measured best: the
  if (nk <= 8) ns = 1;  // up to 512 contexts one slice is short enough, and it needs no partial sums and no reduction launch (128 x 264: 17.8 -> 15.7 us;
                        // at 520 contexts it is a tie, beyond that the unsplit units are the launch's long pole)
  if (opt(OPT_SK_DQ_SLICES) > 0) ns = opt(OPT_SK_DQ_SLICES);
  if (ns < 1) ns = 1;         // units run next to the dC units, and fewer slices mean fewer partial sums to write and re-read
  if (ns > 64) ns = 64;   // bounds the fp32 partial traffic
  p.ksteps = cdiv(nk, ns);
  p.nslices = cdiv(nk, p.ksteps);
  return p;
}

// The few-rows step without its dScores launch (skinny.h, round 4: sk_sim_kernel writes the tile-local softmax, sk_bwdf_kernel derives
// the row logsumexp and the per-tile factors itself).  nts = statistics tiles, nk = 64-context steps of the dQ units (tile space: the
// packed layout's remapped tiles skip the header rows).  128-column statistics tiles only (a dC unit is one tile), at most 128 of
// them, the dQ slices as the plan cut them and short enough for a unit's factor table (SK_FT tiles).
// Where it is chosen (option sk_fused = 1): B x Nc >= 2^19 -- measured with eight-wave units (profiles/r04_rank_shapes_w8.txt, one box, us per
// step without / with the dScores launch): 128 x 8256 x 768 25.4-26.1 / 29.5; 128 x 6208 x 768 23.7 / 25.5; 128 x 4160 x 768 21.9 / 22.4;
// 128 x 4160 x 1024 23.2 / 24.9; 128 x 8256 x 256 18.8 / 20.2; 96 x 6208 x 768 22.6 / 24.2; 64 x 8256 x 512 / 768 / 1024 20.6 / 20.8,
// 23.4 / 25.7, 29.6 / 32.4; below 2^19 scores (128 x 2112, 64 x 4160) the two plans tie.  (With four-wave units the plan lost below
// 2^20: r04_fused_ab.txt.)  What it always buys is accuracy: the gold terms stay in fp32 (gradients 3e-5 of max |grad| from an fp64
// reference instead of 1e-3).
struct SkFused { bool ok; int ksteps, nslices; bool pair; int tpu, ns; };  // nslices x ksteps: the dQ units' slices; pair: sk_bwdp_kernel with ns slices of tpu statistics tiles
SkFused sk_fused_plan(const SkPlan& sk, int nts, int nk, int B, int Nc, int d) {
  SkFused f{false, 0, 0, false, 0, 0};
  const int mode = opt(OPT_SK_FUSED);
  if (mode == 0 || (mode == 1 && (long)B * Nc < (1l << 19))) return f;
  if (!sk.ok || sk.scols != SK_COLS || nts > 128 || sk.nslices < 1 || sk.nslices > 64) return f;
  f.nslices = sk.nslices;
  f.ksteps = cdiv(nk, f.nslices);  // (a tiling with fewer steps may leave the last slices empty: their units write zero slabs)
  if ((f.ksteps + 1) / 2 + 1 > SK_FT) {
    // the plan's slices span more statistics tiles than a dQ unit's factor table holds (long rows, or few slices at d = 1024).
    // This form can cut its own, 13 steps each (7 tiles + 1) -- but those are shapes with more workgroups than the chip holds at
    // once, and there it measured SLOWER than the plan with the dScores launch (128 x 12352 x 768: 40.0 against 35.7 us;
    // 128 x 8256 x 1024: 33.6 / 31.8; 64 x 8256 x 1024: 30.2 / 29.6): only on request (sk_fused = 2)
    if (mode != 2) return f;
    f.ksteps = 2 * (SK_FT - 1) - 1;
    f.nslices = cdiv(nk, f.ksteps);
  }
  f.ok = f.nslices <= 64 && (f.ksteps + 1) / 2 + 1 <= SK_FT;
  if (f.ok && opt(OPT_SK_PAIR) != 0 && d % SK_QN == 0) {
    // one workgroup per CU (152 KiB of LDS each): at most kNumCU units, at most 16 slabs (one batch of the finishing launch's loads)
    const int ndt = d / SK_QN;
    int maxs = kNumCU / ndt;
    if (maxs > 16) maxs = 16;
    if (maxs < 1) maxs = 1;
    f.tpu = cdiv(nts, maxs);
    f.ns = cdiv(nts, f.tpu);
    f.pair = f.tpu <= SK_PT;
  }
  return f;
}

struct FwdPlan { bool short_rows; int tile, splits, kchunk, nt, tpr, cpt, threads, blocks; };
FwdPlan fwd_plan(int B, int Nc, int d);

// workspace carve-up (all offsets 256-byte aligned)
struct WsLayout {
  size_t header, gold, lse, rloss, part_m, part_s, logits, dq_part, total;
};
WsLayout ws_layout(int B, int Nc, int d) {
  WsLayout w;
  const size_t ntmax = (size_t)cdiv(Nc, 32);
  size_t off = 0;
  w.header = off; off += 256;
  w.gold = off; off += align256((size_t)B * 4);
  w.lse = off; off += align256((size_t)B * 4);    // no-logits forward: row logsumexp for the dScores pass / rank: gold logits
  w.rloss = off; off += align256((size_t)B * 4);  // no-logits forward: row losses / rank: counts
  w.part_m = off; off += align256((size_t)B * ntmax * 4);
  w.part_s = off; off += align256((size_t)B * ntmax * 4);
  {
    const FwdPlan fp = fwd_plan(B, Nc, d);  // short rows: split-K slabs of partial logits (4, or up to 32 for wide vectors)
    const size_t slabs = (Nc <= 4096 && (B <= 64 || fp.short_rows)) ? (size_t)(fp.short_rows && fp.splits > 4 ? fp.splits : 4) : 1;
    // no-logits shapes: the forward never stores S (a caller who wants the logits passes S_out)
    w.logits = off; off += nl_ok(B, Nc, d) ? 0 : align256((size_t)B * Nc * 4 * slabs);
  }
  const DqPlan p = dq_plan(B, Nc, d);
  const SkPlan sk = sk_plan(B, Nc, d);
  int slabs = sk.ok && sk.nslices > p.splits ? sk.nslices : p.splits;
  {
    const Pair128Plan pp = pair128_plan(B, Nc, d);
    if (pp.ok && pp.splits > slabs) slabs = pp.splits;
  }
  if (sk.ok && slabs < 16) slabs = 16;  // (sk_bwdp_kernel: up to 16 slices)
  if (sk.ok && slabs < cdiv(cdiv(Nc, 64), 2 * (SK_FT - 1) - 1)) slabs = cdiv(cdiv(Nc, 64), 2 * (SK_FT - 1) - 1);  // (sk_fused_plan's own slices)
  w.dq_part = off; off += align256((size_t)slabs * B * d * 4);
  w.total = off;
  return w;
}

// Forward plan, a pure function of the shape (both forward launches derive it independently).
//  short rows (the BASELINE training shapes): sim split over K into `splits` slabs, softmax with the row in registers
//  long rows: sim with per-tile statistics, then the streaming gfinal kernel
bool no_short() { return opt(OPT_NO_SHORT) != 0; }
FwdPlan fwd_plan(int B, int Nc, int d) {
  FwdPlan p{};
  // wide: vocabulary-wide vectors (CITADEL router, d = 30522): the contraction is long and the logit matrix small -- the
  // K range is what has to be spread over the chip (B = 128, Nc = 1024 are only 32 tiles of 64 x 64)
  const bool wide = d >= 4096 && Nc <= 4096 && B <= 256;
  p.short_rows = Nc <= 4096 && (B <= 64 || wide) && !no_short() && force_tile() < 0;
  if (!p.short_rows) {
    p.tile = (force_tile() < 0 && big_ok(B, Nc, d)) ? kBigTile : pick_tile(B, Nc, d, 1, 2 * kNumCU);
    p.splits = 1;
    p.kchunk = cdiv(d, kTiles[p.tile].bk) * kTiles[p.tile].bk;
    p.nt = cdiv(Nc, kTiles[p.tile].bn);
    return p;
  }
  p.tile = B <= 32 ? (d >= 256 ? 5 : 3) : 2;
  if (wide && B >= 128) p.tile = 2;  // 64 x 64 (128 x 128 with twice the slabs measured within 5 %: the fp32 ingest is the limiter)
  const int bk = kTiles[p.tile].bk, ksteps = cdiv(d, bk);
  int splits = ksteps < 4 ? ksteps : 4;
  if (wide) {  // one 128 x 128 tile per CU (wide.h; the register-staged kernel then gets more, thinner workgroups), at least 4 K steps
               // each, at most 32 slabs of partial logits
    const int per_split = cdiv(B, WD_B) * cdiv(Nc, WD_B);
    splits = cdiv(kNumCU, per_split);
    if (splits > ksteps / 4) splits = ksteps / 4;
    if (splits > 32) splits = 32;
    if (splits < 4) splits = 4;
  }
  // 768 < Nc <= SS_MAXNC (the BASELINE shape gathered over 3..4 ranks): enough column tiles without a K split, and
  // every workgroup of the fused softmax+backward kernel that follows re-reads the logits -- one slab, not four.  (Up to 768
  // columns the K split still pays: 32 x 528, cfg2 over two ranks, 13.25 -> 12.85 us; 32 x 648 13.75 -> 13.55.)
  if (B <= SS_ROWS && Nc > 768 && Nc <= SS_MAXNC) splits = 1;
  p.kchunk = cdiv(ksteps, splits) * bk;
  p.splits = cdiv(d, p.kchunk);
  const int cpr = Nc / 8;
  p.cpt = cpr > 256 ? 2 : 1;
  int tpr = 16;
  while (tpr * p.cpt < cpr) tpr <<= 1;
  p.tpr = tpr;
  int rpb = 1024 / tpr;
  if (rpb > B) rpb = B;
  if (p.splits > 4 && rpb > cdiv(B, 128)) rpb = cdiv(B, 128);  // many slabs: the launch is the slab read (up to 16 MB) -- >= 128 workgroups
  if (rpb < 1) rpb = 1;
  p.threads = rpb * tpr;
  p.blocks = cdiv(B, rpb);
  return p;
}

// the LDS-DMA fp32 sim (wide.h): vocabulary-wide short-row shapes whose K chunks are whole 32-float steps and whose operands are
// addressable with 32-bit byte offsets
bool wide_sim_ok(int B, int Nc, int d, const FwdPlan& fp) {
  return !opt(OPT_NO_WIDE) && d >= 4096 && Nc <= 4096 && B <= 256 && fp.short_rows && d % WD_KS == 0 && fp.kchunk % WD_KS == 0 &&
         (double)Nc * d * 4 < 4.0e9 && (double)B * d * 4 < 4.0e9;
}

// backward of the same shape class on the units of skinny.h (dprhot_inbatch_bwd): rows in blocks of 32 up to 128, vectors in whole
// 64-column dQ tiles, contexts within one dQ slice (<= 24 ring steps of 64), 32-bit element offsets
bool wide_bwd_ok(int B, int Nc, int d) {
  return !opt(OPT_NO_WIDE_BWD) && !opt(OPT_NO_SKINNY) && force_tile() < 0 && !unfused_bwd() && d >= 4096 && d % 64 == 0 && B <= SK_MAXB &&
         B % 32 == 0 && Nc % 8 == 0 && Nc >= 64 && Nc <= 1536 && (double)Nc * d < 4.0e9;
}

// the loss launch of the no-logits forward -- or, inside the one-call step, a note for launch_dq
int launch_loss_sum(const float* src, int n, float scale, float* out, hipStream_t st) {
  if (g_loss_defer.armed) {
    g_loss_defer.pending = true;
    g_loss_defer.src = src; g_loss_defer.n = n; g_loss_defer.scale = scale; g_loss_defer.out = out;
    return DPRHOT_OK;
  }
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, st, src, n, scale, out);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

// dQ = scale * (slab 0 + slab 1 + ...), in slab order; inside the one-call step the same launch also forms the loss sum (launch_loss_sum)
int launch_slab_sum(const float* part, int splits, int B, int d, float h_scale, const float* d_scale, float* dQ, hipStream_t st) {
  const size_t n4 = (size_t)B * d / 4;
  const int blocks = (int)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256);
  if (g_loss_defer.pending) {
    g_loss_defer.pending = false;
    hipLaunchKernelGGL(splitk_reduce_loss_kernel, dim3(blocks + 1), dim3(256), 0, st, part, splits, n4, h_scale, d_scale, dQ, g_loss_defer.src,
                       g_loss_defer.n, g_loss_defer.scale, g_loss_defer.out);
  } else {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, splits, n4, h_scale, d_scale, dQ);
  }
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int launch_dq(const dprhot_bf16* G, const dprhot_bf16* C, int B, int Nc, int d, float h_scale, const float* d_scale, float* dQ,
              char* ws, const WsLayout& wl, hipStream_t st, bool gemm_too) {
  DqPlan p = dq_plan(B, Nc, d);
  if (p.big) p.tile = 0;  // stand-alone dQ (dprhot_dq): same K slices on the 128x128 tile
  if (gemm_too) {
    GemmArgs a{G, C, B, d, Nc, Nc, d, p.kchunk};
    if (p.splits == 1) {
      EpiScaleF32 epi{dQ, B, d, h_scale, d_scale};
      return launch_gemm<true, false>(p.tile, a, epi, 1, st);
    }
    EpiScaleF32 epi{reinterpret_cast<float*>(ws + wl.dq_part), B, d, 1.0f, nullptr};
    if (int rc = launch_gemm<true, false>(p.tile, a, epi, p.splits, st)) return rc;
  }
  if (p.splits > 1) return launch_slab_sum(reinterpret_cast<const float*>(ws + wl.dq_part), p.splits, B, d, h_scale, d_scale, dQ, st);
  return DPRHOT_OK;
}

// ---- few rows x many contexts: the four launches of skinny.h -------------------------------------------------------
template <int NCH, int COLS, int SLOTS>
int launch_sk_sim_c(const SkSimArgs& a, int grid, hipStream_t st) {
  const size_t lds = sk_sim_lds();
  static AttrOnce attr_done[3];  // benign race: idempotent
  if constexpr (COLS == SK_COLS) {
    if (a.q != nullptr && opt(OPT_SK_SIM_PRIV) != 0 && grid <= kNumCU) {
      // wave-private rings, one workgroup per CU (round 5): as many chunks in flight per wave as 160 KiB of LDS hold next to the q block
      constexpr int PS = NCH <= 12 ? 6 : (NCH <= 14 ? 5 : 4);
      auto kern = sk_simp_kernel<NCH, PS>;
      const size_t ldsp = sk_simp_lds(NCH, PS);
      static AttrOnce attr_p;
      if (!attr_p) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
        attr_p = true;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), ldsp, st, a);
      HIP_TRY(hipGetLastError());
      return DPRHOT_OK;
    }
    if (a.q != nullptr && opt(OPT_SK_SIM_W8) != 0) {
      auto kern = sk_sim_kernel<NCH, true, COLS, SLOTS, 8>;
      if (!attr_done[2]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[2] = true;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
      HIP_TRY(hipGetLastError());
      return DPRHOT_OK;
    }
  }
  if (a.q != nullptr) {
    auto kern = sk_sim_kernel<NCH, true, COLS, SLOTS>;
    if (!attr_done[0]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done[0] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SK_THREADS), lds, st, a);
  } else {
    auto kern = sk_sim_kernel<NCH, false, COLS, SLOTS>;
    if (!attr_done[1]) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done[1] = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SK_THREADS), lds, st, a);
  }
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}
template <int NCH>
int launch_sk_sim(const SkSimArgs& a, int grid, int scols, hipStream_t st) {
  return scols == SK_COLS ? launch_sk_sim_c<NCH, SK_COLS, SK_SLOTS>(a, grid, st) : launch_sk_sim_c<NCH, SK_SCOLS, 2 * SK_SLOTS>(a, grid, st);
}

int sk_step(const float* q, const dprhot_bf16* Cb, dprhot_bf16* Qb, int B, int Nc, int d, const int64_t* y, int64_t y_offset,
            const uint8_t* colmask, float inv_T, float grad_scale, float h_scale, const float* d_scale, float* S_out, float* row_loss,
            float* row_lse, float* loss_sum, dprhot_bf16* G, float* dQ, float* dC_part, char* ws, const WsLayout& wl, const SkPlan& sk,
            hipStream_t st) {
  float* S = S_out ? S_out : reinterpret_cast<float*>(ws + wl.logits);
  float* tile_lse = reinterpret_cast<float*>(ws + wl.part_m);
  float* gold = reinterpret_cast<float*>(ws + wl.gold);
  // packed layout whose ranks hold a whole number of sim tiles: only the real rows are multiplied (SkSimArgs::tiles_per_rank)
  const bool remap = g_packed.base != nullptr && g_packed.rows_c > g_packed.n_ctx && g_packed.n_ctx % sk.scols == 0 &&
                     Nc % g_packed.rows_c == 0;
  const int tpr = remap ? g_packed.n_ctx / sk.scols : 0;
  const int nts = remap ? (Nc / g_packed.rows_c) * tpr : sk.nts;
  SkSimArgs a{q, nullptr, Cb, Qb, B, Nc, d, y, y_offset, colmask, inv_T, S, tile_lse, gold, g_packed.base, g_packed.rows_c,
              g_packed.n_ctx, g_packed.row_bytes, tpr};
  // G == NULL (nobody wants the dScores): three launches -- the sim launch leaves the tile-local softmax in the logit workspace
  const int nk_f = remap ? 2 * nts : cdiv(Nc, 64);
  const SkFused fz = sk_fused_plan(sk, nts, nk_f, B, Nc, d);
  const bool fused = G == nullptr && S_out == nullptr && fz.ok;
  if (G == nullptr && !fused) return fail(DPRHOT_E_INVALID, "few-rows step: G == NULL needs the fused-dScores plan (dprhot_step_wants_g)");
  // the finishing role inside the backward launch (sk_fin_unit): two-kinds form only; a dbg switch that silences units would strand it
  const bool tail = fused && !fz.pair && opt(OPT_SK_TAIL) != 0 && (opt(OPT_SK_DBG) & 7) == 0 && fz.nslices <= 62 && d <= 1024;
  unsigned* const tail_cnt = reinterpret_cast<unsigned*>(ws + wl.header + 128);
  // round 6: no finishing launch at all -- the dQ units add their tiles into dQ, zero-filled by the sim launch (two-kinds form, dQ wanted)
  const bool dq_atomic = fused && !fz.pair && !tail && opt(OPT_SK_DQ_ATOMIC) != 0 && opt(OPT_SK_W8) != 0 && dQ != nullptr && (opt(OPT_SK_DBG) & 15) == 0;
  if (fused) {
    a.S = nullptr;
    a.P = reinterpret_cast<uint16_t*>(ws + wl.logits);
    if (tail) a.zero_me = tail_cnt;
    if (dq_atomic) {
      a.zero_dq = dQ;
      a.zero_n4 = (int)((size_t)B * d / 4);
    }
  }
  const int grid1 = sk.nrb * nts;
  int rc = DPRHOT_OK;
  switch (d / 128) {  // NCH = d / 64
    case 1: rc = launch_sk_sim<2>(a, grid1, sk.scols, st); break;
    case 2: rc = launch_sk_sim<4>(a, grid1, sk.scols, st); break;
    case 3: rc = launch_sk_sim<6>(a, grid1, sk.scols, st); break;
    case 4: rc = launch_sk_sim<8>(a, grid1, sk.scols, st); break;
    case 5: rc = launch_sk_sim<10>(a, grid1, sk.scols, st); break;
    case 6: rc = launch_sk_sim<12>(a, grid1, sk.scols, st); break;
    case 7: rc = launch_sk_sim<14>(a, grid1, sk.scols, st); break;
    case 8: rc = launch_sk_sim<16>(a, grid1, sk.scols, st); break;
    default: return fail(DPRHOT_E_UNSUPPORTED, "skinny step: d=%d", d);
  }
  if (rc) return rc;
  if (fused) {
    if (g_dq_part != nullptr) return fail(DPRHOT_E_INVALID, "few-rows step without dScores: dQ is finished by the step itself (dprhot_train_dq_slabs = 0)");
    float* part = reinterpret_cast<float*>(ws + wl.dq_part);
    const int ndq = fz.nslices * (d / SK_QN), ndq_pad = (ndq + 7) & ~7, ndc = nts * (d / SK_DN);
    SkBwdFArgs b{a.P, Qb, Cb, B, Nc, d, tile_lse, gold, y, y_offset, nts, tpr, g_packed.rows_c, g_packed.n_ctx, grad_scale, h_scale, d_scale,
                 dC_part, g_dc_bf16 ? 1 : 0, g_packed.stamp_src != nullptr ? g_packed.rows_c : 0, g_packed.n_ctx, loss_sum, g_loss_scale,
                 row_loss, row_lse, fz.ksteps, fz.nslices, part, dQ, ndq_pad, opt(OPT_NT_STORES) ? 1 : 0, opt(OPT_SK_DBG)};
    b.reg_scale = opt(OPT_SK_DC_REGSCALE) != 0 ? 1 : 0;
    b.dq_atomic = dq_atomic ? 1 : 0;
    const size_t lds = sk_bwdf_lds();
    static AttrOnce attr_done[4];
    auto launch = [&](auto kern, int slot, int threads) -> int {
      if (!attr_done[slot]) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[slot] = true;
      }
      if (tail) {
        b.tail_cnt = tail_cnt;
        b.nfin = cdiv(B, threads / 256);
        b.tail_fence = opt(OPT_SK_TAIL) == 2 ? 1 : 0;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)(ndq_pad + ndc + b.nfin)), dim3((unsigned)threads), lds, st, b);
      return DPRHOT_OK;
    };
    static AttrOnce attr_pair[2];
    if (fz.pair) {
      // one kind of unit: (slice of fz.tpu statistics tiles) x (64 columns of d), one workgroup per CU
      b.ksteps = fz.tpu;
      b.nslices = fz.ns;
      const size_t ldsp = sk_bwdp_lds();
      const unsigned gridp = (unsigned)(fz.ns * (d / SK_QN));
      if (nts <= 64) {
        if (!attr_pair[0]) {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sk_bwdp_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
          attr_pair[0] = true;
        }
        hipLaunchKernelGGL(sk_bwdp_kernel<8>, dim3(gridp), dim3(512), ldsp, st, b);
      } else {
        if (!attr_pair[1]) {
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sk_bwdp_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
          attr_pair[1] = true;
        }
        hipLaunchKernelGGL(sk_bwdp_kernel<16>, dim3(gridp), dim3(512), ldsp, st, b);
      }
    } else {
      const bool w8 = opt(OPT_SK_W8) != 0;
      if (dq_atomic && w8) {  // (eight-wave units only: the form production runs)
        static AttrOnce attr_at[2];
        auto launch_at = [&](auto kern, int slot) -> int {
          if (!attr_at[slot]) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_at[slot] = true;
          }
          hipLaunchKernelGGL(kern, dim3((unsigned)(ndq_pad + ndc)), dim3(512), lds, st, b);
          return DPRHOT_OK;
        };
        rc = nts <= 64 ? launch_at(sk_bwdf_kernel<8, 8, true>, 0) : launch_at(sk_bwdf_kernel<16, 8, true>, 1);
      } else if (nts <= 64) rc = w8 ? launch(sk_bwdf_kernel<8, 8>, 2, 512) : launch(sk_bwdf_kernel<8, 4>, 0, SK_THREADS);
      else rc = w8 ? launch(sk_bwdf_kernel<16, 8>, 3, 512) : launch(sk_bwdf_kernel<16, 4>, 1, SK_THREADS);
      if (rc) return rc;
    }
    HIP_TRY(hipGetLastError());
    if (!tail && (opt(OPT_SK_DBG) & 8)) {
      // TIMING EXPERIMENT ONLY (sk_dbg & 8): the plain slab sum in the finishing launch's place (wrong dQ) -- which launch carries the gaps?
      const size_t n4 = (size_t)B * d / 4;
      hipLaunchKernelGGL(sk_dq_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, part, fz.nslices, n4, h_scale, d_scale, dQ);
      HIP_TRY(hipGetLastError());
    } else if (!tail && !dq_atomic) {
      // one workgroup per row where the row fits 256 threads (d <= 1024): the row's statistics are derived once, not once per part
      const int fthreads = d / 4 >= 256 ? 256 : cdiv(d / 4, 64) * 64;
      const int parts = cdiv(d / 4, fthreads);
      SkFinArgs f{part, fz.pair ? fz.ns : fz.nslices, fz.ksteps, nk_f, B, d, tile_lse, nts, gold, y, y_offset, Cb, grad_scale, h_scale, d_scale, dQ, parts,
                  fz.pair ? 1 : 0};
      hipLaunchKernelGGL(sk_dq_finish_kernel, dim3((unsigned)(B * parts)), dim3((unsigned)fthreads), 0, st, f);
      HIP_TRY(hipGetLastError());
    }
    return DPRHOT_OK;
  }
  float* rl = row_loss ? row_loss : reinterpret_cast<float*>(ws + wl.rloss);  // the backward launch forms the loss from the row losses
  {
    const int parts = B >= 128 ? 2 : (B >= 64 ? 4 : 8);  // >= 256 workgroups; a part is at most 4 x 256 chunks of 8 columns
    int pp = parts;
    while (cdiv(Nc / 8, pp) > 4 * SK_THREADS) pp *= 2;
    SkGArgs g{S, tile_lse, gold, nts, B, Nc, y, y_offset, grad_scale, G, rl, row_lse, pp};
    hipLaunchKernelGGL(sk_g_kernel, dim3((unsigned)(B * pp)), dim3(SK_THREADS), 0, st, g);
    HIP_TRY(hipGetLastError());
  }
  {
    float* part = g_dq_part != nullptr ? g_dq_part : reinterpret_cast<float*>(ws + wl.dq_part);
    const int ndq = sk.nslices * (d / SK_QN), ndq_pad = (ndq + 7) & ~7, ndc = sk.nt * (d / SK_DN);
    SkBwdArgs b{G, Qb, Cb, B, Nc, d, h_scale, d_scale, dC_part, rl, loss_sum, g_loss_scale, g_dc_bf16 ? 1 : 0,
                g_packed.stamp_src != nullptr ? g_packed.rows_c : 0, g_packed.n_ctx, sk.ksteps, sk.nslices, part, dQ, ndq_pad, opt(OPT_NT_STORES) ? 1 : 0};
    const size_t lds = sk_bwd_lds();
    static AttrOnce attr_done;
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sk_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done = true;
    }
    hipLaunchKernelGGL(sk_bwd_kernel, dim3((unsigned)(ndq_pad + ndc)), dim3(SK_THREADS), lds, st, b);
    HIP_TRY(hipGetLastError());
    if (g_dq_part == nullptr && sk.nslices > 1) {  // (else the caller's finishing launch forms dQ from the slabs: dprhot_rescale_grads;
                                                   //  one slice: the dQ units stored the scaled sum themselves)
      const size_t n4 = (size_t)B * d / 4;
      hipLaunchKernelGGL(sk_dq_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, part, sk.nslices, n4, h_scale, d_scale, dQ);
      HIP_TRY(hipGetLastError());
    }
  }
  return DPRHOT_OK;
}

}  // namespace

extern "C" {

int dprhot_version(void) { return DPRHOT_VERSION; }
const char* dprhot_last_error(void) { return g_err; }

int dprhot_set_option(const char* name, int value) {
  REQUIRE(name != nullptr, "NULL name");
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptDefs[i].name) == 0) {
      __atomic_store_n(&g_opt[i], value, __ATOMIC_RELAXED);
      __atomic_add_fetch(&g_opt_epoch, 1, __ATOMIC_RELAXED);
      return DPRHOT_OK;
    }
  return fail(DPRHOT_E_INVALID, "unknown option '%s'", name);
}

long long dprhot_options_epoch(void) { return __atomic_load_n(&g_opt_epoch, __ATOMIC_RELAXED); }

int dprhot_get_option(const char* name, int* h_value) {
  REQUIRE(name != nullptr && h_value != nullptr, "NULL pointer");
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptDefs[i].name) == 0) {
      *h_value = opt((OptId)i);
      return DPRHOT_OK;
    }
  return fail(DPRHOT_E_INVALID, "unknown option '%s'", name);
}

int dprhot_workspace_bytes(int B, int Nc, int d, size_t* h_out) {
  REQUIRE(h_out != nullptr, "h_out is NULL");
  if (int rc = check_shape(B, Nc, d)) return rc;
  *h_out = ws_layout(B, Nc, d).total;
  return DPRHOT_OK;
}

int dprhot_cast_bf16(const float* src, dprhot_bf16* dst, size_t n, void* stream) {
  REQUIRE(src && dst, "NULL pointer");
  REQUIRE(n % 8 == 0, "n=%zu must be a multiple of 8", n);
  REQUIRE(aligned16(src) && aligned16(dst), "pointers must be 16-byte aligned");
  if (n == 0) return DPRHOT_OK;
  const size_t n4 = n / 4, tiles = (n4 + 256 * CAST_UT - 1) / (256 * CAST_UT);
  REQUIRE(tiles < (1ull << 31), "n=%zu", n);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, src, dst, n4);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_prep(const float* q, size_t nq, dprhot_bf16* Qb, const float* c, size_t nc, dprhot_bf16* Cdst, void* stream) {
  REQUIRE(q && Qb && c && Cdst, "NULL pointer");
  REQUIRE(nq % 8 == 0 && nc % 8 == 0, "element counts must be multiples of 8 (nq=%zu nc=%zu)", nq, nc);
  REQUIRE(aligned16(q) && aligned16(Qb) && aligned16(c) && aligned16(Cdst), "pointers must be 16-byte aligned");
  const size_t n8 = (nq + nc) / 8;
  if (n8 == 0) return DPRHOT_OK;
  const int blocks = (int)((n8 + 255) / 256 > 2048 ? 2048 : (n8 + 255) / 256);
  hipLaunchKernelGGL(cast2_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q, Qb, nq / 8, c, Cdst, nc / 8);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_packed_rows(int n_ctx, int d, int* h_rows) {
  REQUIRE(h_rows != nullptr && n_ctx > 0 && d > 0 && d % 8 == 0, "bad argument");
  const int extra = cdiv(n_ctx, 2 * d);           // rows needed for n_ctx mask bytes
  // gathered column count stays a multiple of 8 -- and, from 2048 contexts per rank on (round 6), of 64: the 256 x 256 backward wants
  // whole 64-deep K steps over the gathered axis, and W x 8200 columns (8192 contexts + 8 header rows) sent a large-batch multi-rank
  // step to the 128 x 128 pair kernel; the (at most 56) extra rows are masked columns like every header row
  const int align = n_ctx >= 2048 ? 64 : 8;
  *h_rows = cdiv(n_ctx + extra, align) * align;
  return DPRHOT_OK;
}

int dprhot_pack_ctx(const float* c, const uint8_t* mask, int n_ctx, int d, dprhot_bf16* send, void* stream) {
  REQUIRE(c && send, "NULL pointer");
  REQUIRE(n_ctx > 0 && d > 0 && d % 8 == 0, "bad shape n_ctx=%d d=%d", n_ctx, d);
  REQUIRE(aligned16(c) && aligned16(send), "pointers must be 16-byte aligned");
  int rows_c = 0;
  dprhot_packed_rows(n_ctx, d, &rows_c);
  const size_t n8 = (size_t)rows_c * d / 8;
  const int blocks = (int)((n8 + 255) / 256 > 2048 ? 2048 : (n8 + 255) / 256);
  hipLaunchKernelGGL(pack_ctx_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c, mask, n_ctx, d, rows_c, send);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_unpack_mask(const dprhot_bf16* gathered, int W, int n_ctx, int d, uint8_t* colmask, void* stream) {
  REQUIRE(gathered && colmask, "NULL pointer");
  REQUIRE(W > 0 && n_ctx > 0 && d > 0 && d % 8 == 0, "bad shape W=%d n_ctx=%d d=%d", W, n_ctx, d);
  int rows_c = 0;
  dprhot_packed_rows(n_ctx, d, &rows_c);
  const int total = W * rows_c;
  hipLaunchKernelGGL(unpack_mask_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     gathered, W, n_ctx, d, rows_c, colmask);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_sim_fwd(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const uint8_t* colmask, float inv_T,
                   float* S, void* stream) {
  REQUIRE(Q && C && S, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(Q) && aligned16(C) && aligned16(S), "pointers must be 16-byte aligned");
  if (g8_ok(B, Nc, d) && !opt(OPT_NO_8P_STORE)) {
    // large score matrices (validation with the logits wanted, the head chunk of a retrieval): the phase-interleaved kernel with
    // the store epilogue that writes whole rows (Epi8Store); the round-1 256 x 256 kernel stays behind the option
    Epi8Store epi;
    static_cast<Epi8Base&>(epi) = g8_base(Q, B, Nc, nullptr, 0, colmask, inv_T, nullptr);
    epi.S = S;
    GemmArgs a8{Q, C, B, Nc, d, d, d, d};
    return launch_g8<Epi8Store, 2>(a8, epi, (hipStream_t)stream);
  }
  const int tile = (force_tile() < 0 && big_ok(B, Nc, d)) ? kBigTile : pick_tile(B, Nc, d, 1, 2 * kNumCU);
  GemmArgs a{Q, C, B, Nc, d, d, d, cdiv(d, kTiles[tile].bk) * kTiles[tile].bk};
  EpiSim epi{S, colmask, B, Nc, inv_T, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0};
  epi = with_packed_mask(epi);
  return launch_gemm<true, true>(tile, a, epi, 1, (hipStream_t)stream);
}

int dprhot_softmax_ce_fwd_bwd(const float* S, int B, int Nc, const int64_t* y, int64_t y_offset, float grad_scale,
                              const int64_t* row_win_start, int win_len, float* row_loss, float* row_lse, dprhot_bf16* G,
                              void* stream) {
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
  const WsLayout wl = ws_layout(B, Nc, d);
  if (dq_plan(B, Nc, d).splits > 1) {
    if (workspace == nullptr || workspace_bytes < wl.total)
      return fail(DPRHOT_E_WORKSPACE, "dq needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
    REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  }
  return launch_dq(G, C, B, Nc, d, h_scale, d_scale, dQ, static_cast<char*>(workspace), wl, (hipStream_t)stream, true);
}

int dprhot_dc(const dprhot_bf16* G, const dprhot_bf16* Q, int B, int Nc, int d, float h_scale, const float* d_scale,
              float* dC_part, void* stream) {
  REQUIRE(G && Q && dC_part, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(G) && aligned16(Q) && aligned16(dC_part), "pointers must be 16-byte aligned");
  // A(m = ctx column, k = query row) = G[k][m]  (mn-major, lda = Nc);  B(k, n) = Q[k][n]  (mn-major, ldb = d)
  GemmArgs a{G, Q, Nc, d, B, Nc, d, cdiv(B, 64) * 64};
  const EpiScaleF32 epi = with_loss_stamp(EpiScaleF32{dC_part, Nc, d, h_scale, d_scale});
  const int tile = force_tile() >= 0 && force_tile() <= 3 ? force_tile() : dc_tile(B, Nc, d);
  return launch_gemm<false, false>(tile, a, epi, 1, (hipStream_t)stream);
}

int dprhot_rank_of_gold(const float* S, int rows, int cols, const int64_t* y, int64_t y_offset, int64_t* rank, void* stream) {
  REQUIRE(S && y && rank, "NULL pointer");
  REQUIRE(rows > 0 && cols > 0, "bad shape rows=%d cols=%d", rows, cols);
  if (cols % 4 != 0) return fail(DPRHOT_E_UNSUPPORTED, "cols=%d must be a multiple of 4", cols);
  REQUIRE(aligned16(S), "S must be 16-byte aligned");
  hipLaunchKernelGGL(rank_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, S, rows, cols, y, y_offset, rank);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_pairwise_fwd(const float* q, const float* c, const uint8_t* mask, int B, int M, int d, float* S, void* stream) {
  REQUIRE(q && c && S, "NULL pointer");
  REQUIRE(B > 0 && M > 0 && d > 0, "bad shape B=%d M=%d d=%d", B, M, d);
  REQUIRE((long)B * M <= 0x7fffffffL, "too many pairs");
  hipLaunchKernelGGL(pairwise_fwd_kernel, dim3((unsigned)(B * M)), dim3(256), 0, (hipStream_t)stream, q, c, mask, B, M, d, S);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_pairwise_bwd(const float* g, const float* q, const float* c, int B, int M, int d, float* dq, float* dc, void* stream) {
  REQUIRE(g && q && c, "NULL pointer");
  REQUIRE(B > 0 && M > 0 && d > 0 && B <= 65535, "bad shape B=%d M=%d d=%d", B, M, d);
  hipLaunchKernelGGL(pairwise_bwd_kernel, dim3((unsigned)cdiv(d, 512), (unsigned)B), dim3(256), 0, (hipStream_t)stream, g, q, c, B, M, d, dq, dc);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

static int launch_topk(const TopkArgs& p, int rows, hipStream_t st) {
  if (p.k <= TK_KSMALL) {
    hipLaunchKernelGGL((topk_stream_kernel<1024, TK_KSMALL, 4>), dim3(rows), dim3(256), tk_lds_bytes(1024), st, p);
  } else if (p.k <= TK_KMAX) {
    hipLaunchKernelGGL((topk_stream_kernel<4096, TK_KMAX, 8>), dim3(rows), dim3(256), tk_lds_bytes(4096), st, p);
  } else {  // 1024 < k <= 4096: 8192 slots
    auto kern = topk_stream_kernel<8192, TK_KWIDE, 8>;
    static AttrOnce attr_done;
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tk_lds_bytes(8192)));
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(rows), dim3(256), tk_lds_bytes(8192), st, p);
  }
  return DPRHOT_OK;
}

int dprhot_topk_update(const float* S, int rows, int cols, int64_t ld, int64_t col_offset, int k, float* values,
                       int64_t* indices, int first, void* stream) {
  REQUIRE(S && values && indices, "NULL pointer");
  REQUIRE(rows > 0 && cols > 0 && ld >= cols && k > 0 && k <= TK_KWIDE, "bad shape rows=%d cols=%d ld=%lld k=%d", rows, cols,
          (long long)ld, k);
  REQUIRE(col_offset >= 0, "negative col_offset");
  TopkArgs p{S, rows, cols, (long long)ld, (long long)col_offset, k, values, indices, first ? 1 : 0, nullptr, nullptr};
  if (int rc = launch_topk(p, rows, (hipStream_t)stream)) return rc;
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_topk(const float* S, int rows, int cols, int k, float* values, int64_t* indices, void* stream) {
  REQUIRE(k <= cols, "k=%d > cols=%d", k, cols);
  return dprhot_topk_update(S, rows, cols, cols, 0, k, values, indices, 1, stream);
}

// workspace of dprhot_search: [nq x chunk] fp32 (scores of the first chunk / candidate values) + [nq x chunk] int32
// (candidate columns) + [nq] int32 (candidate counts)
static size_t search_ws_bytes(int nq, int chunk) {
  return (size_t)nq * (size_t)chunk * 8 + ((size_t)nq * 4 + 255) / 256 * 256;
}

// ---- k beyond the LDS-resident selection kernels (csrc/wideselect.h): state in HBM, radix select + block sort + merge passes ----
static size_t wsel_ws_bytes(int rows, int k) {
  return align256((size_t)rows * WSEL_REC * 4) + 2 * (align256((size_t)rows * k * 4) + align256((size_t)rows * k * 8));
}
int dprhot_topk_wide_workspace_bytes(int rows, int k, size_t* h_out) {
  REQUIRE(h_out != nullptr && rows > 0 && k > 0, "bad argument");
  *h_out = wsel_ws_bytes(rows, k);
  return DPRHOT_OK;
}

int dprhot_topk_update_wide(const float* S, int rows, int cols, int64_t ld, int64_t col_offset, int k, float* values, int64_t* indices, int first,
                            void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(S && values && indices && workspace, "NULL pointer");
  REQUIRE(rows > 0 && cols > 0 && k > 0 && ld >= cols, "bad shape rows=%d cols=%d k=%d ld=%lld", rows, cols, k, (long long)ld);
  REQUIRE((long long)k + cols < (1ll << 31), "k + cols must fit 31 bits");
  if (workspace_bytes < wsel_ws_bytes(rows, k)) return fail(DPRHOT_E_WORKSPACE, "topk_update_wide needs %zu workspace bytes, got %zu", wsel_ws_bytes(rows, k), workspace_bytes);
  REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  char* ws = static_cast<char*>(workspace);
  unsigned* rec = reinterpret_cast<unsigned*>(ws);
  size_t off = align256((size_t)rows * WSEL_REC * 4);
  float* Av = reinterpret_cast<float*>(ws + off); off += align256((size_t)rows * k * 4);
  int64_t* Ai = reinterpret_cast<int64_t*>(ws + off); off += align256((size_t)rows * k * 8);
  float* Bv = reinterpret_cast<float*>(ws + off); off += align256((size_t)rows * k * 4);
  int64_t* Bi = reinterpret_cast<int64_t*>(ws + off);
  WselArgs a{S, rows, cols, (long long)ld, (long long)col_offset, k, values, indices, first, rec, Av, Ai};
  hipLaunchKernelGGL(wsel_select_kernel, dim3((unsigned)rows), dim3(WSEL_THREADS), 0, st, a);
  hipLaunchKernelGGL(wsel_collect_kernel, dim3((unsigned)rows), dim3(WSEL_THREADS), 0, st, a);
  const int nblk = cdiv(k, WSEL_BLOCK);
  const size_t lds = (size_t)WSEL_BLOCK * 12;
  static AttrOnce attr_done;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(wsel_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(wsel_sort_kernel, dim3((unsigned)rows, (unsigned)nblk), dim3(WSEL_THREADS), lds, st, Av, Ai, k);
  int passes = 0;
  for (long long L = WSEL_BLOCK; L < k; L <<= 1) ++passes;
  const dim3 mgrid((unsigned)cdiv(k, WSEL_THREADS), (unsigned)rows);
  const float* sv = Av;
  const int64_t* si = Ai;
  if (passes == 0) {  // one block: the sorted block IS the new state (a merge pass against an empty partner run copies it)
    hipLaunchKernelGGL(wsel_merge_kernel, mgrid, dim3(WSEL_THREADS), 0, st, sv, si, values, indices, k, WSEL_BLOCK, rec);
  }
  long long L = WSEL_BLOCK;
  for (int j = 0; j < passes; ++j, L <<= 1) {
    const bool last = j == passes - 1;
    float* dv = last ? values : ((j & 1) == 0 ? Bv : Av);
    int64_t* di = last ? indices : ((j & 1) == 0 ? Bi : Ai);
    hipLaunchKernelGGL(wsel_merge_kernel, mgrid, dim3(WSEL_THREADS), 0, st, sv, si, dv, di, k, (int)L, last ? rec : nullptr);
    sv = dv;
    si = di;
  }
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_search_workspace_bytes(int nq, int chunk, size_t* h_out) {
  REQUIRE(h_out != nullptr, "NULL out pointer");
  REQUIRE(nq > 0 && chunk > 0 && chunk % 8 == 0, "bad shape nq=%d chunk=%d", nq, chunk);
  *h_out = search_ws_bytes(nq, chunk);
  return DPRHOT_OK;
}

int dprhot_search(const dprhot_bf16* Q, int nq, const dprhot_bf16* C, int64_t n_ctx, int d, int64_t id_offset, int k, int chunk,
                  float* values, int64_t* indices, int first, void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(Q && C && values && indices, "NULL pointer");
  REQUIRE(nq > 0 && n_ctx > 0 && d > 0 && d % 8 == 0, "bad shape nq=%d n_ctx=%lld d=%d", nq, (long long)n_ctx, d);
  REQUIRE(n_ctx % 8 == 0 && chunk > 0 && chunk % 8 == 0, "n_ctx=%lld and chunk=%d must be multiples of 8", (long long)n_ctx, chunk);
  REQUIRE(k > 0 && k <= TK_KMAX, "k=%d out of range (1..%d)", k, TK_KMAX);
  REQUIRE(id_offset >= 0, "negative id_offset");
  REQUIRE(aligned16(Q) && aligned16(C), "pointers must be 16-byte aligned");
  const size_t need = search_ws_bytes(nq, chunk);
  if (workspace == nullptr || workspace_bytes < need)
    return fail(DPRHOT_E_WORKSPACE, "search needs %zu workspace bytes (dprhot_search_workspace_bytes), got %zu", need, workspace_bytes);
  REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  float* S = static_cast<float*>(workspace);
  int* cand_j = reinterpret_cast<int*>(static_cast<char*>(workspace) + (size_t)nq * chunk * 4);
  int* cnt = reinterpret_cast<int*>(static_cast<char*>(workspace) + (size_t)nq * chunk * 8);
  const bool unfused = opt(OPT_SEARCH_UNFUSED) != 0;
  HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)nq * sizeof(int), st));
  // An empty state is started from a SHORT head (its scores are materialised and selected from: cost proportional to the head), the
  // filtered chunks behind it may be as long as the workspace allows: every chunk costs one merge launch, and a merge rewrites the
  // whole state (k = 1000: 57 us each; 31 chunks of 65536 spent 1.7 ms of a 5.6 ms search merging)
  // Behind the head every chunk is as long as everything scored before it (doubling, up to `chunk`): a chunk that covers the
  // fraction f of the corpus behind a prefix s leaves ~k f / s candidates per row against the threshold of its start -- f = s keeps
  // that at ~k (the merge's fast path holds 2048), f = 4 s left 4 k and a 790 us merge.
  const int head = chunk < 65536 ? chunk : 65536;
  for (int64_t j0 = 0, step = 0; j0 < n_ctx; j0 += step) {
    step = (first && j0 == 0) ? head : (first ? (j0 < chunk ? (j0 / 8 * 8 > head ? j0 / 8 * 8 : head) : chunk) : chunk);
    if (step > chunk) step = chunk;
    const int cols = (int)((n_ctx - j0 < step) ? (n_ctx - j0) : step);
    const dprhot_bf16* Cj = C + (size_t)j0 * d;
    if ((first && j0 == 0) || unfused) {
      // nothing to filter against yet: materialise this chunk's scores and select from them
      if (int rc = dprhot_sim_fwd(Q, nq, Cj, cols, d, nullptr, 1.0f, S, stream)) return rc;
      if (int rc = dprhot_topk_update(S, nq, cols, cols, id_offset + j0, k, values, indices, (first && j0 == 0) ? 1 : 0, stream))
        return rc;
      continue;
    }
    // scores that cannot enter the top-k never leave the GEMM tile
    if (g8_ok(nq, cols, d)) {  // the phase-interleaved kernel (gemm8p.h); thresholds arrive with the tile's input words
      // (Round 4 merged warm chunks in groups of up to four -- option search_group: up to four chunks filtered against the same stale
      //  thresholds appended to ONE candidate list of `chunk` entries per row, with no bound on the append: a corpus whose later
      //  chunks beat the current k-th value could write past its row.  It had measured no gain -- 4.76 ms either way at 1024 x 2 M,
      //  k = 1000 -- and is gone; one merge per chunk, at most `cols` candidates per row and chunk, is what the workspace is sized for.)
      {
        const Epi8Filter e8{values, indices, k, nq, cols, (long long)(id_offset + j0), cnt, S, cand_j, Q, 0};
        GemmArgs a8{Q, Cj, nq, cols, d, d, d, d};
        if (int rc = launch_g8(a8, e8, st)) return rc;
      }
      TopkArgs p8{S, nq, cols, (long long)cols, (long long)(id_offset + j0), k, values, indices, 0, cand_j, cnt};
      launch_topk(p8, nq, st);
      HIP_TRY(hipGetLastError());
      continue;
    }
    const int tile = (force_tile() < 0 && big_ok(nq, cols, d)) ? kBigTile : pick_tile(nq, cols, d, 1, 2 * kNumCU);
    GemmArgs a{Q, Cj, nq, cols, d, d, d, cdiv(d, kTiles[tile].bk) * kTiles[tile].bk};
    EpiFilter epi{values, indices, k, nq, cols, (long long)(id_offset + j0), cnt, S, cand_j};
    if (int rc = launch_gemm<true, true>(tile, a, epi, 1, st)) return rc;
    TopkArgs p{S, nq, cols, (long long)cols, (long long)(id_offset + j0), k, values, indices, 0, cand_j, cnt};
    launch_topk(p, nq, st);
    HIP_TRY(hipGetLastError());
  }
  return DPRHOT_OK;
}

// launch 1 of the fused forward: logits + per-tile softmax statistics + gold logit (and clears the loss
// accumulator words the next launch adds into)
int dprhot_sim_stats(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                     const uint8_t* colmask, float inv_T, float* S_out, void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(Q && C && y, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(Q) && aligned16(C) && (S_out == nullptr || aligned16(S_out)), "pointers must be 16-byte aligned");
  const WsLayout wl = ws_layout(B, Nc, d);
  if (workspace == nullptr || workspace_bytes < wl.total)
    return fail(DPRHOT_E_WORKSPACE, "sim_stats needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
  REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  char* ws = static_cast<char*>(workspace);
  if (S_out == nullptr && nl_ok(B, Nc, d)) {
    // no-logits plan: strip statistics + gold logit only; dprhot_softmax_finish turns them into logsumexp / loss and
    // dprhot_dscores recomputes the logits into G
    Epi8Stats epi;
    static_cast<Epi8Base&>(epi) = g8_base(Q, B, Nc, y, y_offset, colmask, inv_T, reinterpret_cast<float*>(ws + wl.gold));
    epi.part_m = reinterpret_cast<float*>(ws + wl.part_m);
    epi.part_s = reinterpret_cast<float*>(ws + wl.part_s);
    epi.npart = cdiv(Nc, G2_B) * 4;
    GemmArgs a8{Q, C, B, Nc, d, d, d, d};
    return launch_g8(a8, epi, (hipStream_t)stream);
  }
  const FwdPlan fp = fwd_plan(B, Nc, d);
  GemmArgs a{Q, C, B, Nc, d, d, d, fp.kchunk};
  if (fp.short_rows) {  // partial logits per K split; the softmax launch sums them (and fills S_out if asked)
    EpiSim epi{reinterpret_cast<float*>(ws + wl.logits), colmask, B, Nc, inv_T, nullptr, nullptr, nullptr, 0, nullptr,
               reinterpret_cast<unsigned long long*>(ws + wl.header), 2, (size_t)B * Nc};
    epi = with_packed_mask(epi);
    return launch_gemm<true, true>(fp.tile, a, epi, fp.splits, (hipStream_t)stream);
  }
  float* S = S_out ? S_out : reinterpret_cast<float*>(ws + wl.logits);
  EpiSim epi{S, colmask, B, Nc, inv_T, reinterpret_cast<float*>(ws + wl.part_m), reinterpret_cast<float*>(ws + wl.part_s),
             y, y_offset, reinterpret_cast<float*>(ws + wl.gold), reinterpret_cast<unsigned long long*>(ws + wl.header), 2};
  epi = with_packed_mask(epi);
  return launch_gemm<true, true>(fp.tile, a, epi, 1, (hipStream_t)stream);
}

// launch 1 with fp32 operands: q [B,d] fp32 always; c [Nc,d] fp32 when non-NULL (single rank: no gather), else
// Cb is the (gathered) bf16 input.  Qb (and Cb when c is given) receive the bf16 copies the backward reads.
int dprhot_sim_stats_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d, const int64_t* y,
                         int64_t y_offset, const uint8_t* colmask, float inv_T, float* S_out, void* workspace,
                         size_t workspace_bytes, void* stream) {
  REQUIRE(q && Qb && Cb && y, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(q) && aligned16(Qb) && aligned16(Cb) && (c == nullptr || aligned16(c)) && (S_out == nullptr || aligned16(S_out)),
          "pointers must be 16-byte aligned");
  const WsLayout wl = ws_layout(B, Nc, d);
  if (workspace == nullptr || workspace_bytes < wl.total)
    return fail(DPRHOT_E_WORKSPACE, "sim_stats needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
  REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  char* ws = static_cast<char*>(workspace);
  if (B > 128) {
    // Beyond the latency-bound sizes reading fp32 operands in the GEMM costs more than it saves (twice the staging
    // registers and bytes per tile, measured 250 vs 580 TFLOP/s at 8192^2): cast once, then the bf16 kernels.
    if (c != nullptr) {
      if (int rc = dprhot_prep(q, (size_t)B * d, Qb, c, (size_t)Nc * d, Cb, stream)) return rc;
    } else {
      if (int rc = dprhot_cast_bf16(q, Qb, (size_t)B * d, stream)) return rc;
    }
    return dprhot_sim_stats(Qb, B, Cb, Nc, d, y, y_offset, colmask, inv_T, S_out, workspace, workspace_bytes, stream);
  }
  const FwdPlan fp = fwd_plan(B, Nc, d);
  GemmArgs a{reinterpret_cast<const uint16_t*>(q), c ? reinterpret_cast<const uint16_t*>(c) : Cb, B, Nc, d, d, d, fp.kchunk};
  a.Acopy = Qb;
  a.Bcopy = c ? Cb : nullptr;
  if (fp.short_rows) {
    if (c != nullptr && wide_sim_ok(B, Nc, d, fp) && g_packed.base == nullptr) {
      // vocabulary-wide fp32 operands: LDS-DMA ring of fp32 tiles, bf16 rounding on the fragment read (wide.h)
      WideSimArgs wa{q, c, Qb, Cb, B, Nc, d, colmask, inv_T, reinterpret_cast<float*>(ws + wl.logits), (size_t)B * Nc, fp.kchunk,
                     reinterpret_cast<unsigned long long*>(ws + wl.header), 2, opt(OPT_WIDE_NOCOPY), cdiv(Nc, WD_B), cdiv(B, WD_B)};
      static AttrOnce attr_done;
      if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_sim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wd_lds_bytes));
        attr_done = true;
      }
      hipLaunchKernelGGL(wide_sim_kernel, dim3((unsigned)(wa.nbx * wa.nby * fp.splits)), dim3(WD_THREADS), wd_lds_bytes, (hipStream_t)stream, wa);
      HIP_TRY(hipGetLastError());
      return DPRHOT_OK;
    }
    EpiSim epi{reinterpret_cast<float*>(ws + wl.logits), colmask, B, Nc, inv_T, nullptr, nullptr, nullptr, 0, nullptr,
               reinterpret_cast<unsigned long long*>(ws + wl.header), 2, (size_t)B * Nc};
    epi = with_packed_mask(epi);
    return c ? launch_sim_f32<true, true>(fp.tile, a, epi, fp.splits, (hipStream_t)stream)
             : launch_sim_f32<true, false>(fp.tile, a, epi, fp.splits, (hipStream_t)stream);
  }
  float* S = S_out ? S_out : reinterpret_cast<float*>(ws + wl.logits);
  EpiSim epi{S, colmask, B, Nc, inv_T, reinterpret_cast<float*>(ws + wl.part_m), reinterpret_cast<float*>(ws + wl.part_s),
             y, y_offset, reinterpret_cast<float*>(ws + wl.gold), reinterpret_cast<unsigned long long*>(ws + wl.header), 2};
  epi = with_packed_mask(epi);
  return c ? launch_sim_f32<true, true>(fp.tile, a, epi, 1, (hipStream_t)stream)
           : launch_sim_f32<true, false>(fp.tile, a, epi, 1, (hipStream_t)stream);
}

// launch 2 of the fused forward: logsumexp from the statistics, G in one pass over S, loss numerator
int dprhot_softmax_finish(const float* S_in, int B, int Nc, int d, const int64_t* y, int64_t y_offset, float grad_scale,
                          float* row_loss, float* row_lse, float* loss_sum, dprhot_bf16* G, void* workspace,
                          size_t workspace_bytes, void* stream) {
  REQUIRE(y && loss_sum, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE((G == nullptr || aligned16(G)) && (S_in == nullptr || aligned16(S_in)), "pointers must be 16-byte aligned");
  const WsLayout wl = ws_layout(B, Nc, d);
  if (workspace == nullptr || workspace_bytes < wl.total)
    return fail(DPRHOT_E_WORKSPACE, "softmax_finish needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
  char* ws = static_cast<char*>(workspace);
  if (S_in == nullptr && nl_ok(B, Nc, d)) {
    // no-logits plan (same test as dprhot_sim_stats): logsumexp and loss from the strip statistics; there are no logits to turn
    // into G here -- dprhot_dscores recomputes them
    if (G != nullptr) return fail(DPRHOT_E_UNSUPPORTED, "softmax_finish: no logits at B=%d Nc=%d d=%d (no-logits forward): G comes from dprhot_dscores", B, Nc, d);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(g8_lse_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, reinterpret_cast<const float*>(ws + wl.part_m),
                       reinterpret_cast<const float*>(ws + wl.part_s), cdiv(Nc, G2_B) * 4, reinterpret_cast<const float*>(ws + wl.gold), B,
                       reinterpret_cast<float*>(ws + wl.lse), row_lse, row_loss, reinterpret_cast<float*>(ws + wl.rloss));
    HIP_TRY(hipGetLastError());
    return launch_loss_sum(reinterpret_cast<const float*>(ws + wl.rloss), B, g_loss_scale, loss_sum, st);
  }
  const FwdPlan fp = fwd_plan(B, Nc, d);  // same plan as dprhot_sim_stats -> same intermediate layout
  if (fp.short_rows) {
    // S_in, when given, is where the caller wants the summed logits (the slabs always live in the workspace)
    GShortArgs g{reinterpret_cast<const float*>(ws + wl.logits), fp.splits, (size_t)B * Nc, B, Nc, y, y_offset, grad_scale,
                 const_cast<float*>(S_in), row_loss, row_lse, G, reinterpret_cast<unsigned long long*>(ws + wl.header), loss_sum, fp.tpr, g_loss_scale};
    if (fp.blocks > 65535) return fail(DPRHOT_E_UNSUPPORTED, "softmax_finish: B=%d rows need %d workgroups (max 65535)", B, fp.blocks);
    if (fp.cpt == 1) hipLaunchKernelGGL(gfinal_short_kernel<1>, dim3(fp.blocks), dim3(fp.threads), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(gfinal_short_kernel<2>, dim3(fp.blocks), dim3(fp.threads), 0, (hipStream_t)stream, g);
    HIP_TRY(hipGetLastError());
    return DPRHOT_OK;
  }
  const float* S = S_in ? S_in : reinterpret_cast<const float*>(ws + wl.logits);
  const int nt = fp.nt;
  if (nt > kGfMaxPairs) return fail(DPRHOT_E_UNSUPPORTED, "Nc=%d too long for the statistics buffer (%d column tiles)", Nc, nt);
  GFinalArgs g{S, B, Nc, y, y_offset, grad_scale, reinterpret_cast<const float*>(ws + wl.part_m),
               reinterpret_cast<const float*>(ws + wl.part_s), nt, reinterpret_cast<const float*>(ws + wl.gold), row_loss, row_lse,
               G, reinterpret_cast<unsigned long long*>(ws + wl.header), loss_sum, g_loss_scale};
  const bool thin = (long)B * (Nc / 8) <= 1L << 18;  // latency-bound sizes: one chunk per thread, many small workgroups
  int rpb, xblocks;
  gfinal_geometry(Nc, thin ? 1 : 8, &rpb, &xblocks);
  if (cdiv(B, rpb) > 65535)  // grid.y limit, and the arrival ticket of the loss accumulation is 16 bits wide
    return fail(DPRHOT_E_UNSUPPORTED, "softmax_finish: B=%d rows need %d row blocks (max 65535)", B, cdiv(B, rpb));
  dim3 grid(xblocks, cdiv(B, rpb));
  if (thin) hipLaunchKernelGGL(gfinal_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(gfinal_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, g);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

// dScores of the no-logits forward: G = (softmax(S) - onehot) * grad_scale with S recomputed tile by tile (the same GEMM as
// dprhot_sim_stats, bit-identical accumulators).  row_lse: the rows' logsumexp, or NULL to use the values dprhot_softmax_finish
// left in the workspace.
int dprhot_dscores(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                   const uint8_t* colmask, float inv_T, float grad_scale, const float* row_lse, dprhot_bf16* G, void* workspace,
                   size_t workspace_bytes, void* stream) {
  REQUIRE(Q && C && y && G, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(Q) && aligned16(C) && aligned16(G), "pointers must be 16-byte aligned");
  if (!nl_ok(B, Nc, d)) return fail(DPRHOT_E_UNSUPPORTED, "dscores: B=%d Nc=%d d=%d is not a no-logits shape (use dprhot_softmax_finish)", B, Nc, d);
  const WsLayout wl = ws_layout(B, Nc, d);
  if (row_lse == nullptr) {
    if (workspace == nullptr || workspace_bytes < wl.total)
      return fail(DPRHOT_E_WORKSPACE, "dscores needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
    row_lse = reinterpret_cast<const float*>(static_cast<char*>(workspace) + wl.lse);
  }
  Epi8G epi;
  static_cast<Epi8Base&>(epi) = g8_base(Q, B, Nc, y, y_offset, colmask, inv_T, nullptr);
  epi.row_lse = row_lse;
  epi.G = G;
  epi.grad_scale = grad_scale;
  GemmArgs a8{Q, C, B, Nc, d, d, d, d};
  return launch_g8<Epi8G, 2>(a8, epi, (hipStream_t)stream);
}

// Rank of every row's gold column in the stable descending order of its scores (dpr_task.py:235-246) straight from the
// embeddings: at no-logits shapes the score matrix is never written (gold logits from a gathered mini GEMM, then a count-greater
// epilogue inside the similarity GEMM); smaller problems materialise S in the workspace and count there.
int dprhot_sim_rank(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                    const uint8_t* colmask, float inv_T, int64_t* rank, void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(Q && C && y && rank, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(Q) && aligned16(C), "pointers must be 16-byte aligned");
  const WsLayout wl = ws_layout(B, Nc, d);
  if (workspace == nullptr || workspace_bytes < wl.total)
    return fail(DPRHOT_E_WORKSPACE, "sim_rank needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
  REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  char* ws = static_cast<char*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  if (!nl_ok(B, Nc, d)) {
    float* S = reinterpret_cast<float*>(ws + wl.logits);
    if (int rc = dprhot_sim_fwd(Q, B, C, Nc, d, colmask, inv_T, S, stream)) return rc;
    return dprhot_rank_of_gold(S, B, Nc, y, y_offset, rank, stream);
  }
  float* gold = reinterpret_cast<float*>(ws + wl.lse);
  int* count = reinterpret_cast<int*>(ws + wl.rloss);
  Epi8Count epi;
  static_cast<Epi8Base&>(epi) = g8_base(Q, B, Nc, y, y_offset, colmask, inv_T, nullptr);
  epi.gold_val = gold;
  epi.count = count;
  hipLaunchKernelGGL(g8_gold_kernel, dim3(cdiv(B, 32)), dim3(64), 0, st, Q, C, B, Nc, d, y, y_offset, epi.sim, inv_T, gold);
  HIP_TRY(hipMemsetAsync(count, 0, (size_t)B * sizeof(int), st));
  GemmArgs a8{Q, C, B, Nc, d, d, d, d};
  if (int rc = launch_g8(a8, epi, st)) return rc;
  hipLaunchKernelGGL(g8_rank_finish_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, count, B, rank);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

// compute_rank_metrics AND the cross-entropy of the same scores (dpr_task.py:224-227, :296-299) in one call: at no-logits shapes ONE
// pass of the similarity GEMM whose epilogue counts and keeps the strip statistics (Epi8CountStats); elsewhere the scores go to the
// workspace once and both readers stream them.
int dprhot_sim_rank_loss(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                         const uint8_t* colmask, float inv_T, int64_t* rank, float* row_loss, float* row_lse, float* loss_sum,
                         void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(Q && C && y && rank && loss_sum, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  REQUIRE(aligned16(Q) && aligned16(C), "pointers must be 16-byte aligned");
  const WsLayout wl = ws_layout(B, Nc, d);
  if (workspace == nullptr || workspace_bytes < wl.total)
    return fail(DPRHOT_E_WORKSPACE, "sim_rank_loss needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
  REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
  char* ws = static_cast<char*>(workspace);
  hipStream_t st = (hipStream_t)stream;
  if (!nl_ok(B, Nc, d)) {
    if (int rc = dprhot_sim_rank(Q, B, C, Nc, d, y, y_offset, colmask, inv_T, rank, workspace, workspace_bytes, stream)) return rc;
    return dprhot_inbatch_fwd(Q, B, C, Nc, d, y, y_offset, colmask, inv_T, 1.0f, nullptr, row_loss, row_lse, loss_sum, nullptr, workspace,
                              workspace_bytes, stream);
  }
  float* gold = reinterpret_cast<float*>(ws + wl.gold);
  int* count = reinterpret_cast<int*>(ws + wl.rloss);  // (the row losses take this place once the ranks have been read out of it)
  Epi8CountStats epi;
  static_cast<Epi8Base&>(epi) = g8_base(Q, B, Nc, y, y_offset, colmask, inv_T, nullptr);
  epi.gold_val = gold;
  epi.count = count;
  epi.part_m = reinterpret_cast<float*>(ws + wl.part_m);
  epi.part_s = reinterpret_cast<float*>(ws + wl.part_s);
  epi.npart = cdiv(Nc, G2_B) * 4;
  hipLaunchKernelGGL(g8_gold_kernel, dim3(cdiv(B, 32)), dim3(64), 0, st, Q, C, B, Nc, d, y, y_offset, epi.sim, inv_T, gold);
  HIP_TRY(hipMemsetAsync(count, 0, (size_t)B * sizeof(int), st));
  GemmArgs a8{Q, C, B, Nc, d, d, d, d};
  if (int rc = launch_g8(a8, epi, st)) return rc;
  hipLaunchKernelGGL(g8_rank_finish_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, count, B, rank);
  hipLaunchKernelGGL(g8_lse_kernel, dim3(cdiv(B, 4)), dim3(256), 0, st, epi.part_m, epi.part_s, epi.npart, gold, B,
                     reinterpret_cast<float*>(ws + wl.lse), row_lse, row_loss, reinterpret_cast<float*>(ws + wl.rloss));
  HIP_TRY(hipGetLastError());
  return launch_loss_sum(reinterpret_cast<const float*>(ws + wl.rloss), B, g_loss_scale, loss_sum, st);
}

int dprhot_inbatch_fwd(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                       const uint8_t* colmask, float inv_T, float grad_scale, float* S_out, float* row_loss, float* row_lse,
                       float* loss_sum, dprhot_bf16* G, void* workspace, size_t workspace_bytes, void* stream) {
  if (S_out == nullptr && G != nullptr && nl128_ok(B, Nc, d)) {
    // the same one-pass forward on the 128 x 128 LDS-DMA tile (round 6): the shapes whose 256-wide tiles cannot fill the chip
    REQUIRE(Q && C && y && loss_sum, "NULL pointer");
    if (int rc = check_shape(B, Nc, d)) return rc;
    REQUIRE(aligned16(Q) && aligned16(C) && aligned16(G), "pointers must be 16-byte aligned");
    const WsLayout wl = ws_layout(B, Nc, d);
    if (workspace == nullptr || workspace_bytes < wl.total)
      return fail(DPRHOT_E_WORKSPACE, "inbatch_fwd needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
    REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    EpiSimP epi;
    static_cast<EpiSim&>(epi) = with_packed_mask(EpiSim{nullptr, colmask, B, Nc, inv_T, reinterpret_cast<float*>(ws + wl.part_m),
                                                        reinterpret_cast<float*>(ws + wl.part_s), y, y_offset,
                                                        reinterpret_cast<float*>(ws + wl.gold), nullptr, 0});
    epi.P = G;
    epi.npart = cdiv(Nc, 64);
    GemmArgs a{Q, C, B, Nc, d, d, d, d};
    auto kern = gemm128d_kernel<true, true, EpiSimP, false>;
    static AttrOnce attr_done;
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g1_lds_bytes));
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(cdiv(Nc, 128), cdiv(B, 128), 1), dim3(256), g1_lds_bytes, st, a, epi);
    hipLaunchKernelGGL(g8_lse_p2g_kernel, dim3((unsigned)B), dim3(256), 0, st, epi.part_m, epi.part_s, epi.npart,
                       reinterpret_cast<const float*>(ws + wl.gold), B, Nc, y, y_offset, grad_scale, reinterpret_cast<float*>(ws + wl.lse), row_lse, row_loss,
                       reinterpret_cast<float*>(ws + wl.rloss), G);
    HIP_TRY(hipGetLastError());
    return launch_loss_sum(reinterpret_cast<const float*>(ws + wl.rloss), B, g_loss_scale, loss_sum, st);
  }
  if (S_out == nullptr && G != nullptr && nl_ok(B, Nc, d) && opt(OPT_NL_P16) != 0) {
    // no-logits forward in ONE pass of the GEMM (round 6): strip statistics + fp16 softmax numerators into the G buffer (Epi8StatsP),
    // then one row kernel: logsumexp / loss, and the numerators rescaled into the bf16 dScores in place
    REQUIRE(Q && C && y && loss_sum, "NULL pointer");
    if (int rc = check_shape(B, Nc, d)) return rc;
    REQUIRE(aligned16(Q) && aligned16(C) && aligned16(G), "pointers must be 16-byte aligned");
    const WsLayout wl = ws_layout(B, Nc, d);
    if (workspace == nullptr || workspace_bytes < wl.total)
      return fail(DPRHOT_E_WORKSPACE, "inbatch_fwd needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
    REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
    char* ws = static_cast<char*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    Epi8StatsP epi;
    static_cast<Epi8Base&>(epi) = g8_base(Q, B, Nc, y, y_offset, colmask, inv_T, reinterpret_cast<float*>(ws + wl.gold));
    epi.part_m = reinterpret_cast<float*>(ws + wl.part_m);
    epi.part_s = reinterpret_cast<float*>(ws + wl.part_s);
    epi.npart = cdiv(Nc, G2_B) * 4;
    epi.P = G;
    GemmArgs a8{Q, C, B, Nc, d, d, d, d};
    if (opt(OPT_P16_STAGED) == 2 || (opt(OPT_P16_STAGED) == 1 && ((size_t)Nc * 2) % ((size_t)128 << 10) != 0)) {
      Epi8StatsPS es;
      static_cast<Epi8Stats&>(es) = static_cast<const Epi8Stats&>(epi);
      es.P = G;
      if (int rc = launch_g8<Epi8StatsPS, 2>(a8, es, st)) return rc;
    } else if (int rc = launch_g8<Epi8StatsP, 2>(a8, epi, st)) return rc;  // (two-phase schedule, as every storing epilogue; the four-phase instantiation spilled two registers)
    hipLaunchKernelGGL(g8_lse_p2g_kernel, dim3((unsigned)B), dim3(256), 0, st, epi.part_m, epi.part_s, epi.npart,
                       reinterpret_cast<const float*>(ws + wl.gold), B, Nc, y, y_offset, grad_scale, reinterpret_cast<float*>(ws + wl.lse), row_lse, row_loss,
                       reinterpret_cast<float*>(ws + wl.rloss), G);
    HIP_TRY(hipGetLastError());
    return launch_loss_sum(reinterpret_cast<const float*>(ws + wl.rloss), B, g_loss_scale, loss_sum, st);
  }
  if (S_out == nullptr && nl_ok(B, Nc, d)) {
    // no-logits forward: statistics pass -> logsumexp / loss -> dScores pass (logits recomputed); S is never in memory
    if (int rc = dprhot_sim_stats(Q, B, C, Nc, d, y, y_offset, colmask, inv_T, nullptr, workspace, workspace_bytes, stream)) return rc;
    if (int rc = dprhot_softmax_finish(nullptr, B, Nc, d, y, y_offset, grad_scale, row_loss, row_lse, loss_sum, nullptr, workspace,
                                       workspace_bytes, stream))
      return rc;
    if (G == nullptr) return DPRHOT_OK;
    return dprhot_dscores(Q, B, C, Nc, d, y, y_offset, colmask, inv_T, grad_scale, nullptr, G, workspace, workspace_bytes, stream);
  }
  if (int rc = dprhot_sim_stats(Q, B, C, Nc, d, y, y_offset, colmask, inv_T, S_out, workspace, workspace_bytes, stream)) return rc;
  return dprhot_softmax_finish(S_out, B, Nc, d, y, y_offset, grad_scale, row_loss, row_lse, loss_sum, G, workspace, workspace_bytes,
                               stream);
}

int dprhot_inbatch_fwd_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d, const int64_t* y,
                           int64_t y_offset, const uint8_t* colmask, float inv_T, float grad_scale, float* S_out, float* row_loss,
                           float* row_lse, float* loss_sum, dprhot_bf16* G, void* workspace, size_t workspace_bytes, void* stream) {
  if (S_out == nullptr && B > 128 && (nl_ok(B, Nc, d) || (G != nullptr && nl128_ok(B, Nc, d)))) {  // cast once, then the no-logits forward on the bf16 copies
    REQUIRE(q && Qb && Cb && y, "NULL pointer");
    if (c != nullptr) {
      if (int rc = dprhot_prep(q, (size_t)B * d, Qb, c, (size_t)Nc * d, Cb, stream)) return rc;
    } else {
      if (int rc = dprhot_cast_bf16(q, Qb, (size_t)B * d, stream)) return rc;
    }
    return dprhot_inbatch_fwd(Qb, B, Cb, Nc, d, y, y_offset, colmask, inv_T, grad_scale, nullptr, row_loss, row_lse, loss_sum, G, workspace,
                              workspace_bytes, stream);
  }
  if (int rc = dprhot_sim_stats_f32(q, c, Qb, Cb, B, Nc, d, y, y_offset, colmask, inv_T, S_out, workspace, workspace_bytes, stream))
    return rc;
  return dprhot_softmax_finish(S_out, B, Nc, d, y, y_offset, grad_scale, row_loss, row_lse, loss_sum, G, workspace, workspace_bytes,
                               stream);
}

int dprhot_inbatch_bwd(const dprhot_bf16* G, const dprhot_bf16* Q, const dprhot_bf16* C, int B, int Nc, int d, float h_scale,
                       const float* d_scale, float* dQ, float* dC_part, void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(G && Q && C, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // (the long context axis under fewer than 512 rows: both GEMMs on the 128 x 128 engine -- see `long_axis` below; 256 x 65536 x 768:
  //  146 us against 174 with the dQ units on the 256 x 256 kernel and 252 for the pair)
  // (round 6, with the LDS-DMA 128 x 128 tile: also 512 <= B < 1024 from Nc >= 32 B on -- 512 x 16384 56 against 65 us with the dQ units on
  //  the 256 x 256 kernel and 73 for the pair, 512 x 32768 90 / 100 / 116: profiles/r06_bwd_plan_ab.txt)
  const bool long_axis_small = ((B < 512 && Nc >= 56 * 1024) || (B >= 512 && B < 1024 && (long)Nc >= 32L * B)) && !sk_plan(B, Nc, d).ok &&
                               !wide_bwd_ok(B, Nc, d) && big_bwd_ok(B, Nc, d) && !pair128_use(B, Nc, d);
  if (dQ == nullptr || dC_part == nullptr || unfused_bwd() || force_tile() >= 0 || long_axis_small) {
    if (dC_part != nullptr)
      if (int rc = dprhot_dc(G, Q, B, Nc, d, h_scale, d_scale, dC_part, stream)) return rc;
    if (dQ != nullptr)
      if (int rc = dprhot_dq(G, C, B, Nc, d, h_scale, d_scale, dQ, workspace, workspace_bytes, stream)) return rc;
    return DPRHOT_OK;
  }
  REQUIRE(aligned16(G) && aligned16(Q) && aligned16(C) && aligned16(dQ) && aligned16(dC_part), "pointers must be 16-byte aligned");
  if (g_packed.stamp_src == nullptr) {
    // Shapes of the few-rows plan: the same backward units as the one-call step (dC tiles and split-K dQ tiles side by side, then the
    // slab reduction) for callers that run sim / loss / backward as separate calls (a subclass's own loss on sim_score's logits):
    // 128 x 8192 x 768 23 -> ~15 us against the generic pair kernel.
    const SkPlan sk = sk_plan(B, Nc, d);
    if (sk.ok) {
      const WsLayout wl = ws_layout(B, Nc, d);
      char* ws = static_cast<char*>(workspace);
      if (sk.nslices > 1 && (ws == nullptr || workspace_bytes < wl.total))
        return fail(DPRHOT_E_WORKSPACE, "inbatch_bwd needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
      float* part = sk.nslices > 1 ? reinterpret_cast<float*>(ws + wl.dq_part) : nullptr;
      const int ndq = sk.nslices * (d / SK_QN), ndq_pad = (ndq + 7) & ~7, ndc = sk.nt * (d / SK_DN);
      SkBwdArgs b{G, Q, C, B, Nc, d, h_scale, d_scale, dC_part, nullptr, nullptr, 1.0f, 0, 0, 0, sk.ksteps, sk.nslices, part, dQ, ndq_pad,
                  opt(OPT_NT_STORES) ? 1 : 0};
      const size_t lds = sk_bwd_lds();
      static AttrOnce attr_done;
      if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sk_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
      }
      hipLaunchKernelGGL(sk_bwd_kernel, dim3((unsigned)(ndq_pad + ndc)), dim3(SK_THREADS), lds, st, b);
      HIP_TRY(hipGetLastError());
      if (sk.nslices > 1) {
        const size_t n4 = (size_t)B * d / 4;
        hipLaunchKernelGGL(sk_dq_reduce_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, part, sk.nslices, n4, h_scale, d_scale, dQ);
        HIP_TRY(hipGetLastError());
      }
      return DPRHOT_OK;
    }
  }
  if (wide_bwd_ok(B, Nc, d) && g_packed.stamp_src == nullptr) {
    // Vocabulary-wide vectors under few rows (CITADEL router: 128 x 1024 x 30528): every unit of either GEMM is a short K loop in front
    // of a large store -- the unit shapes of skinny.h (whole operand footprint in flight by LDS-DMA, stores through LDS in whole
    // lines), dC tiles and dQ tiles side by side in one launch; the contexts fit one dQ slice, so no partial sums and no reduction.
    // The 141 MB of fp32 gradients leave with non-temporal stores: nobody re-reads them inside the step, and written normally they push
    // the step's operands out of the 256 MB Infinity Cache (measured, scratch/router_ab.py: step 116 -> 99.5 us; the same policy on the
    // 8192^2 launches -- stored logits, 256 x 256 backward -- LOST 5-10 % and is not used there: profiles/r03_nt_stores_ab.txt).
    const int ksteps = cdiv(Nc, 64);
    const int ndq = d / SK_QN, ndq_pad = (ndq + 7) & ~7, ndc = cdiv(Nc, SK_COLS) * cdiv(d, SK_DN);
    SkBwdArgs b{G, Q, C, B, Nc, d, h_scale, d_scale, dC_part, nullptr, nullptr, 1.0f, 0, 0, 0, ksteps, 1, nullptr, dQ, ndq_pad, opt(OPT_NT_STORES) ? 1 : 0};
    const size_t lds = sk_bwd_lds();
    static AttrOnce attr_done;
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sk_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done = true;
    }
    hipLaunchKernelGGL(sk_bwd_kernel, dim3((unsigned)(ndq_pad + ndc)), dim3(SK_THREADS), lds, st, b);
    HIP_TRY(hipGetLastError());
    return DPRHOT_OK;
  }
  const WsLayout wl = ws_layout(B, Nc, d);
  if (pair128_use(B, Nc, d)) {
    // [dC tiles | dQ tiles x K slices] on the 128 x 128 LDS-DMA tile (round 6): the shapes under the 256 x 256 gate ran this pair on the
    // register-staged tiles (2048 x 4096 x 768: 69.5 us for 26 GFLOP)
    const Pair128Plan pp = pair128_plan(B, Nc, d);
    char* ws = static_cast<char*>(workspace);
    if (pp.splits > 1 && (ws == nullptr || workspace_bytes < wl.total))
      return fail(DPRHOT_E_WORKSPACE, "inbatch_bwd needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
    GemmArgs a1{G, Q, Nc, d, B, Nc, d, B};
    const EpiScaleF32 e1 = with_loss_stamp(EpiScaleF32{dC_part, Nc, d, h_scale, d_scale});
    GemmArgs a2{G, C, B, d, Nc, Nc, d, pp.kchunk};
    const EpiScaleF32 e2 = pp.splits == 1 ? EpiScaleF32{dQ, B, d, h_scale, d_scale}
                                          : EpiScaleF32{reinterpret_cast<float*>(ws + wl.dq_part), B, d, 1.0f, nullptr};
    auto kern = gemm128d_pair_kernel<EpiScaleF32>;
    static AttrOnce attr_done;
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g1_lds_bytes));
      attr_done = true;
    }
    const int nbx1 = cdiv(d, 128), nby1 = cdiv(Nc, 128), nbx2 = cdiv(d, 128), nby2 = cdiv(B, 128);
    const long grid = (long)nbx1 * nby1 + (long)nbx2 * nby2 * pp.splits;
    if (grid > 0x7fffffffL) return fail(DPRHOT_E_UNSUPPORTED, "grid too large");
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), g1_lds_bytes, st, a1, e1, nbx1, nby1, a2, e2, nbx2, nby2, pp.splits);
    HIP_TRY(hipGetLastError());
    if (pp.splits > 1) return launch_slab_sum(reinterpret_cast<const float*>(ws + wl.dq_part), pp.splits, B, d, h_scale, d_scale, dQ, st);
    return DPRHOT_OK;
  }
  const DqPlan p = dq_plan(B, Nc, d);
  char* ws = static_cast<char*>(workspace);
  if (p.splits > 1 && (ws == nullptr || workspace_bytes < wl.total))
    return fail(DPRHOT_E_WORKSPACE, "inbatch_bwd needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
  // one launch: [dC tiles | dQ tiles x splits]
  GemmArgs a1{G, Q, Nc, d, B, Nc, d, cdiv(B, 64) * 64};
  const EpiScaleF32 e1 = with_loss_stamp(EpiScaleF32{dC_part, Nc, d, h_scale, d_scale});
  GemmArgs a2{G, C, B, d, Nc, Nc, d, p.kchunk};
  EpiScaleF32 e2 = p.splits == 1 ? EpiScaleF32{dQ, B, d, h_scale, d_scale}
                                 : EpiScaleF32{reinterpret_cast<float*>(ws + wl.dq_part), B, d, 1.0f, nullptr};
  if (p.big) {
    auto kern = gemm256_bwd_kernel<EpiScaleF32>;
    static AttrOnce attr_done;  // benign race: idempotent
    if (!attr_done) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2_lds_total));
      attr_done = true;
    }
    // Few query rows against a long context axis (512 <= B <= 2048, Nc >= 32 B: e.g. 8192 global queries x 8 contexts seen from one
    // of 8 ranks): the dC tiles of the pair launch are only K = B deep -- 8 to 32 K steps between a pipeline fill and a 256 KiB store
    // -- next to dQ units several times as long (the K slices of dQ stop at 16), and the launch loses to dC on the 128 x 128 engine in a
    // launch of its own (dprhot_dc) followed by this launch with its dQ units only.  Measured, pair / dC apart, us (d = 768,
    // scratch/bwd_hybrid_ab.py, profiles/r05_bwd_long_axis.txt): 1024 x 32768 181 / 154, 1024 x 49152 292 / 236, 1024 x 65536 482 / 299,
    // 512 x 16384 73 / 70, 512 x 32768 117 / 112, 512 x 65536 356 / 202, 2048 x 65536 611 / 555, 1024 x 32768 x 1024 217 / 197; the
    // pair stays where it wins: 1024 x 8192 43 / 71, 1024 x 16384 82 / 96, 2048 x 16384 122 / 146, 2048 x 32768 228 / 289, 512 x 8192 33 / 50.
    const bool long_axis = B >= 512 && B <= 2048 && (long)Nc >= 32L * B;
    const bool no8 = opt(OPT_NO_8PB) != 0;
    // (round 6: B and Nc multiples of 64 suffice -- a unit with an odd number of K steps skips the MFMAs of its last half pair; the packed
    //  multi-rank layout's column counts are multiples of 64, rarely of 128)
    const bool ok8 = !no8 && B % 64 == 0 && Nc % 64 == 0 && p.kchunk % 128 == 0 && (double)B * Nc < 2.0e9 && (double)Nc * d < 2.0e9;
    a1.kchunk = B;  // dC: one K range (B % 64 == 0)
    if (long_axis && ok8 && (opt(OPT_DC_ALONE_8P) == 2 || (opt(OPT_DC_ALONE_8P) == 1 && ((size_t)Nc * 2) % ((size_t)128 << 10) == 0))) {
      // (A/B, round 6) the dC tiles alone on the phase-interleaved kernel: no dQ units four times as long next to them
      auto k8 = gemm8p_bwd_kernel<Epi8Scale>;
      static AttrOnce attr8a_done;
      if (!attr8a_done) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g8_lds_total));
        attr8a_done = true;
      }
      const Epi8Scale s1{e1.out, e1.M, e1.N, e1.h_scale, e1.d_scale, e1.stamp_src, e1.stamp_period, e1.stamp_row};
      const Epi8Scale s2{e2.out, e2.M, e2.N, e2.h_scale, e2.d_scale, e2.stamp_src, e2.stamp_period, e2.stamp_row};
      const int nbx1a = cdiv(d, G2_B), nby1a = cdiv(Nc, G2_B);
      hipLaunchKernelGGL(k8, dim3((unsigned)(nbx1a * nby1a)), dim3(G2_THREADS), g8_lds_total, st, a1, s1, nbx1a, nby1a, a2, s2, cdiv(d, G2_B), 0, p.splits);
      HIP_TRY(hipGetLastError());
    } else if (long_axis) {
      if (int rc = dprhot_dc(G, Q, B, Nc, d, h_scale, d_scale, dC_part, stream)) return rc;
    }
    const int nbx1 = cdiv(d, G2_B), nby1 = long_axis ? 0 : cdiv(Nc, G2_B), nbx2 = cdiv(d, G2_B), nby2 = cdiv(B, G2_B);
    const int grid = nbx1 * nby1 + nbx2 * nby2 * p.splits;
    if (ok8) {
      // the phase-interleaved schedule (gemm8pb.h): an even number of K steps per unit, byte offsets in 32 bits
      auto k8 = gemm8p_bwd_kernel<Epi8Scale>;
      static AttrOnce attr8_done;
      if (!attr8_done) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g8_lds_total));
        attr8_done = true;
      }
      const Epi8Scale s1{e1.out, e1.M, e1.N, e1.h_scale, e1.d_scale, e1.stamp_src, e1.stamp_period, e1.stamp_row};
      const Epi8Scale s2{e2.out, e2.M, e2.N, e2.h_scale, e2.d_scale, e2.stamp_src, e2.stamp_period, e2.stamp_row};
      hipLaunchKernelGGL(k8, dim3((unsigned)grid), dim3(G2_THREADS), g8_lds_total, st, a1, s1, nbx1, nby1, a2, s2, nbx2, nby2, p.splits);
      HIP_TRY(hipGetLastError());
      return launch_dq(G, C, B, Nc, d, h_scale, d_scale, dQ, ws, wl, st, /*gemm_too=*/false);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(G2_THREADS), g2_lds_total, st, a1, e1, nbx1, nby1, a2, e2, nbx2, nby2, p.splits);
    HIP_TRY(hipGetLastError());
    return launch_dq(G, C, B, Nc, d, h_scale, d_scale, dQ, ws, wl, st, /*gemm_too=*/false);
  }
  const int t1 = dc_tile(B, Nc, d);
  const int rc = use_tr() ? launch_pair_tr<true>(t1, p.tile, a1, e1, a2, e2, p.splits, st)
                          : launch_pair_tr<false>(t1, p.tile, a1, e1, a2, e2, p.splits, st);
  if (rc) return rc;
  return launch_dq(G, C, B, Nc, d, h_scale, d_scale, dQ, ws, wl, st, /*gemm_too=*/false);
}

// A whole training step (forward AND backward) in one call.  At the latency-bound shapes (step_small.h) it is TWO
// launches: the sim GEMM (partial-logit slabs) and one kernel that does softmax-CE, dScores and both backward GEMMs;
// every other shape runs the three launches of dprhot_inbatch_fwd_f32 + dprhot_inbatch_bwd.
static bool small_step_ok(int B, int Nc, int d) {
  const FwdPlan fp = fwd_plan(B, Nc, d);
  // two row blocks (32 < B <= 64) up to 256 columns only -- measured (scratch/rb_probe.py, two launches against three): 64 x 256 x 768
  // 13.3 against 16.2 us, 64 x 128 x 1024 13.5 / 14.0, but 64 x 512 18.6 / 17.5 and 64 x 768 24.0 / 22.8: every workgroup repeats
  // BOTH blocks' softmax, and beyond 256 columns that second helping costs more than the launch it saves
  return !opt(OPT_NO_SMALL_STEP) && B <= SS_MAXB && Nc <= (B <= SS_ROWS ? SS_MAXNC : 256) && d % 16 == 0 && fp.short_rows && fp.splits <= 4 &&
         !unfused_bwd();
}

int dprhot_step_wants_g(int B, int Nc, int d, int* h_wants) {
  REQUIRE(h_wants != nullptr, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  // 0 where the few-rows plan takes the fused-dScores form -- judged on the plain column tiling: the packed layout's remapped tiles
  // are never more and their slices never longer.  A pure function of the shape, like every plan here.
  const SkPlan sk = sk_plan(B, Nc, d);
  *h_wants = sk_fused_plan(sk, cdiv(Nc, SK_COLS), cdiv(Nc, 64), B, Nc, d).ok ? 0 : 1;
  return DPRHOT_OK;
}

int dprhot_fwd_one_pass(int B, int Nc, int d, int* h_kind) {
  REQUIRE(h_kind != nullptr, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  *h_kind = nl128_ok(B, Nc, d) ? 2 : (nl_ok(B, Nc, d) && opt(OPT_NL_P16) != 0 ? 1 : 0);
  return DPRHOT_OK;
}

int dprhot_fwd_no_logits(int B, int Nc, int d, int* h_nl) {
  REQUIRE(h_nl != nullptr, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  *h_nl = nl_ok(B, Nc, d) ? 1 : 0;
  return DPRHOT_OK;
}

int dprhot_inbatch_step_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d, const int64_t* y,
                            int64_t y_offset, const uint8_t* colmask, float inv_T, float grad_scale, float h_scale, const float* d_scale,
                            float* S_out, float* row_loss, float* row_lse, float* loss_sum, dprhot_bf16* G, float* dQ, float* dC_part,
                            void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(loss_sum && dQ && dC_part, "NULL pointer (loss_sum, dQ and dC_part are required)");
  REQUIRE(aligned16(G) && aligned16(dQ) && aligned16(dC_part), "pointers must be 16-byte aligned");
  if (int rc = check_shape(B, Nc, d)) return rc;
  if (G == nullptr) {
    int wants = 1;
    if (int rc = dprhot_step_wants_g(B, Nc, d, &wants)) return rc;
    REQUIRE(wants == 0, "G == NULL at B=%d Nc=%d d=%d: this shape's plan materialises the dScores (dprhot_step_wants_g = 1)", B, Nc, d);
  }
  {
    const SkPlan sk = sk_plan(B, Nc, d);
    if (sk.ok) {
      REQUIRE(q && Qb && Cb && y, "NULL pointer");
      REQUIRE(aligned16(q) && aligned16(Qb) && aligned16(Cb) && (c == nullptr || aligned16(c)) && (S_out == nullptr || aligned16(S_out)),
              "pointers must be 16-byte aligned");
      const WsLayout wl = ws_layout(B, Nc, d);
      if (workspace == nullptr || workspace_bytes < wl.total)
        return fail(DPRHOT_E_WORKSPACE, "inbatch_step needs %zu workspace bytes, got %zu", wl.total, workspace_bytes);
      REQUIRE(aligned16(workspace), "workspace must be 16-byte aligned");
      if (c != nullptr)  // single rank: the contexts arrive as fp32 -- one cast, then they are the bf16 matrix of every launch
        if (int rc = dprhot_cast_bf16(c, Cb, (size_t)Nc * d, stream)) return rc;
      return sk_step(q, Cb, Qb, B, Nc, d, y, y_offset, colmask, inv_T, grad_scale, h_scale, d_scale, S_out, row_loss, row_lse, loss_sum, G,
                     dQ, dC_part, static_cast<char*>(workspace), wl, sk, (hipStream_t)stream);
    }
  }
  if (!small_step_ok(B, Nc, d)) {
    // (single rank only: the packed step's backward stamps the finished loss into dC_part, so it must exist before that launch)
    g_loss_defer.armed = g_packed.stamp_src == nullptr && opt(OPT_LOSS_WITH_DQ) != 0;
    g_loss_defer.pending = false;
    int rc = dprhot_inbatch_fwd_f32(q, c, Qb, Cb, B, Nc, d, y, y_offset, colmask, inv_T, grad_scale, S_out, row_loss, row_lse, loss_sum, G,
                                    workspace, workspace_bytes, stream);
    g_loss_defer.armed = false;
    if (rc == DPRHOT_OK) rc = dprhot_inbatch_bwd(G, Qb, Cb, B, Nc, d, h_scale, d_scale, dQ, dC_part, workspace, workspace_bytes, stream);
    if (g_loss_defer.pending) {  // a backward without a slab-combining launch (or one that failed): the loss launch after all
      g_loss_defer.pending = false;
      const int rc2 = launch_loss_sum(g_loss_defer.src, g_loss_defer.n, g_loss_defer.scale, g_loss_defer.out, (hipStream_t)stream);
      if (rc == DPRHOT_OK) rc = rc2;
    }
    return rc;
  }
  if (int rc = dprhot_sim_stats_f32(q, c, Qb, Cb, B, Nc, d, y, y_offset, colmask, inv_T, nullptr, workspace, workspace_bytes, stream))
    return rc;
  const WsLayout wl = ws_layout(B, Nc, d);
  const FwdPlan fp = fwd_plan(B, Nc, d);
  char* ws = static_cast<char*>(workspace);
  StepSmallArgs a{reinterpret_cast<const float*>(ws + wl.logits), fp.splits, (size_t)B * Nc, B, Nc, d, y, y_offset, grad_scale,
                  Qb, Cb, h_scale, d_scale, dQ, dC_part, S_out, row_loss, row_lse, loss_sum, G,
                  g_packed.stamp_src != nullptr ? g_packed.rows_c : 0, g_packed.n_ctx, g_loss_scale};
  // 16 columns of d per workgroup: every workgroup repeats the softmax, the narrow tile only shortens the GEMM / store tail
  constexpr int tw = 16;
  const size_t lds = step_small_lds(Nc, tw);
  const int ncp = (Nc + 31) / 32 * 32;
  const dim3 grid(d / tw), block(1024);
  hipStream_t st = (hipStream_t)stream;
#define DPRHOT_SS_LAUNCH_N(CPT, NS, NRB)                                                                                      \
  do {                                                                                                                         \
    auto kern = step_small_kernel<CPT, tw, NS, NRB>;                                                                             \
    static size_t attr_dev[64] = {}; /* per device; benign race: idempotent */                                                 \
    size_t& attr = attr_dev[lds > 48 * 1024 ? AttrOnce::cur() : 0];                                                            \
    if (lds > 48 * 1024 && attr < lds) {                                                                                       \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      attr = lds;                                                                                                              \
    }                                                                                                                          \
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                                         \
  } while (0)
#define DPRHOT_SS_LAUNCH(CPT, NS)                      \
  do {                                                 \
    if (B <= SS_ROWS) DPRHOT_SS_LAUNCH_N(CPT, NS, 1);  \
    else DPRHOT_SS_LAUNCH_N(CPT, NS, 2);               \
  } while (0)
  if (lds > 160 * 1024 || fp.splits > 4) return fail(DPRHOT_E_UNSUPPORTED, "small step: Nc=%d needs %zu bytes of LDS", Nc, lds);
  if (ncp <= 256) {
    if (fp.splits <= 1) DPRHOT_SS_LAUNCH(1, 1);
    else if (fp.splits == 2) DPRHOT_SS_LAUNCH(1, 2);
    else if (fp.splits == 3) DPRHOT_SS_LAUNCH(1, 3);
    else DPRHOT_SS_LAUNCH(1, 4);
  } else if (ncp <= 512) {
    if (fp.splits <= 1) DPRHOT_SS_LAUNCH(2, 1);
    else if (fp.splits == 2) DPRHOT_SS_LAUNCH(2, 2);
    else if (fp.splits == 3) DPRHOT_SS_LAUNCH(2, 3);
    else DPRHOT_SS_LAUNCH(2, 4);
  } else if (ncp <= 768) {
    if (fp.splits <= 1) DPRHOT_SS_LAUNCH(3, 1);
    else if (fp.splits == 2) DPRHOT_SS_LAUNCH(3, 2);
    else if (fp.splits == 3) DPRHOT_SS_LAUNCH(3, 3);
    else DPRHOT_SS_LAUNCH(3, 4);
  } else {
    DPRHOT_SS_LAUNCH_N(5, 1, 1);  // (B <= 32 only: small_step_ok; above 768 columns the sim launch writes one slab: fwd_plan)
  }
#undef DPRHOT_SS_LAUNCH_N
#undef DPRHOT_SS_LAUNCH
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

// World size > 1, everything after the all-gather in ONE call: the column mask is read from the packed buffer (no
// unpack launch) and this rank's loss numerator rides in dC_part (no loss all-reduce): see include/dprhot.h.
int dprhot_inbatch_step_packed_f32(const float* q, const dprhot_bf16* gathered, dprhot_bf16* Qb, int B, int W, int rank, int n_ctx,
                                   int d, const int64_t* y, float inv_T, float grad_scale, float h_scale, const float* d_scale,
                                   float* row_loss, float* row_lse, float* loss_sum, dprhot_bf16* G, float* dQ, float* dC_part,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(q && gathered && Qb && y && loss_sum, "NULL pointer");
  REQUIRE(W > 0 && rank >= 0 && rank < W && n_ctx > 0 && d > 0 && d % 8 == 0, "bad argument W=%d rank=%d n_ctx=%d d=%d", W, rank, n_ctx, d);
  int rows_c = 0;
  if (int rc = dprhot_packed_rows(n_ctx, d, &rows_c)) return rc;
  PackedSpec ps;
  ps.base = reinterpret_cast<const uint8_t*>(gathered);
  ps.rows_c = rows_c;
  ps.n_ctx = n_ctx;
  ps.row_bytes = d * 2;
  ps.stamp_src = loss_sum;
  PackedScope scope(ps);
  // the gathered buffer IS the context matrix: [W * rows_c, d] bf16, mask rows = always-masked columns
  return dprhot_inbatch_step_f32(q, nullptr, Qb, const_cast<dprhot_bf16*>(gathered), B, W * rows_c, d, y, (int64_t)rank * rows_c,
                                 nullptr, inv_T, grad_scale, h_scale, d_scale, nullptr, row_loss, row_lse, loss_sum, G, dQ, dC_part,
                                 workspace, workspace_bytes, stream);
}

// ---- gradient all-reduce of the towers (reference hook: dpr_task.py:90-92) -------------------------------
static int gc_blocks(size_t groups) {
  const size_t want = (groups + 255) / 256;
  return (int)(want < 1 ? 1 : (want > (size_t)kNumCU * 8 ? (size_t)kNumCU * 8 : want));
}

// ---- the step as the autograd operator runs it ----------------------------------------------------------------------------------
static int train_dc_kind_ok(int dc_kind, int B, int Nc, int d) {
  if (dc_kind == GC_FP32) return DPRHOT_OK;
  if (dc_kind == GC_BF16 && sk_plan(B, Nc, d).ok) return DPRHOT_OK;  // the skinny step's dC units have a bf16 epilogue
  return fail(DPRHOT_E_UNSUPPORTED, "train_step: dc_kind=%d at B=%d Nc=%d d=%d: this shape's plan writes fp32 dC_part only (ask for dc_kind = 2)", dc_kind,
              B, Nc, d);
}

int dprhot_train_dq_slabs(int B, int Nc, int d, int* h_nslabs) {
  REQUIRE(h_nslabs != nullptr, "NULL pointer");
  if (int rc = check_shape(B, Nc, d)) return rc;
  const SkPlan sk = sk_plan(B, Nc, d);
  *h_nslabs = sk.ok && sk.nslices > 1 ? sk.nslices : 0;  // the other plans (and a one-slice few-rows plan) do not split dQ or combine it themselves
  // where the step can run without the dScores (dprhot_step_wants_g = 0) its slabs are slice-normalised and need the row statistics to
  // be combined: the step's own finishing launch does that, nothing is left to the caller
  if (sk_fused_plan(sk, cdiv(Nc, SK_COLS), cdiv(Nc, 64), B, Nc, d).ok) *h_nslabs = 0;
  return DPRHOT_OK;
}

static int train_dq_part_ok(const float* dq_part, int B, int Nc, int d) {
  if (dq_part == nullptr) return DPRHOT_OK;
  REQUIRE(aligned16(dq_part), "dq_part must be 16-byte aligned");
  int n = 0;
  if (int rc = dprhot_train_dq_slabs(B, Nc, d, &n)) return rc;
  if (n == 0) return fail(DPRHOT_E_INVALID, "train_step: dq_part given but B=%d Nc=%d d=%d leaves no slabs (dprhot_train_dq_slabs = 0)", B, Nc, d);
  return DPRHOT_OK;
}

int dprhot_train_step_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d, const int64_t* y,
                          int64_t y_offset, const uint8_t* colmask, float inv_T, float grad_scale, float loss_scale, const float* d_scale,
                          float* row_loss, float* row_lse, float* loss_out, dprhot_bf16* G, float* dQ, float* dq_part, void* dC_part,
                          int dc_kind, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_shape(B, Nc, d)) return rc;
  if (int rc = train_dc_kind_ok(dc_kind, B, Nc, d)) return rc;
  if (int rc = train_dq_part_ok(dq_part, B, Nc, d)) return rc;
  LossScaleScope ls(loss_scale);
  DcKindScope dk(dc_kind == GC_BF16);
  DqPartScope dp(dq_part);
  return dprhot_inbatch_step_f32(q, c, Qb, Cb, B, Nc, d, y, y_offset, colmask, inv_T, grad_scale, 1.0f, d_scale, nullptr, row_loss, row_lse,
                                 loss_out, G, dQ, static_cast<float*>(dC_part), workspace, workspace_bytes, stream);
}

int dprhot_train_step_packed_f32(const float* q, const dprhot_bf16* gathered, dprhot_bf16* Qb, int B, int W, int rank, int n_ctx, int d,
                                 const int64_t* y, float inv_T, float grad_scale, float loss_scale, const float* d_scale, float* row_loss,
                                 float* row_lse, float* loss_out, dprhot_bf16* G, float* dQ, float* dq_part, void* dC_part, int dc_kind,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  REQUIRE(W > 0 && n_ctx > 0 && d > 0 && d % 8 == 0, "bad argument W=%d n_ctx=%d d=%d", W, n_ctx, d);
  int rows_c = 0;
  if (int rc = dprhot_packed_rows(n_ctx, d, &rows_c)) return rc;
  if (int rc = train_dc_kind_ok(dc_kind, B, W * rows_c, d)) return rc;
  if (int rc = train_dq_part_ok(dq_part, B, W * rows_c, d)) return rc;
  LossScaleScope ls(loss_scale);
  DcKindScope dk(dc_kind == GC_BF16);
  DqPartScope dp(dq_part);
  return dprhot_inbatch_step_packed_f32(q, gathered, Qb, B, W, rank, n_ctx, d, y, inv_T, grad_scale, 1.0f, d_scale, row_loss, row_lse, loss_out,
                                        G, dQ, static_cast<float*>(dC_part), workspace, workspace_bytes, stream);
}

int dprhot_rescale_grads(float* dQ, size_t n_dq, const float* dq_part, int nslabs, void* dC, size_t n_dc, int dc_kind, const float* go,
                         const float* used, float* out2, void* stream) {
  REQUIRE(go && used && out2, "NULL pointer");
  REQUIRE(out2 != go && out2 != used && out2 + 1 != go && out2 + 1 != used, "out2 must not alias go / used");
  REQUIRE((dQ != nullptr || n_dq == 0) && (dC != nullptr || n_dc == 0), "NULL gradient with a non-zero count");
  REQUIRE(n_dq % 8 == 0 && n_dc % 8 == 0, "counts must be multiples of 8 (n_dq=%zu n_dc=%zu)", n_dq, n_dc);
  REQUIRE(dc_kind == GC_FP32 || dc_kind == GC_BF16, "dc_kind=%d (2 fp32, 0 bf16)", dc_kind);
  REQUIRE(nslabs >= 0 && (nslabs == 0 || (dq_part != nullptr && n_dq > 0)), "nslabs=%d needs dq_part and dQ", nslabs);
  REQUIRE(aligned16(dQ) && aligned16(dC) && aligned16(dq_part), "pointers must be 16-byte aligned");
  // the common case leaves after two scalar loads: as few workgroups as still stream at full rate when the scale did change
  const size_t groups = (n_dq + n_dc) / 8;
  int nb = gc_blocks(groups / 4 + 1);
  if (nb > 2 * kNumCU) nb = 2 * kNumCU;
  int nqb = 0;
  if (nslabs > 0) {  // one 16-byte group of dQ per thread: the slab loads of a group are the latency chain
    nqb = (int)((n_dq / 8 + 255) / 256);
    nb += nqb;
  }
  const dim3 grid(nb), block(256);
  if (dc_kind == GC_FP32)
    hipLaunchKernelGGL(rescale_grads_kernel<GC_FP32>, grid, block, 0, (hipStream_t)stream, dQ, n_dq / 8, dq_part, nslabs, nqb, dC, n_dc / 8, go, used, out2);
  else
    hipLaunchKernelGGL(rescale_grads_kernel<GC_BF16>, grid, block, 0, (hipStream_t)stream, dQ, n_dq / 8, dq_part, nslabs, nqb, dC, n_dc / 8, go, used, out2);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_grad_pack(const float* bucket, size_t n, float scale, int wire, void* send, size_t n_padded, void* stream) {
  REQUIRE(bucket && send, "NULL pointer");
  REQUIRE(n > 0 && n_padded >= n && n_padded % 8 == 0, "bad sizes n=%zu n_padded=%zu (n_padded %% 8 == 0, >= n)", n, n_padded);
  REQUIRE(wire >= GC_BF16 && wire <= GC_FP32, "wire=%d (0 bf16, 1 fp16, 2 fp32)", wire);
  REQUIRE(aligned16(bucket) && aligned16(send), "pointers must be 16-byte aligned");
  const size_t tiles = (n_padded / 4 + 256 * GC_UT - 1) / (256 * GC_UT);
  REQUIRE(tiles < (1ull << 31), "n_padded=%zu", n_padded);
  const dim3 grid((unsigned)tiles), block(256);  // one workgroup per contiguous tile (gradcomm.h)
  hipStream_t st = (hipStream_t)stream;
  if (wire == GC_BF16) hipLaunchKernelGGL(grad_pack_kernel<GC_BF16>, grid, block, 0, st, bucket, n, scale, send, n_padded);
  else if (wire == GC_FP16) hipLaunchKernelGGL(grad_pack_kernel<GC_FP16>, grid, block, 0, st, bucket, n, scale, send, n_padded);
  else hipLaunchKernelGGL(grad_pack_kernel<GC_FP32>, grid, block, 0, st, bucket, n, scale, send, n_padded);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_grad_sum_shards(const void* recv, int W, size_t shard, int wire, int out_kind, void* out, void* stream) {
  REQUIRE(recv && out, "NULL pointer");
  REQUIRE(W > 0 && shard > 0 && shard % 8 == 0, "bad sizes W=%d shard=%zu (shard %% 8 == 0)", W, shard);
  REQUIRE(wire >= GC_BF16 && wire <= GC_FP32 && (out_kind == wire || out_kind == GC_FP32), "wire=%d out_kind=%d (out = wire kind or fp32)", wire,
          out_kind);
  REQUIRE(aligned16(recv) && aligned16(out), "pointers must be 16-byte aligned");
  const dim3 grid(gc_blocks(shard / 8)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define DPRHOT_GC_SUM(WK, OK) hipLaunchKernelGGL((grad_sum_shards_kernel<WK, OK>), grid, block, 0, st, recv, W, shard, out)
  if (wire == GC_BF16 && out_kind == GC_BF16) DPRHOT_GC_SUM(GC_BF16, GC_BF16);
  else if (wire == GC_BF16) DPRHOT_GC_SUM(GC_BF16, GC_FP32);
  else if (wire == GC_FP16 && out_kind == GC_FP16) DPRHOT_GC_SUM(GC_FP16, GC_FP16);
  else if (wire == GC_FP16) DPRHOT_GC_SUM(GC_FP16, GC_FP32);
  else DPRHOT_GC_SUM(GC_FP32, GC_FP32);
#undef DPRHOT_GC_SUM
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

int dprhot_grad_unpack(const void* full, int kind, float* bucket, size_t n, void* stream) {
  REQUIRE(full && bucket && n > 0, "bad argument");
  REQUIRE(kind >= GC_BF16 && kind <= GC_FP32, "kind=%d (0 bf16, 1 fp16, 2 fp32)", kind);
  REQUIRE(aligned16(full) && aligned16(bucket), "pointers must be 16-byte aligned");
  const size_t tiles = (n / 4 + 256 * GC_UT - 1) / (256 * GC_UT);
  REQUIRE(tiles < (1ull << 31), "n=%zu", n);
  const dim3 grid((unsigned)(tiles > 0 ? tiles : 1)), block(256);  // one workgroup per contiguous tile (gradcomm.h)
  hipStream_t st = (hipStream_t)stream;
  if (kind == GC_BF16) hipLaunchKernelGGL(grad_unpack_kernel<GC_BF16>, grid, block, 0, st, full, bucket, n);
  else if (kind == GC_FP16) hipLaunchKernelGGL(grad_unpack_kernel<GC_FP16>, grid, block, 0, st, full, bucket, n);
  else hipLaunchKernelGGL(grad_unpack_kernel<GC_FP32>, grid, block, 0, st, full, bucket, n);
  HIP_TRY(hipGetLastError());
  return DPRHOT_OK;
}

}  // extern "C"

// ---- optional communicator: the path's collectives issued straight on the caller's stream ----
// torch.distributed wraps every collective in a stream hand-over (event to RCCL's stream and back) and ~17 us of host
// work; the path's three collectives are tiny and sit on the critical path of a ~10 us step.  RCCL is taken from the
// process (the librccl.so PyTorch already loaded) with dlopen/dlsym -- no link-time dependency, and the selftest and
// single-GPU users never touch it.
namespace {
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
struct RcclApi {
  int (*GetUniqueId)(RcclUniqueId*);
  int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int);
  int (*CommDestroy)(RcclComm);
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t);
  int (*ReduceScatter)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
  const char* (*GetErrorString)(int);
  int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t);
  int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  bool ok, p2p;
};
constexpr int kRcclInt8 = 0, kRcclFloat16 = 6, kRcclFloat32 = 7, kRcclBfloat16 = 9, kRcclSum = 0;

const RcclApi& rccl() {
  static const RcclApi api = []() {
    RcclApi a{};
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
    if (!h) return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.ReduceScatter = reinterpret_cast<decltype(a.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.ReduceScatter && a.AllReduce && a.GetErrorString;
    a.p2p = a.ok && a.Send && a.Recv && a.GroupStart && a.GroupEnd;
    return a;
  }();
  return api;
}
struct DprhotComm { RcclComm comm; int W, rank; };
#define RCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    const int r_ = (expr);                                                                          \
    if (r_ != 0) return fail(DPRHOT_E_HIP, "%s: %s", #expr, rccl().GetErrorString(r_));             \
  } while (0)
}  // namespace

extern "C" {

int dprhot_comm_unique_id(void* id128) {
  REQUIRE(id128 != nullptr, "NULL pointer");
  if (!rccl().ok) return fail(DPRHOT_E_UNSUPPORTED, "librccl.so (ncclGetUniqueId ...) not found in this process");
  RCCL_TRY(rccl().GetUniqueId(static_cast<RcclUniqueId*>(id128)));
  return DPRHOT_OK;
}

int dprhot_comm_init(const void* id128, int W, int rank, void** h_out) {
  REQUIRE(id128 && h_out && W > 0 && rank >= 0 && rank < W, "bad argument");
  if (!rccl().ok) return fail(DPRHOT_E_UNSUPPORTED, "librccl.so not found in this process");
  RcclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  RcclComm c = nullptr;
  RCCL_TRY(rccl().CommInitRank(&c, W, id, rank));  // collective over all W ranks, on the current HIP device
  *h_out = new DprhotComm{c, W, rank};
  return DPRHOT_OK;
}

int dprhot_comm_destroy(void* h) {
  if (h == nullptr) return DPRHOT_OK;
  DprhotComm* c = static_cast<DprhotComm*>(h);
  const int r = rccl().CommDestroy(c->comm);
  delete c;
  if (r != 0) return fail(DPRHOT_E_HIP, "ncclCommDestroy: %s", rccl().GetErrorString(r));
  return DPRHOT_OK;
}

int dprhot_allgather_ctx(void* h, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  REQUIRE(h && send && recv && bytes_per_rank > 0, "bad argument");
  DprhotComm* c = static_cast<DprhotComm*>(h);
  RCCL_TRY(rccl().AllGather(send, recv, bytes_per_rank, kRcclInt8, c->comm, (hipStream_t)stream));
  return DPRHOT_OK;
}

int dprhot_reducescatter_dc(void* h, const float* send, float* recv, size_t count_per_rank, void* stream) {
  REQUIRE(h && send && recv && count_per_rank > 0, "bad argument");
  DprhotComm* c = static_cast<DprhotComm*>(h);
  RCCL_TRY(rccl().ReduceScatter(send, recv, count_per_rank, kRcclFloat32, kRcclSum, c->comm, (hipStream_t)stream));
  return DPRHOT_OK;
}

int dprhot_reducescatter_rows(void* h, const void* send, void* recv, size_t count_per_rank, int kind, void* stream) {
  REQUIRE(h && send && recv && count_per_rank > 0, "bad argument");
  REQUIRE(kind >= GC_BF16 && kind <= GC_FP32, "kind=%d (0 bf16, 1 fp16, 2 fp32)", kind);
  DprhotComm* c = static_cast<DprhotComm*>(h);
  const int dt = kind == GC_BF16 ? kRcclBfloat16 : (kind == GC_FP16 ? kRcclFloat16 : kRcclFloat32);
  RCCL_TRY(rccl().ReduceScatter(send, recv, count_per_rank, dt, kRcclSum, c->comm, (hipStream_t)stream));
  return DPRHOT_OK;
}

int dprhot_allreduce_sum(void* h, float* buf, size_t count, void* stream) {
  REQUIRE(h && buf && count > 0, "bad argument");
  DprhotComm* c = static_cast<DprhotComm*>(h);
  RCCL_TRY(rccl().AllReduce(buf, buf, count, kRcclFloat32, kRcclSum, c->comm, (hipStream_t)stream));
  return DPRHOT_OK;
}

// ---- the same two collectives as direct all-pairs exchanges (SURVEY.md section 8(e) "Topology") ---------------------------------
// One RCCL group of W sends and W receives (the self pair included: a device copy): every peer's transfer is its own point-to-point
// operation, so on the fully connected node the W - 1 transfers of a rank travel over W - 1 different xGMI links at once.
int dprhot_comm_has_allpairs(void* h) {
  REQUIRE(h != nullptr, "bad argument");
  return rccl().p2p ? 1 : 0;
}

int dprhot_allgather_allpairs(void* h, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  REQUIRE(h && send && recv && bytes_per_rank > 0, "bad argument");
  if (!rccl().p2p) return fail(DPRHOT_E_UNSUPPORTED, "ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd not found in this process's librccl.so");
  DprhotComm* c = static_cast<DprhotComm*>(h);
  hipStream_t st = (hipStream_t)stream;
  RCCL_TRY(rccl().GroupStart());
  int rc = 0;
  for (int k = 0; k < c->W && rc == 0; ++k) {
    rc = rccl().Send(send, bytes_per_rank, kRcclInt8, k, c->comm, st);
    if (rc == 0) rc = rccl().Recv(static_cast<char*>(recv) + (size_t)k * bytes_per_rank, bytes_per_rank, kRcclInt8, k, c->comm, st);
  }
  const int rc2 = rccl().GroupEnd();  // (always closed: an open group would swallow the caller's next collective)
  if (rc != 0) return fail(DPRHOT_E_HIP, "ncclSend / ncclRecv: %s", rccl().GetErrorString(rc));
  if (rc2 != 0) return fail(DPRHOT_E_HIP, "ncclGroupEnd: %s", rccl().GetErrorString(rc2));
  return DPRHOT_OK;
}

int dprhot_reducescatter_allpairs(void* h, const void* send, void* tmp, void* recv, size_t count_per_rank, int kind, int out_kind, void* stream) {
  REQUIRE(h && send && tmp && recv && count_per_rank > 0, "bad argument");
  REQUIRE(kind >= GC_BF16 && kind <= GC_FP32 && (out_kind == kind || out_kind == GC_FP32), "kind=%d out_kind=%d (0 bf16, 1 fp16, 2 fp32; out = kind or fp32)", kind, out_kind);
  REQUIRE(count_per_rank % 8 == 0, "count_per_rank=%zu must be a multiple of 8", count_per_rank);
  if (!rccl().p2p) return fail(DPRHOT_E_UNSUPPORTED, "ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd not found in this process's librccl.so");
  DprhotComm* c = static_cast<DprhotComm*>(h);
  hipStream_t st = (hipStream_t)stream;
  const size_t es = kind == GC_FP32 ? 4 : 2;
  const int dt = kind == GC_BF16 ? kRcclBfloat16 : (kind == GC_FP16 ? kRcclFloat16 : kRcclFloat32);
  RCCL_TRY(rccl().GroupStart());
  int rc = 0;
  for (int k = 0; k < c->W && rc == 0; ++k) {
    rc = rccl().Send(static_cast<const char*>(send) + (size_t)k * count_per_rank * es, count_per_rank, dt, k, c->comm, st);
    if (rc == 0) rc = rccl().Recv(static_cast<char*>(tmp) + (size_t)k * count_per_rank * es, count_per_rank, dt, k, c->comm, st);
  }
  const int rc2 = rccl().GroupEnd();
  if (rc != 0) return fail(DPRHOT_E_HIP, "ncclSend / ncclRecv: %s", rccl().GetErrorString(rc));
  if (rc2 != 0) return fail(DPRHOT_E_HIP, "ncclGroupEnd: %s", rccl().GetErrorString(rc2));
  return dprhot_grad_sum_shards(tmp, c->W, count_per_rank, kind, out_kind, recv, stream);  // fp32 accumulation, rank order
}

}  // extern "C"
