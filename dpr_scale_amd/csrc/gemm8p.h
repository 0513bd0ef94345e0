// gemm8p.h -- the 256x256 similarity GEMM with a phase-interleaved K loop (dpr_task.py:98-105 at evaluation / retrieval /
// large-batch sizes): D[M,N] = A[M,K] * B[N,K]^T, both operands k-major bf16, persistent workgroups.
//
// gemm256.h runs one barrier per K step: DMA the next step, wait for everything (vmcnt(0)), barrier, 24 fragment reads, 64
// MFMAs -- all eight waves read LDS at the same time and then all queue on the matrix pipe (45 % of the dense peak in the K
// loop alone).  Here a K step (64 deep) is four phases, one per 64 x 32 quadrant of the wave's 128 x 64 output, and a phase is
//     [ fragment reads of ONE half-tile | LDS-DMA of ONE future half-tile | counted vmcnt ]  barrier
//     [ 16 MFMAs ]                                                                            barrier
// The wave groups wm = 0 and wm = 1 (one wave of each on every SIMD) run one barrier apart, so one group's LDS / DMA section
// sits under the other group's MFMA section.  The DMA runs six phases (1.5 K steps) ahead of its first read and is never
// drained inside the loop: vmcnt(10) leaves the five youngest half-tiles in flight across the barriers.
//
// Half-tiles.  A K step of an operand is two 16 KiB images [128 rows][64 k]: A-half h holds the rows
// {wm * 128 + h * 64 + j}, B-half h the columns {wn * 64 + h * 32 + j}: exactly what ALL waves need for their quadrant row /
// column h.  Quadrant order (0,0) (0,1) (1,1) (1,0): one operand half changes per phase.  Per K step t (LDS buffer t & 1):
//     phase  reads (-> registers)   MFMA quadrant        DMA issued (K step, half)
//     p0     A0(t)                  (0,0) A0 x B0(t)     A1(t+1)
//     p1     B1(t)                  (0,1) A0 x B1        B0(t+2)
//     p2     A1(t)                  (1,1) A1 x B1        A0(t+2)
//     p3     B0(t+1)                (1,0) A1 x B0(t)     B1(t+2)
// Every half-tile is read exactly six phases after its DMA was issued, every DMA overwrites an image two phases after its
// last read (one phase would race with the lagging wave group), and every wait sits one phase before the read it covers (a
// wave's vmcnt only covers its own share of the image; the barrier after it covers everybody's).
// K steps beyond the tile's last belong to the workgroup's NEXT tile (persistent: the pipeline never drains).
//
// MFMA shape: v_mfma_f32_32x32x16_bf16 (the matrix pipe's better-fed shape: 2382 vs 2075 TFLOP/s in the instruction
// microbenchmark; measured here 148.7 k vs 163.1 k cycles per XCD at 8192^2 x 768), operands swapped so that a lane holds runs of
// 4 consecutive columns of one row (G8Acc): row statistics reduce inside a lane plus ONE cross-lane step, stores are 8 / 16
// bytes per lane.  Every output element is the same chain of k slices in increasing order as in gemm256.h, and the logits are
// bit-identical to that kernel's (scratch/g8probe.hip checks it).
//
// What did not work (profiles/r02_g8_ablation.txt): the wave groups in step (-33 %), one barrier per phase (-21 %), a 4-wave
// workgroup with one wave per SIMD and no hand-over at all (scratch/negative/gemm1w.h: the barrier then idles the pipe, -33 %).
// The loop is at 66 % MFMA-busy cycles with a null epilogue; its barrier / MFMA skeleton alone (no DMA, no reads) reaches 69-72 %.
// It is POWER-limited: on zero-filled operands the same launch is 47 % faster (1.90 vs 1.29 PFLOP/s), and the two-phase variant
// below (SCHED = 2: 16 MFMAs per section, half the hand-overs) is bit-identical and exactly as fast.
#pragma once
#include "gemm256.h"
#include "rowwise.h"

namespace dprhot {

constexpr int G8_HALF = 128 * 64;                                   // elements of a half-tile image
constexpr size_t g8_tiles_bytes = (size_t)2 * 4 * G8_HALF * 2;      // 2 buffers x {A0, A1, B0, B1} = 128 KiB
constexpr size_t g8_scratch_bytes = 18 * 1024;                      // epilogue scratch (Epi8Store: 8 waves x 8 rows x 68 floats)
constexpr size_t g8_meta_bytes = (size_t)2 * 1024 * sizeof(int);    // 2 x 1024 words of per-tile epilogue inputs
constexpr size_t g8_lds_total = g8_tiles_bytes + g8_scratch_bytes + g8_meta_bytes;  // 154 KiB

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void g8_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g8_wait_lgkm0() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);  // register-only MFMAs may not be hoisted above the wait (guide 5.4 rule 18)
}
__device__ __forceinline__ void g8_bar() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// LDS accesses of the epilogues go through inline asm: hipcc orders a plain LDS access behind every LDS-DMA in flight with
// s_waitcnt vmcnt(0) (it cannot tell the epilogue words from the tile images), which would drain the next tile's prefetch.
typedef __attribute__((address_space(3))) int g8_lds_int;
__device__ __forceinline__ unsigned g8_lds_addr(const void* p) { return (unsigned)(uintptr_t)(g8_lds_int*)p; }
__device__ __forceinline__ void g8_lds_write(void* p, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(g8_lds_addr(p)), "v"(v) : "memory");
}
__device__ __forceinline__ int g8_lds_read(const void* p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(g8_lds_addr(p)) : "memory");
  return v;
}
typedef int g8_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ g8_i32x4 g8_lds_read4(const void* p) {
  g8_i32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(g8_lds_addr(p)) : "memory");
  return v;
}

// What an epilogue sees of one tile.
struct Tile8 {
  int m0, n0;       // tile origin
  int wm, wn;       // wave position (2 x 4): rows wm * 128.., columns wn * 64..
  int lane, tid;
  int bx, by, nbx;  // tile indices, column-tile count
  float* scratch;   // LDS, g8_scratch_bytes (gemm8p) / unused (gemm2w)
  int* meta;        // LDS: the tile's input words, fetched by LDS-DMA from Epi::meta_src while the previous tile was computed
  int ncol = G2_B;  // tile columns: 256 (gemm8p.h, 4 wave strips) or 128 (gemm2w.h, 2 wave strips)
  int nthr = G2_THREADS;
};

// The wave's 128 x 64 accumulators: v[a][b] (a < 4, b < 2) of 16 registers; lane (i = lane & 31, h = lane >> 5), register r:
//   row wm*128 + a*32 + i,   column wn*64 + b*32 + (r >> 2)*8 + h*4 + (r & 3)
// i.e. a lane holds, of each of its 4 rows, eight runs of 4 consecutive columns; lanes i and i + 32 share a row.
struct G8Acc {
  f32x16 v[4][2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[a][b][r] = 0.f;
  }
};

// Epilogue interface (all members __device__):
//   const void* meta_src(int m0, int n0, int e) const   e < 1024: 4-byte aligned, always valid source address of input word e of the
//                                                       tile at (m0, n0); by convention e < 256: one word per column n0 + e,
//                                                       256 <= e < 512 and 512 <= e < 768: two words per row m0 + (e & 255)
//   void finish(G8Acc& acc, const Tile8&) const         every wave executes the same number of barriers in it; it runs with the next
//                                                       tile's first K steps in flight: LDS only through g8_lds_* (asm), tile images
//                                                       untouched; global stores allowed (vmcnt retires in order, loads and stores
//                                                       alike: scratch/g8probe.hip's order probe), plain global loads are not
//                                                       (hipcc would wait vmcnt(0) for them: the whole DMA pipeline)

// LDS fragment: lane reads 16 bytes of image row r0 + (lane & 31), 16-byte chunk kk*2 + (lane >> 5) of the 64-deep K step
// (v_mfma_f32_32x32x16_bf16 operand: 32 rows x 16 k); conflict-free under the c ^ ((r >> 1) & 7) swizzle (the 16 lanes served
// together cover both row parities x all 8 swizzle values).
__device__ __forceinline__ bf16x8 g8_frag32(const uint16_t* T, int r0, int kk, int lane) {
  const int row = r0 + (lane & 31);
  return *reinterpret_cast<const bf16x8*>(T + row * 64 + (((kk * 2 + (lane >> 5)) ^ ((row >> 1) & 7)) << 3));
}

#define G8_RD_A(IMG)                                                                                                      \
  if constexpr (!(VAR & 16)) {                                                                                            \
    _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_) _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_)                     \
        af[a_ * 4 + k_] = g8_frag32((IMG), wm * 64 + a_ * 32, k_, lane);                                                  \
  }
#define G8_RD_B(DST, IMG)                                                                                                 \
  if constexpr (!(VAR & 16)) {                                                                                            \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) DST[k_] = g8_frag32((IMG), wn * 32, k_, lane);                       \
  }
#define G8_MM(AH, BH, BQ)                                                                                                 \
  if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);                                                                \
  _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) _Pragma("unroll") for (int a_ = 0; a_ < 2; ++a_)                       \
      acc.v[(AH) * 2 + a_][(BH)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BQ[k_], af[a_ * 4 + k_], acc.v[(AH) * 2 + a_][(BH)], 0, 0, 0); \
  if constexpr (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);

// VAR: ablation switches of scratch/g8probe.hip (timing only, results are garbage): 1 no s_setprio, 2 wave groups in step,
// 4 one barrier per phase, 8 no DMA after the prologue, 16 no fragment reads; the library always builds VAR = 0.
template <class Epi, int VAR = 0, int SCHED = 4>
__global__ __launch_bounds__(G2_THREADS, 2) void gemm8p_kernel(GemmArgs p, Epi epi, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = p.K / G2_BK;  // the launcher guarantees K % 128 == 0: an even number of K steps
  const int ntiles = nbx * nby;
  float* const scratch = reinterpret_cast<float*>(smem + 8 * G8_HALF);
  int* const meta0 = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + g8_scratch_bytes);

  // Per-lane source offsets (bytes; the launcher guarantees operands below 4 GiB) of this wave's two DMA instructions per
  // half-tile, for the tile the DMA front is in: instruction j covers the image rows (wave * 2 + j) * 8 + (lane >> 3); the 16-byte
  // chunk c of image row r lives at position c ^ ((r >> 1) & 7) and the DMA writes lane-linearly, so the lane at position
  // (lane & 7) fetches chunk (lane & 7) ^ ((r >> 1) & 7)  (source swizzle).  Plain scalars: an array selected at run time would
  // live in scratch memory, and a scratch load in the K loop is a vmcnt(0).
  unsigned oa00, oa01, oa10, oa11, ob00, ob01, ob10, ob11;
  auto aim = [&](int bx, int by) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));  // opaque: nothing derived here is kept live (or spilled) across the K loop
    const int ln = t_ & 63;
    const int lr0 = (wave * 2 + 0) * 8 + (ln >> 3), lr1 = lr0 + 8;
    const int c0 = ((ln & 7) ^ ((lr0 >> 1) & 7)) * 16, c1 = ((ln & 7) ^ ((lr1 >> 1) & 7)) * 16;
    const int ar0 = by * G2_B + (lr0 >> 6) * 128 + (lr0 & 63), ar1 = by * G2_B + (lr1 >> 6) * 128 + (lr1 & 63);
    const int br0 = bx * G2_B + (lr0 >> 5) * 64 + (lr0 & 31), br1 = bx * G2_B + (lr1 >> 5) * 64 + (lr1 & 31);
    const unsigned la = (unsigned)p.lda * 2u, lb = (unsigned)p.ldb * 2u;
    oa00 = (unsigned)min(ar0, p.M - 1) * la + c0;
    oa01 = (unsigned)min(ar1, p.M - 1) * la + c1;
    oa10 = (unsigned)min(ar0 + 64, p.M - 1) * la + c0;
    oa11 = (unsigned)min(ar1 + 64, p.M - 1) * la + c1;
    ob00 = (unsigned)min(br0, p.N - 1) * lb + c0;
    ob01 = (unsigned)min(br1, p.N - 1) * lb + c1;
    ob10 = (unsigned)min(br0 + 32, p.N - 1) * lb + c0;
    ob11 = (unsigned)min(br1 + 32, p.N - 1) * lb + c1;
  };
  // the 1024 input words of the epilogue of the tile at (bx, by) -> meta buffer mb: two 4-byte LDS-DMAs per lane
  auto fetch_meta = [&](int bx, int by, int* mb) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(by * G2_B, bx * G2_B, t_), (g2_lds_ptr*)(mb + wave * 64), 4, 0, 0);
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)epi.meta_src(by * G2_B, bx * G2_B, t_ + 512), (g2_lds_ptr*)(mb + 512 + wave * 64), 4, 0, 0);
  };
  // image addresses: buffer par, {A0, A1, B0, B1}
  auto img = [&](int par, int which) { return smem + (par * 4 + which) * G8_HALF; };
  bool dma_off = false;  // VAR & 8
  // Stage one half-tile of K step TT (relative to the tile being computed; TT >= nt: the DMA front is in the next tile, whose
  // offsets are in place by then) and wait until at most the five youngest half-tiles are in flight.  Never a branch: past the
  // workgroup's last tile the front keeps fetching the same tile again into images nobody reads (a scalar branch in every load
  // section costs more than the loads it would skip); the kernel drains vmcnt before it ends.
#define G8_STAGE(P, O0, O1, TT, IMG)                                                                                              \
  if ((VAR & 8) && dma_off) {                                                                                                     \
  } else {                                                                                                                        \
    const char* base_ = reinterpret_cast<const char*>(P) + (size_t)(((TT) >= nt ? (TT) - nt : (TT)) * (G2_BK * 2));               \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(base_ + (size_t)(O0)), (g2_lds_ptr*)((IMG) + (wave * 2 + 0) * 512), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(base_ + (size_t)(O1)), (g2_lds_ptr*)((IMG) + (wave * 2 + 1) * 512), 16, 0, 0); \
    g8_wait_vm<10>();                                                                                                             \
  }

  // the same without the wait (SCHED 2 issues one or three half-tiles per load section and waits once)
#define G8_STAGE_NW(P, O0, O1, TT, IMG)                                                                                           \
  if (!((VAR & 8) && dma_off)) {                                                                                                  \
    const char* base_ = reinterpret_cast<const char*>(P) + (size_t)(((TT) >= nt ? (TT) - nt : (TT)) * (G2_BK * 2));               \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(base_ + (size_t)(O0)), (g2_lds_ptr*)((IMG) + (wave * 2 + 0) * 512), 16, 0, 0); \
    __builtin_amdgcn_global_load_lds((g2_gbl_ptr*)(base_ + (size_t)(O1)), (g2_lds_ptr*)((IMG) + (wave * 2 + 1) * 512), 16, 0, 0); \
  }

  int tile = blockIdx.x, bx, by;
  g2_tile_of(tile, nbx, nby, bx, by);
  aim(bx, by);
  int next = tile, nbx_ = bx, nby_ = by;
  bool has_next = false;

  bf16x8 af[8], bq[2][4];  // B0(t) lives in bq[t & 1], B1(t) in the other set; B0(t+1) replaces B1(t) in p3
  G8Acc acc;
  acc.zero();

  // ---- prologue: the first tile's epilogue inputs, then the DMAs the steady state would have issued before phase 0, in its order
  fetch_meta(bx, by, meta0);
  if constexpr (SCHED == 4) {
    G8_STAGE(p.B, ob00, ob01, 0, img(0, 2));
    G8_STAGE(p.A, oa00, oa01, 0, img(0, 0));
    G8_STAGE(p.B, ob10, ob11, 0, img(0, 3));
    G8_STAGE(p.A, oa10, oa11, 0, img(0, 1));
    G8_STAGE(p.B, ob00, ob01, 1, img(1, 2));
    G8_STAGE(p.A, oa00, oa01, 1, img(1, 0));  // ... vmcnt(10): B0(0) has landed
    g8_bar();
    g8_bar();
    G8_RD_B(bq[0], img(0, 2));
    G8_STAGE(p.B, ob10, ob11, 1, img(1, 3));  // ... A0(0) has landed
    g8_wait_lgkm0();
    g8_bar();
    g8_bar();
  } else {
    G8_STAGE_NW(p.B, ob00, ob01, 0, img(0, 2)); G8_STAGE_NW(p.A, oa00, oa01, 0, img(0, 0)); G8_STAGE_NW(p.B, ob10, ob11, 0, img(0, 3));
    G8_STAGE_NW(p.A, oa10, oa11, 0, img(0, 1));
    G8_STAGE_NW(p.B, ob00, ob01, 1, img(1, 2)); G8_STAGE_NW(p.A, oa00, oa01, 1, img(1, 0)); G8_STAGE_NW(p.B, ob10, ob11, 1, img(1, 3));
    g8_wait_vm<8>();  // A0, B0, B1 of K step 0 (and the epilogue words) have landed
    g8_bar();
    g8_bar();
  }

  int par = 0;  // meta buffer of the current tile
  dma_off = true;
  while (true) {
    if (!(VAR & 2) && wm == 1) g8_bar();  // this wave group runs one barrier behind from here on
    for (int t = 0; t < nt; t += 2) {
#define G8_KSTEP(PAR, T, SWITCH)                                                                                 \
  {                                                                                                              \
    /* p0 */                                                                                                     \
    G8_RD_A(img(PAR, 0));                                                                                        \
    G8_STAGE(p.A, oa10, oa11, (T) + 1, img((PAR) ^ 1, 1));                                                       \
    if (SWITCH) {                                                                                                \
      /* from here on the DMA front is in the workgroup's next tile */                                          \
      next = tile + (int)gridDim.x;                                                                              \
      has_next = next < ntiles;                                                                                  \
      if (has_next) {                                                                                            \
        g2_tile_of(next, nbx, nby, nbx_, nby_);                                                                  \
        aim(nbx_, nby_);                                                                                         \
        fetch_meta(nbx_, nby_, meta0 + (par ^ 1) * 1024);                                                        \
      }                                                                                                          \
    }                                                                                                            \
    g8_bar();                                                                                                    \
    g8_wait_lgkm0();                                                                                             \
    G8_MM(0, 0, bq[PAR]);                                                                                        \
    if constexpr (!(VAR & 4)) g8_bar();                                                                          \
    /* p1 */                                                                                                     \
    G8_RD_B(bq[(PAR) ^ 1], img(PAR, 3));                                                                         \
    G8_STAGE(p.B, ob00, ob01, (T) + 2, img(PAR, 2));                                                             \
    g8_bar();                                                                                                    \
    g8_wait_lgkm0();                                                                                             \
    G8_MM(0, 1, bq[(PAR) ^ 1]);                                                                                  \
    if constexpr (!(VAR & 4)) g8_bar();                                                                          \
    /* p2 */                                                                                                     \
    G8_RD_A(img(PAR, 1));                                                                                        \
    G8_STAGE(p.A, oa00, oa01, (T) + 2, img(PAR, 0));                                                             \
    g8_bar();                                                                                                    \
    g8_wait_lgkm0();                                                                                             \
    G8_MM(1, 1, bq[(PAR) ^ 1]);                                                                                  \
    if constexpr (!(VAR & 4)) g8_bar();                                                                          \
    /* p3 */                                                                                                     \
    G8_RD_B(bq[(PAR) ^ 1], img((PAR) ^ 1, 2));                                                                   \
    G8_STAGE(p.B, ob10, ob11, (T) + 2, img(PAR, 3));                                                             \
    g8_bar();                                                                                                    \
    g8_wait_lgkm0();                                                                                             \
    G8_MM(1, 0, bq[PAR]);                                                                                        \
    if constexpr (!(VAR & 4)) g8_bar();                                                                          \
  }
      // SCHED 2: a K step is TWO phases of 16 MFMAs (the matrix pipe changes hands 4 instead of 8 times per K step):
      //   X(t): reads A0, B0, B1 (t)    DMA A1(t+1)             MFMAs (0,0) (0,1)
      //   Y(t): reads A1 (t)            DMA A0, B0, B1 (t+2)    MFMAs (1,0) (1,1)
      // Every load section retires its reads (lgkmcnt(0)) BEFORE its barrier, so an image may be overwritten ONE phase after its
      // last read (the lagging wave group has retired its reads before the barrier the leading group's next section starts
      // behind); every half-tile has two phases (~1200 MFMA cycles) between its DMA and the wait that covers it: vmcnt(8).
#define G8_KSTEP2(PAR, T, SWITCH)                                                                                \
  {                                                                                                              \
    /* X */                                                                                                      \
    G8_RD_A(img(PAR, 0));                                                                                        \
    G8_RD_B(bq[0], img(PAR, 2));                                                                                 \
    G8_RD_B(bq[1], img(PAR, 3));                                                                                 \
    G8_STAGE_NW(p.A, oa10, oa11, (T) + 1, img((PAR) ^ 1, 1));                                                    \
    g8_wait_vm<8>();                                                                                             \
    if (SWITCH) {                                                                                                \
      next = tile + (int)gridDim.x;                                                                              \
      has_next = next < ntiles;                                                                                  \
      if (has_next) {                                                                                            \
        g2_tile_of(next, nbx, nby, nbx_, nby_);                                                                  \
        aim(nbx_, nby_);                                                                                         \
        fetch_meta(nbx_, nby_, meta0 + (par ^ 1) * 1024);                                                        \
      }                                                                                                          \
    }                                                                                                            \
    g8_wait_lgkm0();                                                                                             \
    g8_bar();                                                                                                    \
    G8_MM(0, 0, bq[0]);                                                                                          \
    G8_MM(0, 1, bq[1]);                                                                                          \
    g8_bar();                                                                                                    \
    /* Y */                                                                                                      \
    G8_RD_A(img(PAR, 1));                                                                                        \
    G8_STAGE_NW(p.B, ob00, ob01, (T) + 2, img(PAR, 2));                                                          \
    G8_STAGE_NW(p.A, oa00, oa01, (T) + 2, img(PAR, 0));                                                          \
    G8_STAGE_NW(p.B, ob10, ob11, (T) + 2, img(PAR, 3));                                                          \
    g8_wait_vm<8>();                                                                                             \
    g8_wait_lgkm0();                                                                                             \
    g8_bar();                                                                                                    \
    G8_MM(1, 0, bq[0]);                                                                                          \
    G8_MM(1, 1, bq[1]);                                                                                          \
    g8_bar();                                                                                                    \
  }
      if constexpr (SCHED == 4) {
        G8_KSTEP(0, t, t == nt - 2);
        G8_KSTEP(1, t + 1, false);
      } else {
        G8_KSTEP2(0, t, t == nt - 2);
        G8_KSTEP2(1, t + 1, false);
      }
    }
    if (!(VAR & 2) && wm == 0) g8_bar();  // both wave groups in step again

    // ---- epilogue of (bx, by)
    int te = tid;
    asm volatile("" : "+v"(te));  // opaque: the epilogue's lane arithmetic starts here, not above the K loop
    const Tile8 tc{by * G2_B, bx * G2_B, wm, wn, te & 63, te, bx, by, nbx, scratch, meta0 + par * 1024};
    // last tile: nothing of this workgroup may still be writing LDS when its place on the CU is handed on -- drained HERE, ahead of
    // the epilogue's stores, which may outlive the workgroup (a launch with one workgroup per tile relies on that: the stores
    // of a finished workgroup drain under the prologue of its successor)
    if (!has_next) g8_wait_vm<0>();
    epi.finish(acc, tc);
    acc.zero();
    if (!has_next) break;
    par ^= 1;
    tile = next;
    bx = nbx_;
    by = nby_;
  }
}

// ---- cross-lane step of the accumulator layout: lanes i and i + 32 hold the same row ------------------------------------------
// v_permlane32_swap D, S trades D's upper 32 lanes with S's lower 32; with D = S = x the pair becomes ([lo lo], [hi hi]): every
// lane holds x[l] and x[l ^ 32].
__device__ __forceinline__ float g8_max_x32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float g8_sum_x32(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ int g8_isum_x32(int v) {
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)(r[0] + r[1]);
}

// Inputs shared by the sim epilogues.  Raw words arrive by LDS-DMA (meta_src); meta_fix() turns them, one entry per thread, into
//   meta[e]          (e < 256)  1 where tile column e is masked or outside the matrix, else 0
//   meta[256 + e]    gold column of tile row e (global column index), -1: none / row outside the matrix
//   meta[512 + e]    second row word, untouched here (Epi8G: row logsumexp, Epi8Count: gold logit)
//   meta[768 + w]    (w < 4) != 0 when a gold column of the rows fixed by wave w falls inside this tile
struct Epi8Base {
  EpiSim sim;         // mask source (colmask / packed layout), M, N, inv_T, y, y_offset, gold
  const void* dummy;  // any valid device address: source of the words that have no input (no mask, no labels)

  __device__ __forceinline__ const uint8_t* mask_byte(int n) const {
    if (sim.packed != nullptr) {
      const int r = n / sim.p_rows_c, j = n - r * sim.p_rows_c;
      return sim.packed + (size_t)(r * sim.p_rows_c + sim.p_n_ctx) * sim.p_row_bytes + min(j, sim.p_n_ctx - 1);
    }
    return sim.colmask != nullptr ? sim.colmask + n : nullptr;
  }
  __device__ __forceinline__ const void* base_src(int m0, int n0, int e) const {
    if (e < 256) {
      const uint8_t* b = mask_byte(min(n0 + e, sim.N - 1));
      return b != nullptr ? reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(b) & ~(uintptr_t)3) : dummy;
    }
    if (e < 512 && sim.y != nullptr) return reinterpret_cast<const int*>(sim.y) + 2 * min(m0 + e - 256, sim.M - 1);
    return dummy;
  }
  __device__ __forceinline__ const void* meta_src(int m0, int n0, int e) const { return base_src(m0, n0, e); }
  __device__ __forceinline__ void meta_fix(const Tile8& t) const {
    if (t.tid < 256) {  // row entry tid
      const int e = t.tid;
      const int rawy = g8_lds_read(t.meta + 256 + e);
      const int yi = (sim.y != nullptr && t.m0 + e < sim.M) ? rawy + (int)sim.y_offset : -1;
      g8_lds_write(t.meta + 256 + e, yi);
      const bool hit = yi >= t.n0 && yi < t.n0 + t.ncol;
      const unsigned long long any = __ballot(hit);
      if (t.lane == 0) g8_lds_write(t.meta + 768 + (t.tid >> 6), any != 0ull ? 1 : 0);
    }
    if (t.tid >= t.nthr - t.ncol) {  // column entry: the last ncol threads (512 threads: a different half than the rows)
      const int e = t.tid - (t.nthr - t.ncol), n = t.n0 + e;
      const int raw = g8_lds_read(t.meta + e);
      const uint8_t* b = mask_byte(min(n, sim.N - 1));
      int flag = n >= sim.N ? 1 : 0;
      if (b != nullptr) flag |= ((raw >> ((reinterpret_cast<uintptr_t>(b) & 3) * 8)) & 0xff) != 0 ? 1 : 0;
      if (sim.packed != nullptr) flag |= (min(n, sim.N - 1) % sim.p_rows_c) >= sim.p_n_ctx ? 1 : 0;
      g8_lds_write(t.meta + e, flag);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g8_bar();
  }
  __device__ __forceinline__ bool tile_has_gold(const Tile8& t) const {
    const g8_i32x4 f = g8_lds_read4(t.meta + 768);
    return (f[0] | f[1] | f[2] | f[3]) != 0;
  }
  // tile column of the lane's run q (0..3) of fragment column block b: 4 consecutive columns start here
  __device__ __forceinline__ static int run_col(const Tile8& t, int b, int q) { return t.wn * 64 + b * 32 + q * 8 + (t.lane >> 5) * 4; }
  // madd[b*4 + q][j]: 0, or -inf where that column is masked
  __device__ __forceinline__ void col_madd(const Tile8& t, float (&madd)[8][4]) const {
    // the eight runs of the lane sit at run_col(t, 0, 0) + b * 32 + q * 8 words: ONE address, eight reads in flight, one wait (round 6:
    // eight read + wait pairs were eight dependent trips to the LDS, ~1 k cycles per tile, in front of an epilogue nothing overlaps)
    g8_i32x4 f[8];
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:96\n\t"
        "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
        : "v"(g8_lds_addr(t.meta + run_col(t, 0, 0)))
        : "memory");
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) madd[c][j] = f[c][j] != 0 ? -INFINITY : 0.f;
  }
};

typedef float g8_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void g8_lds_write4f(void* p, g8_f32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(g8_lds_addr(p)), "v"(v) : "memory");
}

// fp32 logits (sim_score with a caller buffer): S = acc / T, masked columns -inf   (dpr_task.py:104,211).
// The accumulator layout gives a lane 16-byte runs of 32 different rows: stored as they are, every store instruction touches 64
// lines for 16 bytes each.  Eight rows at a time go through a per-wave LDS patch instead (the 16 lanes that hold them write, all
// 64 read back row-major): every store instruction then writes four whole 256-byte rows of the wave's 64 columns.
struct Epi8Store : Epi8Base {
  float* S;
  __device__ __forceinline__ void finish(G8Acc& acc, const Tile8& t) const {
    meta_fix(t);
    constexpr int TS = 64 + 4;
    const int i = t.lane & 31, h = t.lane >> 5;
    float madd[8][4];
    col_madd(t, madd);
    float* const patch = t.scratch + (t.wm * 4 + t.wn) * (8 * TS);
    const int rr = t.lane >> 4, cq = t.lane & 15;
    const int n = t.n0 + t.wn * 64 + cq * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int sgrp = 0; sgrp < 4; ++sgrp) {
        if ((i >> 3) == sgrp) {
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              g8_f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = acc.v[a][b][q * 4 + j] * sim.inv_T + madd[b * 4 + q][j];
              g8_lds_write4f(patch + (i & 7) * TS + b * 32 + q * 8 + h * 4, v);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = rr + 4 * it;
          const g8_i32x4 w = g8_lds_read4(patch + row * TS + cq * 4);
          const int m = t.m0 + t.wm * 128 + a * 32 + sgrp * 8 + row;
          if (m < sim.M && n < sim.N) {  // N % 8 == 0: a run of four columns is inside or outside as a whole
            float4 o;
            o.x = __int_as_float(w[0]); o.y = __int_as_float(w[1]); o.z = __int_as_float(w[2]); o.w = __int_as_float(w[3]);
            *reinterpret_cast<float4*>(S + (size_t)m * sim.N + n) = o;
          }
        }
      }
    }
  }
};

constexpr float kG8Log2e = 1.4426950408889634f, kG8Ln2 = 0.6931471805599453f;

// Training forward WITHOUT the logits (dpr_task.py:211-212): per (row, 64-column wave strip) the maximum and sum exp(S - max)
// go to part_m / part_s [M][npart] (npart = column tiles x wave strips per tile), the gold logit to sim.gold.  The logits themselves are never
// stored: 8192 x 65536 of them are 2 GiB each way; the backward recomputes them tile by tile (Epi8G).
struct Epi8Stats : Epi8Base {
  float* part_m;
  float* part_s;
  int npart;
  __device__ __forceinline__ void finish(G8Acc& acc, const Tile8& t) const {
    meta_fix(t);
    const int i = t.lane & 31, h = t.lane >> 5;
    float madd[8][4];
    col_madd(t, madd);
    const bool anygold = sim.y != nullptr && tile_has_gold(t);
    const float s2 = sim.inv_T * kG8Log2e;
    const f32x2 s2v = {s2, s2};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x2 v[8][2];  // log2 units
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x2 x = {acc.v[a][b][q * 4 + u * 2], acc.v[a][b][q * 4 + u * 2 + 1]};
            const f32x2 md = {madd[b * 4 + q][u * 2], madd[b * 4 + q][u * 2 + 1]};
            v[b * 4 + q][u] = x * s2v + md;
            mx = fmaxf(mx, fmaxf(v[b * 4 + q][u][0], v[b * 4 + q][u][1]));
          }
      mx = g8_max_x32(mx);
      const float mref = mx == -INFINITY ? 0.f : mx;
      const f32x2 mr = {mref, mref};
      f32x2 sm2 = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x2 d = v[c][u] - mr;
          const f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};
          sm2 += e;
        }
      const float sm = g8_sum_x32(sm2[0] + sm2[1]);
      const int m = t.m0 + t.wm * 128 + a * 32 + i;
      if (h == 0 && m < sim.M) {
        const size_t at = (size_t)m * npart + t.bx * (t.ncol >> 6) + t.wn;
        part_m[at] = mx * kG8Ln2;
        part_s[at] = sm;
      }
      if (anygold) {
        // the lane holds the gold column of its row iff rel = b*32 + q*8 + j with j < 4 (bit 2 clear): one select chain, one store
        const int yi = g8_lds_read(t.meta + 256 + t.wm * 128 + a * 32 + i);
        const int rel = yi - (t.n0 + t.wn * 64 + h * 4);
        const int idx = ((rel >> 5) << 4) | (((rel >> 3) & 3) << 2) | (rel & 3);  // register index b*16 + q*4 + j
        float gv = 0.f, gm = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool hit = idx == b * 16 + r;
            gv = hit ? acc.v[a][b][r] : gv;
            gm = hit ? madd[b * 4 + (r >> 2)][r & 3] : gm;
          }
        if (rel >= 0 && rel < 64 && (rel & 4) == 0 && m < sim.M) sim.gold[m] = gm != 0.f ? -INFINITY : gv * sim.inv_T;
      }
    }
  }
};

// Training forward WITH the dScores wanted, in ONE pass of the GEMM (round 6): Epi8Stats' strip statistics, and the tile's own
// softmax numerators P = exp(S - strip max) leave the tile as fp16 (2 bytes per score, into the buffer that will hold G).  The row
// kernel g8_lse_p2g_kernel then derives the row logsumexp from the strip statistics and rescales the row IN PLACE:
//     G_ij = (P_ij * exp(strip max - lse_i) - onehot) * grad_scale  -> bf16.
// One GEMM pass + one streaming pass (4 bytes per score) instead of two GEMM passes (Epi8Stats, then Epi8G recomputing every logit):
// 8192^2 x 768: ~115 + ~50 us against 101 + 111.  fp16, not bf16, and scaled by 2^14: every strip holds a 1.0 (its maximum), so the
// numerators live in (0, 2^14] -- 11 significant bits down to 2^-14 * 2^14 = 1 ... i.e. relative 2^-12 down to 6e-5 * 2^-14 = 3.7e-9 of
// the strip maximum, absolute 2^-38 below that: the ONE rounding that matters stays G's own bf16 rounding, as in Epi8G.
constexpr float kG8PScale = 16384.f, kG8PInvScale = 1.f / 16384.f;
typedef _Float16 g8_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned g8_cvt_pk_f16(float a, float b) {  // one v_cvt_pk_f16_f32 (round to nearest even)
  g8_h2 v;
  v[0] = (_Float16)a;
  v[1] = (_Float16)b;
  return __builtin_bit_cast(unsigned, v);
}
typedef unsigned g8_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void g8_lds_write2(void* p, unsigned a, unsigned b) {
  const g8_u32x2 v = {a, b};
  asm volatile("ds_write_b64 %0, %1" ::"v"(g8_lds_addr(p)), "v"(v) : "memory");
}
// STAGED: the numerators leave through the wave's LDS patch, half a strip of sixteen rows at a time -- every store instruction writes
// 64-byte pieces of 16 rows instead of 32-byte pieces of 32 rows; else straight from the registers.
template <bool STAGED>
struct Epi8StatsPT : Epi8Stats {
  uint16_t* P;  // [M][N] fp16 bit patterns
  __device__ __forceinline__ void finish(G8Acc& acc, const Tile8& t) const {
    meta_fix(t);
    const int i = t.lane & 31, h = t.lane >> 5;
    float madd[8][4];
    col_madd(t, madd);
    const bool anygold = sim.y != nullptr && tile_has_gold(t);
    const float s2 = sim.inv_T * kG8Log2e;
    const f32x2 s2v = {s2, s2};
    const f32x2 psc = {kG8PScale, kG8PScale};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x2 v[8][2];  // log2 units
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x2 x = {acc.v[a][b][q * 4 + u * 2], acc.v[a][b][q * 4 + u * 2 + 1]};
            const f32x2 md = {madd[b * 4 + q][u * 2], madd[b * 4 + q][u * 2 + 1]};
            v[b * 4 + q][u] = x * s2v + md;
            mx = fmaxf(mx, fmaxf(v[b * 4 + q][u][0], v[b * 4 + q][u][1]));
          }
      mx = g8_max_x32(mx);
      const float mref = mx == -INFINITY ? 0.f : mx;
      const f32x2 mr = {mref, mref};
      f32x2 sm2 = {0.f, 0.f};
      const int m = t.m0 + t.wm * 128 + a * 32 + i;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        unsigned pk[4][2];  // run q: two dwords = 4 fp16 = columns b*32 + q*8 + h*4 ..
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x2 d = v[b * 4 + q][u] - mr;
            const f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};  // exp2(-inf) == 0 at masked columns
            sm2 += e;  // (same values, same order as Epi8Stats: the statistics of the two plans are bit-identical)
            const f32x2 es = e * psc;
            pk[q][u] = g8_cvt_pk_f16(es[0], es[1]);
          }
        if constexpr (STAGED) {
          // half a strip (32 columns = 64 bytes of a row) of sixteen rows at a time through the wave's LDS patch
          constexpr int TSH = 40;  // halfs per patch row (80 bytes: 16-byte aligned reads; the 32 writers of a pass land on distinct banks)
          uint16_t* const patch = reinterpret_cast<uint16_t*>(t.scratch + (t.wm * 4 + t.wn) * (8 * 68));
          const int rr = t.lane >> 2, c4 = t.lane & 3;
          const int n = t.n0 + t.wn * 64 + b * 32 + c4 * 8;
#pragma unroll
          for (int grp = 0; grp < 2; ++grp) {
            if ((i >> 4) == grp) {
#pragma unroll
              for (int q = 0; q < 4; ++q) g8_lds_write2(patch + (i & 15) * TSH + q * 8 + h * 4, pk[q][0], pk[q][1]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const g8_i32x4 w = g8_lds_read4(patch + rr * TSH + c4 * 8);
            const int mm = t.m0 + t.wm * 128 + a * 32 + grp * 16 + rr;
            if (mm < sim.M && n < sim.N)
              *reinterpret_cast<uint4*>(P + (size_t)mm * sim.N + n) = make_uint4((unsigned)w[0], (unsigned)w[1], (unsigned)w[2], (unsigned)w[3]);
          }
        } else {
          // lanes i and i + 32 trade half-runs: every lane stores whole 16-byte runs (Epi8G, guide T21)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const auto w0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
            const auto w1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
            const int n = t.n0 + t.wn * 64 + b * 32 + (2 * pr + h) * 8;
            if (m < sim.M && n < sim.N) *reinterpret_cast<uint4*>(P + (size_t)m * sim.N + n) = make_uint4(w0[0], w1[0], w0[1], w1[1]);
          }
        }
      }
      const float sm = g8_sum_x32(sm2[0] + sm2[1]);
      if (h == 0 && m < sim.M) {
        const size_t at = (size_t)m * npart + t.bx * (t.ncol >> 6) + t.wn;
        part_m[at] = mx * kG8Ln2;
        part_s[at] = sm;
      }
      if (anygold) {
        const int yi = g8_lds_read(t.meta + 256 + t.wm * 128 + a * 32 + i);
        const int rel = yi - (t.n0 + t.wn * 64 + h * 4);
        const int idx = ((rel >> 5) << 4) | (((rel >> 3) & 3) << 2) | (rel & 3);  // register index b*16 + q*4 + j
        float gv = 0.f, gm = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool hit = idx == b * 16 + r;
            gv = hit ? acc.v[a][b][r] : gv;
            gm = hit ? madd[b * 4 + (r >> 2)][r & 3] : gm;
          }
        if (rel >= 0 && rel < 64 && (rel & 4) == 0 && m < sim.M) sim.gold[m] = gm != 0.f ? -INFINITY : gv * sim.inv_T;
      }
    }
  }
};
using Epi8StatsP = Epi8StatsPT<false>;
using Epi8StatsPS = Epi8StatsPT<true>;

// Backward, first half (autograd of dpr_task.py:211-212 into the scores): the logits are recomputed (same GEMM, bit-identical
// accumulators) and leave the tile as G = (softmax - onehot) * grad_scale in bf16, 2 bytes per score instead of the 4 + 4 + 2 of
// store / re-read / G.  row_lse: natural-log logsumexp of every row (g8_lse_kernel).
struct Epi8G : Epi8Base {
  const float* row_lse;  // [M]
  uint16_t* G;           // [M][N] bf16
  float grad_scale;
  __device__ __forceinline__ const void* meta_src(int m0, int n0, int e) const {
    if (e >= 512 && e < 768) return row_lse + min(m0 + e - 512, sim.M - 1);
    return base_src(m0, n0, e);
  }
  __device__ __forceinline__ void finish(G8Acc& acc, const Tile8& t) const {
    meta_fix(t);
    const int i = t.lane & 31, h = t.lane >> 5;
    float madd[8][4];
    col_madd(t, madd);
    const bool anygold = sim.y != nullptr && tile_has_gold(t);
    const float s2 = sim.inv_T * kG8Log2e;
    const f32x2 s2v = {s2, s2};
    const f32x2 gs = {grad_scale, grad_scale};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int lrow = t.wm * 128 + a * 32 + i;
      const int m = t.m0 + lrow;
      const float lse2 = __int_as_float(g8_lds_read(t.meta + 512 + lrow)) * kG8Log2e;
      int rel = -1;
      if (anygold) rel = g8_lds_read(t.meta + 256 + lrow) - (t.n0 + t.wn * 64 + h * 4);
      const f32x2 l2 = {lse2, lse2};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        unsigned pk[4][2];  // run q: two dwords = 4 bf16 = columns q*8 + h*4 ..
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x2 x = {acc.v[a][b][q * 4 + u * 2], acc.v[a][b][q * 4 + u * 2 + 1]};
            const f32x2 md = {madd[b * 4 + q][u * 2], madd[b * 4 + q][u * 2 + 1]};
            const f32x2 d = (x * s2v + md) - l2;
            f32x2 e = {__builtin_amdgcn_exp2f(d[0]), __builtin_amdgcn_exp2f(d[1])};  // exp2(-inf) == 0 at masked columns
            if (anygold) {
              if (rel == b * 32 + q * 8 + u * 2) e[0] -= 1.0f;
              if (rel == b * 32 + q * 8 + u * 2 + 1) e[1] -= 1.0f;
            }
            const f32x2 g = e * gs;
            pk[q][u] = cvt_pk_bf16(g[0], g[1]);
          }
        // Lanes i and i + 32 hold the two halves of every 8-column run.  One v_permlane32_swap per dword pairs run 2p (kept by the
        // lower lane) with run 2p + 1 (kept by the upper lane): every lane then owns one WHOLE run = 16 contiguous bytes, and the
        // two lanes of a row write 32 contiguous bytes -- half the store instructions, twice the bytes per request (guide T21).
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const auto w0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
          const auto w1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
          const int n = t.n0 + t.wn * 64 + b * 32 + (2 * pr + h) * 8;
          if (m < sim.M && n < sim.N) *reinterpret_cast<uint4*>(G + (size_t)m * sim.N + n) = make_uint4(w0[0], w1[0], w0[1], w1[1]);
        }
      }
    }
  }
};

// Validation rank metrics without the score matrix (dpr_task.py:235-246): rank = 1 + #{S > gold} + #{S == gold, column < gold
// column} -- the position of the gold column in the reference's stable descending sort.  gold_val[m] must be the SAME number
// this GEMM produces for (m, y[m]): g8_gold_kernel runs the identical MFMA sequence on the gathered context rows.
struct Epi8Count : Epi8Base {
  const float* gold_val;  // [M]
  int* count;             // [M], zeroed by the caller; += per wave strip
  __device__ __forceinline__ const void* meta_src(int m0, int n0, int e) const {
    if (e >= 512 && e < 768) return gold_val + min(m0 + e - 512, sim.M - 1);
    return base_src(m0, n0, e);
  }
  __device__ __forceinline__ void finish(G8Acc& acc, const Tile8& t) const {
    meta_fix(t);
    const int i = t.lane & 31, h = t.lane >> 5;
    float madd[8][4];
    col_madd(t, madd);  // columns outside the matrix are flagged too: -inf there, never counted (they lie behind every gold column)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int lrow = t.wm * 128 + a * 32 + i;
      const int m = t.m0 + lrow;
      const float gv = __int_as_float(g8_lds_read(t.meta + 512 + lrow));
      int c = 0;
      bool tie = false;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // fma(acc, 1/T, 0) is acc / T as the logit store rounds it; fma(acc, 1/T, -inf) is the masked column's -inf
            const float v = fmaf(acc.v[a][b][q * 4 + j], sim.inv_T, madd[b * 4 + q][j]);
            c += v > gv ? 1 : 0;
            tie |= v == gv;
          }
      if (__ballot(tie) != 0ull) {  // exact ties (always: the tile that holds the gold column itself): lower column index first
        const int ycol = g8_lds_read(t.meta + 256 + lrow);
        const int nbase = t.n0 + t.wn * 64 + h * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float v = fmaf(acc.v[a][b][q * 4 + j], sim.inv_T, madd[b * 4 + q][j]);
              c += (v == gv && nbase + b * 32 + q * 8 + j < ycol) ? 1 : 0;
            }
      }
      c = g8_isum_x32(c);
      if (h == 0 && m < sim.M && c != 0) atomicAdd(count + m, c);
    }
  }
};

// Validation in ONE pass over the scores (dpr_task.py:224-227 per batch, :296-299 per epoch: compute_rank_metrics AND self.loss on the
// same score matrix): the count of Epi8Count and the strip statistics of Epi8Stats from the same accumulators -- the Nq x Nc GEMM runs
// once instead of twice (round 3: 1.80 ms for the two passes at 8192 x 65536 against 1.62 ms through a stored 2 GiB score matrix).  The
// gold logit both need is gold_val (g8_gold_kernel: bit-identical to this GEMM's element), so neither half looks for it in the tile.
struct Epi8CountStats : Epi8Base {
  const float* gold_val;  // [M]
  int* count;             // [M], zeroed by the caller
  float* part_m;
  float* part_s;
  int npart;
  __device__ __forceinline__ const void* meta_src(int m0, int n0, int e) const {
    if (e >= 512 && e < 768) return gold_val + min(m0 + e - 512, sim.M - 1);
    return base_src(m0, n0, e);
  }
  __device__ __forceinline__ void finish(G8Acc& acc, const Tile8& t) const {
    meta_fix(t);
    const int i = t.lane & 31, h = t.lane >> 5;
    float madd[8][4];
    col_madd(t, madd);
    const float s2 = sim.inv_T * kG8Log2e;
    const f32x2 s2v = {s2, s2};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int lrow = t.wm * 128 + a * 32 + i;
      const int m = t.m0 + lrow;
      const float gv = __int_as_float(g8_lds_read(t.meta + 512 + lrow));
      // ---- rank: exactly Epi8Count's comparisons (the logit as the store would round it)
      int c = 0;
      bool tie = false;
      f32x2 v[8][2];  // log2 units, for the statistics
      float mx = -INFINITY;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float sv = fmaf(acc.v[a][b][q * 4 + j], sim.inv_T, madd[b * 4 + q][j]);
            c += sv > gv ? 1 : 0;
            tie |= sv == gv;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x2 x = {acc.v[a][b][q * 4 + u * 2], acc.v[a][b][q * 4 + u * 2 + 1]};
            const f32x2 md = {madd[b * 4 + q][u * 2], madd[b * 4 + q][u * 2 + 1]};
            v[b * 4 + q][u] = x * s2v + md;
            mx = fmaxf(mx, fmaxf(v[b * 4 + q][u][0], v[b * 4 + q][u][1]));
          }
        }
      if (__ballot(tie) != 0ull) {
        const int ycol = g8_lds_read(t.meta + 256 + lrow);
        const int nbase = t.n0 + t.wn * 64 + h * 4;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float sv = fmaf(acc.v[a][b][q * 4 + j], sim.inv_T, madd[b * 4 + q][j]);
              c += (sv == gv && nbase + b * 32 + q * 8 + j < ycol) ? 1 : 0;
            }
      }
      c = g8_isum_x32(c);
      if (h == 0 && m < sim.M && c != 0) atomicAdd(count + m, c);
      // ---- loss: exactly Epi8Stats' strip statistics
      mx = g8_max_x32(mx);
      const float mref = mx == -INFINITY ? 0.f : mx;
      const f32x2 mr = {mref, mref};
      f32x2 sm2 = {0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 8; ++cc)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x2 dd = v[cc][u] - mr;
          const f32x2 e = {__builtin_amdgcn_exp2f(dd[0]), __builtin_amdgcn_exp2f(dd[1])};
          sm2 += e;
        }
      const float sm = g8_sum_x32(sm2[0] + sm2[1]);
      if (h == 0 && m < sim.M) {
        const size_t at = (size_t)m * npart + t.bx * (t.ncol >> 6) + t.wn;
        part_m[at] = mx * kG8Ln2;
        part_s[at] = sm;
      }
    }
  }
};

// Retrieval epilogue (run_retrieval_pytorch.py:149-150 without the score matrix; EpiFilter of gemm_bf16.h on this kernel): a score
// only leaves the tile when it ranks ahead of the row's current k-th best (score desc, passage id asc); such scores are appended to
// the row's candidate list, which the top-k merge kernel folds into the state.  The thresholds arrive with the tile's input words.
struct Epi8Filter {
  const float* kth_val;     // state values  [M][k]
  const int64_t* kth_idx;   // state ids     [M][k]  (-1: slot unfilled)
  int k;
  int M, N;
  long long col_offset;     // passage id of column 0
  int* cnt;                 // [M] candidates appended so far (the merge kernel resets it)
  float* cand_v;            // [M][N]
  int* cand_j;              // [M][N]
  const void* dummy;
  int j_bias = 0;           // added to the column written to cand_j: a chunk appended to the candidate lists of the chunk(s) before
                            // it (dprhot_search merges groups of warm chunks once; the merge sees columns relative to the group's start)
  __device__ __forceinline__ const void* meta_src(int m0, int n0, int e) const {
    if (e >= 512 && e < 768) return kth_val + (size_t)min(m0 + e - 512, M - 1) * k + (k - 1);
    return dummy;
  }
  __device__ __forceinline__ void finish(G8Acc& acc, const Tile8& t) const {
    const int i = t.lane & 31, h = t.lane >> 5;
    // Screen by the maximum of each 16-value run (one row, 16 of the tile's columns): v_max3 chains instead of a compare per value.
    // A warm chunk still leaves a wave tile a few qualifiers (~12.5 * k / 100 / chunk index per 128 x 64 scores), so the walk below
    // is guarded per run as well: only runs that hold one are walked.
    float tv[4], mx[4][2];
    bool any = false;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      tv[a] = __int_as_float(g8_lds_read(t.meta + 512 + t.wm * 128 + a * 32 + i));
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float m3 = fmaxf(fmaxf(acc.v[a][b][0], acc.v[a][b][1]), acc.v[a][b][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) m3 = fmaxf(fmaxf(m3, acc.v[a][b][r]), acc.v[a][b][r + 1]);
        mx[a][b] = fmaxf(m3, acc.v[a][b][15]);
        any |= mx[a][b] >= tv[a];
      }
    }
    if (__ballot(any) == 0ull) return;  // wave-uniform: the lanes of a row trade counts below
    // Append path.  Round 2 took one RETURNING global atomic per candidate inside a divergent walk -- serialised round trips to L2:
    // a k = 1000 search leaves ~60 candidates per 128 x 64 wave tile in its second chunk and that chunk's GEMM took 471 us instead
    // of 98.  Now per 32-row block: every lane counts its qualifiers (a bit per value: no memory), the two lanes of a row add their
    // counts with one v_permlane32_swap, ONE atomic per row reserves the run (all rows of the block in the same instruction), and
    // the values are written behind each other from a running offset.
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int m = t.m0 + t.wm * 128 + a * 32 + i;
      unsigned mask = 0u;  // bit b * 16 + r
      if (m < M) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (!(mx[a][b] >= tv[a])) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = t.n0 + t.wn * 64 + b * 32 + (r >> 2) * 8 + h * 4 + (r & 3);
            const float v = acc.v[a][b][r];
            if (!(n < N && v >= tv[a])) continue;
            bool take = v > tv[a];
            if (!take) {  // exact tie with the k-th best -> lower passage id wins (-1: slot unfilled)
              const long long ti = kth_idx[(size_t)m * k + k - 1];
              take = ti < 0 || col_offset + n < ti;
            }
            if (take) mask |= 1u << (b * 16 + r);
          }
        }
      }
      const int mine = __builtin_popcount(mask);
      const auto cc = __builtin_amdgcn_permlane32_swap((unsigned)mine, (unsigned)mine, false, false);  // (count of lane i, of lane i + 32)
      const int total = (int)(cc[0] + cc[1]);
      int base = 0;
      if (h == 0 && total > 0) base = atomicAdd(&cnt[m], total);
      const auto bb = __builtin_amdgcn_permlane32_swap((unsigned)base, (unsigned)base, false, false);
      int pos = (int)bb[0] + (h != 0 ? (int)cc[0] : 0);
      if (mask == 0u) continue;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (((mask >> (b * 16)) & 0xffffu) == 0u) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if ((mask >> (b * 16 + r)) & 1u) {
            cand_v[(size_t)m * N + pos] = acc.v[a][b][r];
            cand_j[(size_t)m * N + pos] = j_bias + t.n0 + t.wn * 64 + b * 32 + (r >> 2) * 8 + h * 4 + (r & 3);
            ++pos;
          }
        }
      }
    }
  }
};

// ---- the small kernels around the GEMM passes -------------------------------------------------------------------------------------
// Row logsumexp from the strip statistics of Epi8Stats, row loss = lse - gold (dpr_task.py:212, CrossEntropyLoss per row).
// One wave per row; the loss sum is a second tiny launch (reduce_sum_kernel, fixed order).  (Summing inside this kernel was tried
// both ways: a last-arriving workgroup that re-reads the row losses needs an agent-scope fence per workgroup -- 146 us instead of
// 6 + 6; one fixed-point ticket atomic per workgroup (loss_ticket_add) serialises 2048 atomics on one address -- 28 us.)
__global__ __launch_bounds__(256) void g8_lse_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s, int npart,
                                                     const float* __restrict__ gold, int M, float* __restrict__ lse_ws,
                                                     float* __restrict__ row_lse, float* __restrict__ row_loss, float* __restrict__ loss_ws) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* pm = part_m + (size_t)row * npart;
  const float* ps = part_s + (size_t)row * npart;
  float mx = -INFINITY;
  for (int k = lane; k < npart; k += 64) mx = fmaxf(mx, pm[k]);
  mx = wave_max(mx);
  float sm = 0.f;
  if (mx != -INFINITY)
    for (int k = lane; k < npart; k += 64) {
      const float m = pm[k];
      if (m != -INFINITY) sm += ps[k] * __expf(m - mx);
    }
  sm = wave_sum(sm);
  if (lane == 0) {
    const float lse = mx == -INFINITY ? -INFINITY : mx + logf(sm);
    const float loss = lse - gold[row];
    lse_ws[row] = lse;
    loss_ws[row] = loss;
    if (row_lse != nullptr) row_lse[row] = lse;
    if (row_loss != nullptr) row_loss[row] = loss;
  }
}

// Second half of the one-pass forward (Epi8StatsP): one workgroup per row.  Wave 0 derives the row logsumexp EXACTLY as g8_lse_kernel
// does (same lanes, same order: the loss of the two plans is bit-identical); then the row's fp16 numerators become the bf16 dScores in
// place, 16 bytes per thread and step: G = (P * 2^-14 * exp(strip max - lse) - onehot) * grad_scale.  The gold column gets its -1 whether
// masked or not, as in Epi8G; a row with no unmasked column (lse = -inf) has P == 0 everywhere and gets factor 0, not NaN.
typedef _Float16 g8_h8 __attribute__((ext_vector_type(8)));
constexpr int kP2gMaxPart = 1024, kP2gPre = 4;
__global__ __launch_bounds__(256) void g8_lse_p2g_kernel(const float* __restrict__ part_m, const float* __restrict__ part_s, int npart,
                                                         const float* __restrict__ gold, int M, int N, const int64_t* __restrict__ y,
                                                         int64_t y_offset, float grad_scale, float* __restrict__ lse_ws, float* __restrict__ row_lse,
                                                         float* __restrict__ row_loss, float* __restrict__ loss_ws, uint16_t* __restrict__ PG) {
  __shared__ float bc;
  __shared__ float s_pm[kP2gMaxPart], s_ps[kP2gMaxPart];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float* pm = part_m + (size_t)row * npart;
  const float* ps = part_s + (size_t)row * npart;
  uint16_t* prow = PG + (size_t)row * N;
  // Every read this workgroup can issue without knowing the logsumexp goes out first (the launch at 1024 x 8192 is three dependent
  // trips to memory otherwise -- strip maxima, strip sums, then the numerators: 8.2 us for 32 MB): the first kP2gPre 16-byte groups of
  // each thread, and the strip statistics into LDS.
  uint4 raw[kP2gPre];
#pragma unroll
  for (int u = 0; u < kP2gPre; ++u) {
    const int c8 = tid + u * 256;
    if (c8 * 8 < N) raw[u] = *reinterpret_cast<const uint4*>(prow + c8 * 8);
  }
  const int yc = (int)(y[row] + y_offset);
  const bool staged = npart <= kP2gMaxPart;
  if (staged)
    for (int k = tid; k < npart; k += 256) {
      s_pm[k] = pm[k];
      s_ps[k] = ps[k];
    }
  __syncthreads();
  if (tid < 64) {  // (the sums in g8_lse_kernel's order: the logsumexp of the two forward plans is bit-identical)
    float mx = -INFINITY;
    for (int k = lane; k < npart; k += 64) mx = fmaxf(mx, staged ? s_pm[k] : pm[k]);
    mx = wave_max(mx);
    float sm = 0.f;
    if (mx != -INFINITY)
      for (int k = lane; k < npart; k += 64) {
        const float m = staged ? s_pm[k] : pm[k];
        if (m != -INFINITY) sm += (staged ? s_ps[k] : ps[k]) * __expf(m - mx);
      }
    sm = wave_sum(sm);
    if (lane == 0) {
      const float lse = mx == -INFINITY ? -INFINITY : mx + logf(sm);
      const float loss = lse - gold[row];
      lse_ws[row] = lse;
      loss_ws[row] = loss;
      if (row_lse != nullptr) row_lse[row] = lse;
      if (row_loss != nullptr) row_loss[row] = loss;
      bc = lse;
    }
  }
  __syncthreads();
  const float lse = bc;
  const float fs = grad_scale * kG8PInvScale;
  auto rescale = [&](const uint4 in, int col) {
    const float pmv = staged ? s_pm[col >> 6] : pm[col >> 6];
    const float f = lse == -INFINITY ? 0.f : __expf(pmv - lse) * fs;
    const g8_h8 hv = __builtin_bit_cast(g8_h8, in);
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      g[j] = (float)hv[j] * f;
      if (col + j == yc) g[j] -= grad_scale;
    }
    *reinterpret_cast<uint4*>(prow + col) = make_uint4(cvt_pk_bf16(g[0], g[1]), cvt_pk_bf16(g[2], g[3]), cvt_pk_bf16(g[4], g[5]), cvt_pk_bf16(g[6], g[7]));
  };
#pragma unroll
  for (int u = 0; u < kP2gPre; ++u) {
    const int c8 = tid + u * 256;
    if (c8 * 8 < N) rescale(raw[u], c8 * 8);
  }
  for (int c8 = tid + kP2gPre * 256; c8 * 8 < N; c8 += 256) rescale(*reinterpret_cast<const uint4*>(prow + c8 * 8), c8 * 8);
}

// The logit of every row's gold column, bit-identical to what gemm8p_kernel accumulates for that element: one wave per 32 rows runs
// the same v_mfma_f32_32x32x16_bf16 chain (k slices of 16 in increasing order, from zero) on the rows of Q against the GATHERED rows
// C[y[m]] and keeps the diagonal.  Lane (i, h) holds D[n' = 8*(r >> 2) + 4*h + (r & 3)][m' = i]: the diagonal element of row i is
// register ((i >> 3) << 2) | (i & 3) of the lane with h == (i >> 2) & 1.
__global__ __launch_bounds__(64) void g8_gold_kernel(const uint16_t* __restrict__ Q, const uint16_t* __restrict__ C, int M, int N, int K,
                                                     const int64_t* __restrict__ y, int64_t y_offset, EpiSim mask_src, float inv_T,
                                                     float* __restrict__ gold) {
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  const int m = min(blockIdx.x * 32 + i, M - 1);
  const int yc = (int)(y[m] + y_offset);
  const int c = min(max(yc, 0), N - 1);
  const uint16_t* qa = Q + (size_t)m * K + h * 8;
  const uint16_t* cb = C + (size_t)c * K + h * 8;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 16) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(qa + k);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(cb + k);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
  }
  const int want = ((i >> 3) << 2) | (i & 3);
  float v = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) v = want == r ? acc[r] : v;
  if (h == ((i >> 2) & 1) && blockIdx.x * 32 + i < M) {
    const bool masked = yc < 0 || yc >= N || mask_src.mask_at(c) != 0;
    gold[m] = masked ? -INFINITY : v * inv_T;
  }
}

// rank[m] = 1 + count[m]
__global__ void g8_rank_finish_kernel(const int* __restrict__ count, int M, int64_t* __restrict__ rank) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m < M) rank[m] = 1 + (int64_t)count[m];
}

}  // namespace dprhot
