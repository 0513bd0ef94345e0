/* TEST INFRASTRUCTURE ONLY -- plain C restatement of dpr-scale's in-batch contrastive step (CPU).
 *
 * Same role and same rules as oracle/inbatch_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so, and only as the checker / the timed CPU baseline ("port").  It is
 * pinned to the reference through tests/golden/*.npz (tests/test_oracle_c.py).
 *
 * Restates, per rank, in fp32 storage with double accumulation:
 *   dpr_scale/task/dpr_task.py:98-105   sim_score: S = Q C^T, S[:, mask] = -inf
 *   dpr_scale/task/dpr_task.py:211      S /= T
 *   dpr_scale/task/dpr_task.py:46,212   CrossEntropyLoss (mean over the GLOBAL query count Nq)
 *   autograd of the above               G = (softmax - onehot) * gscale ; dQ = G C ; dC = G^T Q
 *   dpr_scale/task/dpr_task.py:235-246  rank of the gold column (stable, lower index first)
 * Build: gcc -O3 -fopenmp -shared -fPIC oracle/inbatch_oracle.c -o oracle/liboracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* S[B,Nc] = inv_T * Q[B,d] C[Nc,d]^T ; masked columns -> -inf   (dpr_task.py:98-105,211) */
void oracle_sim(const float* Q, int B, const float* C, int Nc, int d, const uint8_t* colmask, float inv_T, float* S) {
#pragma omp parallel for schedule(static)
  for (int j = 0; j < Nc; ++j) {
    const float* c = C + (size_t)j * d;
    for (int i = 0; i < B; ++i) {
      const float* q = Q + (size_t)i * d;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int k = 0;
      for (; k + 8 <= d; k += 8)
        for (int u = 0; u < 8; ++u) acc[u] += q[k + u] * c[k + u];
      double a = 0;
      for (int u = 0; u < 8; ++u) a += acc[u];
      for (; k < d; ++k) a += (double)q[k] * c[k];
      S[(size_t)i * Nc + j] = (colmask && colmask[j]) ? -INFINITY : (float)(a * inv_T);
    }
  }
}

/* row CE (dpr_task.py:212) + dScores: returns sum_i row_loss[i]; G may be NULL */
double oracle_softmax_ce(const float* S, int B, int Nc, const int64_t* y, int64_t y_offset, float gscale, float* row_loss,
                         float* row_lse, float* G) {
  double total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
  for (int i = 0; i < B; ++i) {
    const float* s = S + (size_t)i * Nc;
    double m = -INFINITY, sum = 0;
    for (int j = 0; j < Nc; ++j) m = s[j] > m ? s[j] : m;
    for (int j = 0; j < Nc; ++j) sum += exp((double)s[j] - m);
    const double lse = m + log(sum);
    const int64_t yi = y[i] + y_offset;
    const double l = lse - s[yi];
    if (row_lse) row_lse[i] = (float)lse;
    if (row_loss) row_loss[i] = (float)l;
    total += l;
    if (G) {
      float* g = G + (size_t)i * Nc;
      for (int j = 0; j < Nc; ++j) g[j] = (float)((exp((double)s[j] - lse) - (j == yi ? 1.0 : 0.0)) * gscale);
    }
  }
  return total;
}

/* dQ[B,d] = G[B,Nc] C[Nc,d] */
void oracle_dq(const float* G, const float* C, int B, int Nc, int d, float* dQ) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < B; ++i) {
    float* o = dQ + (size_t)i * d;
    memset(o, 0, sizeof(float) * d);
    for (int j = 0; j < Nc; ++j) {
      const float g = G[(size_t)i * Nc + j];
      if (g == 0.0f) continue;
      const float* c = C + (size_t)j * d;
      for (int k = 0; k < d; ++k) o[k] += g * c[k];
    }
  }
}

/* dC[Nc,d] = G^T[Nc,B] Q[B,d] */
void oracle_dc(const float* G, const float* Q, int B, int Nc, int d, float* dC) {
#pragma omp parallel for schedule(static)
  for (int j = 0; j < Nc; ++j) {
    float* o = dC + (size_t)j * d;
    memset(o, 0, sizeof(float) * d);
    for (int i = 0; i < B; ++i) {
      const float g = G[(size_t)i * Nc + j];
      if (g == 0.0f) continue;
      const float* q = Q + (size_t)i * d;
      for (int k = 0; k < d; ++k) o[k] += g * q[k];
    }
  }
}

/* dpr_task.py:235-246 without the sort */
void oracle_rank_of_gold(const float* S, int rows, int cols, const int64_t* y, int64_t y_offset, int64_t* rank) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < rows; ++i) {
    const float* s = S + (size_t)i * cols;
    const int64_t yi = y[i] + y_offset;
    const float g = s[yi];
    int64_t r = 1;
    for (int j = 0; j < cols; ++j) r += (s[j] > g) || (s[j] == g && j < yi);
    rank[i] = r;
  }
}

/* One rank's whole step (what bench.py times as the CPU baseline): returns sum of row losses. */
double oracle_train_step(const float* Q, int B, const float* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                         const uint8_t* colmask, float inv_T, float gscale, float* S, float* G, float* dQ, float* dC) {
  oracle_sim(Q, B, C, Nc, d, colmask, inv_T, S);
  const double total = oracle_softmax_ce(S, B, Nc, y, y_offset, gscale, NULL, NULL, G);
  oracle_dq(G, C, B, Nc, d, dQ);
  oracle_dc(G, Q, B, Nc, d, dC);
  return total;
}
