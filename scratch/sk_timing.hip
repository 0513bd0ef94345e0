// experiment: per-workgroup wall-clock stamps of the skinny.h kernels at the cfg3-per-rank shape (build with hipcc -O3; not shipped)
#ifndef DPRHOT_TIMING
#define DPRHOT_TIMING 2  // 2: with the in-loop stamps of the dQ units (12 more registers); 1: per-workgroup stamps only
#endif
#include <hip/hip_runtime.h>
__device__ unsigned long long g_dprhot_tm[64];
__device__ unsigned long long g_dprhot_tmb[4 * 4096 * 8];
#include "../dpr_scale_amd/csrc/dprhot.hip"
#include <algorithm>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, K = argc > 2 ? atoi(argv[2]) : 8, d = argc > 3 ? atoi(argv[3]) : 768, W = 8;
  const bool no_g = argc > 4 && atoi(argv[4]) != 0;  // 1: G == NULL -- the plan without the dScores launch (round 4)
  if (argc > 5) dprhot_set_option("sk_dbg", atoi(argv[5]));        // 1: dC units leave at once, 2: dQ units leave at once
  if (argc > 6) dprhot_set_option("sk_dq_slices", atoi(argv[6]));  // context slices of the dQ units
  if (argc > 7) dprhot_set_option("sk_w8", atoi(argv[7]));         // 1: eight waves per workgroup in the fused backward launch
  if (argc > 9) dprhot_set_option("sk_sim_w8", atoi(argv[9]));     // 0: four waves per workgroup in the sim launch
  if (argc > 8) dprhot_set_option("sk_pair", atoi(argv[8]));       // 1: one kind of unit (sk_bwdp_kernel; its stamps print as "dc")
  const int n_ctx = B * K;
  int rows_c; dprhot_packed_rows(n_ctx, d, &rows_c);
  const int Nc = W * rows_c;
  float *q, *c, *dq, *dc; uint16_t *Qb, *Cb, *G, *send; int64_t* y; uint8_t* m; float *loss, *lse, *sum, *go; void* ws; size_t wsb;
  dprhot_workspace_bytes(B, Nc, d, &wsb);
  CK(hipMalloc(&q, (size_t)B * d * 4)); CK(hipMalloc(&c, (size_t)n_ctx * d * 4)); CK(hipMalloc(&Qb, (size_t)B * d * 2));
  CK(hipMalloc(&Cb, (size_t)Nc * d * 2)); CK(hipMalloc(&send, (size_t)rows_c * d * 2)); CK(hipMalloc(&G, (size_t)B * Nc * 2));
  CK(hipMalloc(&dq, (size_t)B * d * 4)); CK(hipMalloc(&dc, (size_t)Nc * d * 4)); CK(hipMalloc(&go, 4));
  CK(hipMalloc(&y, B * 8)); CK(hipMalloc(&m, n_ctx)); CK(hipMalloc(&loss, B * 4)); CK(hipMalloc(&lse, B * 4)); CK(hipMalloc(&sum, 4)); CK(hipMalloc(&ws, wsb));
  std::vector<float> h((size_t)n_ctx * d);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  CK(hipMemcpy(q, h.data(), (size_t)B * d * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(c, h.data(), (size_t)n_ctx * d * 4, hipMemcpyHostToDevice));
  float one = 1.f; CK(hipMemcpy(go, &one, 4, hipMemcpyHostToDevice));
  std::vector<int64_t> hy(B); for (int i = 0; i < B; ++i) hy[i] = i * K; CK(hipMemcpy(y, hy.data(), B * 8, hipMemcpyHostToDevice)); CK(hipMemset(m, 0, n_ctx));
  dprhot_pack_ctx(c, m, n_ctx, d, send, nullptr);
  for (int r = 0; r < W; ++r) CK(hipMemcpy(Cb + (size_t)r * rows_c * d, send, (size_t)rows_c * d * 2, hipMemcpyDeviceToDevice));
  const char* names[4] = {"sim", "dc ", "dq ", "fin"};
  const int nst[4] = {5, 6, 4, 5};
  std::vector<unsigned long long> t(4 * 4096 * 8);
  for (int it = 0; it < 4; ++it) {
    CK(hipMemset(ws, 0, 64));
    hipMemsetAsync(nullptr, 0, 0, nullptr);
    unsigned long long* dptr; CK(hipGetSymbolAddress((void**)&dptr, HIP_SYMBOL(g_dprhot_tmb)));
    CK(hipMemset(dptr, 0, t.size() * 8));
    int rc = dprhot_inbatch_step_packed_f32(q, Cb, Qb, B, W, 3, n_ctx, d, y, 1.f, 1.f / (W * B), 1.f, go, loss, lse, sum, no_g ? nullptr : G, dq, dc, ws, wsb, nullptr);
    if (rc) { printf("rc=%d %s\n", rc, dprhot_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(t.data(), dptr, t.size() * 8, hipMemcpyDeviceToHost));
    if (it < 2) continue;
    { unsigned long long g0 = ~0ull; for (int k = 0; k < 3; ++k) for (int b = 0; b < 4096; ++b) { const unsigned long long v = t[((size_t)k * 4096 + b) * 8]; if (v && k > 0) g0 = std::min(g0, v); }
      unsigned long long f0 = ~0ull, f1 = 0; for (int b = 0; b < 4096; ++b) { const unsigned long long* r = &t[((size_t)3 * 4096 + b) * 8]; if (r[0] && r[4]) { f0 = std::min(f0, r[0]); f1 = std::max(f1, r[4]); } }
      if (f1) printf("it%d backward launch: first unit start -> first fin start %.2f us, -> last fin end %.2f us\n", it, (f0 - g0) * 0.01, (f1 - g0) * 0.01); }
    for (int k = 0; k < (DPRHOT_TIMING >= 2 ? 3 : 4); ++k) {
      unsigned long long t0 = ~0ull, t1 = 0; int nb = 0;
      for (int b = 0; b < 4096; ++b) { const unsigned long long* r = &t[((size_t)k * 4096 + b) * 8]; if (r[0]) { ++nb; t0 = std::min(t0, r[0]); t1 = std::max(t1, r[nst[k] - 1]); } }
      if (nb == 0) { printf("it%d %s: no workgroups\n", it, names[k]); continue; }
      printf("it%d %s: %d workgroups, kernel span %.2f us | ", it, names[k], nb, (t1 - t0) * 0.01);
      // start offsets and per-phase averages (10 ns ticks)
      double startavg = 0, startmax = 0; std::vector<double> ph(nst[k], 0.0);
      for (int b = 0; b < 4096; ++b) { const unsigned long long* r = &t[((size_t)k * 4096 + b) * 8]; if (!r[0]) continue;
        const double so = (r[0] - t0) * 0.01; startavg += so; startmax = std::max(startmax, so);
        for (int i = 1; i < nst[k]; ++i) ph[i] += (r[i] - r[i - 1]) * 0.01; }
      printf("start avg %.2f max %.2f | phases(avg us):", startavg / nb, startmax);
      for (int i = 1; i < nst[k]; ++i) printf(" %.2f", ph[i] / nb);
      double dur = 0; for (int b = 0; b < 4096; ++b) { const unsigned long long* r = &t[((size_t)k * 4096 + b) * 8]; if (r[0]) dur += (r[nst[k] - 1] - r[0]) * 0.01; }
      printf(" | wg duration avg %.2f\n", dur / nb);
      if (it == 3) {  // the slowest workgroups: who are they, when did they start, when did they end
        std::vector<std::pair<double, int>> ends;
        for (int b = 0; b < 4096; ++b) { const unsigned long long* r = &t[((size_t)k * 4096 + b) * 8]; if (r[0]) ends.push_back({(r[nst[k] - 1] - t0) * 0.01, b}); }
        std::sort(ends.begin(), ends.end());
        printf("   end times (us since first start): p50 %.2f p90 %.2f p99 %.2f max %.2f | last 8:", ends[ends.size() / 2].first,
               ends[ends.size() * 9 / 10].first, ends[ends.size() * 99 / 100].first, ends.back().first);
        for (size_t i = ends.size() - 8; i < ends.size(); ++i) { const unsigned long long* r = &t[((size_t)k * 4096 + ends[i].second) * 8];
          printf(" wg%d(start %.2f dur %.2f)", ends[i].second, (r[0] - t0) * 0.01, (r[nst[k] - 1] - r[0]) * 0.01); }
        printf("\n");
      }
    }
  }
  int pair_on = 0; dprhot_get_option("sk_pair", &pair_on);
  if (no_g && pair_on) {  // the one-kind units' tile steps (thread 0 of each unit): accumulated time of the parts of a step
    double acc[7] = {0, 0, 0, 0, 0, 0, 0}, steps = 0; int nb = 0;
    for (int b = 0; b < 4096; ++b) {
      const unsigned long long* r2 = &t[((size_t)2 * 4096 + b) * 8];
      if (!r2[7]) continue;
      ++nb; steps += (double)r2[7];
      for (int i = 0; i < 7; ++i) acc[i] += r2[i] * 0.01;
    }
    if (nb) printf("one-kind units, per tile step (us, avg over %d units x %.1f steps): wait for the operands %.3f | barrier %.3f | stores of the tile before %.3f | scale P + barrier %.3f | refill + products + staging %.3f | barrier %.3f | gold rows %.3f\n",
                   nb, steps / nb, acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps, acc[5] / steps, acc[6] / steps);
  } else
  if (no_g) {  // inside the fused dQ units' loop (thread 0 of each unit): accumulated time of the five parts of a step
    double acc[5] = {0, 0, 0, 0, 0}, steps = 0; int nb = 0;
    for (int b = 0; b < 4096; ++b) {
      const unsigned long long* r2 = &t[((size_t)2 * 4096 + b) * 8];
      const unsigned long long* r3 = &t[((size_t)3 * 4096 + b) * 8];
      if (!r2[0] || !r3[2]) continue;
      ++nb; steps += (double)r3[2];
      acc[0] += r2[4] * 0.01; acc[1] += r2[5] * 0.01; acc[2] += r2[6] * 0.01; acc[3] += r3[0] * 0.01; acc[4] += r3[1] * 0.01;
    }
    if (nb) printf("dq loop, per step (us, avg over %d units x %.1f steps): wait for the slot %.3f | barrier %.3f | refill + fragment reads %.3f | MFMA issue + FMAs %.3f | (step 0: table) %.3f\n",
                   nb, steps / nb, acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps, acc[4] / steps);
  }
  float hs; CK(hipMemcpy(&hs, sum, 4, hipMemcpyDeviceToHost)); printf("loss_sum %.4f\n", hs);
  return 0;
}
