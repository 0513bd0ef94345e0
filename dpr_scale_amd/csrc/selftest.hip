// selftest.hip -- standalone (no Python, no torch) check + timing of every libdprhot entry point against
// a double-precision host computation of the same formulas.  Used on the GPU box:
//   ./selftest            all shapes, default kernel selection
//   ./selftest time       plus hipEvent timings
//   DPRHOT_TILE=2 DPRHOT_NO_TR=1 ./selftest      pin a tile / swap the transpose read for 16-bit gathers
// Also dumps what ds_read_b64_tr_b16 returns for a known LDS image ("trdump").
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/dprhot.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e = (x);                                                               \
    if (e != hipSuccess) {                                                            \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);    \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)
#define OK(x)                                                                   \
  do {                                                                          \
    int rc = (x);                                                               \
    if (rc != 0) {                                                              \
      printf("dprhot error %d (%s) at line %d\n", rc, dprhot_last_error(), __LINE__); \
      exit(3);                                                                  \
    }                                                                           \
  } while (0)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) / 9007199254740992.0;
}
static float nrand() {
  double u1 = urand() + 1e-12, u2 = urand();
  return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}

template <class T>
struct Dev {
  T* p = nullptr;
  size_t n = 0;
  explicit Dev(size_t n_) : n(n_) { CK(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16))); }
  ~Dev() { (void)hipFree(p); }
  void up(const std::vector<T>& h) { CK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  std::vector<T> down() const {
    std::vector<T> h(n);
    CK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
};

__global__ void trdump_kernel(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  typedef short v4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) v4 lv4;
  v4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lv4*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)r[j];
}

static int g_fail = 0;
static void report(const char* what, double err, double tol) {
  const bool ok = err <= tol;
  printf("    %-28s err %.3e (tol %.1e) %s\n", what, err, tol, ok ? "ok" : "FAIL");
  if (!ok) ++g_fail;
}

static void run_case(int B, int Nc, int d, int K, float T, bool ragged, bool timing) {
  printf("case B=%d Nc=%d d=%d T=%g ragged=%d\n", B, Nc, d, T, (int)ragged);
  const float dscale = powf((float)d, -0.25f);
  std::vector<uint16_t> hQ((size_t)B * d), hC((size_t)Nc * d);
  std::vector<float> fQ((size_t)B * d), fC((size_t)Nc * d);
  for (size_t i = 0; i < hQ.size(); ++i) { hQ[i] = f2bf(nrand() * dscale); fQ[i] = bf2f(hQ[i]); }
  for (size_t i = 0; i < hC.size(); ++i) { hC[i] = f2bf(nrand() * dscale); fC[i] = bf2f(hC[i]); }
  std::vector<int64_t> hy(B);
  std::vector<uint8_t> hm(Nc, 0);
  for (int i = 0; i < B; ++i) hy[i] = ((int64_t)i * K + 3 * Nc / 8) % Nc;  // not aligned with the row tiling
  if (ragged)
    for (int j = 0; j < Nc; ++j) hm[j] = urand() < 0.05;
  for (int i = 0; i < B; ++i) hm[hy[i]] = 0;

  // host reference in double
  const double invT = 1.0 / T;
  std::vector<double> rS((size_t)B * Nc), rlse(B), rloss(B);
  for (int i = 0; i < B; ++i)
    for (int j = 0; j < Nc; ++j) {
      double a = 0;
      for (int k = 0; k < d; ++k) a += (double)fQ[(size_t)i * d + k] * fC[(size_t)j * d + k];
      rS[(size_t)i * Nc + j] = hm[j] ? -INFINITY : a * invT;
    }
  const double gscale = 1.0 / (B * 3 * T);  // pretend Nq_global = 3B
  std::vector<double> rG((size_t)B * Nc);
  for (int i = 0; i < B; ++i) {
    double m = -INFINITY, s = 0;
    for (int j = 0; j < Nc; ++j) m = std::max(m, rS[(size_t)i * Nc + j]);
    for (int j = 0; j < Nc; ++j) s += exp(rS[(size_t)i * Nc + j] - m);
    rlse[i] = m + log(s);
    rloss[i] = rlse[i] - rS[(size_t)i * Nc + hy[i]];
    for (int j = 0; j < Nc; ++j) rG[(size_t)i * Nc + j] = (exp(rS[(size_t)i * Nc + j] - rlse[i]) - (j == hy[i])) * gscale;
  }

  Dev<uint16_t> dQ_(hQ.size()), dC_(hC.size()), dG((size_t)B * Nc);
  Dev<float> dS((size_t)B * Nc), dloss(B), dlse(B), dsum(1), ddq((size_t)B * d), ddc((size_t)Nc * d), dgo(1);
  Dev<int64_t> dy(B), drank(B);
  Dev<uint8_t> dm(Nc);
  dQ_.up(hQ); dC_.up(hC); dy.up(hy); dm.up(hm);
  size_t wsb = 0;
  OK(dprhot_workspace_bytes(B, Nc, d, &wsb));
  Dev<char> ws(wsb);
  std::vector<float> go = {1.75f};
  dgo.up(go);

  // ---- sim ----
  OK(dprhot_sim_fwd(dQ_.p, B, dC_.p, Nc, d, dm.p, 1.0f / T, dS.p, nullptr));
  CK(hipDeviceSynchronize());
  {
    auto S = dS.down();
    double err = 0, mx = 0;
    int badinf = 0;
    for (size_t i = 0; i < S.size(); ++i) {
      if (isinf(rS[i])) { badinf += !(isinf(S[i]) && S[i] < 0); continue; }
      err = std::max(err, fabs(S[i] - rS[i]));
      mx = std::max(mx, fabs(rS[i]));
    }
    report("sim_fwd logits (rel max)", err / mx, 1e-5);
    report("sim_fwd -inf placement", badinf, 0);
  }
  // ---- softmax CE + G ----
  OK(dprhot_softmax_ce_fwd_bwd(dS.p, B, Nc, dy.p, 0, (float)gscale, nullptr, 0, dloss.p, dlse.p, dG.p, nullptr));
  OK(dprhot_reduce_sum(dloss.p, B, 1.0f, dsum.p, nullptr));
  CK(hipDeviceSynchronize());
  std::vector<float> fG((size_t)B * Nc);
  {
    auto lse = dlse.down(); auto loss = dloss.down(); auto G = dG.down(); auto sum = dsum.down();
    double e1 = 0, e2 = 0, e3 = 0, gm = 0, tot = 0;
    for (int i = 0; i < B; ++i) {
      e1 = std::max(e1, fabs(lse[i] - rlse[i]) / std::max(1.0, fabs(rlse[i])));
      e2 = std::max(e2, fabs(loss[i] - rloss[i]) / std::max(1.0, fabs(rloss[i])));
      tot += rloss[i];
    }
    for (size_t i = 0; i < G.size(); ++i) {
      fG[i] = bf2f(G[i]);
      e3 = std::max(e3, fabs(fG[i] - rG[i]));
      gm = std::max(gm, fabs(rG[i]));
    }
    report("row_lse", e1, 2e-6);
    report("row_loss", e2, 1e-5);
    report("loss_sum", fabs(sum[0] - tot) / std::max(1.0, fabs(tot)), 1e-5);
    report("G bf16 (rel max)", e3 / gm, 4.5e-3);
  }
  // ---- dQ / dC against the bf16 G the device produced (isolates the GEMMs) ----
  OK(dprhot_dq(dG.p, dC_.p, B, Nc, d, 2.0f, dgo.p, ddq.p, ws.p, wsb, nullptr));
  OK(dprhot_dc(dG.p, dQ_.p, B, Nc, d, 2.0f, dgo.p, ddc.p, nullptr));
  CK(hipDeviceSynchronize());
  {
    auto q = ddq.down(); auto c = ddc.down();
    const double sc = 2.0 * 1.75;
    double e = 0, mx = 0;
    for (int i = 0; i < B; ++i)
      for (int k = 0; k < d; ++k) {
        double a = 0;
        for (int j = 0; j < Nc; ++j) a += (double)fG[(size_t)i * Nc + j] * fC[(size_t)j * d + k];
        a *= sc;
        e = std::max(e, fabs(q[(size_t)i * d + k] - a));
        mx = std::max(mx, fabs(a));
      }
    report("dQ = s*G*C (rel max)", e / mx, 2e-5);
    e = 0; mx = 0;
    for (int j = 0; j < Nc; ++j)
      for (int k = 0; k < d; ++k) {
        double a = 0;
        for (int i = 0; i < B; ++i) a += (double)fG[(size_t)i * Nc + j] * fQ[(size_t)i * d + k];
        a *= sc;
        e = std::max(e, fabs(c[(size_t)j * d + k] - a));
        mx = std::max(mx, fabs(a));
      }
    report("dC = s*G^T*Q (rel max)", e / mx, 2e-5);
  }
  // ---- rank of gold ----
  OK(dprhot_rank_of_gold(dS.p, B, Nc, dy.p, 0, drank.p, nullptr));
  CK(hipDeviceSynchronize());
  {
    auto S = dS.down(); auto r = drank.down();
    int bad = 0;
    for (int i = 0; i < B; ++i) {
      const float g = S[(size_t)i * Nc + hy[i]];
      int64_t c = 1;
      for (int j = 0; j < Nc; ++j) c += (S[(size_t)i * Nc + j] > g) || (S[(size_t)i * Nc + j] == g && j < hy[i]);
      bad += (c != r[i]);
    }
    report("rank_of_gold mismatches", bad, 0);
  }
  // ---- top-k ----
  {
    const int k = std::min(16, Nc);
    Dev<float> dv((size_t)B * k);
    Dev<int64_t> di((size_t)B * k);
    OK(dprhot_topk(dS.p, B, Nc, k, dv.p, di.p, nullptr));
    CK(hipDeviceSynchronize());
    auto S = dS.down(); auto v = dv.down(); auto idx = di.down();
    int bad = 0;
    std::vector<int> ord(Nc);
    for (int i = 0; i < B; ++i) {
      for (int j = 0; j < Nc; ++j) ord[j] = j;
      const float* row = &S[(size_t)i * Nc];
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return row[a] > row[b]; });
      for (int r = 0; r < k; ++r) bad += (idx[(size_t)i * k + r] != ord[r]) || (v[(size_t)i * k + r] != row[ord[r]]);
    }
    report("topk mismatches", bad, 0);
    // the same columns folded in three ragged pieces must give the same state
    Dev<float> dv2((size_t)B * k);
    Dev<int64_t> di2((size_t)B * k);
    const int cut1 = Nc / 3, cut2 = Nc / 3 + Nc / 2;
    const int cuts[4] = {0, cut1, cut2 < Nc ? cut2 : Nc, Nc};
    bool first = true;
    for (int c = 0; c < 3; ++c) {
      if (cuts[c + 1] == cuts[c]) continue;
      OK(dprhot_topk_update(dS.p + cuts[c], B, cuts[c + 1] - cuts[c], Nc, cuts[c], k, dv2.p, di2.p, first, nullptr));
      first = false;
    }
    CK(hipDeviceSynchronize());
    auto v2 = dv2.down(); auto i2 = di2.down();
    int bad2 = 0;
    for (size_t e = 0; e < (size_t)B * k; ++e) bad2 += (v2[e] != v[e]) || (i2[e] != idx[e]);
    report("topk streamed mismatches", bad2, 0);
  }
  // ---- fused entry points agree with the pieces ----
  {
    Dev<float> l2(B), s2(1), q2((size_t)B * d), c2((size_t)Nc * d);
    Dev<uint16_t> G2((size_t)B * Nc);
    OK(dprhot_inbatch_fwd(dQ_.p, B, dC_.p, Nc, d, dy.p, 0, dm.p, 1.0f / T, (float)gscale, nullptr, l2.p, nullptr, s2.p, G2.p, ws.p, wsb, nullptr));
    CK(hipDeviceSynchronize());
    auto a = G2.down();
    auto l2h = l2.down(); auto s2h = s2.down();
    double e3 = 0, gm = 0, e2 = 0, tot = 0;
    for (size_t i = 0; i < a.size(); ++i) { e3 = std::max(e3, fabs(bf2f(a[i]) - rG[i])); gm = std::max(gm, fabs(rG[i])); }
    for (int i = 0; i < B; ++i) { e2 = std::max(e2, fabs(l2h[i] - rloss[i]) / std::max(1.0, fabs(rloss[i]))); tot += rloss[i]; }
    report("inbatch_fwd G bf16 (rel max)", e3 / gm, 4.5e-3);
    report("inbatch_fwd row_loss", e2, 1e-5);
    report("inbatch_fwd loss_sum", fabs(s2h[0] - tot) / std::max(1.0, fabs(tot)), 1e-5);
    // fp32-operand forward: (q, c fp32) and (q fp32, C already bf16); bf16 copies must equal host RNE
    for (int mode = 0; mode < 2; ++mode) {
      Dev<float> fq((size_t)B * d), fc((size_t)Nc * d);
      fq.up(fQ); fc.up(fC);
      Dev<uint16_t> oq((size_t)B * d), oc((size_t)Nc * d), G3((size_t)B * Nc);
      Dev<float> l3(B), s3(1);
      if (mode == 1) oc.up(hC);
      OK(dprhot_inbatch_fwd_f32(fq.p, mode == 0 ? fc.p : nullptr, oq.p, oc.p, B, Nc, d, dy.p, 0, dm.p, 1.0f / T, (float)gscale, nullptr,
                                l3.p, nullptr, s3.p, G3.p, ws.p, wsb, nullptr));
      CK(hipDeviceSynchronize());
      auto hq = oq.down(); auto hc = oc.down(); auto g3 = G3.down(); auto s3h = s3.down();
      report(mode == 0 ? "fwd_f32(q,c): bf16 copies" : "fwd_f32(q,Cb): bf16 copy",
             (memcmp(hq.data(), hQ.data(), hq.size() * 2) != 0) + (memcmp(hc.data(), hC.data(), hc.size() * 2) != 0), 0);
      report("fwd_f32 G == bf16-input path", memcmp(g3.data(), a.data(), a.size() * 2) != 0, 0);
      report("fwd_f32 loss_sum", fabs(s3h[0] - tot) / std::max(1.0, fabs(tot)), 1e-5);
    }
    OK(dprhot_inbatch_bwd(dG.p, dQ_.p, dC_.p, B, Nc, d, 2.0f, dgo.p, q2.p, c2.p, ws.p, wsb, nullptr));
    CK(hipDeviceSynchronize());
    auto x = q2.down(), y0 = ddq.down(), z = c2.down(), w = ddc.down();
    {  // (the fused entry point may run other kernels than the pieces -- the few-rows units at B <= 128: same sums, other order)
      double eq = 0, mq = 0, ec = 0, mc = 0;
      for (size_t i = 0; i < x.size(); ++i) { eq = std::max(eq, (double)fabs(x[i] - y0[i])); mq = std::max(mq, (double)fabs(y0[i])); }
      for (size_t i = 0; i < z.size(); ++i) { ec = std::max(ec, (double)fabs(z[i] - w[i])); mc = std::max(mc, (double)fabs(w[i])); }
      report("inbatch_bwd dQ vs pieces (rel max)", eq / std::max(mq, 1e-30), 1e-5);
      report("inbatch_bwd dC vs pieces (rel max)", ec / std::max(mc, 1e-30), 1e-5);
    }
    // whole step in one call (two launches at the small shapes): loss / G against the host reference, gradients against
    // dprhot_inbatch_bwd fed with the step's own G
    {
      Dev<float> fq((size_t)B * d), fc((size_t)Nc * d);
      fq.up(fQ); fc.up(fC);
      Dev<uint16_t> oq((size_t)B * d), oc((size_t)Nc * d), G4((size_t)B * Nc);
      Dev<float> l4(B), lse4(B), s4(1), dq4((size_t)B * d), dc4((size_t)Nc * d), dq5((size_t)B * d), dc5((size_t)Nc * d), S4((size_t)B * Nc);
      OK(dprhot_inbatch_step_f32(fq.p, fc.p, oq.p, oc.p, B, Nc, d, dy.p, 0, dm.p, 1.0f / T, (float)gscale, 2.0f, dgo.p, S4.p, l4.p, lse4.p,
                                 s4.p, G4.p, dq4.p, dc4.p, ws.p, wsb, nullptr));
      CK(hipDeviceSynchronize());
      OK(dprhot_inbatch_bwd(G4.p, oq.p, oc.p, B, Nc, d, 2.0f, dgo.p, dq5.p, dc5.p, ws.p, wsb, nullptr));
      CK(hipDeviceSynchronize());
      auto g4 = G4.down(); auto l4h = l4.down(); auto s4h = s4.down();
      auto a4 = dq4.down(), a5 = dq5.down(), b4 = dc4.down(), b5 = dc5.down(), s4m = S4.down(), sref = dS.down();
      double eg = 0, el = 0, eq = 0, mq = 0, ec = 0, mc = 0, es = 0;
      for (size_t i = 0; i < g4.size(); ++i) eg = std::max(eg, fabs(bf2f(g4[i]) - rG[i]));
      for (int i = 0; i < B; ++i) el = std::max(el, fabs(l4h[i] - rloss[i]) / std::max(1.0, fabs(rloss[i])));
      for (size_t i = 0; i < a4.size(); ++i) { eq = std::max(eq, (double)fabs(a4[i] - a5[i])); mq = std::max(mq, (double)fabs(a5[i])); }
      for (size_t i = 0; i < b4.size(); ++i) { ec = std::max(ec, (double)fabs(b4[i] - b5[i])); mc = std::max(mc, (double)fabs(b5[i])); }
      for (size_t i = 0; i < s4m.size(); ++i)
        if (std::isfinite(sref[i]) || std::isfinite(s4m[i])) es = std::max(es, (double)fabs(s4m[i] - sref[i]) / std::max(1.0, (double)fabs(sref[i])));
      report("step G bf16 (rel max)", eg / gm, 4.5e-3);
      report("step row_loss", el, 1e-5);
      report("step loss_sum", fabs(s4h[0] - tot) / std::max(1.0, fabs(tot)), 1e-5);
      report("step logits", es, 1e-5);
      report("step dQ vs bwd(G) (rel max)", eq / std::max(mq, 1e-30), 1e-5);
      report("step dC vs bwd(G) (rel max)", ec / std::max(mc, 1e-30), 1e-5);
    }
  }
  if (timing) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn, double bytes, double flops) {
      for (int i = 0; i < 5; ++i) fn();
      CK(hipDeviceSynchronize());
      const int iters = 50;
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) fn();
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1000.0 / iters;
      printf("    TIME %-22s %9.2f us  %8.1f GB/s  %8.2f TFLOP/s\n", name, us, bytes / us * 1e-3, flops / us * 1e-6);
    };
    const double bn = (double)B * Nc, bd = (double)B * d, nd = (double)Nc * d;
    {
      Dev<float> fq((size_t)B * d), fc((size_t)Nc * d);
      fq.up(fQ); fc.up(fC);
      Dev<uint16_t> oq((size_t)B * d), oc((size_t)Nc * d);
      OK(dprhot_prep(fq.p, (size_t)B * d, oq.p, fc.p, (size_t)Nc * d, oc.p, nullptr));
      CK(hipDeviceSynchronize());
      auto hq = oq.down(); auto hc = oc.down();
      report("prep casts == host RNE", (memcmp(hq.data(), hQ.data(), hq.size() * 2) != 0) + (memcmp(hc.data(), hC.data(), hc.size() * 2) != 0), 0);
      timeit("prep (both casts)", [&] { OK(dprhot_prep(fq.p, (size_t)B * d, oq.p, fc.p, (size_t)Nc * d, oc.p, nullptr)); }, 6 * (bd + nd), 0);
    }
    timeit("sim_fwd", [&] { OK(dprhot_sim_fwd(dQ_.p, B, dC_.p, Nc, d, dm.p, 1.0f / T, dS.p, nullptr)); }, 2 * (bd + nd) + 4 * bn, 2 * bn * d);
    timeit("softmax_ce_fwd_bwd", [&] { OK(dprhot_softmax_ce_fwd_bwd(dS.p, B, Nc, dy.p, 0, (float)gscale, nullptr, 0, dloss.p, dlse.p, dG.p, nullptr)); }, 6 * bn, 0);
    timeit("dq", [&] { OK(dprhot_dq(dG.p, dC_.p, B, Nc, d, 2.0f, dgo.p, ddq.p, ws.p, wsb, nullptr)); }, 2 * bn + 2 * nd + 4 * bd, 2 * bn * d);
    timeit("dc", [&] { OK(dprhot_dc(dG.p, dQ_.p, B, Nc, d, 2.0f, dgo.p, ddc.p, nullptr)); }, 2 * bn + 2 * bd + 4 * nd, 2 * bn * d);
    {
      Dev<float> fq((size_t)B * d), fc((size_t)Nc * d), dq4((size_t)B * d), dc4((size_t)Nc * d);
      fq.up(fQ); fc.up(fC);
      Dev<uint16_t> oq((size_t)B * d), oc((size_t)Nc * d);
      timeit("step_f32 (fwd + bwd)", [&] { OK(dprhot_inbatch_step_f32(fq.p, fc.p, oq.p, oc.p, B, Nc, d, dy.p, 0, dm.p, 1.0f / T, (float)gscale, 1.0f, dgo.p, nullptr, dloss.p, dlse.p, dsum.p, dG.p, dq4.p, dc4.p, ws.p, wsb, nullptr)); }, 0, 6 * bn * d);
    }
    timeit("rank_of_gold", [&] { OK(dprhot_rank_of_gold(dS.p, B, Nc, dy.p, 0, drank.p, nullptr)); }, 4 * bn, 0);
    timeit("inbatch_fwd", [&] { OK(dprhot_inbatch_fwd(dQ_.p, B, dC_.p, Nc, d, dy.p, 0, dm.p, 1.0f / T, (float)gscale, nullptr, dloss.p, dlse.p, dsum.p, dG.p, ws.p, wsb, nullptr)); }, 2 * (bd + nd) + 10 * bn, 2 * bn * d);
    {
      Dev<float> fq((size_t)B * d), fc((size_t)Nc * d);
      fq.up(fQ); fc.up(fC);
      Dev<uint16_t> oq((size_t)B * d), oc((size_t)Nc * d);
      timeit("inbatch_fwd_f32", [&] { OK(dprhot_inbatch_fwd_f32(fq.p, fc.p, oq.p, oc.p, B, Nc, d, dy.p, 0, dm.p, 1.0f / T, (float)gscale, nullptr, dloss.p, dlse.p, dsum.p, dG.p, ws.p, wsb, nullptr)); }, 6 * (bd + nd) + 10 * bn, 2 * bn * d);
    }
    timeit("inbatch_bwd", [&] { OK(dprhot_inbatch_bwd(dG.p, dQ_.p, dC_.p, B, Nc, d, 2.0f, dgo.p, ddq.p, ddc.p, ws.p, wsb, nullptr)); }, 4 * bn + 2 * (bd + nd) + 4 * (bd + nd), 4 * bn * d);
    timeit("fwd+bwd", [&] {
      OK(dprhot_inbatch_fwd(dQ_.p, B, dC_.p, Nc, d, dy.p, 0, dm.p, 1.0f / T, (float)gscale, nullptr, dloss.p, dlse.p, dsum.p, dG.p, ws.p, wsb, nullptr));
      OK(dprhot_inbatch_bwd(dG.p, dQ_.p, dC_.p, B, Nc, d, 2.0f, dgo.p, ddq.p, ddc.p, ws.p, wsb, nullptr)); }, 0, 6 * bn * d);
  }
}

// dprhot_search against a host top-k of the full score matrix (scores from dprhot_sim_fwd: the test is about the
// selection, score values are checked in run_case)
static void search_case(int nq, int n, int d, int k, int chunk, bool quantise, bool timing) {
  printf("search nq=%d n=%d d=%d k=%d chunk=%d%s\n", nq, n, d, k, chunk, quantise ? " (quantised: many ties)" : "");
  std::vector<uint16_t> hq((size_t)nq * d), hc((size_t)n * d);
  for (auto& x : hq) x = f2bf(quantise ? (float)(int)(urand() * 3 - 1) : nrand());
  for (auto& x : hc) x = f2bf(quantise ? (float)(int)(urand() * 3 - 1) : nrand());
  Dev<uint16_t> dq(hq.size()), dc(hc.size());
  dq.up(hq); dc.up(hc);
  size_t wsb = 0;
  OK(dprhot_search_workspace_bytes(nq, chunk, &wsb));
  Dev<float> dS((size_t)nq * n), dv((size_t)nq * k), ws(wsb / sizeof(float));
  Dev<int64_t> di((size_t)nq * k);
  OK(dprhot_sim_fwd(dq.p, nq, dc.p, n, d, nullptr, 1.0f, dS.p, nullptr));
  OK(dprhot_search(dq.p, nq, dc.p, n, d, 1000, k, chunk, dv.p, di.p, 1, ws.p, ws.n * sizeof(float), nullptr));
  CK(hipDeviceSynchronize());
  auto S = dS.down(); auto v = dv.down(); auto idx = di.down();
  int bad = 0;
  std::vector<int> ord(n);
  for (int i = 0; i < nq; ++i) {
    for (int j = 0; j < n; ++j) ord[j] = j;
    const float* row = &S[(size_t)i * n];
    std::partial_sort(ord.begin(), ord.begin() + k, ord.end(),
                      [&](int a, int b) { return row[a] > row[b] || (row[a] == row[b] && a < b); });
    for (int r = 0; r < k; ++r) bad += (idx[(size_t)i * k + r] != 1000 + ord[r]) || (v[(size_t)i * k + r] != row[ord[r]]);
  }
  report("search top-k mismatches", bad, 0);
  if (timing) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) OK(dprhot_search(dq.p, nq, dc.p, n, d, 0, k, chunk, dv.p, di.p, 1, ws.p, ws.n * sizeof(float), nullptr));
    CK(hipEventRecord(e0, nullptr));
    const int it = 5;
    for (int w = 0; w < it; ++w) OK(dprhot_search(dq.p, nq, dc.p, n, d, 0, k, chunk, dv.p, di.p, 1, ws.p, ws.n * sizeof(float), nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= it;
    printf("    search: %.3f ms  (%.1f TFLOP/s scoring, %.2f M query-passage scores/us)\n", ms, 2.0 * nq * n * d / ms * 1e-9,
           (double)nq * n / ms * 1e-9);
    // the pieces
    CK(hipEventRecord(e0, nullptr));
    for (int w = 0; w < it; ++w) OK(dprhot_sim_fwd(dq.p, nq, dc.p, chunk, d, nullptr, 1.0f, ws.p, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("    one chunk: sim %.1f us", ms / it * 1e3);
    CK(hipEventRecord(e0, nullptr));
    for (int w = 0; w < it; ++w) OK(dprhot_topk_update(ws.p, nq, chunk, chunk, 0, k, dv.p, di.p, 1, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf(", top-k from empty %.1f us", ms / it * 1e3);
    CK(hipEventRecord(e0, nullptr));
    for (int w = 0; w < it; ++w) OK(dprhot_topk_update(ws.p, nq, chunk, chunk, chunk, k, dv.p, di.p, 0, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf(", top-k warm %.1f us (%.0f GB/s)\n", ms / it * 1e3, (double)nq * chunk * 4 / (ms / it) * 1e-6);
  }
}

int main(int argc, char** argv) {
  const bool timing = argc > 1 && strstr(argv[1], "time");
  const bool big = argc > 1 && strstr(argv[1], "big");
  // this binary (not the library) reads the environment: DPRHOT_TILE / DPRHOT_NO_TR / DPRHOT_BIG_MIN -> dprhot_set_option
  {
    const char* const env_opt[3][2] = {{"DPRHOT_TILE", "tile"}, {"DPRHOT_NO_TR", "no_tr"}, {"DPRHOT_BIG_MIN", "big_min"}};
    for (int i = 0; i < 3; ++i)
      if (const char* e = getenv(env_opt[i][0])) dprhot_set_option(env_opt[i][1], atoi(e));
  }
  printf("libdprhot version %d  DPRHOT_TILE=%s DPRHOT_NO_TR=%s\n", dprhot_version(), getenv("DPRHOT_TILE") ? getenv("DPRHOT_TILE") : "-",
         getenv("DPRHOT_NO_TR") ? getenv("DPRHOT_NO_TR") : "-");
  {
    Dev<uint16_t> o(256);
    hipLaunchKernelGGL(trdump_kernel, dim3(1), dim3(64), 0, nullptr, o.p);
    CK(hipDeviceSynchronize());
    auto h = o.down();
    printf("trdump (lane: 4 values; LDS image value = element index, lane l address = 4*l):\n");
    int expect_ok = 1;
    for (int l = 0; l < 64; ++l) {
      if (l < 20 || l >= 60) printf("  lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      for (int j = 0; j < 4; ++j) expect_ok &= (h[l * 4 + j] == (l & 15) + j * 16 + (l >> 4) * 64);
    }
    printf("trdump matches documented layout: %s\n", expect_ok ? "yes" : "NO");
  }
  run_case(4, 8, 128, 2, 1.0f, false, false);
  run_case(32, 256, 768, 8, 1.0f, false, timing);
  run_case(37, 264, 136, 7, 0.5f, true, false);
  run_case(100, 1000, 200, 10, 1.0f, true, false);
  run_case(64, 1024, 1024, 2, 1.0f, false, timing);
  run_case(8, 512, 768, 8, 0.05f, true, timing);
  run_case(32, 2112, 768, 66, 1.0f, true, timing);   // cfg2 per rank at W=8 incl. the mask rows (two chunks per thread)
  run_case(16, 4096, 128, 256, 2.0f, true, false);
  if (timing || big) run_case(128, 8192, 768, 8, 1.0f, true, timing);
  run_case(32, 528, 768, 16, 1.0f, true, timing);   // cfg2 gathered over 2 and 4 ranks: one slab, fused softmax+backward
  run_case(32, 1056, 768, 33, 1.0f, true, timing);
  run_case(16, 1152, 64, 72, 0.5f, true, false);
  run_case(5, 40, 64, 8, 1.0f, true, false);       // small-step shapes with ragged Nc (not a multiple of 32 / 16)
  run_case(32, 264, 192, 8, 0.5f, true, false);
  run_case(20, 488, 128, 24, 1.0f, true, false);
  run_case(300, 1000, 192, 4, 1.0f, true, false);   // ragged in M and N; with DPRHOT_BIG_MIN=1 through the 256x256 kernel
  run_case(520, 520, 128, 1, 0.5f, false, false);
  run_case(128, 320, 200, 2, 1.0f, true, false);    // B, Nc multiples of 64: with DPRHOT_BIG_MIN=1 the 256x256 backward pair,
  run_case(320, 1088, 136, 3, 1.0f, true, false);   // ragged tiles in every dimension, dQ split over K
  search_case(16, 5000 / 8 * 8, 64, 10, 1024, true, false);
  search_case(40, 30000, 128, 100, 8192, false, false);
  search_case(3, 20000, 768, 128, 4096, true, false);
  search_case(300, 24576, 128, 50, 8192, true, false);  // ragged row tile; with DPRHOT_BIG_MIN=1 the persistent 256x256 filter GEMM
  if (timing || big) search_case(1024, 1 << 20, 768, 100, 65536, false, timing);
  printf(g_fail ? "SELFTEST FAILED (%d)\n" : "SELFTEST PASSED\n", g_fail);
  return g_fail ? 1 : 0;
}
