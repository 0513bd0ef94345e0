// experiment: in-kernel wall-clock stamps of the sim kernel (build with -DDPRHOT_TIMING, not shipped)
#define DPRHOT_TIMING 1
#include <hip/hip_runtime.h>
__device__ unsigned long long g_dprhot_tm[64];
#include "../dpr_scale_amd/csrc/dprhot.hip"
#include <vector>
#include <stdio.h>
#include <string.h>
int main() {
  const int B = 32, Nc = 256, d = 768;
  float *q, *c; uint16_t *Qb, *Cb, *G; int64_t* y; uint8_t* m; float *loss, *lse, *sum; void* ws; size_t wsb;
  dprhot_workspace_bytes(B, Nc, d, &wsb);
  hipMalloc(&q, B * d * 4); hipMalloc(&c, Nc * d * 4); hipMalloc(&Qb, B * d * 2); hipMalloc(&Cb, Nc * d * 2); hipMalloc(&G, B * Nc * 2);
  hipMalloc(&y, B * 8); hipMalloc(&m, Nc); hipMalloc(&loss, B * 4); hipMalloc(&lse, B * 4); hipMalloc(&sum, 4); hipMalloc(&ws, wsb);
  std::vector<float> h(Nc * d, 0.01f); hipMemcpy(q, h.data(), B * d * 4, hipMemcpyHostToDevice); hipMemcpy(c, h.data(), Nc * d * 4, hipMemcpyHostToDevice);
  std::vector<int64_t> hy(B); for (int i = 0; i < B; ++i) hy[i] = i * 8; hipMemcpy(y, hy.data(), B * 8, hipMemcpyHostToDevice); hipMemset(m, 0, Nc);
  for (int mode = 0; mode < 3; ++mode) {
    for (int it = 0; it < 6; ++it) {
      if (mode == 0) dprhot_inbatch_fwd_f32(q, c, Qb, Cb, B, Nc, d, y, 0, m, 1.f, 1.f / B, nullptr, loss, lse, sum, G, ws, wsb, nullptr);
      else if (mode == 1) { dprhot_prep(q, B * d, Qb, c, Nc * d, Cb, nullptr); dprhot_inbatch_fwd(Qb, B, Cb, Nc, d, y, 0, m, 1.f, 1.f / B, nullptr, loss, lse, sum, G, ws, wsb, nullptr); }
      else { dprhot_prep(q, B * d, Qb, c, Nc * d, Cb, nullptr); dprhot_sim_fwd(Qb, B, Cb, Nc, d, m, 1.f, (float*)ws, nullptr); }
      float go = 1.f; (void)go;
      hipDeviceSynchronize();
      unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_dprhot_tm), sizeof(t));
      printf("%s it%d ticks(10ns): begin->firstdata %llu  cvt+ldsstore %llu  barrier %llu  kloop %llu  barrier %llu  epilogue %llu  total %llu\n",
             mode == 0 ? "f32 " : (mode == 1 ? "bf16" : "nost"), it, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[6] - t[0]);
    }
  }
  return 0;
}
