// experiment: in-kernel wall-clock stamps of the fused softmax+backward kernel (build with -DDPRHOT_TIMING, not shipped)
#define DPRHOT_TIMING 1
#include <hip/hip_runtime.h>
__device__ unsigned long long g_dprhot_tm[64];
#include "../dpr_scale_amd/csrc/dprhot.hip"
#include <vector>
#include <stdio.h>
int main(int argc, char** argv) {
  const int B = 32, Nc = argc > 1 ? atoi(argv[1]) : 256, d = 768;
  float *q, *c, *dq, *dc; uint16_t *Qb, *Cb, *G; int64_t* y; uint8_t* m; float *loss, *lse, *sum, *go; void* ws; size_t wsb;
  dprhot_workspace_bytes(B, Nc, d, &wsb);
  hipMalloc(&q, B * d * 4); hipMalloc(&c, Nc * d * 4); hipMalloc(&Qb, B * d * 2); hipMalloc(&Cb, Nc * d * 2); hipMalloc(&G, B * Nc * 2);
  hipMalloc(&dq, B * d * 4); hipMalloc(&dc, Nc * d * 4); hipMalloc(&go, 4);
  hipMalloc(&y, B * 8); hipMalloc(&m, Nc); hipMalloc(&loss, B * 4); hipMalloc(&lse, B * 4); hipMalloc(&sum, 4); hipMalloc(&ws, wsb);
  std::vector<float> h(Nc * d, 0.01f); hipMemcpy(q, h.data(), B * d * 4, hipMemcpyHostToDevice); hipMemcpy(c, h.data(), Nc * d * 4, hipMemcpyHostToDevice);
  float one = 1.f; hipMemcpy(go, &one, 4, hipMemcpyHostToDevice);
  std::vector<int64_t> hy(B); for (int i = 0; i < B; ++i) hy[i] = i * 8; hipMemcpy(y, hy.data(), B * 8, hipMemcpyHostToDevice); hipMemset(m, 0, Nc);
  for (int it = 0; it < 8; ++it) {
    dprhot_inbatch_step_f32(q, c, Qb, Cb, B, Nc, d, y, 0, m, 1.f, 1.f / B, 1.f, go, nullptr, loss, lse, sum, G, dq, dc, ws, wsb, nullptr);
    hipDeviceSynchronize();
    unsigned long long t[16]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_dprhot_tm), sizeof(t));
    printf("it%d ticks(10ns): sim begin->end %llu | gap %llu | issue loads %llu  wait+lds store %llu  slab sum(wait) %llu  softmax+G %llu  barrier %llu  dQ %llu  dC %llu  total %llu\n",
           it, t[6] - t[0], t[8] - t[6], t[9] - t[8], t[10] - t[9], t[11] - t[10], t[12] - t[11], t[13] - t[12], t[14] - t[13], t[15] - t[14], t[15] - t[8]);
  }
  return 0;
}
