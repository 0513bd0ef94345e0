// gradcomm.h -- the three HBM streams of the towers' gradient all-reduce (the reference registers torch's
// fp16_compress_hook, dpr_scale/task/dpr_task.py:90-92).  The all-pairs decomposition of dpr_scale_amd/comm_hooks.py moves a DDP
// bucket as   pack -> all-to-all -> sum of the W received shards -> all-gather -> unpack ;   the three local legs are one pass each:
//   grad_pack_kernel        send[i] = wire(bucket[i] * scale) for i < n, 0 up to n_pad          (4 B read, 2 B written per element)
//   grad_sum_shards_kernel  out[i]  = sum_r recv[r][i], fp32 accumulation in rank order r = 0..W-1 (deterministic), one rounding
//   grad_unpack_kernel      bucket[i] = float(full[i]) for i < n
// Wire kinds: 0 = bf16 (RNE, v_cvt_pk_bf16_f32), 1 = fp16 (RNE; the reference's format), 2 = fp32.
// Every global access is a 16-byte vector with consecutive lanes; each thread keeps U independent 16-byte groups in flight.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowwise.h"

namespace dprhot {

constexpr int GC_BF16 = 0, GC_FP16 = 1, GC_FP32 = 2;

__device__ __forceinline__ uint32_t gc_pk_f16(float a, float b) {
  const __half2 h = __floats2half2_rn(a, b);
  return __builtin_bit_cast(uint32_t, h);
}
template <int KIND>
__device__ __forceinline__ uint32_t gc_pack2(float a, float b) {
  if constexpr (KIND == GC_BF16) return pk_bf16(a, b);
  else return gc_pk_f16(a, b);
}
template <int KIND>
__device__ __forceinline__ void gc_unpack2(uint32_t w, float& a, float& b) {
  if constexpr (KIND == GC_BF16) {
    a = __uint_as_float(w << 16);
    b = __uint_as_float(w & 0xffff0000u);
  } else {
    const float2 f = __half22float2(__builtin_bit_cast(__half2, w));
    a = f.x;
    b = f.y;
  }
}
// 8 consecutive values of a 2-byte kind (one 16-byte load) or of fp32 (two)
// NT: non-temporal loads -- for a source that is read once and is larger than the Infinity Cache (the fp32 bucket of the pack leg:
// 131 -> 115 us for 110 M elements).  Measured and NOT used elsewhere: non-temporal loads of the wire buffers (unpack 125 -> 135 us,
// shard sum 38 -> 45) and non-temporal stores anywhere in these legs (unpack 125 -> 210 us).
template <int KIND, bool NT = false>
__device__ __forceinline__ void gc_load8(const void* base, size_t c, float (&v)[8]) {
  if constexpr (KIND == GC_FP32) {
    typedef float gc_f4 __attribute__((ext_vector_type(4)));
    gc_f4 a, b;
    if constexpr (NT) {
      a = __builtin_nontemporal_load(reinterpret_cast<const gc_f4*>(base) + 2 * c);
      b = __builtin_nontemporal_load(reinterpret_cast<const gc_f4*>(base) + 2 * c + 1);
    } else {
      a = reinterpret_cast<const gc_f4*>(base)[2 * c];
      b = reinterpret_cast<const gc_f4*>(base)[2 * c + 1];
    }
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    const uint4 w = reinterpret_cast<const uint4*>(base)[c];
    gc_unpack2<KIND>(w.x, v[0], v[1]);
    gc_unpack2<KIND>(w.y, v[2], v[3]);
    gc_unpack2<KIND>(w.z, v[4], v[5]);
    gc_unpack2<KIND>(w.w, v[6], v[7]);
  }
}
template <int KIND>
__device__ __forceinline__ void gc_store8(void* base, size_t c, const float (&v)[8]) {
  if constexpr (KIND == GC_FP32) {
    reinterpret_cast<float4*>(base)[2 * c] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(base)[2 * c + 1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    reinterpret_cast<uint4*>(base)[c] =
        make_uint4(gc_pack2<KIND>(v[0], v[1]), gc_pack2<KIND>(v[2], v[3]), gc_pack2<KIND>(v[4], v[5]), gc_pack2<KIND>(v[6], v[7]));
  }
}
template <int KIND>
__device__ __forceinline__ void gc_store1(void* base, size_t i, float v) {
  if constexpr (KIND == GC_FP32) reinterpret_cast<float*>(base)[i] = v;
  else if constexpr (KIND == GC_BF16) reinterpret_cast<uint16_t*>(base)[i] = (uint16_t)(pk_bf16(v, 0.f) & 0xffffu);
  else reinterpret_cast<uint16_t*>(base)[i] = (uint16_t)(gc_pk_f16(v, 0.f) & 0xffffu);
}
template <int KIND>
__device__ __forceinline__ float gc_load1(const void* base, size_t i) {
  if constexpr (KIND == GC_FP32) return reinterpret_cast<const float*>(base)[i];
  else {
    float a, b;
    gc_unpack2<KIND>(reinterpret_cast<const uint16_t*>(base)[i], a, b);
    return a;
  }
}

constexpr int GC_U = 4;  // 16-byte groups per thread and trip (x2 loads each for fp32 sources)

// send[0..n_pad) <- wire(bucket[0..n) * scale), zeros beyond n.  One workgroup per contiguous tile of 256 x GC_UT groups of four values
// (see grad_unpack_kernel: the persistent grid-stride form of this leg ran at 0.72).  n_pad % 8 == 0.
#ifndef GC_UT_EXP
#define GC_UT_EXP 2
#endif
constexpr int GC_UT = GC_UT_EXP;  // (tile sweep of the unpack leg, 110 M elements: 16 / 8 / 4 / 2 groups per thread 124 / 117 / 111 / 110 us)
template <int WIRE>
__global__ __launch_bounds__(256) void grad_pack_kernel(const float* __restrict__ bucket, size_t n, float scale, void* __restrict__ send,
                                                        size_t n_pad) {
  typedef float gc_f4 __attribute__((ext_vector_type(4)));
  const size_t n4 = n / 4, p4 = n_pad / 4;
  const size_t base = (size_t)blockIdx.x * (256 * GC_UT) + threadIdx.x;
  gc_f4 v[GC_UT];
#pragma unroll
  for (int u = 0; u < GC_UT; ++u) {
    const size_t g = base + u * 256;
    if (g < n4) {
      v[u] = __builtin_nontemporal_load(reinterpret_cast<const gc_f4*>(bucket) + g);  // (read once, larger than the Infinity Cache)
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[u][e] = (g < p4 && g * 4 + e < n) ? bucket[g * 4 + e] : 0.f;  // the ragged end, then zero padding
    }
  }
#pragma unroll
  for (int u = 0; u < GC_UT; ++u) {
    const size_t g = base + u * 256;
    if (g < p4) {
      const gc_f4 w = v[u] * scale;
      if constexpr (WIRE == GC_FP32) reinterpret_cast<gc_f4*>(send)[g] = w;
      else reinterpret_cast<uint2*>(send)[g] = make_uint2(gc_pack2<WIRE>(w[0], w[1]), gc_pack2<WIRE>(w[2], w[3]));
    }
  }
}

// out[0..shard) <- sum over r of recv[r * shard + i]: fp32 accumulation in rank order, ONE rounding into OUT.  shard % 8 == 0.
template <int WIRE, int OUT>
__global__ __launch_bounds__(256) void grad_sum_shards_kernel(const void* __restrict__ recv, int W, size_t shard, void* __restrict__ out) {
  const size_t s8 = shard / 8;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < s8; c += stride) {
    float acc[8];
    gc_load8<WIRE>(recv, c, acc);
    int r = 1;
    for (; r + 3 < W; r += 4) {  // four ranks' groups in flight
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) gc_load8<WIRE>(recv, (size_t)(r + u) * s8 + c, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[u][e];
    }
    for (; r < W; ++r) {
      float v[8];
      gc_load8<WIRE>(recv, (size_t)r * s8 + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
    gc_store8<OUT>(out, c, acc);
  }
}

// bucket[0..n) <- float(full[0..n))
// One workgroup per contiguous tile of 256 x GC_UT groups of FOUR values (8 bytes in, 16 out per thread and group): every load
// instruction of a wave covers 512 contiguous bytes, every store 1 KiB without holes, and a workgroup's GC_UT groups are neighbours.
// The persistent grid-stride form with 8 values per thread (16 bytes in, two 16-byte stores 32 bytes apart out; its GC_U groups a
// whole grid apart) ran at 0.59-0.62 of the HBM rate where torch's own widening copy reaches 0.72 on the same box (bench.py:
// grad_hook.torch_streams_same_box); trading halves between lanes to densify ITS stores had made it slower (145 us against 130).
template <int KIND>
__global__ __launch_bounds__(256) void grad_unpack_kernel(const void* __restrict__ full, float* __restrict__ bucket, size_t n) {
  const size_t n4 = n / 4;
  const size_t base = (size_t)blockIdx.x * (256 * GC_UT) + threadIdx.x;
  if constexpr (KIND == GC_FP32) {
    float4 v[GC_UT];
#pragma unroll
    for (int u = 0; u < GC_UT; ++u)
      if (base + u * 256 < n4) v[u] = reinterpret_cast<const float4*>(full)[base + u * 256];
#pragma unroll
    for (int u = 0; u < GC_UT; ++u)
      if (base + u * 256 < n4) reinterpret_cast<float4*>(bucket)[base + u * 256] = v[u];
  } else {
    uint2 w[GC_UT];
#pragma unroll
    for (int u = 0; u < GC_UT; ++u)
      if (base + u * 256 < n4) w[u] = reinterpret_cast<const uint2*>(full)[base + u * 256];
#pragma unroll
    for (int u = 0; u < GC_UT; ++u)
      if (base + u * 256 < n4) {
        float4 v;
        gc_unpack2<KIND>(w[u].x, v.x, v.y);
        gc_unpack2<KIND>(w[u].y, v.z, v.w);
        reinterpret_cast<float4*>(bucket)[base + u * 256] = v;
      }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bucket[n4 * 4 + threadIdx.x] = gc_load1<KIND>(full, n4 * 4 + threadIdx.x);
}

// ---- the autograd operator's grad_output fix-up ------------------------------------------------------------------------------------
// The training step computes dQ / dC_part in the forward call, scaled by the grad_output it EXPECTS (`used`: the value the previous
// backward saw -- AMP's loss scale changes once in thousands of steps, plain backward() is always 1).  backward() launches this
// kernel: every workgroup reads the two scalars and leaves when they agree; only when the scale really changed are the gradients
// multiplied by go / used.  The first thread also publishes out2[0] = go (what the gradients are now scaled by) and out2[1] = the
// value the NEXT forward should expect (go when it is a finite, normal, non-zero number, else 1).  out2 is a fresh 2-float buffer,
// never one of the inputs: no workgroup can read a value another one has already replaced.
// With nslabs > 0 the step left dQ as split-K partial slabs (dprhot_train_dq_slabs): workgroups [0, nqb) form dQ = go * sum of the
// slabs in slab order (bit-reproducible) -- the reduction launch of the step rides in this one -- and the rest treat dC as above.
template <int DCK>
__global__ __launch_bounds__(256) void rescale_grads_kernel(float* __restrict__ dQ, size_t nq8, const float* __restrict__ dq_part, int nslabs,
                                                            int nqb, void* __restrict__ dC, size_t nc8, const float* __restrict__ go,
                                                            const float* __restrict__ used, float* __restrict__ out2) {
  const float g = go[0], u = used[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out2[0] = g;
    const float a = fabsf(g);
    out2[1] = (a >= 1.17549435e-38f && a <= 3.0e38f) ? g : 1.0f;
  }
  if (nslabs > 0 && (int)blockIdx.x < nqb) {
    const size_t stride = (size_t)nqb * blockDim.x;
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < nq8; c += stride) {
      float acc[8];
      gc_load8<GC_FP32>(dq_part, c, acc);
      int z = 1;
      for (; z + 3 < nslabs; z += 4) {  // four slabs' groups in flight
        float v[4][8];
#pragma unroll
        for (int w = 0; w < 4; ++w) gc_load8<GC_FP32>(dq_part, (size_t)(z + w) * nq8 + c, v[w]);
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += v[w][e];
      }
      for (; z < nslabs; ++z) {
        float v[8];
        gc_load8<GC_FP32>(dq_part, (size_t)z * nq8 + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= g;
      gc_store8<GC_FP32>(dQ, c, acc);
    }
    return;
  }
  if (g == u) return;
  const float ratio = g / u;
  const int b0 = nslabs > 0 ? nqb : 0;
  const size_t first = nslabs > 0 ? nq8 : 0;  // dQ is already exact when it came from the slabs
  const size_t stride = (size_t)(gridDim.x - b0) * blockDim.x;
  for (size_t c = first + (blockIdx.x - b0) * (size_t)blockDim.x + threadIdx.x; c < nq8 + nc8; c += stride) {
    float v[8];
    if (c < nq8) {
      gc_load8<GC_FP32>(dQ, c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= ratio;
      gc_store8<GC_FP32>(dQ, c, v);
    } else {
      gc_load8<DCK>(dC, c - nq8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= ratio;
      gc_store8<DCK>(dC, c - nq8, v);
    }
  }
}

}  // namespace dprhot
