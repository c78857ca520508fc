// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ddpo_hip.h"

#define DDPO_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return DDPO_ELAUNCH;               \
  } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS. Result valid in all threads.
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = (lane < nw) ? red[lane] : 0.f;
  t = wave_sum(t);
  return t;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}

// ---- fp32 -> bf16 hi / lo split of the bf16x3 datapath (x ~= hi + lo, both bf16, round-to-nearest-even conversions)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// split 4 floats into 4 bf16 "hi" (2 dwords) and 4 bf16 "lo" (2 dwords)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  hi.x = cvt_pk_bf16(v.x, v.y);
  hi.y = cvt_pk_bf16(v.z, v.w);
  const float r0 = v.x - __uint_as_float(hi.x << 16), r1 = v.y - __uint_as_float(hi.x & 0xFFFF0000u);
  const float r2 = v.z - __uint_as_float(hi.y << 16), r3 = v.w - __uint_as_float(hi.y & 0xFFFF0000u);
  lo.x = cvt_pk_bf16(r0, r1);
  lo.y = cvt_pk_bf16(r2, r3);
}

// Element offset of (row, channel c) in a bf16 ACTIVATION plane of `rows` rows.  ld > 0: row-major, row stride ld.  ld == 0 (ABI v6):
// k-blocked, (C / 32, rows, 32) — the 32 channels of one k-tile of CONSECUTIVE rows are consecutive memory, so an LDS-DMA piece
// of the plane-fed GEMM (16 consecutive rows x 64 B) is 1 KiB contiguous = 8 full cache lines instead of 16 half lines.
__host__ __device__ __forceinline__ int64_t plane_off(int64_t row, int c, int ld, int64_t rows) {
  return ld ? row * ld + c : ((int64_t)(c >> 5) * rows + row) * 32 + (c & 31);
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
