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

// Butterfly reductions over the 64 lanes: v += partner(lane ^ 32), ^16, ^8, ^4, ^2, ^1 — the association order every "bit-reproducible"
// statement in this tree rests on — as six `ds_bpermute_b32` (the LDS crossbar).  (Round 5: a DPP / v_permlane{32,16}_swap form was 3x faster per
// reduction, 193 -> 64 ns, but its two lane-swap steps did not return lane ^ 32 / ^ 16 on hardware in either of its two spellings — 67 % of the
// lanes differed from this butterfly, profiles/r05_first_call.log — and the norms it would have served are 3 % of the sampling step: deleted.)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t lane_xor1_u32(uint32_t x) { return __shfl_xor(x, 1, 64); }
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

// gelu(x, approximate=tanh) = 0.5 x (1 + tanh u), u = sqrt(2/pi) (x + 0.044715 x^3), evaluated through the identity
// 0.5 (1 + tanh u) = sigmoid(2u) = 1 / (1 + 2^(-2 log2(e) u)) on v_exp_f32 + v_rcp_f32 (~6 instructions; saturates correctly: 2^(+big) = inf ->
// 1 / inf = 0, 2^(-big) = 0 -> 1).  Rounds 1-4 called libm's tanhf (~40 instructions with its range reduction and branches): 32 evaluations per
// lane in the fused GEGLU output stage of FF1 were 1786 VALU instructions per lane and 128 x 128 tile against 918 now.  Measured on MI355X
// (round 5, `kernel_probe gelu`, profiles/r05_first_call.log): against float64 on [-12, 12] max abs error 5.2e-7 (tanhf form 4.3e-7), error
// relative to max(|gelu|, 1e-3) 1.3e-6 (tanhf form 7.7e-5: 1 + tanh(u) cancels for negative x), 2.57x faster per evaluation; sampling +1.0 %
// interleaved on one box.
__device__ __forceinline__ float sigmoid_2u_fast(float u) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  return x * sigmoid_2u_fast(u);
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

// ---- "f16mx" operand format (ABI v7): x ~= h + l with h = f16(x) (round to nearest even) and the 8-bit parts the CROSS terms of
// a product use: a*b ~= a_h*b_h (16-bit MFMA) + a_h8*b_l8 + a_l8*b_h8 (ONE block-scaled 8-bit MFMA, v_mfma_scale_f32_32x32x64_f8f6f4,
// lanes 0-31 carrying the first pair and lanes 32-63 the second).  ACTIVATIONS: e5m2 at a FIXED scale — h8 = e5m2(h) (the f16 value
// rounded to two mantissa bits; same exponent range, so no block scale has to be computed, stored or fetched), l8 = e5m2(l * 2^11)
// (|l| <= 2^-11 |x|).  Two planes of the geometry of the bf16 hi / lo planes: `p16` holds h, `p8` holds per 32-channel block the
// 64 bytes [h8 c0-15 | l8 c0-15 | h8 c16-31 | l8 c16-31] — 16-byte chunk q of a row is what lane half (q & 1) of the 8-bit MFMA reads for
// its k half (q >> 1), i.e. the fragment reads use the SAME LDS offsets as the 16-bit plane's.  WEIGHTS: e4m3 with one power-of-two
// scale per output column, chunks [l8 | h8 | l8 | h8] (ddpo_pack_weights_f16mx).
typedef _Float16 mx_half2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mx_split4(const float4 v, uint2& h16, uint32_t& h8, uint32_t& l8) {
  // Saturation (tests/test_gpu_f16mx.py::test_operands_beyond_the_f16_range_...): the value is clamped to the f16 range FIRST and every
  // part is derived from the clamped value (the low part of an unclamped 7e4 would be 4496 * 2^11 = inf in e5m2), and the e5m2 image of h is taken
  // from h clamped to e5m2's largest finite value 57344 (f16 values above 61440 round to e5m2 infinity otherwise; the cross terms are 2^-3 relative,
  // so the clamp costs such a value nothing it had).  Values inside +-57344 — every activation of the model — produce the same bits as before.
  const float lim = 65504.f, lim8 = 57344.f;
  const float vx = fminf(fmaxf(v.x, -lim), lim), vy = fminf(fmaxf(v.y, -lim), lim), vz = fminf(fmaxf(v.z, -lim), lim), vw = fminf(fmaxf(v.w, -lim), lim);
  const _Float16 a = (_Float16)vx, b = (_Float16)vy, c = (_Float16)vz, e = (_Float16)vw;
  h16.x = __builtin_bit_cast(uint32_t, mx_half2{a, b});
  h16.y = __builtin_bit_cast(uint32_t, mx_half2{c, e});
  const float ha = (float)a, hb = (float)b, hc = (float)c, he = (float)e;
  int p = __builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(ha, -lim8), lim8), fminf(fmaxf(hb, -lim8), lim8), 0, false);
  h8 = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(fminf(fmaxf(hc, -lim8), lim8), fminf(fmaxf(he, -lim8), lim8), p, true);
  p = __builtin_amdgcn_cvt_pk_bf8_f32((vx - ha) * 2048.f, (vy - hb) * 2048.f, 0, false);
  l8 = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32((vz - hc) * 2048.f, (vw - he) * 2048.f, p, true);
}
// 4 consecutive channels c .. c+3 (c % 4 == 0) of `row` into the two f16mx planes (p16 / p8 addressed like bf16 planes: plane_off)
__device__ __forceinline__ void mx_store4(uint16_t* __restrict__ p16, uint16_t* __restrict__ p8, int64_t row, int c, int ld, int64_t rows,
                                          const float4 v) {
  uint2 h16;
  uint32_t h8, l8;
  mx_split4(v, h16, h8, l8);
  *reinterpret_cast<uint2*>(p16 + plane_off(row, c, ld, rows)) = h16;
  char* b = reinterpret_cast<char*>(p8 + plane_off(row, c & ~31, ld, rows)) + 2 * (c & 16) + (c & 15);
  *reinterpret_cast<uint32_t*>(b) = h8;
  *reinterpret_cast<uint32_t*>(b + 16) = l8;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
