// mfma_mix_bench — matrix-pipe cost of one (32x32 block, 32 k) unit of an fp32-class product under different operand splittings,
// at the register level (no memory traffic): 2 waves per SIMD, 10 accumulator blocks per wave (the 256x320 tile's wave), random operands.
//   V0  bf16x3          : a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, bf16 32x32x16                      -> 6 MFMAs per unit
//   V1  f16 + MX fp8    : a_h*b_h in f16 32x32x16 (2 MFMAs) + [a_h8 | a_l8] x [b_l8 ; b_h8] as ONE scaled 32x32x64 fp8 MFMA
//   V2  f16 + MX fp6    : the same with the cross terms in fp6 (e2m3)
//   V3  f16 x1          : a_h*b_h only (lower bound of the f16 part)
//   V4  bf16 x1         : single pass (the reference's TPU default precision)
// Prints ns and shader cycles per unit per SIMD, the sustained clock, and the equivalent algorithmic TFLOP/s of the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline uint32_t hsh(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int V>
__global__ void __launch_bounds__(512) mix_kernel(int iters, float* __restrict__ out, unsigned long long* __restrict__ times) {
  constexpr int NB = 10;                       // accumulator blocks per wave (2 x 5 of the 64x160 wave tile)
  const uint32_t t = blockIdx.x * 512 + threadIdx.x;
  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // operand fragments: 2 row fragments (A) and 5 column fragments (B), each in every format a variant needs; random bits, finite values
  u32x4 a16[2][2], b16[5][2];                  // [frag][plane]: 8 x 16-bit values (bf16 hi/lo or f16 hi)
  i32x8 a8[2], b8[5];                          // 32 x 8-bit (or 6-bit in the low 24 bytes) values
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) a16[i][p][e] = hsh(t * 131u + i * 17u + p * 5u + e) & 0xBFFFBFFFu;
#pragma unroll
    for (int e = 0; e < 8; ++e) a8[i][e] = (int)(hsh(t * 977u + i * 29u + e) & 0xBFBFBFBFu);
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) b16[j][p][e] = hsh(t * 313u + j * 19u + p * 7u + e + 99u) & 0xBFFFBFFFu;
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[j][e] = (int)(hsh(t * 611u + j * 23u + e + 7u) & 0xBFBFBFBFu);
  }
  const int sc = 127;                          // E8M0 scale 2^0
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {           // one "k-tile" of 32 = 2 x K16 for the 16-bit parts
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          f32x16& c = acc[i * 5 + j];
          if constexpr (V == 0) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16[i][1]), __builtin_bit_cast(bf16x8, b16[j][0]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16[i][0]), __builtin_bit_cast(bf16x8, b16[j][1]), c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16[i][0]), __builtin_bit_cast(bf16x8, b16[j][0]), c, 0, 0, 0);
          } else if constexpr (V == 4) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16[i][0]), __builtin_bit_cast(bf16x8, b16[j][0]), c, 0, 0, 0);
          } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a16[i][0]), __builtin_bit_cast(f16x8, b16[j][0]), c, 0, 0, 0);
            if (ks == 1) {
              if constexpr (V == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], c, 0, 0, 0, sc, 0, sc);
              if constexpr (V == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], c, 2, 2, 0, sc, 0, sc);
            }
          }
        }
      // keep the operands "fresh" so nothing is hoisted: rotate one register per fragment (cheap VALU, as a fragment reload would be)
      a16[ks & 1][0][0] ^= (uint32_t)it & 0x3F003Fu;
      b8[ks][0] ^= it & 0x0F0F0F0F;
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[t] = s;
  if (threadIdx.x == 0) { times[blockIdx.x * 2] = c1 - c0; times[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int V>
static void run(const char* name, int iters, float* out, unsigned long long* times, double flop_units_mfma) {
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mix_kernel<V>, dim3(256), dim3(512), 0, 0, iters / 10, out, times);
  HIP_OK(hipDeviceSynchronize());
  double best = 1e30, clk = 0, cyc = 0;
  for (int rep = 0; rep < 3; ++rep) {
    HIP_OK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mix_kernel<V>, dim3(256), dim3(512), 0, 0, iters, out, times);
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(512);
    HIP_OK(hipMemcpy(t.data(), times, 512 * 8, hipMemcpyDeviceToHost));
    double c = 0, r = 0;
    for (int i = 0; i < 256; ++i) { c += (double)t[2 * i]; r += (double)t[2 * i + 1]; }
    if (ms < best) { best = ms; clk = c / (r * 0.01); cyc = c / 256; }
  }
  // units: per iteration a wave does 10 blocks x (K = 32): per SIMD 2 waves -> 20 units per iteration
  const double units = (double)iters * 20.0;
  const double alg_flops = 2.0 * 32 * 32 * 32 * (double)iters * 10.0 * 8.0 * 256.0;        // algorithmic flops of the launch (8 waves x 256 CUs)
  printf("%-18s %7.3f ms | %6.1f cycles and %6.2f ns per (block, 32 k) unit per SIMD | clock %4.0f MHz | %6.1f algorithmic TFLOP/s\n", name, best,
         cyc / units, best * 1e6 / units, clk, alg_flops / (best * 1e-3) / 1e12);
  (void)flop_units_mfma;
}

int main() {
  float* out; unsigned long long* times;
  HIP_OK(hipMalloc(&out, 256 * 512 * 4)); HIP_OK(hipMalloc(&times, 512 * 8));
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("bf16x3", iters, out, times, 6);
    run<1>("f16 + MX fp8", iters, out, times, 3);
    run<2>("f16 + MX fp6", iters, out, times, 3);
    run<3>("f16 x1", iters, out, times, 2);
    run<4>("bf16 x1", iters, out, times, 2);
  }
  return 0;
}
