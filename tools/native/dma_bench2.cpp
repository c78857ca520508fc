// dma_bench2 — the L2 -> LDS operand stream as a function of BYTES IN FLIGHT per CU (round 6).
// dma_bench (round 3) replays the k-loop's request pattern with ONE k-tile in flight and a full drain per k-tile, so what it measures is
// (bytes in flight) / (loaded latency), not what the L2s / the fabric can deliver.  Here every workgroup (8 waves, one per CU, 256 workgroups)
// keeps S - 1 k-tiles of PA + PW pieces (1 KiB each) per wave in flight in an S-stage ring with a COUNTED vmcnt (never 0 inside the loop) and a
// raw s_barrier per k-tile; no MFMAs, no fragment reads.
//   A pieces: rows of the workgroup's own 256-row tile of a row-major (rows, C) bf16 plane (16 rows x 64 B per piece), re-read for nine "taps"
//             (the 3x3 convolution's activation stream: L2 / Infinity-Cache traffic);
//   W pieces: k-blocked (K/32, N, 32) planes, the same for every workgroup (L2 hits).
// Prints us per k-tile, GB/s per CU, TB/s over the chip, and the shader clock.      Build: make -C tools/native dma_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PA, int PW, int S>
__global__ void __launch_bounds__(512) dma_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ w, int C, int N, int nk,
                                                  unsigned long long* __restrict__ times) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int P = PA + PW, STAGE = 8 * P * 1024;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane >> 2;
  const uint32_t lc16 = (uint32_t)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
  const uint64_t ap = reinterpret_cast<uint64_t>(a), wp = reinterpret_cast<uint64_t>(w);
  const u32x4 rs_a = {(uint32_t)ap, (uint32_t)(ap >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
  const u32x4 rs_w = {(uint32_t)wp, (uint32_t)(wp >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  uint32_t avoff[PA > 0 ? PA : 1], bvoff[PW > 0 ? PW : 1];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = blockIdx.x * (8 * PA * 16) + 16 * (wv + 8 * i) + lr;
    avoff[i] = (uint32_t)row * (uint32_t)C * 2u + lc16;
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int n = (16 * (wv + 8 * i) + lr) % N;
    bvoff[i] = (uint32_t)n * 64u + lc16;
  }
  auto fill = [&](int stage, int kt) {
    const uint32_t so_a = (uint32_t)((kt % (C / 32)) * 64);
    const uint32_t so_w = (uint32_t)kt * (uint32_t)N * 64u;
    const uint32_t l0 = lds0 + stage * STAGE + wv * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i)
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(l0 + i * 8192), "v"(avoff[i]), "s"(rs_a), "s"(so_a) : "memory");
#pragma unroll
    for (int i = 0; i < PW; ++i)
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(l0 + (PA + i) * 8192), "v"(bvoff[i]), "s"(rs_w), "s"(so_w) : "memory");
  };
  uint32_t sink = 0;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int s = 0; s < S - 1; ++s) fill(s, s);
  int st_rd = 0, st_wr = S - 1;
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    // the oldest of the S - 1 tiles in flight has landed; the S - 2 younger ones may still be on their way
    if (kt + S - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P * (S - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + S - 1 < nk) fill(st_wr, kt + S - 1);
    if (lane == 0 && wv == 0) { uint32_t tok; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tok) : "v"((uint32_t)(lds0 + st_rd * STAGE))); sink += tok; }
    st_rd = st_rd + 1 == S ? 0 : st_rd + 1;
    st_wr = st_wr + 1 == S ? 0 : st_wr + 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { times[blockIdx.x * 4] = t0; times[blockIdx.x * 4 + 1] = t1 + (sink == 0x12345u ? 1 : 0); times[blockIdx.x * 4 + 2] = c1 - c0; }
}

template <int PA, int PW, int S>
static void run(const uint16_t* a, const uint16_t* w, int C, int N, int nk, unsigned long long* times, const char* what) {
  constexpr int P = PA + PW;
  const size_t lds = (size_t)S * 8 * P * 1024;
  if (lds > 160 * 1024) return;
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<PA, PW, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double best = 1e30, mhz = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((dma_kernel<PA, PW, S>), dim3(256), dim3(512), lds, 0, a, w, C, N, nk, times);
    HIP_OK(hipDeviceSynchronize());
    std::vector<unsigned long long> t(1024);
    HIP_OK(hipMemcpy(t.data(), times, 1024 * 8, hipMemcpyDeviceToHost));
    double s = 0, c = 0;
    for (int i = 0; i < 256; ++i) { s += (t[4 * i + 1] - t[4 * i]) * 0.01; c += (double)t[4 * i + 2]; }
    if (s / 256 < best) { best = s / 256; mhz = c / s; }
  }
  const double bytes = 8.0 * P * 1024;
  printf("%-28s PA=%d PW=%d stages=%d: %3d KB per k-tile, %3d KB in flight per CU: %6.3f us per k-tile, %6.1f GB/s per CU, %5.2f TB/s chip, clock %.0f MHz\n", what, PA, PW, S,
         (int)(bytes / 1024), (int)(bytes * (S - 1) / 1024), best / nk, bytes / (best / nk) / 1e3, bytes * 256 / (best / nk) / 1e6, mhz);
  fflush(stdout);
}

int main() {
  // activation plane: 256 workgroups x up to 512 rows x C = 320 channels (bf16) = 84 MB; weights (K/32, 320, 32): nk k-tiles x 20 KB
  const int C = 320, N = 320, nk = 360;
  uint16_t *a, *w; unsigned long long* times;
  const size_t abytes = (size_t)256 * 1024 * C * 2 + 65536, wbytes = (size_t)nk * N * 64 + 65536;
  HIP_OK(hipMalloc(&a, abytes)); HIP_OK(hipMalloc(&w, wbytes)); HIP_OK(hipMalloc(&times, 1024 * 8));
  HIP_OK(hipMemset(a, 1, abytes)); HIP_OK(hipMemset(w, 1, wbytes));
  // the shipped tall f16mx tile's mix (per wave 4 A + 5 W pieces): one tile in flight = today's loop
  run<4, 5, 2>(a, w, C, N, nk, times, "tall mix (4A+5W)");
  // half k-tiles of the same mix, 2..4 stages
  run<2, 3, 2>(a, w, C, N, nk, times, "half tall mix (2A+3W)");
  run<2, 3, 3>(a, w, C, N, nk, times, "half tall mix (2A+3W)");
  run<2, 3, 4>(a, w, C, N, nk, times, "half tall mix (2A+3W)");
  // activations only / weights only, depth sweep at 16 KB per k-tile
  run<2, 0, 2>(a, w, C, N, nk, times, "activations only");
  run<2, 0, 3>(a, w, C, N, nk, times, "activations only");
  run<2, 0, 5>(a, w, C, N, nk, times, "activations only");
  run<2, 0, 8>(a, w, C, N, nk, times, "activations only");
  run<2, 0, 10>(a, w, C, N, nk, times, "activations only");
  run<0, 2, 2>(a, w, C, N, nk, times, "weights only (shared)");
  run<0, 2, 3>(a, w, C, N, nk, times, "weights only (shared)");
  run<0, 2, 5>(a, w, C, N, nk, times, "weights only (shared)");
  run<0, 2, 8>(a, w, C, N, nk, times, "weights only (shared)");
  run<0, 2, 10>(a, w, C, N, nk, times, "weights only (shared)");
  // the halo loader's mix: weights every k-tile, activations amortised over nine taps (~ 0.7 A pieces per k-tile: here 1 A + 5 W)
  run<1, 5, 2>(a, w, C, N, nk, times, "halo mix (1A+5W)");
  run<1, 5, 3>(a, w, C, N, nk, times, "halo mix (1A+5W)");
  run<0, 5, 2>(a, w, C, N, nk, times, "weights only, 40 KB tiles");
  run<0, 5, 3>(a, w, C, N, nk, times, "weights only, 40 KB tiles");
  run<0, 5, 4>(a, w, C, N, nk, times, "weights only, 40 KB tiles");
  HIP_OK(hipFree(a)); HIP_OK(hipFree(w)); HIP_OK(hipFree(times));
  return 0;
}
