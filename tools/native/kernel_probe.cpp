// kernel_probe — torch-free micro-benchmark + spot-check of libddpo_hip.so through its C ABI (include/ddpo_hip.h).
//
// Starts in milliseconds (no Python, no `import torch`), so a whole A/B sweep of the GEMM tuning knobs fits into a minute
// of GPU-box time.  Inputs are a counter-based hash evaluated identically on host and device: nothing is copied to the
// device, and the host reference (double precision) is evaluated only on a few hundred sampled outputs per case.
//
//   kernel_probe gemm [batch=16] [iters=10]     bf16x3 implicit-GEMM conv / dense shapes of one SD-1.5 U-Net forward
//   kernel_probe gemm2 [batch=16] [iters=10]    plane-fed (LDS-DMA) kernel vs the fp32-fed one: timing + bitwise comparison
//                                               PROBE_COLD=1: flush the caches before every timed launch and re-read only the
//                                               activations (weights cold, as in the model)
//   kernel_probe vae [batch=8] [iters=5]        the VAE decoder's convolutions (512 x 512 output), fp32-fed vs plane-fed
//   kernel_probe mx [batch=16] [iters=10]       f16mx plane-fed datapath vs bf16x3 plane-fed: timing, contract check against the decoded
//                                               planes, accuracy against fp64
//   kernel_probe wgrad [batch=16] [iters=10]    bf16x3 weight gradients: wide 128x320 tile vs the 128x128 kernel, fp32 and plane operands
//   kernel_probe attn [batch=16] [iters=10]     bf16x3 / fp32 flash-attention shapes of the same forward
//   kernel_probe gelu                           gelu(tanh): libm tanhf form vs the v_exp / v_rcp sigmoid form of csrc/common.h (shipped since round 5) against float64, and their VALU cost
//   kernel_probe stream                         plain copy of 21 ... 336 MB: what the memory system gives the bytes of a short-reduction layer
//   kernel_probe ppo                            scoring-mode log-prob + PPO-clip + grouped micro-batches vs a host loop
//
// Build: make -C tools/native   (hipcc --offload-arch=gfx950; links ../../ddpo_amd/libddpo_hip.so)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/ddpo_hip.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)
#define ABI_OK(x)                                                       \
  do {                                                                  \
    int r_ = (x);                                                       \
    if (r_ != DDPO_OK) {                                                \
      fprintf(stderr, "%s:%d %s -> %d\n", __FILE__, __LINE__, #x, r_);  \
      exit(3);                                                          \
    }                                                                   \
  } while (0)

// value i of stream `seed`, uniform in [-1, 1): the same bits on host and device
__host__ __device__ inline float hval(uint32_t seed, uint64_t i) {
  uint32_t x = (uint32_t)i * 2654435761u ^ seed ^ ((uint32_t)(i >> 32) * 0x9E3779B9u);
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (float)(int32_t)x * (1.0f / 2147483648.0f);
}
__global__ void fill_kernel(float* p, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = scale * hval(seed, (uint64_t)i);
}

struct Dev {                      // device buffer filled with scale * hval(seed, i)
  float* p = nullptr;
  int64_t n = 0;
  uint32_t seed = 0;
  float scale = 1.f;
  Dev() {}
  Dev(int64_t n_, uint32_t seed_, float scale_) : n(n_), seed(seed_), scale(scale_) {
    HIP_OK(hipMalloc(&p, std::max<int64_t>(n, 4) * sizeof(float)));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, p, n, seed, scale);
    HIP_OK(hipGetLastError());
  }
  float at(int64_t i) const { return scale * hval(seed, (uint64_t)i); }
  void release() { if (p) HIP_OK(hipFree(p)); p = nullptr; }
};
static void* dalloc(size_t bytes) { void* p; HIP_OK(hipMalloc(&p, std::max<size_t>(bytes, 16))); HIP_OK(hipMemset(p, 0, std::max<size_t>(bytes, 16))); return p; }

static float time_ms(int iters, const std::function<void()>& fn) {
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) fn();
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) fn();
  HIP_OK(hipEventRecord(e1, 0));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
  return ms / iters;
}

// "In-model" timing (PROBE_COLD=1): before every timed launch the caches are flushed (768 MB streamed through L2 and the
// 256 MB Infinity Cache) and the ACTIVATION buffers are read back in — in the model the activations were just written by the
// previous kernel while the layer's weights come from HBM on every launch.  Only the launch itself is inside the event pair.
__global__ void warm_kernel(const uint4* p, int64_t n16, unsigned int* sink) {
  unsigned int acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
struct WarmBuf { const void* p; size_t bytes; };
static float time_cold_ms(int iters, const std::function<void()>& fn, const std::vector<WarmBuf>& warm) {
  static float* flush = nullptr;
  static unsigned int* sink = nullptr;
  const int64_t flush_n = (int64_t)768 << 18;          // 768 MB of floats
  if (!flush) { HIP_OK(hipMalloc(&flush, flush_n * 4)); HIP_OK(hipMalloc(&sink, 4)); }
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  fn();
  double total = 0.0;
  for (int i = 0; i < iters; ++i) {
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, flush, flush_n, 99u + i, 1.0f);
    for (const WarmBuf& w : warm) hipLaunchKernelGGL(warm_kernel, dim3(1024), dim3(256), 0, 0, (const uint4*)w.p, (int64_t)(w.bytes / 16), sink);
    HIP_OK(hipEventRecord(e0, 0));
    fn();
    HIP_OK(hipEventRecord(e1, 0));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    total += ms;
  }
  HIP_OK(hipEventDestroy(e0)); HIP_OK(hipEventDestroy(e1));
  return (float)(total / iters);
}
static bool probe_wkblk() { const char* e = getenv("PROBE_WKBLK"); return e && atoi(e) != 0; }     // k-blocked weight planes (w_layout = 1)
static void pack_w(const float* w, int K, int N, int Kp, uint16_t* hi, uint16_t* lo, ddpo_gemm_desc* d0, ddpo_gemm_desc* d1 = nullptr) {
  if (probe_wkblk()) {
    ABI_OK(ddpo_pack_weights_bf16_kblocked(w, K, N, hi, lo, nullptr));
    d0->w_layout = 1;
    if (d1) d1->w_layout = 1;
  } else {
    ABI_OK(ddpo_pack_weights_bf16(w, K, N, Kp, hi, lo, nullptr, nullptr, nullptr));
  }
}
static bool probe_cold() { const char* e = getenv("PROBE_COLD"); return e && atoi(e) != 0; }

// ------------------------------------------------------------------------------------------------ gemm / conv
struct ConvCase { int H, Cin, Cout, ks, stride, ups; };      // square H x H source, pad = ks / 2; ks == 0: dense with M = B*H rows
static int g_fail = 0;

static void run_gemm(int B, int H, int Cin, int Cout, int ks, int stride, int ups, int iters, void* ws, size_t ws_bytes) {
  const bool conv = ks > 0;
  const int pad = ks / 2;
  const int VH = ups ? 2 * H : H;
  const int OH = conv ? (VH + 2 * pad - ks) / stride + 1 : 0;
  const int64_t M = conv ? (int64_t)B * OH * OH : (int64_t)B * H;
  const int K = conv ? ks * ks * Cin : Cin, N = Cout, Kp = (K + 7) / 8 * 8;
  Dev src(conv ? (int64_t)B * H * H * Cin : M * K, 11, 1.0f), w((int64_t)K * N, 12, 1.0f / sqrtf((float)K)), bias(N, 13, 0.5f);
  float* out = (float*)dalloc((size_t)M * N * 4);
  uint16_t* hi = (uint16_t*)dalloc((size_t)N * ((Kp + 31) / 32 * 32) * 2);
  uint16_t* lo = (uint16_t*)dalloc((size_t)N * ((Kp + 31) / 32 * 32) * 2);
  ddpo_gemm_desc d;
  memset(&d, 0, sizeof(d));
  pack_w(w.p, K, N, Kp, hi, lo, &d);
  d.src = src.p; d.ld_src = conv ? Cin : K;
  d.bias = bias.p; d.out = out; d.ld_out = N; d.alpha = 1.f;
  d.M = (int)M; d.N = N; d.K = K;
  if (conv) { d.ksize = ks; d.stride = stride; d.pad = pad; d.upsample = ups; d.B = B; d.H = H; d.W = H; d.Cin = Cin; d.OH = OH; d.OW = OH; }
  const float ms = time_ms(iters, [&] { ABI_OK(ddpo_gemm_conv_fwd_bf16(&d, hi, lo, Kp, 3, ws, ws_bytes, nullptr)); });
  // spot check: 96 sampled outputs against a double-precision host evaluation of the same hashed inputs
  const int NS = 96;
  std::vector<float> got(NS);
  std::vector<int64_t> ms_(NS);
  std::vector<int> ns_(NS);
  double max_err = 0.0, ref_sq = 0.0;
  for (int s = 0; s < NS; ++s) {
    const int64_t m = (s < 8) ? (s < 4 ? s : M - 1 - (s - 4)) : (int64_t)((hval(77, s) * 0.5 + 0.5) * (double)M) % M;   // corners + random
    const int n = (s < 8) ? (s & 1 ? N - 1 - s : s) % N : (int)((hval(78, s) * 0.5 + 0.5) * N) % N;
    ms_[s] = m; ns_[s] = n;
    HIP_OK(hipMemcpy(&got[s], out + m * N + n, 4, hipMemcpyDeviceToHost));
    double acc = 0.0;
    if (conv) {
      const int64_t b = m / ((int64_t)OH * OH), rem = m % ((int64_t)OH * OH);
      const int oy = (int)(rem / OH), ox = (int)(rem % OH);
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
          const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
          if (iy < 0 || iy >= VH || ix < 0 || ix >= VH) continue;
          const int sy = ups ? iy >> 1 : iy, sx = ups ? ix >> 1 : ix;
          const int64_t pix = (b * H + sy) * H + sx;
          for (int ci = 0; ci < Cin; ++ci)
            acc += (double)src.at(pix * Cin + ci) * (double)w.at((int64_t)((ky * ks + kx) * Cin + ci) * N + n);
        }
    } else {
      for (int k = 0; k < K; ++k) acc += (double)src.at(m * K + k) * (double)w.at((int64_t)k * N + n);
    }
    acc += bias.at(n);
    max_err = std::max(max_err, fabs(acc - (double)got[s]));
    ref_sq += acc * acc;
  }
  const double rms = sqrt(ref_sq / NS), rel = max_err / (rms + 1e-30);
  const double tf = 2.0 * (double)M * N * K / (ms * 1e-3) / 1e12;
  const bool ok = rel < 2e-4;                       // bf16x3: ~1e-5 of the output scale
  if (!ok) ++g_fail;
  if (conv) printf("conv %dx%d s%d up%d %5d->%5d @%3d^2 B%-3d: %8.3f ms %7.1f TF  err/rms %.1e %s\n", ks, ks, stride, ups, Cin, Cout, H, B, ms, tf, rel, ok ? "" : "FAIL");
  else printf("gemm M=%7lld K=%5d N=%5d       : %8.3f ms %7.1f TF  err/rms %.1e %s\n", (long long)M, K, N, ms, tf, rel, ok ? "" : "FAIL");
  fflush(stdout);
  src.release(); w.release(); bias.release();
  HIP_OK(hipFree(out)); HIP_OK(hipFree(hi)); HIP_OK(hipFree(lo));
}

static int probe_gemm(int B, int iters) {
  const size_t ws_bytes = 64u << 20;
  void* ws = dalloc(ws_bytes);
  // the convolutions of an SD-1.5 U-Net forward at 64x64 latents (ResBlock convs per level, skip-concat inputs, down / up samplers)
  const ConvCase convs[] = {{64, 320, 320, 3, 1, 0},  {32, 640, 640, 3, 1, 0},   {16, 1280, 1280, 3, 1, 0}, {8, 1280, 1280, 3, 1, 0},
                            {64, 960, 320, 3, 1, 0},  {64, 640, 320, 3, 1, 0},   {32, 1920, 640, 3, 1, 0},  {32, 1280, 640, 3, 1, 0},
                            {32, 960, 640, 3, 1, 0},  {16, 2560, 1280, 3, 1, 0}, {16, 1920, 1280, 3, 1, 0}, {8, 2560, 1280, 3, 1, 0},
                            {32, 320, 640, 3, 1, 0},  {16, 640, 1280, 3, 1, 0},  {32, 640, 640, 3, 1, 1},   {16, 1280, 1280, 3, 1, 1},
                            {8, 1280, 1280, 3, 1, 1}, {64, 320, 320, 3, 2, 0},   {32, 640, 640, 3, 2, 0},   {16, 1280, 1280, 3, 2, 0},
                            {64, 320, 320, 1, 1, 0},  {64, 960, 320, 1, 1, 0},   {32, 1920, 640, 1, 1, 0},  {16, 2560, 1280, 1, 1, 0}};
  for (const ConvCase& c : convs) run_gemm(B, c.H, c.Cin, c.Cout, c.ks, c.stride, c.ups, iters, ws, ws_bytes);
  // dense layers: rows per sample x K x N (q/k/v/out projections, FF1 (unfused shape), FF2, cross-attention k/v, time embedding)
  const int dense[][3] = {{4096, 320, 320}, {4096, 320, 2560}, {4096, 1280, 320}, {1024, 640, 640}, {1024, 640, 5120}, {1024, 2560, 640},
                          {256, 1280, 1280}, {256, 1280, 10240}, {256, 5120, 1280}, {64, 1280, 1280}, {64, 1280, 10240}, {64, 5120, 1280},
                          {77, 768, 320}, {77, 768, 640}, {77, 768, 1280}, {1, 1280, 1280}};
  for (auto& g : dense) run_gemm(B, g[0], g[1], g[2], 0, 1, 0, iters, ws, ws_bytes);
  HIP_OK(hipFree(ws));
  return g_fail;
}

// ------------------------------------------------------------------------------------------------ plane-fed gemm / conv (LDS-DMA)
__global__ void diff_kernel(const float* a, const float* b, int64_t n, unsigned long long* cnt, float* maxabs) {
  unsigned long long c = 0;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (__float_as_uint(a[i]) != __float_as_uint(b[i])) { ++c; m = fmaxf(m, fabsf(a[i] - b[i])); }
  }
  if (c) { atomicAdd(cnt, c); atomicMax(reinterpret_cast<unsigned int*>(maxabs), __float_as_uint(m)); }
}

// fp32-fed kernel vs plane-fed kernel on the same layer: timing of both, bitwise comparison of the two outputs
static int g_npass = 3;      // mode x1: single-pass bf16 (both lo planes NULL at the plane-fed entry, npass = 1 at the fp32-fed one)
static void run_gemm2(int B, int H, int Cin, int Cout, int ks, int stride, int ups, int iters, void* ws, size_t ws_bytes) {
  const bool conv = ks > 0;
  const int pad = ks / 2;
  const int VH = ups ? 2 * H : H;
  const int OH = conv ? (VH + 2 * pad - ks) / stride + 1 : 0;
  const int64_t M = conv ? (int64_t)B * OH * OH : (int64_t)B * H;
  const int K = conv ? ks * ks * Cin : Cin, N = Cout, Kp = (K + 7) / 8 * 8;
  const int64_t arows = conv ? (int64_t)B * H * H : M;
  const int acols = conv ? Cin : K;
  Dev src(arows * acols, 11, 1.0f), w((int64_t)K * N, 12, 1.0f / sqrtf((float)K)), bias(N, 13, 0.5f), res(M * N, 14, 1.0f);
  float* out1 = (float*)dalloc((size_t)M * N * 4);
  float* out2 = (float*)dalloc((size_t)M * N * 4);
  uint16_t *hi = (uint16_t*)dalloc((size_t)N * ((Kp + 31) / 32 * 32) * 2), *lo = (uint16_t*)dalloc((size_t)N * ((Kp + 31) / 32 * 32) * 2);
  uint16_t *ah = (uint16_t*)dalloc((size_t)arows * acols * 2), *al = (uint16_t*)dalloc((size_t)arows * acols * 2);
  ABI_OK(ddpo_split_planes_bf16(src.p, acols, ah, al, acols, arows, acols, nullptr));
  ddpo_gemm_desc d;
  memset(&d, 0, sizeof(d));
  pack_w(w.p, K, N, Kp, hi, lo, &d);
  d.src = src.p; d.ld_src = acols;
  d.bias = bias.p; d.residual = res.p; d.ld_res = N; d.ld_out = N; d.alpha = 1.f;
  d.M = (int)M; d.N = N; d.K = K;
  if (conv) { d.ksize = ks; d.stride = stride; d.pad = pad; d.upsample = ups; d.B = B; d.H = H; d.W = H; d.Cin = Cin; d.OH = OH; d.OW = OH; }
  ddpo_gemm_desc d1 = d, d2 = d;
  d1.out = out1; d2.out = out2;
  const bool cold = probe_cold();
  const std::vector<WarmBuf> warm1 = {{src.p, (size_t)arows * acols * 4}, {res.p, (size_t)M * N * 4}};
  const std::vector<WarmBuf> warm2 = {{ah, (size_t)arows * acols * 2}, {al, (size_t)arows * acols * 2}, {res.p, (size_t)M * N * 4}};
  auto f1 = [&] { ABI_OK(ddpo_gemm_conv_fwd_bf16(&d1, hi, lo, Kp, g_npass, ws, ws_bytes, nullptr)); };
  const float ms1 = cold ? time_cold_ms(iters, f1, warm1) : time_ms(iters, f1);
  if (g_npass == 1) { HIP_OK(hipFree(al)); HIP_OK(hipFree(lo)); al = nullptr; lo = nullptr; }      // the single pass must not touch them
  const int rc = ddpo_gemm_conv_fwd_bf16_planes(&d2, ah, al, acols, hi, lo, Kp, ws, ws_bytes, nullptr);
  if (rc != DDPO_OK) {
    printf("%s K=%d N=%d: plane-fed entry returned %d (layer stays on the fp32-fed kernel)\n", conv ? "conv" : "gemm", K, N, rc);
  } else {
    auto f2 = [&] { ABI_OK(ddpo_gemm_conv_fwd_bf16_planes(&d2, ah, al, acols, hi, lo, Kp, ws, ws_bytes, nullptr)); };
    const float ms2 = cold ? time_cold_ms(iters, f2, warm2) : time_ms(iters, f2);
    unsigned long long* cnt = (unsigned long long*)dalloc(8);
    float* mx = (float*)dalloc(4);
    hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, out1, out2, M * N, cnt, mx);
    unsigned long long cnt_h = 0; float mx_h = 0.f;
    HIP_OK(hipMemcpy(&cnt_h, cnt, 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&mx_h, mx, 4, hipMemcpyDeviceToHost));
    const double fl = 2.0 * (double)M * N * K;
    if (cnt_h) ++g_fail;
    if (conv) printf("conv %dx%d s%d up%d %5d->%5d @%3d^2 B%-3d:", ks, ks, stride, ups, Cin, Cout, H, B);
    else printf("gemm M=%7lld K=%5d N=%5d       :", (long long)M, K, N);
    printf(" fp32-fed %7.3f ms %6.1f TF | planes %7.3f ms %6.1f TF (x%.2f) | %llu of %lld outputs differ (max %.2e) %s\n", ms1, fl / ms1 / 1e9, ms2,
           fl / ms2 / 1e9, ms1 / ms2, cnt_h, (long long)(M * N), mx_h, cnt_h ? "FAIL" : "bit-identical");
    HIP_OK(hipFree(cnt)); HIP_OK(hipFree(mx));
  }
  fflush(stdout);
  src.release(); w.release(); bias.release(); res.release();
  HIP_OK(hipFree(out1)); HIP_OK(hipFree(out2)); HIP_OK(hipFree(hi)); HIP_OK(hipFree(ah));
  if (lo) HIP_OK(hipFree(lo));
  if (al) HIP_OK(hipFree(al));
}

// single-pass bf16 (BASELINE configs[4]'s dtype): fp32-fed NPASS = 1 kernel vs the plane-fed one (round 6), SD-1.5 at 64x64 and SD-2.1 at 96x96 latents
static int probe_x1(int B, int iters) {
  const size_t ws_bytes = 64u << 20;
  void* ws = dalloc(ws_bytes);
  g_npass = 1;
  setenv("PROBE_WKBLK", "1", 1);
  const ConvCase convs[] = {{64, 320, 320, 3, 1, 0}, {32, 640, 640, 3, 1, 0}, {16, 1280, 1280, 3, 1, 0}, {8, 1280, 1280, 3, 1, 0}, {64, 960, 320, 3, 1, 0},
                            {32, 640, 640, 3, 1, 1}, {64, 320, 320, 3, 2, 0}, {64, 320, 320, 1, 1, 0},
                            {96, 320, 320, 3, 1, 0}, {48, 640, 640, 3, 1, 0}, {24, 1280, 1280, 3, 1, 0}, {12, 1280, 1280, 3, 1, 0}, {96, 960, 320, 3, 1, 0},
                            {48, 1920, 640, 3, 1, 0}, {24, 2560, 1280, 3, 1, 0}, {48, 640, 640, 3, 1, 1}, {20, 64, 96, 3, 1, 0}};
  for (const ConvCase& c : convs) run_gemm2(B, c.H, c.Cin, c.Cout, c.ks, c.stride, c.ups, iters, ws, ws_bytes);
  const int dense[][3] = {{4096, 320, 320}, {4096, 320, 2560}, {4096, 1280, 320}, {1024, 640, 640}, {1024, 2560, 640}, {256, 1280, 1280}, {256, 5120, 1280},
                          {9216, 320, 320}, {9216, 320, 2560}, {9216, 1280, 320}, {2304, 640, 640}, {2304, 2560, 640}, {576, 1280, 1280}, {37, 96, 72}};
  for (auto& g : dense) run_gemm2(B, g[0], g[1], g[2], 0, 1, 0, iters, ws, ws_bytes);
  g_npass = 3;
  HIP_OK(hipFree(ws));
  return g_fail;
}

// the VAE decoder's convolutions at 512 x 512 output (batch = images per decode call): few channels, millions of rows
static int probe_vae(int B, int iters) {
  const size_t ws_bytes = 64u << 20;
  void* ws = dalloc(ws_bytes);
  const ConvCase convs[] = {{64, 512, 512, 3, 1, 0},  {128, 512, 512, 3, 1, 0}, {64, 512, 512, 3, 1, 1},  {256, 512, 256, 3, 1, 0},
                            {256, 256, 256, 3, 1, 0}, {128, 512, 512, 3, 1, 1}, {512, 256, 128, 3, 1, 0}, {512, 128, 128, 3, 1, 0},
                            {256, 256, 256, 3, 1, 1}, {256, 512, 256, 1, 1, 0}, {512, 256, 128, 1, 1, 0}};
  for (const ConvCase& c : convs) run_gemm2(B, c.H, c.Cin, c.Cout, c.ks, c.stride, c.ups, iters, ws, ws_bytes);
  HIP_OK(hipFree(ws));
  return 0;
}

static int probe_gemm2(int B, int iters) {
  const size_t ws_bytes = 64u << 20;
  void* ws = dalloc(ws_bytes);
  const ConvCase convs[] = {{64, 320, 320, 3, 1, 0}, {32, 640, 640, 3, 1, 0}, {16, 1280, 1280, 3, 1, 0}, {8, 1280, 1280, 3, 1, 0},
                            {64, 960, 320, 3, 1, 0}, {32, 1920, 640, 3, 1, 0}, {16, 2560, 1280, 3, 1, 0}, {32, 320, 640, 3, 1, 0},
                            {32, 640, 640, 3, 1, 1}, {64, 320, 320, 3, 2, 0},  {64, 320, 320, 1, 1, 0},   {20, 64, 96, 3, 1, 0}};
  // PROBE_ONLY=c<i> / d<i>: run a single convolution / dense case (PMC passes: one kernel pair per process)
  const char* only = getenv("PROBE_ONLY");
  const int only_i = only && only[0] ? atoi(only + 1) : -1;
  int ci = 0;
  for (const ConvCase& c : convs) {
    if (!only || (only[0] == 'c' && ci == only_i)) run_gemm2(B, c.H, c.Cin, c.Cout, c.ks, c.stride, c.ups, iters, ws, ws_bytes);
    ++ci;
  }
  const int dense[][3] = {{4096, 320, 320}, {4096, 320, 2560}, {4096, 1280, 320}, {1024, 640, 5120}, {1024, 2560, 640},
                          {256, 1280, 10240}, {256, 5120, 1280}, {64, 1280, 1280}, {77, 768, 320}, {1, 1280, 1280}, {37, 96, 72},
                          {1024, 640, 640}, {256, 1280, 1280}, {4096, 320, 960}};
  int di = 0;
  for (auto& g : dense) {
    if (!only || (only[0] == 'd' && di == only_i)) run_gemm2(B, g[0], g[1], g[2], 0, 1, 0, iters, ws, ws_bytes);
    ++di;
  }
  HIP_OK(hipFree(ws));
  return g_fail;
}


// ------------------------------------------------------------------------------------------------ f16mx plane-fed datapath
static double dec_f16(uint16_t b) {
  const int s = b >> 15, e = (b >> 10) & 31, m = b & 1023;
  double v = e == 0 ? ldexp((double)m, -24) : (e == 31 ? INFINITY : ldexp((double)(m | 1024), e - 25));
  return s ? -v : v;
}
static double dec_e5m2(uint8_t b) { return dec_f16((uint16_t)b << 8); }
static double dec_e4m3(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  double v = e == 0 ? ldexp((double)m, -9) : ldexp((double)(m | 8), e - 10);
  return s ? -v : v;
}
// bf16x3 plane-fed kernel vs f16mx plane-fed kernel on the same layer: timing of both; f16mx spot-checked (a) against the exact product
// of the DECODED planes (the kernel's contract: layouts, lane pairing, block scales — fp32-accumulation error only) and (b) against the
// double-precision product of the fp32 inputs (the datapath's accuracy)
static void run_mx(int B, int H, int Cin, int Cout, int ks, int stride, int ups, int iters, void* ws, size_t ws_bytes) {
  const bool conv = ks > 0;
  const int pad = ks / 2;
  const int VH = ups ? 2 * H : H;
  const int OH = conv ? (VH + 2 * pad - ks) / stride + 1 : 0;
  const int64_t M = conv ? (int64_t)B * OH * OH : (int64_t)B * H;
  const int K = conv ? ks * ks * Cin : Cin, N = Cout, Kb = (K + 31) / 32;
  const int64_t arows = conv ? (int64_t)B * H * H : M;
  const int acols = conv ? Cin : K;
  Dev src(arows * acols, 11, 1.0f), w((int64_t)K * N, 12, 1.0f / sqrtf((float)K)), bias(N, 13, 0.5f), res(M * N, 14, 1.0f);
  float* out1 = (float*)dalloc((size_t)M * N * 4);
  float* out2 = (float*)dalloc((size_t)M * N * 4);
  const size_t wpl = (size_t)Kb * N * 64, apl = (size_t)arows * acols * 2;
  uint16_t *hi = (uint16_t*)dalloc(wpl), *lo = (uint16_t*)dalloc(wpl), *w16 = (uint16_t*)dalloc(wpl), *w8 = (uint16_t*)dalloc(wpl);
  uint8_t* wsc = (uint8_t*)dalloc(N);
  uint16_t *ah = (uint16_t*)dalloc(apl), *al = (uint16_t*)dalloc(apl), *a16 = (uint16_t*)dalloc(apl), *a8 = (uint16_t*)dalloc(apl);
  ABI_OK(ddpo_split_planes_bf16(src.p, acols, ah, al, acols, arows, acols, nullptr));
  ABI_OK(ddpo_split_planes_f16mx(src.p, acols, a16, a8, acols, arows, acols, nullptr));
  ABI_OK(ddpo_pack_weights_bf16_kblocked(w.p, K, N, hi, lo, nullptr));
  ABI_OK(ddpo_pack_weights_f16mx(w.p, K, N, w16, w8, wsc, nullptr));
  ddpo_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.w_layout = 1;
  d.bias = bias.p; d.residual = res.p; d.ld_res = N; d.ld_out = N; d.alpha = 1.f;
  d.M = (int)M; d.N = N; d.K = K;
  if (conv) { d.ksize = ks; d.stride = stride; d.pad = pad; d.upsample = ups; d.B = B; d.H = H; d.W = H; d.Cin = Cin; d.OH = OH; d.OW = OH; }
  ddpo_gemm_desc d1 = d, d2 = d;
  d1.out = out1; d2.out = out2; d2.w_scale = wsc;
  auto f1 = [&] { ABI_OK(ddpo_gemm_conv_fwd_bf16_planes(&d1, ah, al, acols, hi, lo, 0, ws, ws_bytes, nullptr)); };
  auto f2 = [&] { ABI_OK(ddpo_gemm_conv_fwd_f16mx_planes(&d2, a16, a8, acols, w16, w8, ws, ws_bytes, nullptr)); };
  // f16mx tile routing (csrc/gemm_bf16.hip dispatch, DDPO_MX_TALL read per launch): 0 = 128-row tiles only, default = the 256x320 tall tile where
  // its grid fills the chip.  Timed back to back, and the outputs compared BIT FOR BIT over the whole tensor (same per-accumulator order
  // f16 ks 0, f16 ks 1, MX on both tiles).
  const char* keep = getenv("DDPO_MX_TALL");
  const std::string keep_s = keep ? keep : "";
  float ms_mode[2] = {0.f, 0.f};
  long long ndiff = 0;
  std::vector<float> h_ref((size_t)M * N), h_cmp((size_t)M * N);
  for (int mode = 0; mode < 2; ++mode) {
    setenv("DDPO_MX_TALL", mode == 0 ? "0" : "1", 1);
    HIP_OK(hipMemset(out2, 0xFF, (size_t)M * N * 4));
    ms_mode[mode] = time_ms(iters, f2);
    HIP_OK(hipMemcpy((mode == 0 ? h_ref : h_cmp).data(), out2, (size_t)M * N * 4, hipMemcpyDeviceToHost));
  }
  for (size_t q = 0; q < h_ref.size(); ++q) ndiff += memcmp(&h_ref[q], &h_cmp[q], 4) != 0;
  if (keep) setenv("DDPO_MX_TALL", keep_s.c_str(), 1); else unsetenv("DDPO_MX_TALL");
  const float ms1 = time_ms(iters, f1), ms2 = time_ms(iters, f2);
  printf("   f16mx, 128-row tiles only %.3f ms | with the tall-tile rule %.3f ms (x%.2f, %s)\n", ms_mode[0], ms_mode[1], ms_mode[0] / ms_mode[1],
         ndiff == 0 ? "bit-identical" : "DIFFERS");
  if (ndiff) ++g_fail;
  std::vector<uint16_t> h_a16(apl / 2), h_a8(apl / 2), h_w16(wpl / 2), h_w8(wpl / 2);
  std::vector<uint8_t> h_sc(N);
  HIP_OK(hipMemcpy(h_a16.data(), a16, apl, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(h_a8.data(), a8, apl, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_w16.data(), w16, wpl, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(h_w8.data(), w8, wpl, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_sc.data(), wsc, N, hipMemcpyDeviceToHost));
  const uint8_t* a8b = reinterpret_cast<const uint8_t*>(h_a8.data());
  const uint8_t* w8b = reinterpret_cast<const uint8_t*>(h_w8.data());
  const int NS = 64;
  double err_planes = 0.0, err_true = 0.0, err_b3 = 0.0, ref_sq = 0.0;
  for (int s = 0; s < NS; ++s) {
    const int64_t m = (s < 8) ? (s < 4 ? s : M - 1 - (s - 4)) : (int64_t)((hval(77, s) * 0.5 + 0.5) * (double)M) % M;
    const int n = (s < 8) ? (s & 1 ? N - 1 - s : s) % N : (int)((hval(78, s) * 0.5 + 0.5) * N) % N;
    float got = 0.f, got3 = 0.f;
    HIP_OK(hipMemcpy(&got, out2 + m * N + n, 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&got3, out1 + m * N + n, 4, hipMemcpyDeviceToHost));
    double acc_p = 0.0, acc_t = 0.0;
    const double sn = ldexp(1.0, (int)h_sc[n] - 127);
    auto term = [&](int64_t arow, int ci, int k) {            // activation (arow, ci) x weight (k, n)
      const int64_t ao = arow * acols + ci, wo = ((int64_t)(k >> 5) * N + n);
      const double a_h = dec_f16(h_a16[ao]), w_h = dec_f16(h_w16[wo * 32 + (k & 31)]);
      const int64_t ab = (arow * acols + (ci & ~31)) * 2 + 2 * (ci & 16) + (ci & 15);       // chunks [h8 | l8 | h8 | l8]
      const double a_h8 = dec_e5m2(a8b[ab]), a_l8 = dec_e5m2(a8b[ab + 16]) / 2048.0;
      const int wq = 2 * (k & 16) + (k & 15);                                              // chunks [l8 | h8 | l8 | h8]
      const double w_l8 = dec_e4m3(w8b[wo * 64 + wq]) * sn / 2048.0, w_h8 = dec_e4m3(w8b[wo * 64 + wq + 16]) * sn;
      acc_p += a_h * w_h + a_h8 * w_l8 + a_l8 * w_h8;
      acc_t += (double)src.at(ao) * (double)w.at((int64_t)k * N + n);
    };
    if (conv) {
      const int64_t b = m / ((int64_t)OH * OH), rem = m % ((int64_t)OH * OH);
      const int oy = (int)(rem / OH), ox = (int)(rem % OH);
      for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
          const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
          if (iy < 0 || iy >= VH || ix < 0 || ix >= VH) continue;
          const int sy = ups ? iy >> 1 : iy, sx = ups ? ix >> 1 : ix;
          const int64_t pix = (b * H + sy) * H + sx;
          for (int ci = 0; ci < Cin; ++ci) term(pix, ci, (ky * ks + kx) * Cin + ci);
        }
    } else {
      for (int k = 0; k < K; ++k) term(m, k, k);
    }
    const double add = (double)bias.at(n) + (double)res.at(m * N + n);
    err_planes = std::max(err_planes, fabs(acc_p + add - (double)got));
    err_true = std::max(err_true, fabs(acc_t + add - (double)got));
    err_b3 = std::max(err_b3, fabs(acc_t + add - (double)got3));
    ref_sq += acc_t * acc_t;
  }
  const double rms = sqrt(ref_sq / NS);
  const double fl = 2.0 * (double)M * N * K;
  const bool ok = err_planes / rms < 1e-5 && err_true / rms < 3e-4;      // fp32 accumulation over up to 17280 terms; the datapath's ~5e-5
  if (!ok) ++g_fail;
  if (conv) printf("conv %dx%d s%d up%d %5d->%5d @%3d^2 B%-3d:", ks, ks, stride, ups, Cin, Cout, H, B);
  else printf("gemm M=%7lld K=%5d N=%5d       :", (long long)M, K, N);
  printf(" bf16x3 %7.3f ms %6.1f TF | f16mx %7.3f ms %6.1f TF (x%.2f) | f16mx vs decoded planes %.1e, vs fp64 %.1e (bf16x3 %.1e) of rms %s\n", ms1,
         fl / ms1 / 1e9, ms2, fl / ms2 / 1e9, ms1 / ms2, err_planes / rms, err_true / rms, err_b3 / rms, ok ? "" : "FAIL");
  fflush(stdout);
  src.release(); w.release(); bias.release(); res.release();
  for (void* q : {(void*)out1, (void*)out2, (void*)hi, (void*)lo, (void*)w16, (void*)w8, (void*)wsc, (void*)ah, (void*)al, (void*)a16, (void*)a8}) HIP_OK(hipFree(q));
}

static int probe_mx(int B, int iters) {
  const size_t ws_bytes = 64u << 20;
  void* ws = dalloc(ws_bytes);
  const ConvCase convs[] = {{64, 320, 320, 3, 1, 0}, {32, 640, 640, 3, 1, 0}, {16, 1280, 1280, 3, 1, 0}, {8, 1280, 1280, 3, 1, 0},
                            {64, 960, 320, 3, 1, 0}, {32, 1920, 640, 3, 1, 0}, {16, 2560, 1280, 3, 1, 0}, {32, 320, 640, 3, 1, 0},
                            {32, 640, 640, 3, 1, 1}, {64, 320, 320, 3, 2, 0},  {64, 320, 320, 1, 1, 0},   {20, 64, 96, 3, 1, 0}};
  for (const ConvCase& c : convs) run_mx(B, c.H, c.Cin, c.Cout, c.ks, c.stride, c.ups, iters, ws, ws_bytes);
  const int dense[][3] = {{4096, 320, 320}, {4096, 320, 2560}, {4096, 1280, 320}, {1024, 640, 5120}, {1024, 2560, 640},
                          {256, 1280, 10240}, {256, 5120, 1280}, {64, 1280, 1280}, {77, 768, 320}, {1, 1280, 1280}, {37, 96, 72},
                          {1024, 640, 640}, {256, 1280, 1280}, {4096, 320, 960}};
  for (auto& g : dense) run_mx(B, g[0], g[1], g[2], 0, 1, 0, iters, ws, ws_bytes);
  HIP_OK(hipFree(ws));
  return g_fail;
}


// ------------------------------------------------------------------------------------------------ weight gradient (bf16x3)
// dW[k][n] = sum_m A(m, k) dY[m][n]: the wide 128x320 tile (splits = 0: the library chooses) against the 128x128 kernel (explicit split count of
// its own heuristic), fp32 operands and plane operands; spot check against a double-precision host sum
static int old_wgrad_splits(int M, int K, int N) {
  const int tiles = ((K + 127) / 128) * ((N + 127) / 128), max_splits = (M + 255) / 256;
  int best = 1;
  double best_eff = 0.0;
  for (int r = 1; r <= 4; ++r) {
    int cand = std::max(1, std::min((512 * r) / tiles, max_splits));
    const long wgs = (long)tiles * cand;
    const double eff = (double)wgs / (double)(((wgs + 511) / 512) * 512);
    if (eff > best_eff + 0.03 || best_eff == 0.0) { best_eff = eff; best = cand; }
  }
  return best;
}
static void run_wgrad(int B, int H, int Cin, int Cout, int ks, int iters) {
  const bool conv = ks > 0;
  const int pad = ks / 2;
  const int64_t M = conv ? (int64_t)B * H * H : (int64_t)B * H;
  const int K = conv ? ks * ks * Cin : Cin, N = Cout;
  const int acols = conv ? Cin : K;
  Dev src(M * acols, 21, 1.0f), dy(M * N, 22, 1.0f);
  float* dw = (float*)dalloc((size_t)K * N * 4);
  uint16_t *ah = (uint16_t*)dalloc((size_t)M * acols * 2), *al = (uint16_t*)dalloc((size_t)M * acols * 2);
  uint16_t *bh = (uint16_t*)dalloc((size_t)M * N * 2), *bl = (uint16_t*)dalloc((size_t)M * N * 2);
  ABI_OK(ddpo_split_planes_bf16(src.p, acols, ah, al, acols, M, acols, nullptr));
  ABI_OK(ddpo_split_planes_bf16(dy.p, N, bh, bl, N, M, N, nullptr));
  ddpo_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.src = src.p; d.ld_src = acols; d.w = dy.p; d.ld_w = N; d.out = dw; d.ld_out = N; d.alpha = 1.f;
  d.M = (int)M; d.N = N; d.K = K;
  if (conv) { d.ksize = ks; d.stride = 1; d.pad = pad; d.B = B; d.H = H; d.W = H; d.Cin = Cin; d.OH = H; d.OW = H; }
  ddpo_gemm_desc d_old = d;
  d_old.splits = old_wgrad_splits((int)M, K, N);
  const double fl = 2.0 * (double)M * N * K;
  float ms[4];
  ms[0] = time_ms(iters, [&] { ABI_OK(ddpo_gemm_conv_wgrad_bf16x3(&d_old, nullptr)); });
  ms[1] = time_ms(iters, [&] { ABI_OK(ddpo_gemm_conv_wgrad_bf16x3(&d, nullptr)); });
  ms[2] = time_ms(iters, [&] { ABI_OK(ddpo_gemm_conv_wgrad_bf16x3_planes(&d_old, ah, al, bh, bl, nullptr)); });
  ms[3] = time_ms(iters, [&] { ABI_OK(ddpo_gemm_conv_wgrad_bf16x3_planes(&d, ah, al, bh, bl, nullptr)); });
  double rel[2] = {0, 0};
  for (int v = 0; v < 2; ++v) {              // one clean accumulation each: fp32-fed, plane-fed (library's choice of kernel)
    HIP_OK(hipMemset(dw, 0, (size_t)K * N * 4));
    if (v == 0) ABI_OK(ddpo_gemm_conv_wgrad_bf16x3(&d, nullptr));
    else ABI_OK(ddpo_gemm_conv_wgrad_bf16x3_planes(&d, ah, al, bh, bl, nullptr));
    const int NS = 24;
    double max_err = 0.0, ref_sq = 0.0;
    for (int s = 0; s < NS; ++s) {
      const int k = (s < 4) ? (s & 1 ? K - 1 - s : s) : (int)((hval(87, s) * 0.5 + 0.5) * K) % K;
      const int n = (s < 4) ? (s & 2 ? N - 1 - s : s) : (int)((hval(88, s) * 0.5 + 0.5) * N) % N;
      float got;
      HIP_OK(hipMemcpy(&got, dw + (int64_t)k * N + n, 4, hipMemcpyDeviceToHost));
      double acc = 0.0;
      const int tap = conv ? k / Cin : 0, ci = conv ? k % Cin : k;
      const int dky = conv ? tap / ks - pad : 0, dkx = conv ? tap % ks - pad : 0;
      for (int64_t m = 0; m < M; ++m) {
        int64_t am = m;
        if (conv) {
          const int x = (int)(m % H), y = (int)((m / H) % H);
          if (y + dky < 0 || y + dky >= H || x + dkx < 0 || x + dkx >= H) continue;
          am = m + (int64_t)dky * H + dkx;
        }
        acc += (double)src.at(am * acols + ci) * (double)dy.at(m * N + n);
      }
      max_err = std::max(max_err, fabs(acc - (double)got));
      ref_sq += acc * acc;
    }
    rel[v] = max_err / (sqrt(ref_sq / NS) + 1e-30);
  }
  const bool ok = rel[0] < 2e-4 && rel[1] < 2e-4;
  if (!ok) ++g_fail;
  if (conv) printf("wgrad conv %dx%d %5d->%5d @%3d^2 B%-3d:", ks, ks, Cin, Cout, H, B);
  else printf("wgrad gemm M=%7lld K=%5d N=%5d  :", (long long)M, K, N);
  printf(" fp32 128x128 %7.3f ms %6.1f TF -> auto %7.3f ms %6.1f TF (x%.2f) | planes %7.3f ms %6.1f TF -> auto %7.3f ms %6.1f TF (x%.2f) | err/rms %.1e %.1e %s\n",
         ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, ms[0] / ms[1], ms[2], fl / ms[2] / 1e9, ms[3], fl / ms[3] / 1e9, ms[2] / ms[3], rel[0], rel[1], ok ? "" : "FAIL");
  fflush(stdout);
  src.release(); dy.release();
  for (void* q : {(void*)dw, (void*)ah, (void*)al, (void*)bh, (void*)bl}) HIP_OK(hipFree(q));
}
static int probe_wgrad(int B, int iters) {
  const int convs[][4] = {{64, 320, 320, 3}, {32, 640, 640, 3}, {16, 1280, 1280, 3}, {8, 1280, 1280, 3}, {64, 960, 320, 3}, {32, 1920, 640, 3},
                          {16, 2560, 1280, 3}, {32, 320, 640, 3}, {64, 320, 320, 1}, {32, 960, 640, 1}};
  for (auto& c : convs) run_wgrad(B, c[0], c[1], c[2], c[3], iters);
  const int dense[][3] = {{4096, 320, 320}, {4096, 320, 2560}, {4096, 1280, 320}, {1024, 640, 640}, {1024, 2560, 640}, {256, 1280, 1280}, {256, 5120, 1280}};
  for (auto& g : dense) run_wgrad(B, g[0], g[1], g[2], 0, iters);
  return g_fail;
}

// ------------------------------------------------------------------------------------------------ phase timing (kernel_probe_timing)
#ifdef PROBE_TIMING
extern "C" int ddpo_debug_kloop_times(unsigned long long* host, int n_wg);
// One launch of one layer on the instrumented library; per workgroup: entry -> first k-loop barrier (prologue), k-loop, output stage,
// in microseconds (s_memrealtime, 100 MHz) and the shader clock seen over the whole workgroup (s_memtime ticks / us).
static void run_ktime(int B, int H, int Cin, int Cout, int ks, int planes, void* ws, size_t ws_bytes, bool mx = false) {
  const bool conv = ks > 0;
  const int pad = ks / 2, OH = conv ? H : 0;
  const int64_t M = conv ? (int64_t)B * OH * OH : (int64_t)B * H;
  const int K = conv ? ks * ks * Cin : Cin, N = Cout, Kp = (K + 7) / 8 * 8;
  const int64_t arows = conv ? (int64_t)B * H * H : M;
  const int acols = conv ? Cin : K;
  Dev src(arows * acols, 11, 1.0f), w((int64_t)K * N, 12, 1.0f / sqrtf((float)K)), bias(N, 13, 0.5f);
  float* out = (float*)dalloc((size_t)M * N * 4);
  uint16_t *hi = (uint16_t*)dalloc((size_t)N * ((Kp + 31) / 32 * 32) * 2), *lo = (uint16_t*)dalloc((size_t)N * ((Kp + 31) / 32 * 32) * 2);
  uint16_t *ah = (uint16_t*)dalloc((size_t)arows * acols * 2), *al = (uint16_t*)dalloc((size_t)arows * acols * 2);
  ABI_OK(ddpo_split_planes_bf16(src.p, acols, ah, al, acols, arows, acols, nullptr));
  ddpo_gemm_desc d;
  memset(&d, 0, sizeof(d));
  pack_w(w.p, K, N, Kp, hi, lo, &d);
  uint8_t* wsc = nullptr;
  if (mx) {                                   // f16mx operator on the same buffers (planes are the same size)
    wsc = (uint8_t*)dalloc(N);
    ABI_OK(ddpo_split_planes_f16mx(src.p, acols, ah, al, acols, arows, acols, nullptr));
    ABI_OK(ddpo_pack_weights_f16mx(w.p, K, N, hi, lo, wsc, nullptr));
    d.w_layout = 1; d.w_scale = wsc;
  }
  d.src = src.p; d.ld_src = acols; d.bias = bias.p; d.out = out; d.ld_out = N; d.alpha = 1.f;
  d.M = (int)M; d.N = N; d.K = K;
  if (conv) { d.ksize = ks; d.stride = 1; d.pad = pad; d.B = B; d.H = H; d.W = H; d.Cin = Cin; d.OH = OH; d.OW = OH; }
  auto launch = [&] {
    if (mx) ABI_OK(ddpo_gemm_conv_fwd_f16mx_planes(&d, ah, al, acols, hi, lo, ws, ws_bytes, nullptr));
    else if (planes) ABI_OK(ddpo_gemm_conv_fwd_bf16_planes(&d, ah, al, acols, hi, lo, Kp, ws, ws_bytes, nullptr));
    else ABI_OK(ddpo_gemm_conv_fwd_bf16(&d, hi, lo, Kp, 3, ws, ws_bytes, nullptr));
  };
  const std::vector<WarmBuf> warm = planes ? std::vector<WarmBuf>{{ah, (size_t)arows * acols * 2}, {al, (size_t)arows * acols * 2}}
                                           : std::vector<WarmBuf>{{src.p, (size_t)arows * acols * 4}};
  const float ms = probe_cold() ? time_cold_ms(3, launch, warm) : time_ms(3, launch);
  HIP_OK(hipDeviceSynchronize());
  const int NW = 16384;
  std::vector<unsigned long long> t((size_t)2 * NW * 8, 0ull);
  // poison, run once, read
  {
    std::vector<unsigned long long> z((size_t)NW * 8, 0ull);
    (void)z;
  }
  launch();
  HIP_OK(hipDeviceSynchronize());
  if (ddpo_debug_kloop_times(t.data(), NW) != 0) { printf("ktime: no timing symbol\n"); return; }
  // workgroups of the LAST launch: those whose entry stamp is within the last launch window (max entry - 1 ms)
  unsigned long long tmax = 0;
  for (int i = 0; i < NW; ++i) tmax = std::max(tmax, t[i * 8 + 4 + 3]);
  double sp = 0, sk = 0, se = 0, mp = 0, mk = 0, me = 0, clk = 0, wv = 0, wb = 0;
  unsigned long long first = ~0ull, last = 0;
  int n = 0;
  for (int i = 0; i < NW; ++i) {
    const unsigned long long* r = &t[i * 8 + 4];
    if (r[3] == 0 || tmax - r[3] > 100000ull) continue;     // older than 1 ms before the end: not this launch
    const double p = (r[1] - r[0]) * 0.01, k = (r[2] - r[1]) * 0.01, e = (r[3] - r[2]) * 0.01;
    sp += p; sk += k; se += e; mp = std::max(mp, p); mk = std::max(mk, k); me = std::max(me, e);
    if (r[3] > r[0]) clk += (double)(t[i * 8 + 3] - t[i * 8 + 0]) / ((r[3] - r[0]) * 0.01);
    wv += (double)t[(size_t)(NW + i) * 8 + 0]; wb += (double)t[(size_t)(NW + i) * 8 + 1];
    first = std::min(first, r[0]); last = std::max(last, r[3]);
    ++n;
  }
  if (conv) printf("ktime conv %dx%d %5d->%5d @%3d^2 B%-3d %s:", ks, ks, Cin, Cout, H, B, mx ? "f16mx " : planes ? "planes" : "fp32  ");
  else printf("ktime gemm M=%7lld K=%5d N=%5d %s:", (long long)M, K, N, planes ? "planes" : "fp32  ");
  const double mhz = clk / n;
  printf(" event %7.1f us | %5d WGs, span %7.1f us | prologue avg %6.1f max %6.1f | k-loop avg %7.1f max %7.1f (%d k-tiles: %.2f us each; wave 0 waits per k-tile: vmcnt/lgkm %.2f us + barrier %.2f us) | output avg %6.1f max %6.1f | clock %.0f MHz\n",
         ms * 1e3, n, (last - first) * 0.01, sp / n, mp, sk / n, mk, K / 32, sk / n / (K / 32), wv / n / mhz / (K / 32), wb / n / mhz / (K / 32), se / n, me, mhz);
  fflush(stdout);
  src.release(); w.release(); bias.release();
  HIP_OK(hipFree(out)); HIP_OK(hipFree(hi)); HIP_OK(hipFree(lo)); HIP_OK(hipFree(ah)); HIP_OK(hipFree(al));
  if (wsc) HIP_OK(hipFree(wsc));
}
// f16mx operator under the timing build (DDPO_DBG_ABL: 1 = no LDS-DMA inside the loop, 2 = no MFMAs, 4 = activation pieces on 2 of 9 k-tiles)
static int probe_ktmx(int B) {
  const size_t ws_bytes = 64u << 20;
  void* ws = dalloc(ws_bytes);
  const ConvCase convs[] = {{64, 320, 320, 3, 1, 0}, {64, 960, 320, 3, 1, 0}, {32, 640, 640, 3, 1, 0}, {32, 1920, 640, 3, 1, 0}, {16, 1280, 1280, 3, 1, 0},
                            {16, 2560, 1280, 3, 1, 0}, {8, 1280, 1280, 3, 1, 0}};
  for (const ConvCase& c : convs) run_ktime(B, c.H, c.Cin, c.Cout, c.ks, 1, ws, ws_bytes, true);
  HIP_OK(hipFree(ws));
  return 0;
}
static int probe_ktime(int B) {
  const size_t ws_bytes = 64u << 20;
  void* ws = dalloc(ws_bytes);
  const ConvCase convs[] = {{64, 320, 320, 3, 1, 0}, {32, 640, 640, 3, 1, 0}, {16, 1280, 1280, 3, 1, 0}, {64, 960, 320, 3, 1, 0}};
  for (const ConvCase& c : convs) for (int pl = 0; pl < 2; ++pl) run_ktime(B, c.H, c.Cin, c.Cout, c.ks, pl, ws, ws_bytes);
  const int dense[][3] = {{4096, 320, 320}, {4096, 1280, 320}, {1024, 640, 640}, {256, 1280, 1280}, {1024, 2560, 640}, {4096, 320, 960}};
  for (auto& g : dense) for (int pl = 0; pl < 2; ++pl) run_ktime(B, g[0], g[1], g[2], 0, pl, ws, ws_bytes);
  HIP_OK(hipFree(ws));
  return 0;
}
#endif

// ------------------------------------------------------------------------------------------------ attention
static void run_attn(int B, int heads, int Nq, int Nk, int d, int iters, bool f16p = false) {
  const int C = heads * d;
  Dev q((int64_t)B * Nq * C, 21, 1.0f), k((int64_t)B * Nk * C, 22, 1.0f), v((int64_t)B * Nk * C, 23, 1.0f);
  float* o = (float*)dalloc((size_t)B * Nq * C * 4);
  const float scale = 1.0f / sqrtf((float)d);
  const bool bf = (d == 8 || d == 16 || d == 40 || d == 64 || d == 80);
  const size_t wsb = bf ? ddpo_attention_fwd_bf16x3_ws_bytes(B, heads, Nk, d) : 0;
  void* ws = wsb ? dalloc(wsb) : nullptr;
  const float ms = time_ms(iters, [&] {
    if (bf && f16p) ABI_OK(ddpo_attention_fwd_f16p(q.p, C, k.p, C, v.p, C, o, C, nullptr, B, heads, Nq, Nk, d, scale, ws, wsb, nullptr));
    else if (bf) ABI_OK(ddpo_attention_fwd_bf16x3(q.p, C, k.p, C, v.p, C, o, C, nullptr, B, heads, Nq, Nk, d, scale, ws, wsb, nullptr));
    else ABI_OK(ddpo_attention_fwd(q.p, C, k.p, C, v.p, C, o, C, nullptr, B, heads, Nq, Nk, d, scale, nullptr));
  });
  double max_err = 0.0;
  std::vector<double> p(Nk);
  std::vector<float> got(d);
  for (int s = 0; s < 12; ++s) {
    const int b = s % B, h = (s * 3) % heads, iq = (s < 2) ? (s ? Nq - 1 : 0) : (int)((hval(31, s) * 0.5 + 0.5) * Nq) % Nq;
    const int64_t qo = ((int64_t)b * Nq + iq) * C + h * d;
    double mx = -1e300;
    for (int j = 0; j < Nk; ++j) {
      double sc = 0.0;
      const int64_t ko = ((int64_t)b * Nk + j) * C + h * d;
      for (int e = 0; e < d; ++e) sc += (double)q.at(qo + e) * (double)k.at(ko + e);
      p[j] = sc * scale;
      mx = std::max(mx, p[j]);
    }
    double den = 0.0;
    for (int j = 0; j < Nk; ++j) { p[j] = exp(p[j] - mx); den += p[j]; }
    HIP_OK(hipMemcpy(got.data(), o + qo, d * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < d; ++e) {
      double acc = 0.0;
      for (int j = 0; j < Nk; ++j) acc += p[j] * (double)v.at(((int64_t)b * Nk + j) * C + h * d + e);
      max_err = std::max(max_err, fabs(acc / den - (double)got[e]));
    }
  }
  const bool ok = max_err < 2e-4;
  if (!ok) ++g_fail;
  printf("attn B%-3d h%d Nq=%5d Nk=%5d d=%3d %s: %8.3f ms %7.1f TF  max abs err %.1e %s\n", B, heads, Nq, Nk, d, bf ? (f16p ? "f16p  " : "bf16x3") : "fp32  ", ms,
         4.0 * B * heads * (double)Nq * Nk * d / (ms * 1e-3) / 1e12, max_err, ok ? "" : "FAIL");
  fflush(stdout);
  q.release(); k.release(); v.release();
  HIP_OK(hipFree(o));
  if (ws) HIP_OK(hipFree(ws));
}

static int probe_attn(int B, int iters) {
  const int cases[][3] = {{4096, 4096, 40}, {4096, 77, 40}, {1024, 1024, 80}, {1024, 77, 80}, {256, 256, 160}, {256, 77, 160}, {64, 64, 160}, {64, 77, 160}};
  for (auto& c : cases) {
    run_attn(B, 8, c[0], c[1], c[2], iters);
    if (c[2] != 160) run_attn(B, 8, c[0], c[1], c[2], iters, true);      // the f16mx datapath's operator (one f16 probability term, 2 PV passes)
  }
  return g_fail;
}

// ------------------------------------------------------------------------------------------------ log-prob + PPO (grouped)
static int probe_ppo() {
  const int k = 4, b = 2, B = k * b, chw = 4 * 64 * 64;
  // scaled-linear SD schedule, float32 sequential cumprod (scheduling_ddim_flax.py create_state)
  std::vector<float> ac(1000);
  {
    const float s0 = sqrtf(0.00085f), s1 = sqrtf(0.012f);
    float c = 1.f;
    for (int i = 0; i < 1000; ++i) {
      const float sb = s0 + (s1 - s0) * (float)i / 999.0f;
      c *= 1.0f - sb * sb;
      ac[i] = c;
    }
  }
  float* d_ac = (float*)dalloc(4000);
  HIP_OK(hipMemcpy(d_ac, ac.data(), 4000, hipMemcpyHostToDevice));
  ddpo_ddim_consts c;
  c.alphas_cumprod = d_ac; c.num_train_timesteps = 1000; c.step_ratio = 20; c.final_alpha_cumprod = ac[0]; c.eta = 1.0f; c.pred_type = DDPO_PRED_EPSILON;
  const int ts_h[B] = {981, 481, 1, 21, 701, 241, 961, 501};
  const float adv_h[B] = {1.5f, -0.7f, 12.0f, -20.0f, 0.3f, -0.2f, 0.9f, -1.4f};
  Dev ec((int64_t)B * chw, 41, 1.f), eu((int64_t)B * chw, 42, 1.f), x((int64_t)B * chw, 43, 1.f), z((int64_t)B * chw, 44, 1.f);
  int32_t* d_ts = (int32_t*)dalloc(B * 4);
  float *d_adv = (float*)dalloc(B * 4), *d_old = (float*)dalloc(B * 4), *xn = (float*)dalloc((size_t)B * chw * 4), *lp0 = (float*)dalloc(B * 4);
  HIP_OK(hipMemcpy(d_ts, ts_h, B * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_adv, adv_h, B * 4, hipMemcpyHostToDevice));
  const float g = 5.0f, clip = 1e-4f;
  ABI_OK(ddpo_ddim_step_fwd(eu.p, ec.p, x.p, z.p, d_ts, g, &c, xn, lp0, B, chw, nullptr));
  std::vector<float> lp0_h(B), old_h(B);
  HIP_OK(hipMemcpy(lp0_h.data(), lp0, B * 4, hipMemcpyDeviceToHost));
  const float doff[B] = {0.f, 5e-5f, -3e-4f, 3e-4f, 3e-4f, -3e-4f, 2e-5f, -8e-5f};
  for (int i = 0; i < B; ++i) old_h[i] = lp0_h[i] + doff[i];
  HIP_OK(hipMemcpy(d_old, old_h.data(), B * 4, hipMemcpyHostToDevice));
  float *dc = (float*)dalloc((size_t)B * chw * 4), *du = (float*)dalloc((size_t)B * chw * 4), *per = (float*)dalloc(B * 16), *info = (float*)dalloc(k * 12);
  ABI_OK(ddpo_ddim_logprob_ppo_fwd_bwd_grouped(ec.p, eu.p, x.p, xn, d_ts, d_old, d_adv, g, clip, 1, &c, dc, du, per, info, B, b, chw, nullptr));
  std::vector<float> dc_h((size_t)B * chw), du_h((size_t)B * chw), per_h(B * 4), info_h(k * 3), xn_h((size_t)B * chw);
  HIP_OK(hipMemcpy(dc_h.data(), dc, dc_h.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(du_h.data(), du, du_h.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(per_h.data(), per, B * 16, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(info_h.data(), info, k * 12, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(xn_h.data(), xn, xn_h.size() * 4, hipMemcpyDeviceToHost));
  // (1) the grouped call equals k ungrouped calls on the row slices, bit for bit
  int fails = 0;
  for (int j = 0; j < k; ++j) {
    const int64_t off = (int64_t)j * b * chw;
    float *dc2 = (float*)dalloc((size_t)b * chw * 4), *du2 = (float*)dalloc((size_t)b * chw * 4), *per2 = (float*)dalloc(b * 16), *info2 = (float*)dalloc(12);
    ABI_OK(ddpo_ddim_logprob_ppo_fwd_bwd(ec.p + off, eu.p + off, x.p + off, xn + off, d_ts + j * b, d_old + j * b, d_adv + j * b, g, clip, 1, &c,
                                         dc2, du2, per2, info2, b, chw, nullptr));
    std::vector<float> a((size_t)b * chw), bb((size_t)b * chw), p2(b * 4), i2(3);
    HIP_OK(hipMemcpy(a.data(), dc2, a.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(bb.data(), du2, bb.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(p2.data(), per2, b * 16, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(i2.data(), info2, 12, hipMemcpyDeviceToHost));
    if (memcmp(a.data(), dc_h.data() + off, a.size() * 4) || memcmp(bb.data(), du_h.data() + off, bb.size() * 4) ||
        memcmp(p2.data(), per_h.data() + j * b * 4, b * 16) || memcmp(i2.data(), info_h.data() + j * 3, 12)) {
      printf("ppo grouped: micro-batch %d differs from the ungrouped call FAIL\n", j);
      ++fails;
    }
    HIP_OK(hipFree(dc2)); HIP_OK(hipFree(du2)); HIP_OK(hipFree(per2)); HIP_OK(hipFree(info2));
  }
  // (2) host evaluation in double of ddpo/training/policy_gradient.py:95-134 per micro-batch
  double worst_lp = 0, worst_g = 0, worst_info = 0;
  for (int j = 0; j < k; ++j) {
    double kl = 0, cf = 0, ls = 0;
    for (int r = 0; r < b; ++r) {
      const int i = j * b + r, t = ts_h[i], p = t - 20;
      const double a_t = ac[t], a_p = p >= 0 ? ac[p] : ac[0];
      const double var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p), sd = sqrt(var), sdc = std::max(sd, 1e-6);
      const double dirc = sqrt(1 - a_p - sd * sd), dmu = dirc - sqrt(a_p) * sqrt(1 - a_t) / sqrt(a_t);
      double acc = 0;
      std::vector<double> diff(chw);
      for (int e = 0; e < chw; ++e) {
        const int64_t ix = (int64_t)i * chw + e;
        const double ee = eu.at(ix) + (double)g * ((double)ec.at(ix) - (double)eu.at(ix));
        const double mu = sqrt(a_p) * ((double)x.at(ix) - sqrt(1 - a_t) * ee) / sqrt(a_t) + dirc * ee;
        diff[e] = (double)xn_h[ix] - mu;
        acc += -(diff[e] * diff[e]) / (2 * sdc * sdc) - log(sdc) - 0.9189385332046727;
      }
      const double lp = acc / chw, A = std::min(std::max((double)adv_h[i], -10.0), 10.0), ratio = exp(lp - (double)old_h[i]);
      const double un = -A * ratio, cl = -A * std::min(std::max(ratio, 1.0 - clip), 1.0 + clip);
      const double dl = (un >= cl) ? -A * ratio / b : 0.0;
      worst_lp = std::max(worst_lp, fabs(lp - per_h[i * 4]));
      kl += (lp - old_h[i]) * (lp - old_h[i]); cf += fabs(ratio - 1.0) > clip ? 1 : 0; ls += std::max(un, cl);
      double gmax = 1e-30, emax = 0;
      for (int e = 0; e < chw; ++e) {
        const double de = dl * diff[e] / (sdc * sdc * chw) * dmu;
        gmax = std::max(gmax, fabs(g * de));
        emax = std::max(emax, std::max(fabs(g * de - dc_h[(int64_t)i * chw + e]), fabs((1 - g) * de - du_h[(int64_t)i * chw + e])));
      }
      // the clip decision sits on |ratio - 1| ~ 1e-4: rows whose fp32 / fp64 log-probs land on opposite sides are skipped
      if (fabs(fabs(ratio - 1.0) - clip) > 2e-5) worst_g = std::max(worst_g, emax / gmax);
    }
    worst_info = std::max(worst_info, fabs(ls / b - info_h[j * 3 + 2]));
    (void)kl; (void)cf;
  }
  const bool ok = fails == 0 && worst_lp < 1e-4 && worst_g < 5e-3 && worst_info < 1e-3;
  printf("ppo grouped (k=%d micro-batches of %d): log-prob abs err %.1e, grad rel err %.1e, loss abs err %.1e %s\n", k, b, worst_lp, worst_g, worst_info,
         ok ? "OK" : "FAIL");
  return ok ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------ streaming yardstick
// What the memory system gives a kernel that only moves the bytes of a short-reduction layer: read `mb` MB, write `mb` MB (float4, grid-stride),
// timed over 3 / 10 / 50 back-to-back launches (3 launches fit the 256 MB Infinity Cache, 50 are steady state) — the yardstick for
// M=65536,K=320,N=320 (84 MB in, 84 MB out).
__global__ void __launch_bounds__(256) stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static int probe_stream() {
  for (int mb : {21, 84, 168, 336}) {
    const int64_t n4 = (int64_t)mb * (1 << 20) / 16;
    float4 *src = (float4*)dalloc((size_t)n4 * 16), *dst = (float4*)dalloc((size_t)n4 * 16);
    for (int grid : {2048, 16384}) {
      printf("stream copy %4d MB in + %4d MB out, grid %5d:", mb, mb, grid);
      for (int it : {3, 10, 50}) {
        const float ms = time_ms(it, [&] { hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, 0, src, dst, n4); });
        printf("  %2d launches %7.1f us (%5.2f TB/s)", it, ms * 1e3, 2.0 * mb * 1.048576e-3 / ms * 1e-3 * 1e3);
      }
      printf("\n");
    }
    HIP_OK(hipFree(src)); HIP_OK(hipFree(dst));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ gelu forms
// gelu(x, approximate=tanh): libm tanhf (rounds 1-4) vs the v_exp_f32 / v_rcp_f32 sigmoid form csrc/common.h ships since round 5 (both copied
// here), against float64 on a dense sweep of [-12, 12] plus large / special arguments; then the VALU cost of 32 evaluations per lane.
__device__ __forceinline__ float gelu_ref_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));
}
__global__ void gelu_check_kernel(const float* __restrict__ x, float* __restrict__ a, float* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { a[i] = gelu_ref_f(x[i]); b[i] = gelu_fast_f(x[i]); }
}
template <bool FAST>
__global__ void __launch_bounds__(256) gelu_cost_kernel(const float* __restrict__ x, float* __restrict__ y, int reps) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = x[(blockIdx.x * blockDim.x + threadIdx.x) * 8 + j];
  for (int r = 0; r < reps; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (FAST ? gelu_fast_f(v[j]) : gelu_ref_f(v[j])) + 0.25f;           // 8 independent chains per lane: throughput, not latency
  float s_ = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s_ += v[j];
  y[blockIdx.x * blockDim.x + threadIdx.x] = s_;
}
static int probe_gelu() {
  const int64_t n = 1 << 22;
  std::vector<float> h(n);
  for (int64_t i = 0; i < n - 16; ++i) h[i] = -12.0f + 24.0f * (float)i / (float)(n - 16);
  const float sp[16] = {0.f, -0.f, 1e-30f, -1e-30f, 1e-6f, -1e-6f, 20.f, -20.f, 100.f, -100.f, 1e10f, -1e10f, 3e38f, -3e38f, INFINITY, -INFINITY};
  for (int j = 0; j < 16; ++j) h[n - 16 + j] = sp[j];
  float *x = (float*)dalloc(n * 4), *a = (float*)dalloc(n * 4), *b = (float*)dalloc(n * 4);
  HIP_OK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gelu_check_kernel, dim3(1024), dim3(256), 0, 0, x, a, b, n);
  std::vector<float> ha(n), hb(n);
  HIP_OK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double ea = 0, eb = 0, ra = 0, rb = 0;                // max absolute error, and max error relative to max(|gelu|, 1e-3)
  int bad_special = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double xd = h[i];
    double ref;
    if (std::isinf(xd)) ref = xd > 0 ? INFINITY : 0.0;  // gelu(-inf) = -inf * 0 is NaN in both forms' arithmetic; the model never produces it
    else ref = 0.5 * xd * (1.0 + std::tanh(0.7978845608028654 * (xd + 0.044715 * xd * xd * xd)));
    if (i >= n - 16) {
      const bool oka = (std::isinf(ref) ? ha[i] == ref : std::fabs(ha[i] - ref) <= 1e-6 * std::max(1.0, std::fabs(ref))) || (std::isinf(xd) && xd < 0);
      const bool okb = (std::isinf(ref) ? hb[i] == ref : std::fabs(hb[i] - ref) <= 1e-6 * std::max(1.0, std::fabs(ref))) || (std::isinf(xd) && xd < 0);
      if (!oka || !okb) { ++bad_special; printf("gelu: special x = %g: tanhf form %g, fast form %g, float64 %g\n", xd, ha[i], hb[i], ref); }
      continue;
    }
    const double den = std::max(std::fabs(ref), 1e-3);
    ea = std::max(ea, std::fabs(ha[i] - ref)); eb = std::max(eb, std::fabs(hb[i] - ref));
    ra = std::max(ra, std::fabs(ha[i] - ref) / den); rb = std::max(rb, std::fabs(hb[i] - ref) / den);
  }
  printf("gelu: against float64 on [-12, 12] (%lld points): tanhf form max abs %.2e rel %.2e | v_exp / v_rcp form max abs %.2e rel %.2e | specials wrong: %d\n",
         (long long)(n - 16), ea, ra, eb, rb, bad_special);
  const int reps = 200, blocks = 1024;
  const float t0 = time_ms(5, [&] { hipLaunchKernelGGL(gelu_cost_kernel<false>, dim3(blocks), dim3(256), 0, 0, x, a, reps); });
  const float t1 = time_ms(5, [&] { hipLaunchKernelGGL(gelu_cost_kernel<true>, dim3(blocks), dim3(256), 0, 0, x, b, reps); });
  printf("gelu: %d x 8 evaluations per lane, 1024 workgroups: tanhf form %.1f us, v_exp / v_rcp form %.1f us (x%.2f)\n", reps, t0 * 1e3, t1 * 1e3, t0 / t1);
  HIP_OK(hipFree(x)); HIP_OK(hipFree(a)); HIP_OK(hipFree(b));
  // pass: the fast form is inside 1e-5 of max(|gelu|, 1e-3) or no worse than the shipped one (whose 1 + tanh(u) cancels for negative x: float32
  // emulation with exact exp2 / reciprocal gives 1.5e-6 for the sigmoid form and 1.5e-4 for the tanh form on this sweep)
  return (rb > std::max(ra, 1e-5) || bad_special) ? 1 : 0;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "gemm";
  const int B = argc > 2 ? atoi(argv[2]) : 16, iters = argc > 3 ? atoi(argv[3]) : 10;
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  printf("# %s (%s, %d CUs), libddpo_hip ABI v%d, mode %s, batch %d, PROBE_WKBLK=%s PROBE_COLD=%s\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, ddpo_abi_version(), mode.c_str(), B, getenv("PROBE_WKBLK") ? getenv("PROBE_WKBLK") : "-",
         getenv("PROBE_COLD") ? getenv("PROBE_COLD") : "-");
  int rc;
  if (mode == "gemm") rc = probe_gemm(B, iters);
  else if (mode == "gemm2") rc = probe_gemm2(B, iters);
  else if (mode == "x1") rc = probe_x1(B, iters);
  else if (mode == "mx") rc = probe_mx(B, iters);
  else if (mode == "vae") rc = probe_vae(B, iters);
  else if (mode == "wgrad") rc = probe_wgrad(B, iters);
  else if (mode == "attn") rc = probe_attn(B, iters);
  else if (mode == "ppo") rc = probe_ppo();
  else if (mode == "stream") rc = probe_stream();
  else if (mode == "gelu") rc = probe_gelu();
#ifdef PROBE_TIMING
  else if (mode == "ktime") rc = probe_ktime(B);
  else if (mode == "ktmx") rc = probe_ktmx(B);
#endif
  else { fprintf(stderr, "usage: kernel_probe gemm|gemm2|attn|ppo [batch] [iters]\n"); return 64; }
  HIP_OK(hipDeviceSynchronize());
  printf("# %s: %s\n", mode.c_str(), rc ? "FAILURES" : "all spot checks passed");
  return rc ? 1 : 0;
}
