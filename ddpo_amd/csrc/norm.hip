// GroupNorm(+SiLU) and LayerNorm forward for NHWC activations (HBM-bound; float4 coalesced rows).
// GroupNorm = nn.GroupNorm(32, eps) of diffusers' FlaxResnetBlock2D / FlaxTransformer2DModel / conv_norm_out;
// LayerNorm = nn.LayerNorm(eps=1e-5) of FlaxBasicTransformerBlock (reference call sites: ddpo_hip.h).
#include "common.h"

#define GN_THREADS 256
#define GN_MAXCOL 4        // C <= 4*256*GN_MAXCOL = 4096
#define GN_MAXG 64

// Pass 1: per-(b, group) sum / sum-of-squares.  grid = (pixel chunks, B).
__global__ void __launch_bounds__(GN_THREADS) gn_stats_kernel(const float* __restrict__ x, int ldx, int HW, int C, int G,
                                                              int pix_per_block, double* __restrict__ ws) {
  __shared__ float s_sum[GN_MAXG], s_sq[GN_MAXG];
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int cpg = C / G;
  const int t = threadIdx.x;
  if (t < G) { s_sum[t] = 0.f; s_sq[t] = 0.f; }
  __syncthreads();
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  float sum[GN_MAXCOL][4], sq[GN_MAXCOL][4];
#pragma unroll
  for (int j = 0; j < GN_MAXCOL; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { sum[j][e] = 0.f; sq[j][e] = 0.f; }
  int ppi, col0, poff, ncol;
  if (C4 <= GN_THREADS) {
    ppi = GN_THREADS / C4;
    poff = t / C4;
    col0 = t - poff * C4;
    ncol = (poff < ppi) ? 1 : 0;
  } else {
    ppi = 1; poff = 0; col0 = t;
    ncol = (C4 - t + GN_THREADS - 1) / GN_THREADS;
  }
  const float* xb = x + (int64_t)b * HW * ldx;
  for (int p = p0 + poff; p < p1; p += ppi) {
    const float* row = xb + (int64_t)p * ldx;
#pragma unroll
    for (int j = 0; j < GN_MAXCOL; ++j) {
      if (j < ncol) {
        const float4 v = *reinterpret_cast<const float4*>(row + ((col0 + j * GN_THREADS) << 2));
        sum[j][0] += v.x; sq[j][0] += v.x * v.x;
        sum[j][1] += v.y; sq[j][1] += v.y * v.y;
        sum[j][2] += v.z; sq[j][2] += v.z * v.z;
        sum[j][3] += v.w; sq[j][3] += v.w * v.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < GN_MAXCOL; ++j) {
    if (j < ncol) {
      const int c = (col0 + j * GN_THREADS) << 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / cpg;
        atomicAdd(&s_sum[g], sum[j][e]);
        atomicAdd(&s_sq[g], sq[j][e]);
      }
    }
  }
  __syncthreads();
  if (t < G) {
    atomicAdd(&ws[((int64_t)b * G + t) * 2 + 0], (double)s_sum[t]);
    atomicAdd(&ws[((int64_t)b * G + t) * 2 + 1], (double)s_sq[t]);
  }
}

// Pass 2: per-(b, c) affine  a = rstd*gamma, s = beta - mean*a
__global__ void gn_finalize_kernel(const double* __restrict__ ws, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ ab, int B, int C, int G, int HW, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  const int cpg = C / G, g = c / cpg;
  const double cnt = (double)cpg * (double)HW;
  const double mean = ws[((int64_t)b * G + g) * 2] / cnt;
  double var = ws[((int64_t)b * G + g) * 2 + 1] / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = rstd * gamma[c];
  ab[2 * (int64_t)i] = a;
  ab[2 * (int64_t)i + 1] = beta[c] - (float)mean * a;
}

// Pass 3: y = act(x*a + s)
template <bool SILU>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                       const float* __restrict__ ab, int B, int HW, int C) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)B * HW * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C4;
    const int c = (int)(i - row * C4) << 2;
    const int b = (int)(row / HW);
    const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
    const float4 k0 = *reinterpret_cast<const float4*>(ab + 2 * ((int64_t)b * C + c));
    const float4 k1 = *reinterpret_cast<const float4*>(ab + 2 * ((int64_t)b * C + c) + 4);
    float4 o;
    o.x = v.x * k0.x + k0.y; o.y = v.y * k0.z + k0.w; o.z = v.z * k1.x + k1.y; o.w = v.w * k1.z + k1.w;
    if (SILU) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
    *reinterpret_cast<float4*>(y + row * ldy + c) = o;
  }
}

extern "C" size_t ddpo_groupnorm_ws_bytes(int B, int C, int G) {
  return (size_t)B * G * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float);
}

extern "C" int ddpo_groupnorm_fwd(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, int B,
                                  int HW, int C, int G, float eps, int fuse_silu, void* ws, void* stream) {
  if (!x || !y || !gamma || !beta || !ws || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > GN_MAXG) return DDPO_EINVAL;
  if ((C & 3) || (C % G) || (ldx & 3) || (ldy & 3) || C > 4 * GN_THREADS * GN_MAXCOL) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ws)) & 15) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  double* sums = reinterpret_cast<double*>(ws);
  float* ab = reinterpret_cast<float*>(sums + (size_t)B * G * 2);
  if (hipMemsetAsync(sums, 0, (size_t)B * G * 2 * sizeof(double), st) != hipSuccess) return DDPO_ELAUNCH;
  const int C4 = C >> 2;
  const int ppi = C4 <= GN_THREADS ? GN_THREADS / C4 : 1;
  int ppb = 32;
  if (ppb < ppi) ppb = ppi;
  ppb = ((ppb + ppi - 1) / ppi) * ppi;
  const int chunks = (HW + ppb - 1) / ppb;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, B), dim3(GN_THREADS), 0, st, x, ldx, HW, C, G, ppb, sums);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, sums, gamma, beta, ab, B, C, G, HW, eps);
  DDPO_LAUNCH_CHECK();
  int64_t blocks = ((int64_t)B * HW * C4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (fuse_silu)
    hipLaunchKernelGGL(gn_apply_kernel<true>, dim3((int)blocks), dim3(256), 0, st, x, ldx, y, ldy, ab, B, HW, C);
  else
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3((int)blocks), dim3(256), 0, st, x, ldx, y, ldy, ab, B, HW, C);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 64*4*LN_MAXV = 2560)
// ------------------------------------------------------------------------------------------------
#define LN_MAXV 10
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int C4 = C >> 2;
  const float* xr = x + (int64_t)row * C;
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c4 = lane + j * 64;
    if (c4 < C4) {
      v[j] = *reinterpret_cast<const float4*>(xr + (c4 << 2));
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c4 = lane + j * 64;
    if (c4 < C4) {
      const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  float* yr = y + (int64_t)row * C;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c4 = lane + j * 64;
    if (c4 < C4) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + (c4 << 2));
      const float4 bt = *reinterpret_cast<const float4*>(beta + (c4 << 2));
      float4 o;
      o.x = (v[j].x - mean) * rstd * g.x + bt.x;
      o.y = (v[j].y - mean) * rstd * g.y + bt.y;
      o.z = (v[j].z - mean) * rstd * g.z + bt.z;
      o.w = (v[j].w - mean) * rstd * g.w + bt.w;
      *reinterpret_cast<float4*>(yr + (c4 << 2)) = o;
    }
  }
}

extern "C" int ddpo_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, int rows, int C, float eps,
                                  void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV) return DDPO_EINVAL;
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, y, gamma, beta, rows, C, eps);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}
