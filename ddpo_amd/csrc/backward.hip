// Backward kernels of the HBM-bound U-Net layers (jax.grad of ddpo/training/policy_gradient.py:138-139 through
// GroupNorm(+SiLU), LayerNorm, GEGLU, SiLU, bias adds and nearest-2x upsampling).  Parameter gradients are
// accumulated atomically straight into the flat gradient buffer (AccumulatingTrainState: grad_acc += g).
#include "common.h"

#define GN_THREADS 256
#define GN_MAXCOL 4
#define GN_MAXC 4096
#define GN_MAXG 64
#define GN_PPB 32

__device__ __forceinline__ float dsilu_f(float z) {
  const float s = 1.f / (1.f + __expf(-z));
  return s * (1.f + z * (1.f - s));
}

// ---- GroupNorm backward, pass 1: per-channel sums of dz and dz*xhat over a pixel chunk; per-group partials
template <bool SILU>
__global__ void __launch_bounds__(GN_THREADS) gn_bwd_stats_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                                  int lddy, const float* __restrict__ ab, const float* __restrict__ mr,
                                                                  const float* __restrict__ gamma, int HW, int C, int G,
                                                                  int pix_per_block, double* __restrict__ part,
                                                                  float* __restrict__ cpart) {
  __shared__ float s_dz[GN_MAXC], s_dzx[GN_MAXC];
  const int b = blockIdx.y;
  const int C4 = C >> 2, cpg = C / G, t = threadIdx.x;
  const int p0 = blockIdx.x * pix_per_block, p1 = min(p0 + pix_per_block, HW);
  int ppi, col0, poff, ncol;
  if (C4 <= GN_THREADS) {
    ppi = GN_THREADS / C4; poff = t / C4; col0 = t - poff * C4; ncol = (poff < ppi) ? 1 : 0;
  } else {
    ppi = 1; poff = 0; col0 = t; ncol = (C4 - t + GN_THREADS - 1) / GN_THREADS;
  }
  float sdz[GN_MAXCOL][4], sdzx[GN_MAXCOL][4], ka[GN_MAXCOL][4], ks[GN_MAXCOL][4], km[GN_MAXCOL][4], kr[GN_MAXCOL][4];
#pragma unroll
  for (int j = 0; j < GN_MAXCOL; ++j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sdz[j][e] = 0.f; sdzx[j][e] = 0.f; ka[j][e] = 0.f; ks[j][e] = 0.f; km[j][e] = 0.f; kr[j][e] = 0.f;
      if (j < ncol) {
        const int c = ((col0 + j * GN_THREADS) << 2) + e;
        ka[j][e] = ab[2 * ((int64_t)b * C + c)];
        ks[j][e] = ab[2 * ((int64_t)b * C + c) + 1];
        const int g = c / cpg;
        km[j][e] = mr[2 * (b * G + g)];
        kr[j][e] = mr[2 * (b * G + g) + 1];
      }
    }
  }
  const float* xb = x + (int64_t)b * HW * ldx;
  const float* db = dy + (int64_t)b * HW * lddy;
  for (int p = p0 + poff; p < p1; p += ppi) {
#pragma unroll
    for (int j = 0; j < GN_MAXCOL; ++j) {
      if (j < ncol) {
        const int c = (col0 + j * GN_THREADS) << 2;
        const float4 xv = *reinterpret_cast<const float4*>(xb + (int64_t)p * ldx + c);
        const float4 dv = *reinterpret_cast<const float4*>(db + (int64_t)p * lddy + c);
        const float* px = &xv.x; const float* pd = &dv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dz = pd[e];
          if (SILU) dz *= dsilu_f(px[e] * ka[j][e] + ks[j][e]);
          const float xh = (px[e] - km[j][e]) * kr[j][e];
          sdz[j][e] += dz;
          sdzx[j][e] += dz * xh;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < GN_MAXCOL; ++j) {
    if (j < ncol) {
      const int idx = poff * C + ((col0 + j * GN_THREADS) << 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s_dz[idx + e] = sdz[j][e]; s_dzx[idx + e] = sdzx[j][e]; }
    }
  }
  __syncthreads();
  // parameter gradients: per-channel partial of this block -> cpart[block][{dgamma | dbeta}][C]; a column-sum pass follows
  float* cp = cpart + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C;
  for (int c = t; c < C; c += GN_THREADS) {
    float a = 0.f, q = 0.f;
    for (int slot = 0; slot < ppi; ++slot) { a += s_dz[slot * C + c]; q += s_dzx[slot * C + c]; }
    cp[c] = q;
    cp[C + c] = a;
  }
  if (t < G) {
    double a = 0.0, q = 0.0;
    for (int c = t * cpg; c < (t + 1) * cpg; ++c) {
      const double gm = (double)gamma[c];
      for (int slot = 0; slot < ppi; ++slot) { a += gm * (double)s_dz[slot * C + c]; q += gm * (double)s_dzx[slot * C + c]; }
    }
    double* o = part + (((int64_t)b * gridDim.x + blockIdx.x) * G + t) * 2;
    o[0] = a; o[1] = q;
  }
}

// pass 2: per (b, group): S1/n, S2/n -> per-(b,c) {p = rstd^2 S2/n, q = rstd S1/n - mean p}
__global__ void __launch_bounds__(64) gn_bwd_finalize_kernel(const double* __restrict__ part, const float* __restrict__ mr,
                                                             float* __restrict__ pq, int chunks, int C, int G, int HW) {
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, lane = threadIdx.x;
  double a = 0.0, q = 0.0;
  for (int ch = lane; ch < chunks; ch += 64) {
    const double* o = part + (((int64_t)b * chunks + ch) * G + g) * 2;
    a += o[0]; q += o[1];
  }
  a = wave_sum_d(a); q = wave_sum_d(q);
  const int cpg = C / G;
  const double n = (double)cpg * (double)HW;
  const float mean = mr[2 * blockIdx.x], rstd = mr[2 * blockIdx.x + 1];
  const float p = rstd * rstd * (float)(q / n);
  const float qq = rstd * (float)(a / n) - mean * p;
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
    pq[2 * ((int64_t)b * C + c)] = p;
    pq[2 * ((int64_t)b * C + c) + 1] = qq;
  }
}

// pass 3: dx = a*dz - x*p - q (+ dx_add)
template <bool SILU>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dy,
                                                           int lddy, const float* __restrict__ ab, const float* __restrict__ pq,
                                                           const float* __restrict__ dx_add, int ld_add, float* __restrict__ dx,
                                                           int lddx, int B, int HW, int C) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)B * HW * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C4;
    const int c = (int)(i - row * C4) << 2;
    const int b = (int)(row / HW);
    const float4 xv = *reinterpret_cast<const float4*>(x + row * ldx + c);
    const float4 dv = *reinterpret_cast<const float4*>(dy + row * lddy + c);
    const float4 a0 = *reinterpret_cast<const float4*>(ab + 2 * ((int64_t)b * C + c));
    const float4 a1 = *reinterpret_cast<const float4*>(ab + 2 * ((int64_t)b * C + c) + 4);
    const float4 q0 = *reinterpret_cast<const float4*>(pq + 2 * ((int64_t)b * C + c));
    const float4 q1 = *reinterpret_cast<const float4*>(pq + 2 * ((int64_t)b * C + c) + 4);
    const float av[4] = {a0.x, a0.z, a1.x, a1.z}, sv[4] = {a0.y, a0.w, a1.y, a1.w};
    const float pv[4] = {q0.x, q0.z, q1.x, q1.z}, qv[4] = {q0.y, q0.w, q1.y, q1.w};
    const float* px = &xv.x; const float* pd = &dv.x;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dx_add) o = *reinterpret_cast<const float4*>(dx_add + row * ld_add + c);
    float* po = &o.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dz = pd[e];
      if (SILU) dz *= dsilu_f(px[e] * av[e] + sv[e]);
      po[e] += av[e] * dz - px[e] * pv[e] - qv[e];
    }
    *reinterpret_cast<float4*>(dx + row * lddx + c) = o;
  }
}

static inline int gn_ppb(int C) {
  const int C4 = C >> 2;
  const int ppi = C4 <= GN_THREADS ? GN_THREADS / C4 : 1;
  int ppb = GN_PPB < ppi ? ppi : GN_PPB;
  return ((ppb + ppi - 1) / ppi) * ppi;
}

extern "C" size_t ddpo_groupnorm_bwd_ws_bytes(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0) return 0;
  const int chunks = (HW + gn_ppb(C) - 1) / gn_ppb(C);
  return (size_t)B * chunks * G * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float) + (size_t)B * chunks * 2 * C * sizeof(float);
}
extern "C" int ddpo_colsum_accum(const float* x, int ldx, int64_t rows, int cols, int rows_per_seg, float* out, void* stream);

extern "C" int ddpo_groupnorm_bwd(const float* x, int ldx, const float* dy, int lddy, const float* stats, const float* gamma, int B,
                                  int HW, int C, int G, int fuse_silu, const float* dx_add, int ld_add, float* dx, int lddx,
                                  float* dgamma, float* dbeta, void* ws, void* stream) {
  if (!x || !dy || !stats || !gamma || !dx || !dgamma || !dbeta || !ws) return DDPO_EINVAL;
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > GN_MAXG || (C & 3) || (C % G) || C > GN_MAXC || B > 65535) return DDPO_EINVAL;
  if ((ldx & 3) || (lddy & 3) || (lddx & 3) || (dx_add && (ld_add & 3))) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  const int C4 = C >> 2, ppb = gn_ppb(C), chunks = (HW + ppb - 1) / ppb;
  const float* ab = stats;
  const float* mr = stats + (size_t)B * C * 2;
  double* part = reinterpret_cast<double*>(ws);
  float* pq = reinterpret_cast<float*>(part + (size_t)B * chunks * G * 2);
  float* cpart = pq + (size_t)B * C * 2;
  if (fuse_silu)
    hipLaunchKernelGGL(gn_bwd_stats_kernel<true>, dim3(chunks, B), dim3(GN_THREADS), 0, st, x, ldx, dy, lddy, ab, mr, gamma, HW, C, G,
                       ppb, part, cpart);
  else
    hipLaunchKernelGGL(gn_bwd_stats_kernel<false>, dim3(chunks, B), dim3(GN_THREADS), 0, st, x, ldx, dy, lddy, ab, mr, gamma, HW, C, G,
                       ppb, part, cpart);
  DDPO_LAUNCH_CHECK();
  {   // second stage of the parameter gradients: column sums of the per-block partials (few atomics per address)
    int rc;
    if (dbeta == dgamma + C) {          // scale and bias are adjacent in the flat gradient buffer: one pass over 2C columns
      rc = ddpo_colsum_accum(cpart, 2 * C, (int64_t)B * chunks, 2 * C, 0, dgamma, stream);
    } else {
      rc = ddpo_colsum_accum(cpart, 2 * C, (int64_t)B * chunks, C, 0, dgamma, stream);
      if (rc == DDPO_OK) rc = ddpo_colsum_accum(cpart + C, 2 * C, (int64_t)B * chunks, C, 0, dbeta, stream);
    }
    if (rc != DDPO_OK) return rc;
  }
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B * G), dim3(64), 0, st, part, mr, pq, chunks, C, G, HW);
  DDPO_LAUNCH_CHECK();
  int64_t blocks = ((int64_t)B * HW * C4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (fuse_silu)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3((int)blocks), dim3(256), 0, st, x, ldx, dy, lddy, ab, pq, dx_add, ld_add, dx,
                       lddx, B, HW, C);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3((int)blocks), dim3(256), 0, st, x, ldx, dy, lddy, ab, pq, dx_add, ld_add, dx,
                       lddx, B, HW, C);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row (statistics recomputed), parameter grads in registers per lane-owned channel
// ------------------------------------------------------------------------------------------------
template <int MAXV>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, int rows, int C, float eps,
                                                            const float* __restrict__ dx_add, float* __restrict__ dx,
                                                            float* __restrict__ cpart) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int C4 = C >> 2;
  float4 gm[MAXV], dg[MAXV], dbt[MAXV];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int c4 = lane + j * 64;
    gm[j] = (c4 < C4) ? *reinterpret_cast<const float4*>(gamma + (c4 << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    dbt[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = wave; row < rows; row += nwaves) {
    const float* xr = x + (int64_t)row * C;
    const float* dr = dy + (int64_t)row * C;
    float4 v[MAXV], d[MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int c4 = lane + j * 64;
      if (c4 < C4) {
        v[j] = *reinterpret_cast<const float4*>(xr + (c4 << 2));
        d[j] = *reinterpret_cast<const float4*>(dr + (c4 << 2));
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      } else {
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f); d[j] = v[j];
      }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      if (lane + j * 64 < C4) {
        const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, e = v[j].w - mean;
        q += (a * a + b * b) + (c * c + e * e);
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      if (lane + j * 64 < C4) {
        float* pv = &v[j].x; float* pd = &d[j].x; const float* pg = &gm[j].x; float* pdg = &dg[j].x; float* pdb = &dbt[j].x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (pv[e] - mean) * rstd;
          pdg[e] += pd[e] * xh;
          pdb[e] += pd[e];
          const float g = pd[e] * pg[e];
          m1 += g; m2 += g * xh;
          pv[e] = xh; pd[e] = g;
        }
      }
    }
    m1 = wave_sum(m1) / (float)C;
    m2 = wave_sum(m2) / (float)C;
    float* oxr = dx + (int64_t)row * C;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int c4 = lane + j * 64;
      if (c4 < C4) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dx_add) o = *reinterpret_cast<const float4*>(dx_add + (int64_t)row * C + (c4 << 2));
        o.x += rstd * (d[j].x - m1 - v[j].x * m2);
        o.y += rstd * (d[j].y - m1 - v[j].y * m2);
        o.z += rstd * (d[j].z - m1 - v[j].z * m2);
        o.w += rstd * (d[j].w - m1 - v[j].w * m2);
        *reinterpret_cast<float4*>(oxr + (c4 << 2)) = o;
      }
    }
  }
  // combine the 4 waves of the block in LDS (fixed order), then one atomic per (block, channel)
  __shared__ float4 s_dg[3][MAXV * 64], s_db[3][MAXV * 64];
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < MAXV; ++j)
    if (w > 0) { s_dg[w - 1][j * 64 + lane] = dg[j]; s_db[w - 1][j * 64 + lane] = dbt[j]; }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int c4 = lane + j * 64;
      if (c4 < C4) {
        float4 a = dg[j], b = dbt[j];
        for (int k = 0; k < 3; ++k) {
          const float4 x = s_dg[k][j * 64 + lane], y = s_db[k][j * 64 + lane];
          a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
          b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
        }
        float* cp = cpart + (int64_t)blockIdx.x * 2 * C;          // [block][{dgamma | dbeta}][C]
        *reinterpret_cast<float4*>(cp + (c4 << 2)) = a;
        *reinterpret_cast<float4*>(cp + C + (c4 << 2)) = b;
      }
    }
  }
}

static inline int ln_bwd_blocks(int rows) {
  int blocks = (rows + 15) / 16;           // ~4 rows per wave: enough waves in flight to stream at HBM rate
  if (blocks > 1024) blocks = 1024;
  return blocks < 1 ? 1 : blocks;
}
extern "C" size_t ddpo_layernorm_bwd_ws_bytes(int rows, int C) { return (size_t)ln_bwd_blocks(rows) * 2 * C * sizeof(float); }

extern "C" int ddpo_layernorm_bwd(const float* x, const float* dy, const float* gamma, int rows, int C, float eps,
                                  const float* dx_add, float* dx, float* dgamma, float* dbeta, void* ws, void* stream) {
  if (!x || !dy || !gamma || !dx || !dgamma || !dbeta || !ws || rows <= 0 || C <= 0 || (C & 3) || C > 2560) return DDPO_EINVAL;
  if (reinterpret_cast<uintptr_t>(ws) & 15) return DDPO_EINVAL;
  const int blocks = ln_bwd_blocks(rows);
  float* cpart = reinterpret_cast<float*>(ws);
  hipStream_t st = as_stream(stream);
  const int nv = ((C >> 2) + 63) / 64;
  if (nv <= 2) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, x, dy, gamma, rows, C, eps, dx_add, dx, cpart);
  else if (nv <= 5) hipLaunchKernelGGL(layernorm_bwd_kernel<5>, dim3(blocks), dim3(256), 0, st, x, dy, gamma, rows, C, eps, dx_add, dx, cpart);
  else hipLaunchKernelGGL(layernorm_bwd_kernel<10>, dim3(blocks), dim3(256), 0, st, x, dy, gamma, rows, C, eps, dx_add, dx, cpart);
  DDPO_LAUNCH_CHECK();
  if (dbeta == dgamma + C) return ddpo_colsum_accum(cpart, 2 * C, blocks, 2 * C, 0, dgamma, stream);
  int rc = ddpo_colsum_accum(cpart, 2 * C, blocks, C, 0, dgamma, stream);
  if (rc != DDPO_OK) return rc;
  return ddpo_colsum_accum(cpart + C, 2 * C, blocks, C, 0, dbeta, stream);
}

// ------------------------------------------------------------------------------------------------
// element-wise backward pieces
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gelu_tanh_fwd_bwd(float x, float& g, float& dg) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  const float th = 2.0f * sigmoid_2u_fast(u) - 1.0f;             // tanh u = 2 sigmoid(2u) - 1 (common.h)
  g = 0.5f * x * (1.f + th);
  dg = 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * k0 * (1.f + 3.f * k1 * x * x);
}

__global__ void __launch_bounds__(256) geglu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                        int64_t rows, int F) {
  const int f4 = F >> 2;
  const int64_t total = rows * f4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / f4;
    const int c = (int)(i - r * f4) << 2;
    const float4 a = *reinterpret_cast<const float4*>(x + r * (2 * (int64_t)F) + c);
    const float4 b = *reinterpret_cast<const float4*>(x + r * (2 * (int64_t)F) + F + c);
    const float4 d = *reinterpret_cast<const float4*>(dy + r * (int64_t)F + c);
    float4 da, dbv;
    const float* pa = &a.x; const float* pb = &b.x; const float* pd = &d.x; float* pda = &da.x; float* pdb = &dbv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g, dg;
      gelu_tanh_fwd_bwd(pb[e], g, dg);
      pda[e] = pd[e] * g;
      pdb[e] = pd[e] * pa[e] * dg;
    }
    *reinterpret_cast<float4*>(dx + r * (2 * (int64_t)F) + c) = da;
    *reinterpret_cast<float4*>(dx + r * (2 * (int64_t)F) + F + c) = dbv;
  }
}
extern "C" int ddpo_geglu_bwd(const float* x, const float* dy, float* dx, int64_t rows, int F, void* stream) {
  if (!x || !dy || !dx || rows <= 0 || F <= 0 || (F & 3)) return DDPO_EINVAL;
  int64_t blocks = (rows * (F >> 2) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, dy, dx, rows, F);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

__global__ void __launch_bounds__(256) silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dx[i] = dy[i] * dsilu_f(x[i]);
}
extern "C" int ddpo_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  if (!x || !dy || !dx || n <= 0) return DDPO_EINVAL;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, dy, dx, n);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// column sums (bias / time-embedding-add gradients): 16 float4 column lanes x 16 row lanes per block
#define CS_ROWS 512
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ x, int ldx, int64_t rows, int cols, int rows_per_seg,
                                                     int chunks_per_seg, float* __restrict__ out) {
  __shared__ float4 red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + cl) << 2;
  const int seg = blockIdx.y / chunks_per_seg, chunk = blockIdx.y - seg * chunks_per_seg;
  const int64_t r0 = (int64_t)seg * rows_per_seg + (int64_t)chunk * CS_ROWS;
  int64_t r1 = r0 + CS_ROWS;
  const int64_t seg_end = (int64_t)(seg + 1) * rows_per_seg;
  if (r1 > seg_end) r1 = seg_end;
  if (r1 > rows) r1 = rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols)
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float4 s = red[0][cl];
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][cl]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    float* o = out + (int64_t)seg * cols + c;
    atomicAdd(o, s.x); atomicAdd(o + 1, s.y); atomicAdd(o + 2, s.z); atomicAdd(o + 3, s.w);
  }
}
extern "C" int ddpo_colsum_accum(const float* x, int ldx, int64_t rows, int cols, int rows_per_seg, float* out, void* stream) {
  if (!x || !out || rows <= 0 || cols <= 0 || (cols & 3) || (ldx & 3)) return DDPO_EINVAL;
  if (rows_per_seg <= 0) rows_per_seg = (int)rows;
  if (rows % rows_per_seg) return DDPO_EINVAL;
  const int nseg = (int)(rows / rows_per_seg);
  const int cps = (rows_per_seg + CS_ROWS - 1) / CS_ROWS;
  if ((int64_t)nseg * cps > 65535) return DDPO_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, nseg * cps), dim3(256), 0, as_stream(stream), x, ldx, rows, cols,
                     rows_per_seg, cps, out);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

__global__ void __launch_bounds__(256) sumpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4) {
  const int64_t total = (int64_t)B * H * W * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    int64_t p = i / C4;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int b = (int)(p / H);
    const float4* s = reinterpret_cast<const float4*>(x) + (((int64_t)b * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c;
    const float4 a = s[0], b2 = s[C4], c2 = s[(int64_t)2 * W * C4], d2 = s[(int64_t)2 * W * C4 + C4];
    float4 o;
    o.x = (a.x + b2.x) + (c2.x + d2.x); o.y = (a.y + b2.y) + (c2.y + d2.y);
    o.z = (a.z + b2.z) + (c2.z + d2.z); o.w = (a.w + b2.w) + (c2.w + d2.w);
    reinterpret_cast<float4*>(y)[i] = o;
  }
}
extern "C" int ddpo_sumpool2x2(const float* x, float* y, int B, int H, int W, int C, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return DDPO_EINVAL;
  int64_t blocks = ((int64_t)B * H * W * (C >> 2) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(sumpool2x2_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, y, B, H, W, C >> 2);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}
extern "C" int ddpo_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return DDPO_EINVAL;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(add_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), a, b, out, n);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}
