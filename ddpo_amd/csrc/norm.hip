// GroupNorm(+SiLU) and LayerNorm forward for NHWC activations (HBM-bound; float4 coalesced rows).
// GroupNorm = nn.GroupNorm(32, eps) of diffusers' FlaxResnetBlock2D / FlaxTransformer2DModel / conv_norm_out;
// LayerNorm = nn.LayerNorm(eps=1e-5) of FlaxBasicTransformerBlock (reference call sites: ddpo_hip.h).
#include "common.h"

#define GN_THREADS 256
#define GN_MAXCOL 4        // C <= 4*256*GN_MAXCOL = 4096
#define GN_MAXC 4096
#define GN_MAXG 64
#define GN_PPB 32          // pixels per stats block (lower bound)

// Pass 1: per-(b, pixel-chunk, group) partial sum / sum-of-squares, reduced in a FIXED order (no atomics, so the
// result is bit-reproducible run to run).  grid = (pixel chunks, B).
__global__ void __launch_bounds__(GN_THREADS) gn_stats_kernel(const float* __restrict__ x, int ldx, int HW, int C, int G,
                                                              int pix_per_block, double* __restrict__ part) {
  __shared__ float s_sum[GN_MAXC], s_sq[GN_MAXC];
  const int b = blockIdx.y;
  const int C4 = C >> 2;
  const int cpg = C / G;
  const int t = threadIdx.x;
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
  // four pixel rows are requested before any is consumed (the loop is latency-bound otherwise); they are accumulated in
  // the original pixel order, so the partial sums are bit-identical to a one-row-at-a-time loop
  constexpr int UNR = 4;
  for (int p = p0 + poff; p < p1; p += UNR * ppi) {
    float4 v[UNR][GN_MAXCOL];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int pu = p + u * ppi;
      const float* row = xb + (int64_t)pu * ldx;
#pragma unroll
      for (int j = 0; j < GN_MAXCOL; ++j)
        v[u][j] = (j < ncol && pu < p1) ? *reinterpret_cast<const float4*>(row + ((col0 + j * GN_THREADS) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      if (p + u * ppi >= p1) break;
#pragma unroll
      for (int j = 0; j < GN_MAXCOL; ++j) {
        if (j < ncol) {
          sum[j][0] += v[u][j].x; sq[j][0] += v[u][j].x * v[u][j].x;
          sum[j][1] += v[u][j].y; sq[j][1] += v[u][j].y * v[u][j].y;
          sum[j][2] += v[u][j].z; sq[j][2] += v[u][j].z * v[u][j].z;
          sum[j][3] += v[u][j].w; sq[j][3] += v[u][j].w * v[u][j].w;
        }
      }
    }
  }
  // per-thread partials -> LDS as [slot][channel] (slot = pixel lane of the thread), then a serial per-group sum
#pragma unroll
  for (int j = 0; j < GN_MAXCOL; ++j) {
    if (j < ncol) {
      const int idx = poff * C + ((col0 + j * GN_THREADS) << 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s_sum[idx + e] = sum[j][e]; s_sq[idx + e] = sq[j][e]; }
    }
  }
  __syncthreads();
  if (t < G) {
    double a = 0.0, q = 0.0;
    for (int slot = 0; slot < ppi; ++slot)
      for (int c = t * cpg; c < (t + 1) * cpg; ++c) { a += (double)s_sum[slot * C + c]; q += (double)s_sq[slot * C + c]; }
    double* o = part + (((int64_t)b * gridDim.x + blockIdx.x) * G + t) * 2;
    o[0] = a; o[1] = q;
  }
}

// Pass 2: one wave per (b, group): fixed-order reduction over the chunks, then the per-(b, c) affine
//   a = rstd*gamma, s = beta - mean*a
__global__ void __launch_bounds__(64) gn_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ ab,
                                                         float* __restrict__ mr, int chunks, int C, int G, int HW, float eps) {
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int lane = threadIdx.x;
  double a = 0.0, q = 0.0;
  for (int ch = lane; ch < chunks; ch += 64) {
    const double* o = part + (((int64_t)b * chunks + ch) * G + g) * 2;
    a += o[0]; q += o[1];
  }
  a = wave_sum_d(a); q = wave_sum_d(q);
  const int cpg = C / G;
  const double cnt = (double)cpg * (double)HW;
  const double mean = a / cnt;
  double var = q / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (lane == 0) { mr[2 * blockIdx.x] = (float)mean; mr[2 * blockIdx.x + 1] = rstd; }
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
    const float k = rstd * gamma[c];
    ab[2 * ((int64_t)b * C + c)] = k;
    ab[2 * ((int64_t)b * C + c) + 1] = beta[c] - (float)mean * k;
  }
}

// Pass 3: y = act(x*a + s)
// PL: instead of the fp32 tensor write its bf16 hi / lo planes (rows, ldy) — the operand format of the plane-fed GEMM
// (ddpo_gemm_conv_fwd_bf16_planes); same arithmetic, so the planes hold exactly the split of the fp32 result.
template <bool SILU, bool PL = false>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                       const float* __restrict__ ab, int B, int HW, int C,
                                                       uint16_t* __restrict__ y_hi = nullptr, uint16_t* __restrict__ y_lo = nullptr,
                                                       int pair = 0) {
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
    if (PL) {
      if (pair == 2) {                    // f16mx planes (common.h): y_hi = f16 plane, y_lo = interleaved e5m2 [h8 | l8] chunks
        mx_store4(y_hi, y_lo, row, c, ldy, (int64_t)B * HW, o);
        continue;
      }
      uint2 h, l;
      split4(o, h, l);
      if (pair) {
        // 16-byte plane stores: lanes 2j / 2j+1 hold adjacent channel quads of one row (C % 8 == 0); the even lane collects
        // both hi halves and writes 8 channels of the hi plane, the odd lane both lo halves and writes the lo plane
        const bool odd = threadIdx.x & 1;
        const uint32_t rx = lane_xor1_u32(odd ? h.x : l.x), ry = lane_xor1_u32(odd ? h.y : l.y);
        if (!odd) *reinterpret_cast<uint4*>(y_hi + plane_off(row, c, ldy, (int64_t)B * HW)) = make_uint4(h.x, h.y, rx, ry);
        else *reinterpret_cast<uint4*>(y_lo + plane_off(row, c - 4, ldy, (int64_t)B * HW)) = make_uint4(rx, ry, l.x, l.y);
      } else {
        *reinterpret_cast<uint2*>(y_hi + plane_off(row, c, ldy, (int64_t)B * HW)) = h;
        *reinterpret_cast<uint2*>(y_lo + plane_off(row, c, ldy, (int64_t)B * HW)) = l;
      }
    } else {
      *reinterpret_cast<float4*>(y + row * ldy + c) = o;
    }
  }
}

// Small feature maps (HW <= 256: the 16x16 and 8x8 levels, 29 of the 61 norms of an SD-1.5 forward): statistics, finalize and apply in ONE
// launch, one workgroup per (image, group), its <= 256 x 80 values in registers (a thread = one channel quad of up to 22 pixels).
// The three-launch form spends 10 + 5 + 7 us on these shapes, all of it launch latency (the data of a level is 1 - 21 MB, L2-resident).
// Same formulas as the three kernels above: float partial sums per thread, the reduction over the group in double in a fixed order (lanes by the
// wave butterfly, then the waves in order), var = E[x^2] - mean^2 in double, a = rstd * gamma, s = beta - mean * a, y = act(x * a + s);
// the saved statistics (ab, mr) are written for the backward as before.  Selected by the layer's geometry ALONE (HW, channels per group), never by
// the batch, so the sampler and the training forward of one layer always run the same arithmetic.
// (A 1024-thread form of the same kernel for the 32x32 level — 16 pixels per thread — was built and measured: 4.199 vs 4.207 images/s for the
// three launches there, profiles/r05_ab_gn_fused3.log; not kept.)
#define GNF_MAXQ 20        // float4 per pixel and group: channels per group <= 80
#define GNF_MAXP 22        // pixels per thread: ceil(256 / floor(256 / GNF_MAXQ))
template <bool SILU, int MODE>        // MODE 0: fp32 output, 1: bf16 hi / lo planes, 2: f16mx planes
__global__ void __launch_bounds__(256) gn_fused_small_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, uint16_t* __restrict__ y_hi,
                                                             uint16_t* __restrict__ y_lo, int ldy, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ ab, float* __restrict__ mr,
                                                             int HW, int C, int G, float eps, int64_t rows_total) {
  // thread = (pixel slot ps, channel quad q) with the quad FASTEST: the lanes of a wave read whole 160 .. 320-byte runs of consecutive pixels'
  // group channels (the first form of this kernel gave a thread one pixel's whole run: 16 bytes per lane and cache line, 23 us per launch)
  __shared__ double s_red[2][4];
  const int g = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int cpg = C / G, cq = cpg >> 2;
  const int pps = blockDim.x / cq;                   // pixels per pass
  const int ps = t / cq, q = t - ps * cq;
  const bool lane_on = ps < pps;
  const int c = g * cpg + (q << 2);
  const float* xb = x + (int64_t)b * HW * ldx + c;
  float4 v[GNF_MAXP];
  float sm = 0.f, sq = 0.f;
#pragma unroll
  for (int i = 0; i < GNF_MAXP; ++i) {
    const int p = ps + i * pps;
    v[i] = (lane_on && p < HW) ? *reinterpret_cast<const float4*>(xb + (int64_t)p * ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
    sm += v[i].x; sq += v[i].x * v[i].x;
    sm += v[i].y; sq += v[i].y * v[i].y;
    sm += v[i].z; sq += v[i].z * v[i].z;
    sm += v[i].w; sq += v[i].w * v[i].w;
  }
  const double wa = wave_sum_d((double)sm), wq = wave_sum_d((double)sq);
  const int lane = t & 63, wid = t >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) { s_red[0][wid] = wa; s_red[1][wid] = wq; }
  __syncthreads();
  double a = 0.0, qq = 0.0;
  for (int w = 0; w < nw; ++w) { a += s_red[0][w]; qq += s_red[1][w]; }
  const double cnt = (double)cpg * (double)HW;
  const double mean = a / cnt;
  double var = qq / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  if (t == 0) { mr[2 * (b * G + g)] = meanf; mr[2 * (b * G + g) + 1] = rstd; }
  if (!lane_on) return;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
  const float4 bt = *reinterpret_cast<const float4*>(beta + c);
  const float kx = rstd * gm.x, ky = rstd * gm.y, kz = rstd * gm.z, kw = rstd * gm.w;
  const float sx = bt.x - meanf * kx, sy = bt.y - meanf * ky, sz = bt.z - meanf * kz, sw = bt.w - meanf * kw;
  if (ps == 0) {                                     // the saved per-channel affine (a = rstd * gamma, s = beta - mean * a) of this quad
    float* o = ab + 2 * ((int64_t)b * C + c);
    *reinterpret_cast<float4*>(o) = make_float4(kx, sx, ky, sy);
    *reinterpret_cast<float4*>(o + 4) = make_float4(kz, sz, kw, sw);
  }
#pragma unroll
  for (int i = 0; i < GNF_MAXP; ++i) {
    const int p = ps + i * pps;
    if (p >= HW) continue;
    const int64_t row = (int64_t)b * HW + p;
    float4 o;
    o.x = v[i].x * kx + sx; o.y = v[i].y * ky + sy; o.z = v[i].z * kz + sz; o.w = v[i].w * kw + sw;
    if (SILU) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
    if (MODE == 2) {
      mx_store4(y_hi, y_lo, row, c, ldy, rows_total, o);
    } else if (MODE == 1) {
      uint2 h, l;
      split4(o, h, l);
      *reinterpret_cast<uint2*>(y_hi + plane_off(row, c, ldy, rows_total)) = h;
      *reinterpret_cast<uint2*>(y_lo + plane_off(row, c, ldy, rows_total)) = l;
    } else {
      *reinterpret_cast<float4*>(y + row * ldy + c) = o;
    }
  }
}
static inline bool gn_fused_small_ok(int HW, int C, int G) {
  const int cpg = C / G;
  return HW <= 256 && (cpg & 3) == 0 && cpg <= 4 * GNF_MAXQ && (C & 3) == 0;
}

static inline int gn_ppb(int C) {
  const int C4 = C >> 2;
  const int ppi = C4 <= GN_THREADS ? GN_THREADS / C4 : 1;
  int ppb = GN_PPB < ppi ? ppi : GN_PPB;
  return ((ppb + ppi - 1) / ppi) * ppi;
}

extern "C" size_t ddpo_groupnorm_ws_bytes(int B, int HW, int C, int G) {
  if (B <= 0 || HW <= 0 || C <= 0 || G <= 0) return 0;
  const int chunks = (HW + gn_ppb(C) - 1) / gn_ppb(C);
  return (size_t)B * chunks * G * 2 * sizeof(double);
}
extern "C" size_t ddpo_groupnorm_stats_floats(int B, int C, int G) { return (size_t)B * C * 2 + (size_t)B * G * 2; }

static int groupnorm_fwd_impl(const float* x, int ldx, float* y, uint16_t* y_hi, uint16_t* y_lo, int ldy, const float* gamma,
                              const float* beta, int B, int HW, int C, int G, float eps, int fuse_silu, void* ws, float* stats,
                              void* stream) {
  const bool planes = y == nullptr;
  const bool mx = (fuse_silu & 2) != 0;               // planes entry only: bit 1 of the flag selects the f16mx plane format (ABI v9)
  fuse_silu &= 1;
  if (mx && (!planes || (C & 31) || (ldy & 31))) return DDPO_EINVAL;
  if (!x || !gamma || !beta || !ws || !stats || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > GN_MAXG) return DDPO_EINVAL;
  if (planes ? (!y_hi || !y_lo) : false) return DDPO_EINVAL;
  if ((C & 3) || (C % G) || (ldx & 3) || (ldy & 3) || C > GN_MAXC || B > 65535) return DDPO_EINVAL;
  if (ldy == 0 && (!planes || (C & 31))) return DDPO_EINVAL;          // ldy == 0: k-blocked planes (C / 32, B*HW, 32)
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(stats)) & 15) return DDPO_EINVAL;
  if (planes ? ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 7) != 0 : (reinterpret_cast<uintptr_t>(y) & 15) != 0)
    return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  const int C4 = C >> 2;
  const int ppb = gn_ppb(C);
  const int chunks = (HW + ppb - 1) / ppb;
  double* part = reinterpret_cast<double*>(ws);
  float* ab = stats;                                  // (B, C, 2): a = rstd*gamma, s = beta - mean*a
  float* mr = stats + (size_t)B * C * 2;              // (B, G, 2): mean, rstd
  // (the one-launch form reads gamma / beta as float4: 16-byte aligned parameter vectors only — ParamStore views are; anything else keeps
  // the three-launch form's scalar reads — ADVICE r05)
  if (gn_fused_small_ok(HW, C, G) && ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0) {
    const dim3 grid(G, B), blk(256);
    const int64_t rt = (int64_t)B * HW;
    const int mode = planes ? (mx ? 2 : 1) : 0;
#define GNF_LAUNCH(S, M) hipLaunchKernelGGL((gn_fused_small_kernel<S, M>), grid, blk, 0, st, x, ldx, y, y_hi, y_lo, ldy, gamma, beta, ab, mr, HW, C, G, eps, rt)
    if (fuse_silu) { if (mode == 0) GNF_LAUNCH(true, 0); else if (mode == 1) GNF_LAUNCH(true, 1); else GNF_LAUNCH(true, 2); }
    else { if (mode == 0) GNF_LAUNCH(false, 0); else if (mode == 1) GNF_LAUNCH(false, 1); else GNF_LAUNCH(false, 2); }
#undef GNF_LAUNCH
    DDPO_LAUNCH_CHECK();
    return DDPO_OK;
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, B), dim3(GN_THREADS), 0, st, x, ldx, HW, C, G, ppb, part);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(64), 0, st, part, gamma, beta, ab, mr, chunks, C, G, HW, eps);
  DDPO_LAUNCH_CHECK();
  int64_t blocks = ((int64_t)B * HW * C4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  uint16_t* const no = nullptr;
  if (planes) {
    // lane-paired 16-byte stores need 8-channel granularity and 16-byte aligned plane rows
    const int pair = mx ? 2 : ((C & 7) == 0 && (ldy & 7) == 0 &&
                                ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15) == 0) ? 1 : 0;
    if (fuse_silu)
      hipLaunchKernelGGL((gn_apply_kernel<true, true>), dim3((int)blocks), dim3(256), 0, st, x, ldx, y, ldy, ab, B, HW, C, y_hi, y_lo, pair);
    else
      hipLaunchKernelGGL((gn_apply_kernel<false, true>), dim3((int)blocks), dim3(256), 0, st, x, ldx, y, ldy, ab, B, HW, C, y_hi, y_lo, pair);
  } else if (fuse_silu) {
    hipLaunchKernelGGL((gn_apply_kernel<true, false>), dim3((int)blocks), dim3(256), 0, st, x, ldx, y, ldy, ab, B, HW, C, no, no, 0);
  } else {
    hipLaunchKernelGGL((gn_apply_kernel<false, false>), dim3((int)blocks), dim3(256), 0, st, x, ldx, y, ldy, ab, B, HW, C, no, no, 0);
  }
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_groupnorm_fwd(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, int B,
                                  int HW, int C, int G, float eps, int fuse_silu, void* ws, float* stats, void* stream) {
  if (!y) return DDPO_EINVAL;
  return groupnorm_fwd_impl(x, ldx, y, nullptr, nullptr, ldy, gamma, beta, B, HW, C, G, eps, fuse_silu, ws, stats, stream);
}

extern "C" int ddpo_groupnorm_fwd_planes(const float* x, int ldx, uint16_t* y_hi, uint16_t* y_lo, int ldy, const float* gamma,
                                         const float* beta, int B, int HW, int C, int G, float eps, int fuse_silu, void* ws,
                                         float* stats, void* stream) {
  if (!y_hi || !y_lo) return DDPO_EINVAL;
  return groupnorm_fwd_impl(x, ldx, nullptr, y_hi, y_lo, ldy, gamma, beta, B, HW, C, G, eps, fuse_silu, ws, stats, stream);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 64*4*LN_MAXV = 2560)
// ------------------------------------------------------------------------------------------------
#define LN_MAXV 10
#define LN_RPW 4          // rows per wave of the batched variant (C <= 512: two float4 per lane and row)
// One row's arithmetic (shared by both kernels: identical lane assignment, summation order and rounding -> identical bits)
template <bool PL, int NV>
__device__ __forceinline__ void ln_row(const float4 (&v)[NV], int lane, int C4, int C, float eps, int64_t row, int64_t rows, float* __restrict__ y,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, uint16_t* __restrict__ y_hi,
                                       uint16_t* __restrict__ y_lo, int pair, int ldy) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = lane + j * 64;
    if (c4 < C4) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = lane + j * 64;
    if (c4 < C4) {
      const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  float* yr = PL ? nullptr : y + row * C;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c4 = lane + j * 64;
    if (c4 < C4) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + (c4 << 2));
      const float4 bt = *reinterpret_cast<const float4*>(beta + (c4 << 2));
      float4 o;
      o.x = (v[j].x - mean) * rstd * g.x + bt.x;
      o.y = (v[j].y - mean) * rstd * g.y + bt.y;
      o.z = (v[j].z - mean) * rstd * g.z + bt.z;
      o.w = (v[j].w - mean) * rstd * g.w + bt.w;
      if (PL && pair == 2) {             // f16mx planes
        mx_store4(y_hi, y_lo, row, c4 << 2, ldy, rows, o);
      } else if (PL) {             // bf16 hi / lo planes (rows, C) instead of the fp32 tensor (see gn_apply_kernel)
        uint2 h, l;
        split4(o, h, l);
        if (pair) {           // lane-paired 16-byte stores (see gn_apply_kernel); C % 8 == 0 keeps both lanes of a pair inside the row
          const bool odd = lane & 1;
          const uint32_t rx = lane_xor1_u32(odd ? h.x : l.x), ry = lane_xor1_u32(odd ? h.y : l.y);
          if (!odd) *reinterpret_cast<uint4*>(y_hi + plane_off(row, c4 << 2, ldy, rows)) = make_uint4(h.x, h.y, rx, ry);
          else *reinterpret_cast<uint4*>(y_lo + plane_off(row, (c4 << 2) - 4, ldy, rows)) = make_uint4(rx, ry, l.x, l.y);
        } else {
          *reinterpret_cast<uint2*>(y_hi + plane_off(row, c4 << 2, ldy, rows)) = h;
          *reinterpret_cast<uint2*>(y_lo + plane_off(row, c4 << 2, ldy, rows)) = l;
        }
      } else {
        *reinterpret_cast<float4*>(yr + (c4 << 2)) = o;
      }
    }
  }
}

template <bool PL>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, float eps, uint16_t* __restrict__ y_hi,
                                                        uint16_t* __restrict__ y_lo, int pair, int ldy = -1) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int C4 = C >> 2;
  const float* xr = x + (int64_t)row * C;
  float4 v[LN_MAXV];
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int c4 = lane + j * 64;
    if (c4 < C4) v[j] = *reinterpret_cast<const float4*>(xr + (c4 << 2));
  }
  ln_row<PL, LN_MAXV>(v, lane, C4, C, eps, row, rows, y, gamma, beta, y_hi, y_lo, pair, ldy);
}

// Narrow rows (C <= 512, i.e. the 64x64 / 32x32-level transformer blocks: 65536 / 16384 rows of 320 / 640 channels): one row per wave leaves
// a single 1 KiB + 256 B request in flight per wave and the kernel latency-bound at ~2.9 TB/s; here a wave requests LN_RPW rows before it
// reduces the first.  Per-row arithmetic is ln_row's: bit-identical to layernorm_kernel.
template <bool PL>
__global__ void __launch_bounds__(256) layernorm_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int rows, int C, float eps, uint16_t* __restrict__ y_hi,
                                                             uint16_t* __restrict__ y_lo, int pair, int ldy = -1) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
  if (row0 >= rows) return;
  const int C4 = C >> 2;
  float4 v[LN_RPW][2];
#pragma unroll
  for (int r = 0; r < LN_RPW; ++r) {
    const float* xr = x + (int64_t)min(row0 + r, rows - 1) * C;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c4 = lane + j * 64;
      if (c4 < C4) v[r][j] = *reinterpret_cast<const float4*>(xr + (c4 << 2));
    }
  }
#pragma unroll
  for (int r = 0; r < LN_RPW; ++r)
    if (row0 + r < rows) ln_row<PL, 2>(v[r], lane, C4, C, eps, row0 + r, rows, y, gamma, beta, y_hi, y_lo, pair, ldy);
}

// C == 320 (the 64x64 level of SD-1.x / SD-2.x: 65536 rows at batch 16, the LayerNorm launches that matter): a row is 80 float4 = 16 lanes x 5,
// so a wave works on FOUR rows at once, one per 16-lane DPP row, every lane busy (one row per wave leaves 48 of 128 lane slots idle), and both
// reductions of a row stay inside its 16 lanes: four DPP steps (row_ror:8, row_shl / shr:4 under bank masks, two quad_perms — each verified on
// MI355X to read exactly lane ^ {8, 4, 2, 1}, profiles/r04_probe_reduce.log; 64 ns per reduction against 193 ns for the six ds_bpermute of wave_sum)
// instead of 12 trips through the LDS crossbar per row.  The lane pairing of the 16-byte plane stores is a quad_perm as well.  RS row sets
// (8 rows) are requested per wave before the first is reduced.  Selected by C ALONE (any row count), so a row's bits do not depend on how many
// rows its launch has — the property the sampler / training-forward equality rests on (tests/test_gpu_kernels.py).
__device__ __forceinline__ float row16_xor(float v, int o) {              // the value of lane (i ^ o) inside the lane's 16-lane row, o in {1, 2, 4, 8}
  const int x = __builtin_bit_cast(int, v);
  int t;
  if (o == 1) t = __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);                 // quad_perm [1, 0, 3, 2]
  else if (o == 2) t = __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);            // quad_perm [2, 3, 0, 1]
  else if (o == 8) t = __builtin_amdgcn_mov_dpp(x, 0x128, 0xF, 0xF, true);           // row_ror:8
  else {
    t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);                   // row_shl:4 -> banks 0, 2 (lane i reads i + 4)
    t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);                   // row_shr:4 -> banks 1, 3 (lane i reads i - 4)
  }
  return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float row16_sum(float v) {
  v += row16_xor(v, 8);
  v += row16_xor(v, 4);
  v += row16_xor(v, 2);
  v += row16_xor(v, 1);
  return v;
}
template <bool PL, int NV, int RS>
__global__ void __launch_bounds__(256) layernorm_rows16_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int rows, float eps, uint16_t* __restrict__ y_hi,
                                                               uint16_t* __restrict__ y_lo, int pair, int ldy) {
  constexpr int C = NV * 64;
  const int lane = threadIdx.x & 63, g = lane >> 4, sub = lane & 15;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (4 * RS);
  if (row0 >= rows) return;
  float4 v[RS][NV];
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    const float* xr = x + (int64_t)min(row0 + 4 * s + g, rows - 1) * C;
#pragma unroll
    for (int j = 0; j < NV; ++j) v[s][j] = *reinterpret_cast<const float4*>(xr + ((sub + 16 * j) << 2));
  }
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    const int64_t row = row0 + 4 * s + g;
    const bool live = row < rows;                      // (a clamped duplicate row still takes part in the DPP steps: no divergence in front of them)
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) sm += (v[s][j].x + v[s][j].y) + (v[s][j].z + v[s][j].w);
    const float mean = row16_sum(sm) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float a = v[s][j].x - mean, b = v[s][j].y - mean, c = v[s][j].z - mean, d = v[s][j].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(row16_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c4 = sub + 16 * j;
      const float4 gm = *reinterpret_cast<const float4*>(gamma + (c4 << 2));
      const float4 bt = *reinterpret_cast<const float4*>(beta + (c4 << 2));
      float4 o;
      o.x = (v[s][j].x - mean) * rstd * gm.x + bt.x;
      o.y = (v[s][j].y - mean) * rstd * gm.y + bt.y;
      o.z = (v[s][j].z - mean) * rstd * gm.z + bt.z;
      o.w = (v[s][j].w - mean) * rstd * gm.w + bt.w;
      if (PL && pair == 2) {
        if (live) mx_store4(y_hi, y_lo, row, c4 << 2, ldy, rows, o);
      } else if (PL) {
        uint2 h, l;
        split4(o, h, l);
        if (pair) {                                    // lane-paired 16-byte stores: lanes 2k / 2k + 1 hold adjacent channel quads of one row
          const bool odd = lane & 1;
          const uint32_t sx = odd ? h.x : l.x, sy = odd ? h.y : l.y;
          const uint32_t rx = (uint32_t)__builtin_amdgcn_mov_dpp((int)sx, 0xB1, 0xF, 0xF, true), ry = (uint32_t)__builtin_amdgcn_mov_dpp((int)sy, 0xB1, 0xF, 0xF, true);
          if (live) {
            if (!odd) *reinterpret_cast<uint4*>(y_hi + plane_off(row, c4 << 2, ldy, rows)) = make_uint4(h.x, h.y, rx, ry);
            else *reinterpret_cast<uint4*>(y_lo + plane_off(row, (c4 << 2) - 4, ldy, rows)) = make_uint4(rx, ry, l.x, l.y);
          }
        } else if (live) {
          *reinterpret_cast<uint2*>(y_hi + plane_off(row, c4 << 2, ldy, rows)) = h;
          *reinterpret_cast<uint2*>(y_lo + plane_off(row, c4 << 2, ldy, rows)) = l;
        }
      } else if (live) {
        *reinterpret_cast<float4*>(y + row * C + (c4 << 2)) = o;
      }
    }
  }
}
/* (C = 640 on the same kernel, NV = 10, one row set: measured 4.2525 vs 4.2479 images/s = nothing, profiles/r05_ab_ln16_640.log — those launches already
 * move their 84 MB at 4.5 - 5.3 TB/s; not kept.) */
#define LN16_C 320          /* NV = 5 */
#define LN16_RS 2

extern "C" int ddpo_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, int rows, int C, float eps,
                                  void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV) return DDPO_EINVAL;
  uint16_t* const no = nullptr;
  if (C == LN16_C)
    hipLaunchKernelGGL((layernorm_rows16_kernel<false, LN16_C / 64, LN16_RS>), dim3((rows + 16 * LN16_RS - 1) / (16 * LN16_RS)), dim3(256), 0, as_stream(stream), x, y,
                       gamma, beta, rows, eps, no, no, 0, C);
  else if (C <= 512 && rows >= 4096)
    hipLaunchKernelGGL(layernorm_rows_kernel<false>, dim3((rows + 4 * LN_RPW - 1) / (4 * LN_RPW)), dim3(256), 0, as_stream(stream), x, y, gamma, beta, rows, C,
                       eps, no, no, 0);
  else
    hipLaunchKernelGGL(layernorm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, y, gamma, beta, rows, C, eps, no, no, 0);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_layernorm_fwd_planes(const float* x, uint16_t* y_hi, uint16_t* y_lo, const float* gamma, const float* beta,
                                         int rows, int C, float eps, int kblocked, void* stream) {
  if (!x || !y_hi || !y_lo || !gamma || !beta || rows <= 0 || C <= 0 || (C & 3) || C > 256 * LN_MAXV) return DDPO_EINVAL;
  const bool mx = (kblocked & 2) != 0;                // bit 1 of the layout flag: f16mx plane format (ABI v9); bit 0: k-blocked storage
  kblocked &= 1;
  if ((kblocked || mx) && (C & 31)) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 7) return DDPO_EINVAL;
  float* const nof = nullptr;
  const int pair = mx ? 2 : ((C & 7) == 0 && ((reinterpret_cast<uintptr_t>(y_hi) | reinterpret_cast<uintptr_t>(y_lo)) & 15) == 0) ? 1 : 0;
  if (C == LN16_C)
    hipLaunchKernelGGL((layernorm_rows16_kernel<true, LN16_C / 64, LN16_RS>), dim3((rows + 16 * LN16_RS - 1) / (16 * LN16_RS)), dim3(256), 0, as_stream(stream), x, nof,
                       gamma, beta, rows, eps, y_hi, y_lo, pair, kblocked ? 0 : C);
  else if (C <= 512 && rows >= 4096)
    hipLaunchKernelGGL(layernorm_rows_kernel<true>, dim3((rows + 4 * LN_RPW - 1) / (4 * LN_RPW)), dim3(256), 0, as_stream(stream), x, nof, gamma, beta, rows, C,
                       eps, y_hi, y_lo, pair, kblocked ? 0 : C);
  else
    hipLaunchKernelGGL(layernorm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), x, nof, gamma, beta, rows, C, eps, y_hi,
                       y_lo, pair, kblocked ? 0 : C);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}
