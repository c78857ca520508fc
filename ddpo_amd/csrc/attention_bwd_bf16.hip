// Attention backward on the bf16 MFMA datapath (v_mfma_f32_32x32x16_bf16, bf16x3 split: every operand hi + lo, three
// passes per product, fp32 accumulate).  Same algorithm as attention_bwd.hip (recompute P from the saved log2-LSE, two
// sweeps, no atomics) and the same operand-chaining trick as attention_bf16.hip: the 32x32 accumulator fragment of the
// score product (row = lane-half-dependent, col = lane&31) is converted in registers into the B operand of the next
// product, whose 8 reduction slots per lane half h in MFMA step u are rows 16u + 4h + {0,1,2,3, 8,9,10,11}.
//   dkdv kernel: workgroup = 4 waves x 32 keys (K, V fragments in registers); sweeps 64-query tiles held in LDS both
//       row-major (A of S = Q K^T and dP = dO V^T) and transposed (A of dV^T += dO^T P and dK^T += Q^T dS)
//   dq kernel:   workgroup = 4 waves x 32 queries (Q, dO fragments in registers); sweeps 64-key tiles: K row-major +
//       transposed, V row-major:  S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - D), dQ^T += K^T dS^T
// Q is pre-scaled by scale*log2(e) wherever it feeds the scores (so dK picks up ln 2 at the end).
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA3(ah, al, bh, bl, c) \
  do { (c) = MFMA32((al), (bh), (c)); (c) = MFMA32((ah), (bl), (c)); (c) = MFMA32((ah), (bh), (c)); } while (0)

__device__ __forceinline__ uint32_t cvtpk(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void split2b(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = cvtpk(a, b);
  lo = cvtpk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}
// 8 floats (held as 8 accumulator registers) -> B-operand fragments hi / lo
__device__ __forceinline__ void frag8(const float* v, bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split2b(v[2 * e], v[2 * e + 1], h[e], l[e]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// row-major fragment: 8 consecutive bf16 at (row, col0)
__device__ __forceinline__ bf16x8 ld_rm(const char* base, int ld, int row, int col0) {
  return *reinterpret_cast<const bf16x8*>(base + (row * ld + col0) * 2);
}
// transposed fragment: image [row][slot] ; slots {s0..s0+3, s0+8..s0+11}
__device__ __forceinline__ bf16x8 ld_tr(const char* base, int ld, int row, int s0) {
  const uint2 a = *reinterpret_cast<const uint2*>(base + (row * ld + s0) * 2);
  const uint2 b = *reinterpret_cast<const uint2*>(base + (row * ld + s0 + 8) * 2);
  return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}
// stage one float4 (4 consecutive feature values of one token) into a row-major image (b64) and, optionally, a transposed one
__device__ __forceinline__ void stage4(const float4 v, char* rm_hi, char* rm_lo, int ldrm, char* tr_hi, char* tr_lo, int ldtr,
                                       int tok, int f0) {
  uint32_t h0, l0, h1, l1;
  split2b(v.x, v.y, h0, l0);
  split2b(v.z, v.w, h1, l1);
  *reinterpret_cast<uint2*>(rm_hi + (tok * ldrm + f0) * 2) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(rm_lo + (tok * ldrm + f0) * 2) = make_uint2(l0, l1);
  if (tr_hi) {
    uint16_t* th = reinterpret_cast<uint16_t*>(tr_hi);
    uint16_t* tl = reinterpret_cast<uint16_t*>(tr_lo);
    th[(f0 + 0) * ldtr + tok] = (uint16_t)(h0 & 0xFFFFu); th[(f0 + 1) * ldtr + tok] = (uint16_t)(h0 >> 16);
    th[(f0 + 2) * ldtr + tok] = (uint16_t)(h1 & 0xFFFFu); th[(f0 + 3) * ldtr + tok] = (uint16_t)(h1 >> 16);
    tl[(f0 + 0) * ldtr + tok] = (uint16_t)(l0 & 0xFFFFu); tl[(f0 + 1) * ldtr + tok] = (uint16_t)(l0 >> 16);
    tl[(f0 + 2) * ldtr + tok] = (uint16_t)(l1 & 0xFFFFu); tl[(f0 + 3) * ldtr + tok] = (uint16_t)(l1 >> 16);
  }
}
// B-operand fragments (registers) of one token's feature vector: lane (token = li, half h) holds features 16s + 8h .. +8
template <int D, int NKS>
__device__ __forceinline__ void load_frags(const float* p, float mul, int h, bf16x8* fh, bf16x8* fl) {
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 16 * s + 8 * h + e;
      v[e] = f < D ? p[f] * mul : 0.f;
    }
    frag8(v, fh[s], fl[s]);
  }
}

// ---------------------------------------------------------------------------------------------- dK, dV
template <int D, int DKP, int DVP>
__global__ void __launch_bounds__(256) attn_bwd_dkdv_bf16_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                                 int ldk, const float* __restrict__ v, int ldv,
                                                                 const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                 const float* __restrict__ dvec, float* __restrict__ dk,
                                                                 float* __restrict__ dv, int heads, int Nq, int Nk,
                                                                 float scale_log2e) {
  constexpr int QT = 64, LDR = DKP + 8, LDT = QT + 4;
  constexpr int NKS = DKP / 16, NDT = DVP / 32;
  constexpr int RM = QT * LDR * 2, TR = DVP * LDT * 2;          // bytes per plane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qh = smem;            char* Ql = Qh + RM;  char* Oh = Ql + RM;  char* Ol = Oh + RM;        // row-major Q', dO
  char* QTh = Ol + RM;        char* QTl = QTh + TR; char* OTh = QTl + TR; char* OTl = OTh + TR;    // transposed
  float* Ls = reinterpret_cast<float*>(OTl + TR);
  float* Dv = Ls + QT;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads, C = heads * D;
  const int key0 = blockIdx.x * 128 + wid * 32;
  const int krow = min(key0 + li, Nk - 1);
  bf16x8 krh[NKS], krl[NKS], vrh[NKS], vrl[NKS];
  load_frags<D, NKS>(k + ((int64_t)b * Nk + krow) * ldk + hd * D, 1.0f, h, krh, krl);
  load_frags<D, NKS>(v + ((int64_t)b * Nk + krow) * ldv + hd * D, 1.0f, h, vrh, vrl);
  for (int i = t; i < (4 * RM + 4 * TR) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);

  f32x16 dvt[NDT], dkt[NDT];
#pragma unroll
  for (int n = 0; n < NDT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvt[n][r] = 0.f; dkt[n][r] = 0.f; }

  const float* qb = q + (int64_t)b * Nq * ldq + hd * D;
  const float* ob = d_o + (int64_t)b * Nq * C + hd * D;
  for (int q0 = 0; q0 < Nq; q0 += QT) {
    __syncthreads();
    for (int i = t; i < QT * (D / 4); i += 256) {
      const int tok = i / (D / 4), c4 = i - tok * (D / 4);
      float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), ov = qv;
      if (q0 + tok < Nq) {
        qv = *reinterpret_cast<const float4*>(qb + (int64_t)(q0 + tok) * ldq + c4 * 4);
        ov = *reinterpret_cast<const float4*>(ob + (int64_t)(q0 + tok) * C + c4 * 4);
        qv.x *= scale_log2e; qv.y *= scale_log2e; qv.z *= scale_log2e; qv.w *= scale_log2e;
      }
      stage4(qv, Qh, Ql, LDR, QTh, QTl, LDT, tok, c4 * 4);
      stage4(ov, Oh, Ol, LDR, OTh, OTl, LDT, tok, c4 * 4);
    }
    if (t < QT) {
      const bool ok = q0 + t < Nq;
      Ls[t] = ok ? lse[(int64_t)bh * Nq + q0 + t] : INFINITY;         // padded queries: P = exp2(S - inf) = 0
      Dv[t] = ok ? dvec[(int64_t)bh * Nq + q0 + t] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 sacc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const bf16x8 ah = ld_rm(Qh, LDR, 32 * sub + li, 16 * s + 8 * h), al = ld_rm(Ql, LDR, 32 * sub + li, 16 * s + 8 * h);
        MFMA3(ah, al, krh[s], krl[s], sacc);
        const bf16x8 bh_ = ld_rm(Oh, LDR, 32 * sub + li, 16 * s + 8 * h), bl_ = ld_rm(Ol, LDR, 32 * sub + li, 16 * s + 8 * h);
        MFMA3(bh_, bl_, vrh[s], vrl[s], dp);
      }
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * h;       // accumulator row = query
        p[r] = exp2f(sacc[r] - Ls[qq]);
        ds[r] = p[r] * (dp[r] - Dv[qq]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 ph, pl, sh, sl;
        frag8(p + 8 * u, ph, pl);
        frag8(ds + 8 * u, sh, sl);
        const int s0 = 32 * sub + 16 * u + 4 * h;
#pragma unroll
        for (int n = 0; n < NDT; ++n) {
          const bf16x8 oh = ld_tr(OTh, LDT, 32 * n + li, s0), ol = ld_tr(OTl, LDT, 32 * n + li, s0);
          MFMA3(oh, ol, ph, pl, dvt[n]);
          const bf16x8 qh = ld_tr(QTh, LDT, 32 * n + li, s0), ql = ld_tr(QTl, LDT, 32 * n + li, s0);
          MFMA3(qh, ql, sh, sl, dkt[n]);
        }
      }
    }
  }
  if (key0 + li < Nk) {
    const float ln2 = 0.6931471805599453f;
    float* pk = dk + ((int64_t)b * Nk + key0 + li) * C + hd * D;
    float* pv = dv + ((int64_t)b * Nk + key0 + li) * C + hd * D;
#pragma unroll
    for (int n = 0; n < NDT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dc = 32 * n + 8 * g + 4 * h;
        if (dc < D) {
          *reinterpret_cast<float4*>(pk + dc) = make_float4(dkt[n][4 * g] * ln2, dkt[n][4 * g + 1] * ln2, dkt[n][4 * g + 2] * ln2, dkt[n][4 * g + 3] * ln2);
          *reinterpret_cast<float4*>(pv + dc) = make_float4(dvt[n][4 * g], dvt[n][4 * g + 1], dvt[n][4 * g + 2], dvt[n][4 * g + 3]);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------- dQ
template <int D, int DKP, int DVP>
__global__ void __launch_bounds__(256) attn_bwd_dq_bf16_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                               int ldk, const float* __restrict__ v, int ldv,
                                                               const float* __restrict__ d_o, const float* __restrict__ lse,
                                                               const float* __restrict__ dvec, float* __restrict__ dq, int heads,
                                                               int Nq, int Nk, float scale, float scale_log2e) {
  constexpr int KT = 64, LDR = DKP + 8, LDT = KT + 4;
  constexpr int NKS = DKP / 16, NDT = DVP / 32;
  constexpr int RM = KT * LDR * 2, TR = DVP * LDT * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kh = smem;  char* Kl = Kh + RM;  char* Vh = Kl + RM;  char* Vl = Vh + RM;
  char* KTh = Vl + RM;  char* KTl = KTh + TR;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads, C = heads * D;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const int qrow = min(q0 + li, Nq - 1);
  bf16x8 qrh[NKS], qrl[NKS], orh[NKS], orl[NKS];
  load_frags<D, NKS>(q + ((int64_t)b * Nq + qrow) * ldq + hd * D, scale_log2e, h, qrh, qrl);
  load_frags<D, NKS>(d_o + ((int64_t)b * Nq + qrow) * C + hd * D, 1.0f, h, orh, orl);
  const float L = lse[(int64_t)bh * Nq + qrow], Dq = dvec[(int64_t)bh * Nq + qrow];
  for (int i = t; i < (4 * RM + 2 * TR) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);

  f32x16 dqt[NDT];
#pragma unroll
  for (int n = 0; n < NDT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[n][r] = 0.f;
  const float* kb = k + (int64_t)b * Nk * ldk + hd * D;
  const float* vb = v + (int64_t)b * Nk * ldv + hd * D;
  for (int kt0 = 0; kt0 < Nk; kt0 += KT) {
    __syncthreads();
    for (int i = t; i < KT * (D / 4); i += 256) {
      const int tok = i / (D / 4), c4 = i - tok * (D / 4);
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kt0 + tok < Nk) {
        kv = *reinterpret_cast<const float4*>(kb + (int64_t)(kt0 + tok) * ldk + c4 * 4);
        vv = *reinterpret_cast<const float4*>(vb + (int64_t)(kt0 + tok) * ldv + c4 * 4);
      }
      stage4(kv, Kh, Kl, LDR, KTh, KTl, LDT, tok, c4 * 4);
      stage4(vv, Vh, Vl, LDR, nullptr, nullptr, 0, tok, c4 * 4);
    }
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      f32x16 sacc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const bf16x8 ah = ld_rm(Kh, LDR, 32 * j + li, 16 * s + 8 * h), al = ld_rm(Kl, LDR, 32 * j + li, 16 * s + 8 * h);
        MFMA3(ah, al, qrh[s], qrl[s], sacc);
        const bf16x8 bh_ = ld_rm(Vh, LDR, 32 * j + li, 16 * s + 8 * h), bl_ = ld_rm(Vl, LDR, 32 * j + li, 16 * s + 8 * h);
        MFMA3(bh_, bl_, orh[s], orl[s], dp);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = kt0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h < Nk;   // accumulator row = key
        const float p = ok ? exp2f(sacc[r] - L) : 0.f;
        ds[r] = p * (dp[r] - Dq);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 sh, sl;
        frag8(ds + 8 * u, sh, sl);
        const int s0 = 32 * j + 16 * u + 4 * h;
#pragma unroll
        for (int n = 0; n < NDT; ++n) {
          const bf16x8 kh = ld_tr(KTh, LDT, 32 * n + li, s0), kl = ld_tr(KTl, LDT, 32 * n + li, s0);
          MFMA3(kh, kl, sh, sl, dqt[n]);
        }
      }
    }
  }
  if (q0 + li < Nq) {
    float* pq = dq + ((int64_t)b * Nq + q0 + li) * C + hd * D;
#pragma unroll
    for (int n = 0; n < NDT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dc = 32 * n + 8 * g + 4 * h;
        if (dc < D)
          *reinterpret_cast<float4*>(pq + dc) = make_float4(dqt[n][4 * g] * scale, dqt[n][4 * g + 1] * scale, dqt[n][4 * g + 2] * scale, dqt[n][4 * g + 3] * scale);
      }
  }
}

// D[b,h,q] = sum_j dO[q][h*d+j] * O[q][h*d+j]   (same pre-pass as the fp32 backward)
__global__ void __launch_bounds__(256) attn_dvec_bf16_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                             float* __restrict__ dvec, int B, int heads, int Nq, int d) {
  const int64_t total = (int64_t)B * Nq * heads;
  const int C = heads * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int hh = (int)(i % heads);
    const int64_t row = i / heads;
    const int b = (int)(row / Nq), qq = (int)(row - (int64_t)b * Nq);
    const float4* po = reinterpret_cast<const float4*>(o + row * C + hh * d);
    const float4* pd = reinterpret_cast<const float4*>(d_o + row * C + hh * d);
    float s = 0.f;
    for (int j = 0; j < d / 4; ++j) {
      const float4 a = po[j], c = pd[j];
      s += (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
    }
    dvec[((int64_t)b * heads + hh) * Nq + qq] = s;
  }
}

template <int D, int DKP, int DVP>
static int launch_bwd_bf16(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* d_o,
                           const float* lse, float* dvec, float* dq, float* dk, float* dv, int B, int heads, int Nq, int Nk,
                           float scale, hipStream_t st) {
  constexpr int LDR = DKP + 8, LDT = 64 + 4;
  constexpr int RM = 64 * LDR * 2, TR = DVP * LDT * 2;
  constexpr int LDS_A = 4 * RM + 4 * TR + 2 * 64 * (int)sizeof(float);
  constexpr int LDS_B = 4 * RM + 2 * TR;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_bf16_kernel<D, DKP, DVP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_A);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_bf16_kernel<D, DKP, DVP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
    attr = true;
  }
  const float sl2 = scale * 1.4426950408889634f;
  int64_t blocks = ((int64_t)B * Nq * heads + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(attn_dvec_bf16_kernel, dim3((int)blocks), dim3(256), 0, st, o, d_o, dvec, B, heads, Nq, D);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_dkdv_bf16_kernel<D, DKP, DVP>), dim3((Nk + 127) / 128, B * heads), dim3(256), LDS_A, st, q, ldq, k, ldk, v,
                     ldv, d_o, lse, dvec, dk, dv, heads, Nq, Nk, sl2);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<D, DKP, DVP>), dim3((Nq + 127) / 128, B * heads), dim3(256), LDS_B, st, q, ldq, k, ldk, v, ldv,
                     d_o, lse, dvec, dq, heads, Nq, Nk, scale, sl2);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_attention_bwd_bf16x3(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                         const float* d_o, const float* lse, float* dvec, float* dq, float* dk, float* dv, int B,
                                         int heads, int Nq, int Nk, int d, float scale, void* stream) {
  if (!q || !k || !v || !o || !d_o || !lse || !dvec || !dq || !dk || !dv) return DDPO_EINVAL;
  if (B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || (ldq & 3) || (ldk & 3) || (ldv & 3) || (long)B * heads > 65535) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
#define BWDB(DD, DK, DV) return launch_bwd_bf16<DD, DK, DV>(q, ldq, k, ldk, v, ldv, o, d_o, lse, dvec, dq, dk, dv, B, heads, Nq, Nk, scale, st)
  switch (d) {
    case 8: BWDB(8, 16, 32);
    case 16: BWDB(16, 16, 32);
    case 40: BWDB(40, 48, 64);
    case 64: BWDB(64, 64, 64);
    case 80: BWDB(80, 80, 96);
    default: return DDPO_EINVAL;
  }
#undef BWDB
}
