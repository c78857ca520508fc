// Attention backward on the 16-bit MFMA datapath.  Same algorithm as attention_bwd.hip (recompute P from the saved log2-LSE, two sweeps, no
// atomics) and the same operand-chaining trick as attention_bf16.hip: the 32x32 accumulator fragment of the score product (row =
// lane-half-dependent, col = lane&31) is converted in registers into the B operand of the next product, whose 8 reduction slots per lane
// half h in MFMA step u are rows 16u + 4h + {0,1,2,3, 8,9,10,11}.
//   dkdv kernel: workgroup = 4 waves x 32 keys (K, V fragments in registers); sweeps 64-query tiles held in LDS both
//       row-major (A of S = Q K^T and dP = dO V^T) and transposed (A of dV^T += dO^T P and dK^T += Q^T dS)
//   dq kernel:   workgroup = 4 waves x 32 queries (Q, dO fragments in registers); sweeps 64-key tiles: K row-major +
//       transposed, V row-major:  S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - D), dQ^T += K^T dS^T
// Q is pre-scaled by scale*log2(e) wherever it feeds the scores (so dK picks up ln 2 at the end).
//
// Arithmetic (round 4).  Both kernels were matrix-pipe bound at three bf16 passes per product (84 + 60 MFMAs per 64 x 32 block pair,
// profiles/r03_final_train_kernel_stats.md).  Only the SCORES keep the three-pass bf16 split (they are exponentiated).  Every other product
// has one operand that is a single f16 term (11 significant bits, round to nearest even) against the f16 hi + lo split of the other
// (2 passes of v_mfma_f32_32x32x16_f16):
//   dP  = dO' V^T      dO' single,  V hi + lo          dV = P'^T dO'     P' single,  dO' hi + lo
//   dK  = dS'^T Q      dS' single,  Q hi + lo          dQ = dS' K        dS' single, K hi + lo
// 62 + 46 MFMAs instead of 84 + 60.  Ranges: P' = exp2(S - L + 14) in (0, 2^14] (as in the forward); dO' = dO * 2^-e with e the exponent of
// max |dO| over the (batch, head) slab — found by the dvec pre-pass (one atomic max per slab) — so |dO'| < 1 whatever the loss scale of the
// micro-batch (PPO gradients differ by 1e4 between timesteps; f16 keeps 14 binades below the slab maximum normal); dS' = 16 P (dP' - D'),
// clamped to the f16 range.  The powers of two come out exactly in the output stage.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA3(ah, al, bh, bl, c) \
  do { (c) = MFMA32((al), (bh), (c)); (c) = MFMA32((ah), (bl), (c)); (c) = MFMA32((ah), (bh), (c)); } while (0)
// single f16 B operand `b` against the f16 hi / lo split of A
#define MFMA2H(ah, al, b, c) \
  do { (c) = MFMA32H((al), (b), (c)); (c) = MFMA32H((ah), (b), (c)); } while (0)
#define BWD_P_SHIFT 14.0f             /* P' = 2^14 P */
#define BWD_DS_SCALE 16.0f            /* dS' = 16 P (dP' - D') */
#define BWD_F16_MAX 60000.0f

__device__ __forceinline__ uint32_t cvtpk(float lo, float hi) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void split2b(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = cvtpk(a, b);
  lo = cvtpk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}
// f16 conversions are left to the compiler (v_cvt_pk_f16_f32, round to nearest even): their inputs include v_exp_f32 results, and the
// TRANS -> VALU wait state is only inserted for instructions the compiler selects itself (see attention_bf16.hip)
__device__ __forceinline__ uint32_t pk_f16(float a, float b) {
  const f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void split2h(float a, float b, uint32_t& hi, uint32_t& lo) {
  a = fminf(fmaxf(a, -BWD_F16_MAX), BWD_F16_MAX);
  b = fminf(fmaxf(b, -BWD_F16_MAX), BWD_F16_MAX);
  hi = pk_f16(a, b);
  const f16x2 h = __builtin_bit_cast(f16x2, hi);
  lo = pk_f16(a - (float)h[0], b - (float)h[1]);
}
// 8 floats (held as 8 accumulator registers) -> bf16 B-operand fragments hi / lo
__device__ __forceinline__ void frag8(const float* v, bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split2b(v[2 * e], v[2 * e + 1], h[e], l[e]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// 8 floats -> ONE f16 B-operand fragment
__device__ __forceinline__ f16x8 frag8h(const float* v) {
  return __builtin_bit_cast(f16x8, make_uint4(pk_f16(v[0], v[1]), pk_f16(v[2], v[3]), pk_f16(v[4], v[5]), pk_f16(v[6], v[7])));
}
// 8 floats -> f16 hi / lo fragments
__device__ __forceinline__ void frag8h2(const float* v, f16x8& hi, f16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split2h(v[2 * e], v[2 * e + 1], h[e], l[e]);
  hi = __builtin_bit_cast(f16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(f16x8, make_uint4(l[0], l[1], l[2], l[3]));
}
// row-major fragment: 8 consecutive 16-bit values at (row, col0)
template <typename V8>
__device__ __forceinline__ V8 ld_rm(const char* base, int ld, int row, int col0) {
  return *reinterpret_cast<const V8*>(base + (row * ld + col0) * 2);
}
// transposed fragment: image [row][slot] ; slots {s0..s0+3, s0+8..s0+11}
template <typename V8>
__device__ __forceinline__ V8 ld_tr(const char* base, int ld, int row, int s0) {
  const uint2 a = *reinterpret_cast<const uint2*>(base + (row * ld + s0) * 2);
  const uint2 b = *reinterpret_cast<const uint2*>(base + (row * ld + s0 + 8) * 2);
  return __builtin_bit_cast(V8, make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ void scatter_tr(char* tr_hi, char* tr_lo, int ldtr, int tok, int f0, uint32_t h0, uint32_t h1, uint32_t l0, uint32_t l1) {
  uint16_t* th = reinterpret_cast<uint16_t*>(tr_hi);
  uint16_t* tl = reinterpret_cast<uint16_t*>(tr_lo);
  th[(f0 + 0) * ldtr + tok] = (uint16_t)(h0 & 0xFFFFu); th[(f0 + 1) * ldtr + tok] = (uint16_t)(h0 >> 16);
  th[(f0 + 2) * ldtr + tok] = (uint16_t)(h1 & 0xFFFFu); th[(f0 + 3) * ldtr + tok] = (uint16_t)(h1 >> 16);
  tl[(f0 + 0) * ldtr + tok] = (uint16_t)(l0 & 0xFFFFu); tl[(f0 + 1) * ldtr + tok] = (uint16_t)(l0 >> 16);
  tl[(f0 + 2) * ldtr + tok] = (uint16_t)(l1 & 0xFFFFu); tl[(f0 + 3) * ldtr + tok] = (uint16_t)(l1 >> 16);
}
// stage one float4 (4 consecutive feature values of one token):
//   RMB: row-major bf16 hi / lo image (score operand)      RMH1: row-major SINGLE f16 image      RMH2: row-major f16 hi / lo image
//   TRH: transposed f16 hi / lo image
template <bool RMB, bool RMH1, bool RMH2, bool TRH>
__device__ __forceinline__ void stage4(const float4 v, char* rm_hi, char* rm_lo, int ldrm, char* rh_hi, char* rh_lo, char* tr_hi, char* tr_lo,
                                       int ldtr, int tok, int f0) {
  if (RMB) {
    uint32_t h0, l0, h1, l1;
    split2b(v.x, v.y, h0, l0);
    split2b(v.z, v.w, h1, l1);
    *reinterpret_cast<uint2*>(rm_hi + (tok * ldrm + f0) * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(rm_lo + (tok * ldrm + f0) * 2) = make_uint2(l0, l1);
  }
  if (RMH1 || RMH2 || TRH) {
    uint32_t h0, l0, h1, l1;
    split2h(v.x, v.y, h0, l0);
    split2h(v.z, v.w, h1, l1);
    if (RMH1 || RMH2) *reinterpret_cast<uint2*>(rh_hi + (tok * ldrm + f0) * 2) = make_uint2(h0, h1);
    if (RMH2) *reinterpret_cast<uint2*>(rh_lo + (tok * ldrm + f0) * 2) = make_uint2(l0, l1);
    if (TRH) scatter_tr(tr_hi, tr_lo, ldtr, tok, f0, h0, h1, l0, l1);
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
template <int D, int NKS, bool SPLIT>
__device__ __forceinline__ void load_frags_h(const float* p, float mul, int h, f16x8* fh, f16x8* fl) {
#pragma unroll
  for (int s = 0; s < NKS; ++s) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 16 * s + 8 * h + e;
      v[e] = f < D ? p[f] * mul : 0.f;
    }
    if (SPLIT) frag8h2(v, fh[s], fl[s]);
    else fh[s] = frag8h(v);
  }
}
// 2^-e for the slab maximum `amax_bits` (float bits of max |dO|, >= 0): max |dO| * mul lies in [0.5, 1); inv = 1 / mul.  amax == 0 -> 1.
__device__ __forceinline__ void slab_scale(uint32_t amax_bits, float& mul, float& inv) {
  uint32_t e = amax_bits >> 23;
  if (e == 0u) { mul = 1.f; inv = 1.f; return; }
  e = e > 250u ? 250u : e;
  mul = __uint_as_float((253u - e) << 23);
  inv = __uint_as_float((e + 1u) << 23);
}

// ---------------------------------------------------------------------------------------------- dK, dV
template <int D, int DKP, int DVP>
__global__ void __launch_bounds__(256) attn_bwd_dkdv_bf16_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                                 int ldk, const float* __restrict__ v, int ldv,
                                                                 const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                 const float* __restrict__ dvec, const uint32_t* __restrict__ amax,
                                                                 float* __restrict__ dk, float* __restrict__ dv, int heads, int Nq, int Nk,
                                                                 float scale_log2e) {
  constexpr int QT = 64, LDR = DKP + 8, LDT = QT + 4;
  constexpr int NKS = DKP / 16, NDT = DVP / 32;
  constexpr int RM = QT * LDR * 2, TR = DVP * LDT * 2;          // bytes per plane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qh = smem;            char* Ql = Qh + RM;  char* Os = Ql + RM;                             // row-major: Q' bf16 hi / lo, dO' single f16
  char* QTh = Os + RM;        char* QTl = QTh + TR; char* OTh = QTl + TR; char* OTl = OTh + TR;    // transposed f16 hi / lo: Q', dO'
  float* Ls = reinterpret_cast<float*>(OTl + TR);
  float* Dv = Ls + QT;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads, C = heads * D;
  const int key0 = blockIdx.x * 128 + wid * 32;
  const int krow = min(key0 + li, Nk - 1);
  float omul, oinv;
  slab_scale(amax[bh], omul, oinv);
  bf16x8 krh[NKS], krl[NKS];
  f16x8 vrh[NKS], vrl[NKS];
  load_frags<D, NKS>(k + ((int64_t)b * Nk + krow) * ldk + hd * D, 1.0f, h, krh, krl);
  load_frags_h<D, NKS, true>(v + ((int64_t)b * Nk + krow) * ldv + hd * D, 1.0f, h, vrh, vrl);
  for (int i = t; i < (3 * RM + 4 * TR) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);

  f32x16 dvt[NDT], dkt[NDT];
#pragma unroll
  for (int n = 0; n < NDT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvt[n][r] = 0.f; dkt[n][r] = 0.f; }

  const float c1 = BWD_DS_SCALE / 16384.0f;                    // dS' = P' * (dP' - D') * c1 with P' = 2^14 P
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
        ov.x *= omul; ov.y *= omul; ov.z *= omul; ov.w *= omul;
      }
      stage4<true, false, false, true>(qv, Qh, Ql, LDR, nullptr, nullptr, QTh, QTl, LDT, tok, c4 * 4);
      stage4<false, true, false, true>(ov, nullptr, nullptr, LDR, Os, nullptr, OTh, OTl, LDT, tok, c4 * 4);
    }
    if (t < QT) {
      const bool ok = q0 + t < Nq;
      Ls[t] = ok ? lse[(int64_t)bh * Nq + q0 + t] - BWD_P_SHIFT : INFINITY;      // padded queries: P' = exp2(S - inf) = 0
      Dv[t] = ok ? dvec[(int64_t)bh * Nq + q0 + t] * omul * c1 : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 sacc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const bf16x8 ah = ld_rm<bf16x8>(Qh, LDR, 32 * sub + li, 16 * s + 8 * h), al = ld_rm<bf16x8>(Ql, LDR, 32 * sub + li, 16 * s + 8 * h);
        MFMA3(ah, al, krh[s], krl[s], sacc);
        const f16x8 os = ld_rm<f16x8>(Os, LDR, 32 * sub + li, 16 * s + 8 * h);
        dp = MFMA32H(os, vrl[s], dp);
        dp = MFMA32H(os, vrh[s], dp);
      }
      float p[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qq = 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * h;       // accumulator row = query
        p[r] = __builtin_amdgcn_exp2f(sacc[r] - Ls[qq]);
        ds[r] = fminf(fmaxf(p[r] * (dp[r] * c1 - Dv[qq]), -BWD_F16_MAX), BWD_F16_MAX);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f16x8 pf = frag8h(p + 8 * u), sf = frag8h(ds + 8 * u);
        const int s0 = 32 * sub + 16 * u + 4 * h;
#pragma unroll
        for (int n = 0; n < NDT; ++n) {
          const f16x8 oh = ld_tr<f16x8>(OTh, LDT, 32 * n + li, s0), ol = ld_tr<f16x8>(OTl, LDT, 32 * n + li, s0);
          MFMA2H(oh, ol, pf, dvt[n]);
          const f16x8 qh = ld_tr<f16x8>(QTh, LDT, 32 * n + li, s0), ql = ld_tr<f16x8>(QTl, LDT, 32 * n + li, s0);
          MFMA2H(qh, ql, sf, dkt[n]);
        }
      }
    }
  }
  if (key0 + li < Nk) {
    const float fk = 0.6931471805599453f * oinv / BWD_DS_SCALE, fv = oinv / 16384.0f;
    float* pk = dk + ((int64_t)b * Nk + key0 + li) * C + hd * D;
    float* pv = dv + ((int64_t)b * Nk + key0 + li) * C + hd * D;
#pragma unroll
    for (int n = 0; n < NDT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dc = 32 * n + 8 * g + 4 * h;
        if (dc < D) {
          *reinterpret_cast<float4*>(pk + dc) = make_float4(dkt[n][4 * g] * fk, dkt[n][4 * g + 1] * fk, dkt[n][4 * g + 2] * fk, dkt[n][4 * g + 3] * fk);
          *reinterpret_cast<float4*>(pv + dc) = make_float4(dvt[n][4 * g] * fv, dvt[n][4 * g + 1] * fv, dvt[n][4 * g + 2] * fv, dvt[n][4 * g + 3] * fv);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------- dQ
template <int D, int DKP, int DVP>
__global__ void __launch_bounds__(256) attn_bwd_dq_bf16_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k,
                                                               int ldk, const float* __restrict__ v, int ldv,
                                                               const float* __restrict__ d_o, const float* __restrict__ lse,
                                                               const float* __restrict__ dvec, const uint32_t* __restrict__ amax,
                                                               float* __restrict__ dq, int heads, int Nq, int Nk, float scale, float scale_log2e) {
  constexpr int KT = 64, LDR = DKP + 8, LDT = KT + 4;
  constexpr int NKS = DKP / 16, NDT = DVP / 32;
  constexpr int RM = KT * LDR * 2, TR = DVP * LDT * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kh = smem;  char* Kl = Kh + RM;  char* Vh = Kl + RM;  char* Vl = Vh + RM;       // row-major: K bf16 hi / lo, V f16 hi / lo
  char* KTh = Vl + RM;  char* KTl = KTh + TR;                                           // transposed f16 hi / lo: K
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads, C = heads * D;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const int qrow = min(q0 + li, Nq - 1);
  float omul, oinv;
  slab_scale(amax[bh], omul, oinv);
  bf16x8 qrh[NKS], qrl[NKS];
  f16x8 ors[NKS];
  load_frags<D, NKS>(q + ((int64_t)b * Nq + qrow) * ldq + hd * D, scale_log2e, h, qrh, qrl);
  load_frags_h<D, NKS, false>(d_o + ((int64_t)b * Nq + qrow) * C + hd * D, omul, h, ors, nullptr);
  const float c1 = BWD_DS_SCALE / 16384.0f;
  const float L = lse[(int64_t)bh * Nq + qrow] - BWD_P_SHIFT, Dq = dvec[(int64_t)bh * Nq + qrow] * omul * c1;
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
      stage4<true, false, false, true>(kv, Kh, Kl, LDR, nullptr, nullptr, KTh, KTl, LDT, tok, c4 * 4);
      stage4<false, false, true, false>(vv, nullptr, nullptr, LDR, Vh, Vl, nullptr, nullptr, 0, tok, c4 * 4);
    }
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      f32x16 sacc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const bf16x8 ah = ld_rm<bf16x8>(Kh, LDR, 32 * j + li, 16 * s + 8 * h), al = ld_rm<bf16x8>(Kl, LDR, 32 * j + li, 16 * s + 8 * h);
        MFMA3(ah, al, qrh[s], qrl[s], sacc);
        const f16x8 bh_ = ld_rm<f16x8>(Vh, LDR, 32 * j + li, 16 * s + 8 * h), bl_ = ld_rm<f16x8>(Vl, LDR, 32 * j + li, 16 * s + 8 * h);
        MFMA2H(bh_, bl_, ors[s], dp);
      }
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = kt0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h < Nk;   // accumulator row = key
        const float p = ok ? __builtin_amdgcn_exp2f(sacc[r] - L) : 0.f;
        ds[r] = fminf(fmaxf(p * (dp[r] * c1 - Dq), -BWD_F16_MAX), BWD_F16_MAX);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f16x8 sf = frag8h(ds + 8 * u);
        const int s0 = 32 * j + 16 * u + 4 * h;
#pragma unroll
        for (int n = 0; n < NDT; ++n) {
          const f16x8 kh = ld_tr<f16x8>(KTh, LDT, 32 * n + li, s0), kl = ld_tr<f16x8>(KTl, LDT, 32 * n + li, s0);
          MFMA2H(kh, kl, sf, dqt[n]);
        }
      }
    }
  }
  if (q0 + li < Nq) {
    const float fq = scale * oinv / BWD_DS_SCALE;
    float* pq = dq + ((int64_t)b * Nq + q0 + li) * C + hd * D;
#pragma unroll
    for (int n = 0; n < NDT; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dc = 32 * n + 8 * g + 4 * h;
        if (dc < D)
          *reinterpret_cast<float4*>(pq + dc) = make_float4(dqt[n][4 * g] * fq, dqt[n][4 * g + 1] * fq, dqt[n][4 * g + 2] * fq, dqt[n][4 * g + 3] * fq);
      }
  }
}

// D[b,h,q] = sum_j dO[q][h*d+j] * O[q][h*d+j]   (same pre-pass as the fp32 backward)  +  amax[b*heads+h] = max |dO| over the slab, as float bits
// (non-negative floats order like unsigned integers; the buffer is zeroed before the launch)
__global__ void __launch_bounds__(256) attn_dvec_bf16_kernel(const float* __restrict__ o, const float* __restrict__ d_o,
                                                             float* __restrict__ dvec, uint32_t* __restrict__ amax, int B, int heads, int Nq, int d) {
  const int64_t total = (int64_t)B * Nq * heads;
  const int C = heads * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int hh = (int)(i % heads);
    const int64_t row = i / heads;
    const int b = (int)(row / Nq), qq = (int)(row - (int64_t)b * Nq);
    const float4* po = reinterpret_cast<const float4*>(o + row * C + hh * d);
    const float4* pd = reinterpret_cast<const float4*>(d_o + row * C + hh * d);
    float s = 0.f, m = 0.f;
    for (int j = 0; j < d / 4; ++j) {
      const float4 a = po[j], c = pd[j];
      s += (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(c.x), fabsf(c.y))), fmaxf(fabsf(c.z), fabsf(c.w)));
    }
    dvec[((int64_t)b * heads + hh) * Nq + qq] = s;
    if (m < INFINITY) {                                   // (NaN / inf gradients propagate through the products anyway)
      uint32_t* slot = amax + (b * heads + hh);
      const uint32_t bits = __float_as_uint(m);
      if (bits > *reinterpret_cast<volatile uint32_t*>(slot)) atomicMax(slot, bits);
    }
  }
}

template <int D, int DKP, int DVP>
static int launch_bwd_bf16(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o, const float* d_o,
                           const float* lse, float* dvec, float* dq, float* dk, float* dv, int B, int heads, int Nq, int Nk,
                           float scale, hipStream_t st) {
  constexpr int LDR = DKP + 8, LDT = 64 + 4;
  constexpr int RM = 64 * LDR * 2, TR = DVP * LDT * 2;
  constexpr int LDS_A = 3 * RM + 4 * TR + 2 * 64 * (int)sizeof(float);
  constexpr int LDS_B = 4 * RM + 2 * TR;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkdv_bf16_kernel<D, DKP, DVP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_A);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_bf16_kernel<D, DKP, DVP>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
    attr = true;
  }
  const float sl2 = scale * 1.4426950408889634f;
  // the slab maxima live behind the B * heads * Nq floats of `dvec` (the caller allocates B * heads * (Nq + 1) floats, include/ddpo_hip.h)
  uint32_t* amax = reinterpret_cast<uint32_t*>(dvec + (size_t)B * heads * Nq);
  if (hipMemsetAsync(amax, 0, (size_t)B * heads * sizeof(uint32_t), st) != hipSuccess) return DDPO_ELAUNCH;
  int64_t blocks = ((int64_t)B * Nq * heads + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(attn_dvec_bf16_kernel, dim3((int)blocks), dim3(256), 0, st, o, d_o, dvec, amax, B, heads, Nq, D);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_dkdv_bf16_kernel<D, DKP, DVP>), dim3((Nk + 127) / 128, B * heads), dim3(256), LDS_A, st, q, ldq, k, ldk, v,
                     ldv, d_o, lse, dvec, amax, dk, dv, heads, Nq, Nk, sl2);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<D, DKP, DVP>), dim3((Nq + 127) / 128, B * heads), dim3(256), LDS_B, st, q, ldq, k, ldk, v, ldv,
                     d_o, lse, dvec, amax, dq, heads, Nq, Nk, scale, sl2);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// ================================================================================================
// The `bf16x3` datapath's backward (ddpo_attention_bwd_bf16x3): EVERY product on three bf16 passes (all operands hi + lo), no scaling — the
// round-1..3 kernels, kept as the reference-accuracy variant (84 + 60 MFMAs per 64 x 32 block pair).
// ================================================================================================
namespace x3 {
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvtpk(float lo, float hi) {       // v_cvt_pk_bf16_f32, selected by the compiler (hazards handled)
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
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

}  // namespace x3

extern "C" int ddpo_attention_bwd_bf16x3(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                                         const float* d_o, const float* lse, float* dvec, float* dq, float* dk, float* dv, int B,
                                         int heads, int Nq, int Nk, int d, float scale, void* stream) {
  if (!q || !k || !v || !o || !d_o || !lse || !dvec || !dq || !dk || !dv) return DDPO_EINVAL;
  if (B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || (ldq & 3) || (ldk & 3) || (ldv & 3) || (long)B * heads > 65535) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
#define BWDB(DD, DK, DV) return x3::launch_bwd_bf16<DD, DK, DV>(q, ldq, k, ldk, v, ldv, o, d_o, lse, dvec, dq, dk, dv, B, heads, Nq, Nk, scale, st)
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

/* The f16mx datapath's backward (see the head of this file); dvec: B * heads * (Nq + 1) floats. */
extern "C" int ddpo_attention_bwd_f16p(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
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
