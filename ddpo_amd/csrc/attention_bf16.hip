// Flash attention forward on the 16-bit MFMA datapath, in two arithmetic variants of the second product (template flag F16P; the first product
// is the same in both).  F16P = false is the `bf16x3` datapath's operator (ddpo_attention_fwd_bf16x3*): P and V split into bf16 hi + lo, three
// passes, ~1e-6 on the output.  F16P = true is the `f16mx` datapath's (ddpo_attention_fwd_f16p*, round 4): described below, ~1e-5.
//   S^T (keys x queries) = K Q^T : bf16x3 — K and the pre-scaled Q split into bf16 hi + lo, three MFMA passes (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi,
//                                  fp32 accumulate, ~1e-5 relative: the scores are exponentiated, they keep the full split)
//   O^T (d x queries)    = V^T P^T: since round 4 the probabilities are ONE f16 term and V is split into f16 hi + lo — two passes of
//                                  v_mfma_f32_32x32x16_f16 instead of three bf16 passes, and one v_cvt_pk_f16_f32 per probability pair instead of
//                                  the six-instruction bf16 hi / lo split.  p = exp2(s - m + 14) lies in (0, 2^14]: everything down to 2^-28 of the
//                                  row maximum is a NORMAL f16 number (11 significant bits, round to nearest even).  The softmax denominator is
//                                  the sum of the SAME rounded probabilities — row D of V^T is all ones wherever the head dim leaves a spare row
//                                  of the 32-row MFMA tile (d = 8, 16, 40, 80: the sum falls out of the second product for free), the sum of the
//                                  values converted back for d = 64 — so O = sum(p~ v) / sum(p~) is an exact convex combination of the values
//                                  with weights perturbed by <= 2^-12 relative: a row dominated by one key returns that value exactly.
// Structure mirrors attention.hip: A = K tile (LDS, [key][dk] bf16), B = Q^T (registers); A = V^T tile (LDS, [d][key] f16, transposed while it
// is staged), B = P^T.  One workgroup = 4 waves x 32 queries; the 32x32 C fragment of S^T (row = key, col = query = lane&31) is converted
// in registers to the B operand of the second product: for lane half h the 8 k-slots of MFMA step u are the keys
// 16u + 4h + {0,1,2,3, 8,9,10,11} (exactly the rows that half holds), and the V^T fragment is read with the same map.
#include "common.h"
#include <cstdlib>


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16_a(float lo, float hi) {      // v_cvt_pk_bf16_f32, selected by the compiler (see cvt_pk_f16_a)
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// Two floats -> packed f16 pair, round to nearest even (v_cvt_pk_f16_f32).  Deliberately NOT inline assembly: the operands are the results
// of v_exp_f32, a transcendental-unit instruction, and gfx950 needs a wait state between a TRANS result and a VALU consumer — the compiler's
// hazard recognizer inserts it for instructions it selects itself and does not look inside an asm statement (round 4: the asm form read stale
// registers whenever the scheduler put the last exponential of a pair directly in front of it: ~10 % errors on long key sequences, NaNs at d = 64).
__device__ __forceinline__ uint32_t cvt_pk_f16_a(float lo, float hi) {
  const f16x2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// two floats -> packed bf16 hi pair and packed bf16 lo pair
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = cvt_pk_bf16_a(a, b);
  lo = cvt_pk_bf16_a(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}
// two floats -> packed f16 hi pair and packed f16 lo pair (the V operand; magnitudes beyond the f16 range saturate)
__device__ __forceinline__ void split2h(float a, float b, uint32_t& hi, uint32_t& lo) {
  const float lim = 65504.f;
  a = fminf(fmaxf(a, -lim), lim);
  b = fminf(fmaxf(b, -lim), lim);
  hi = cvt_pk_f16_a(a, b);
  const f16x2 h = __builtin_bit_cast(f16x2, hi);
  lo = cvt_pk_f16_a(a - (float)h[0], b - (float)h[1]);
}
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define ATTN_P_SHIFT 14.0f            /* p = exp2(s - m + 14): the f16 probabilities use the exponent range [2^-14, 2^14] */
#define ATTN_F16_ONE 0x3C00u

// Online-softmax step shared by the three kernels below (identical instruction sequence -> identical bits): running maximum, the tile's
// probabilities as f16 B-operand fragments (step u = 2j + half uses accumulator registers 8 half .. 8 half + 7 of sub-tile j), and — only when
// the denominator does not come out of the second product (ONES == false) — the sum of the rounded probabilities.
template <bool ONES>
__device__ __forceinline__ void attn_softmax_tile(const f32x16 (&sacc)[2], float& m_run, float& l_run, f16x8 (&ph)[4], float& alpha, bool& grew) {
  float mx = sacc[0][0];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[j][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_new = fmaxf(m_run, mx);
  alpha = __builtin_amdgcn_exp2f(m_run - m_new);
  grew = m_new > m_run;
  m_run = m_new;
  const float m_sh = m_new - ATTN_P_SHIFT;
  float ls = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint32_t pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __builtin_amdgcn_exp2f(sacc[j][8 * half + 2 * e] - m_sh);
        const float p1 = __builtin_amdgcn_exp2f(sacc[j][8 * half + 2 * e + 1] - m_sh);
        pk[e] = cvt_pk_f16_a(p0, p1);
        if (!ONES) {
          const f16x2 q = __builtin_bit_cast(f16x2, pk[e]);
          ls += (float)q[0] + (float)q[1];
        }
      }
      ph[2 * j + half] = __builtin_bit_cast(f16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
    }
  }
  if (!ONES) l_run = l_run * alpha + ls;
}

// O^T += V^T P^T for one 64-key tile: A fragment of step u = V^T[d = 32n + li][16u + 4h + {0..3, 8..11}] (f16 hi / lo planes in LDS), B = ph[u].
// Per accumulator the order is fixed (lo then hi, u ascending) — shared by all three kernels; the fragments of step u + 1 are requested before
// the MFMAs of step u are issued, and the two passes of a step alternate between the accumulators.
template <int NDT, int LDVT, int QB = 1>
__device__ __forceinline__ void attn_pv_tile(const char* Vhi, const char* Vlo, int li, int h, const f16x8 (*ph)[4], f32x16 (*oacc)[NDT]) {
  f16x8 vh[2][NDT], vl[2][NDT];
  auto fetch = [&](int u, int slot) {
#pragma unroll
    for (int n = 0; n < NDT; ++n) {
      const int off = ((32 * n + li) * LDVT + 16 * u + 4 * h) * 2;
      const uint2 a0 = *reinterpret_cast<const uint2*>(Vhi + off), a1 = *reinterpret_cast<const uint2*>(Vhi + off + 16);
      const uint2 c0 = *reinterpret_cast<const uint2*>(Vlo + off), c1 = *reinterpret_cast<const uint2*>(Vlo + off + 16);
      vh[slot][n] = __builtin_bit_cast(f16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
      vl[slot][n] = __builtin_bit_cast(f16x8, make_uint4(c0.x, c0.y, c1.x, c1.y));
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u + 1 < 4) fetch(u + 1, (u + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);       // keep the requests of step u + 1 in front of the MFMAs of step u (the scheduler sinks them otherwise)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int n = 0; n < NDT; ++n) oacc[qb][n] = MFMA32H(vl[u & 1][n], ph[qb][u], oacc[qb][n]);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int n = 0; n < NDT; ++n) oacc[qb][n] = MFMA32H(vh[u & 1][n], ph[qb][u], oacc[qb][n]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// softmax denominator of query li (scaled by 2^14 like the accumulators): row D of O^T where V^T carries the ones row, else the running sum
template <int D, int NDT, bool ONES>
__device__ __forceinline__ float attn_row_sum(const f32x16 (&oacc)[NDT], float l_run, int li) {
  if constexpr (ONES) {
    constexpr int n1 = D / 32, r1 = D % 32;
    static_assert(r1 % 8 == 0 && n1 < NDT, "row D must be register 4 (r1 / 8) of the lanes with h == 0");
    return __shfl(oacc[n1][4 * (r1 / 8)], li, 64);         // registers 4g .. 4g+3 of lane (li, h) are rows 32n + 8g + 4h + {0..3}
  } else {
    return l_run + __shfl_xor(l_run, 32, 64);
  }
}

// ---- the bf16x3 variant of the same two steps (F16P == false): probabilities split into bf16 hi + lo, the denominator summed in fp32 from
// the unrounded values, three MFMA passes per accumulator and step in the order vl*ph, vh*pl, vh*ph
__device__ __forceinline__ void attn_softmax_tile_x3(const f32x16 (&sacc)[2], float& m_run, float& l_run, bf16x8 (&ph)[4], bf16x8 (&pl)[4], float& alpha,
                                                     bool& grew) {
  float mx = sacc[0][0];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[j][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  const float m_new = fmaxf(m_run, mx);
  alpha = __builtin_amdgcn_exp2f(m_run - m_new);
  grew = m_new > m_run;
  m_run = m_new;
  float ls = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __builtin_amdgcn_exp2f(sacc[j][8 * half + 2 * e] - m_new);
        const float p1 = __builtin_amdgcn_exp2f(sacc[j][8 * half + 2 * e + 1] - m_new);
        ls += p0 + p1;
        split2(p0, p1, hi[e], lo[e]);
      }
      ph[2 * j + half] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
      pl[2 * j + half] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
  }
  l_run = l_run * alpha + ls;
}

template <int NDT, int LDVT, int QB = 1>
__device__ __forceinline__ void attn_pv_tile_x3(const char* Vhi, const char* Vlo, int li, int h, const bf16x8 (*ph)[4], const bf16x8 (*pl)[4],
                                                f32x16 (*oacc)[NDT]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int n = 0; n < NDT; ++n) {
      const int off = ((32 * n + li) * LDVT + 16 * u + 4 * h) * 2;
      const uint2 a0 = *reinterpret_cast<const uint2*>(Vhi + off), a1 = *reinterpret_cast<const uint2*>(Vhi + off + 16);
      const uint2 c0 = *reinterpret_cast<const uint2*>(Vlo + off), c1 = *reinterpret_cast<const uint2*>(Vlo + off + 16);
      const bf16x8 vh = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
      const bf16x8 vl = __builtin_bit_cast(bf16x8, make_uint4(c0.x, c0.y, c1.x, c1.y));
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        oacc[qb][n] = MFMA32(vl, ph[qb][u], oacc[qb][n]);
        oacc[qb][n] = MFMA32(vh, pl[qb][u], oacc[qb][n]);
        oacc[qb][n] = MFMA32(vh, ph[qb][u], oacc[qb][n]);
      }
    }
  }
}

// One tile's softmax + second product for either variant (all three kernels call this, so a variant's bits do not depend on the kernel)
template <bool F16P, bool ONES, int NDT, int LDVT>
__device__ __forceinline__ void attn_tile_tail(const f32x16 (&sacc)[2], float& m_run, float& l_run, f32x16 (&oacc)[NDT], const char* Vhi, const char* Vlo,
                                               int li, int h) {
  float alpha;
  bool grew;
  if constexpr (F16P) {
    f16x8 ph[4];
    attn_softmax_tile<ONES>(sacc, m_run, l_run, ph, alpha, grew);
    if (__any(grew)) {                     // the running maximum settles after a few tiles; skip the no-op rescale (alpha == 1)
#pragma unroll
      for (int n = 0; n < NDT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[n][r] *= alpha;
    }
    attn_pv_tile<NDT, LDVT>(Vhi, Vlo, li, h, &ph, &oacc);
  } else {
    bf16x8 ph[4], pl[4];
    attn_softmax_tile_x3(sacc, m_run, l_run, ph, pl, alpha, grew);
    if (__any(grew)) {
#pragma unroll
      for (int n = 0; n < NDT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[n][r] *= alpha;
    }
    attn_pv_tile_x3<NDT, LDVT>(Vhi, Vlo, li, h, &ph, &pl, &oacc);
  }
}

// Where the normalised output rows go.  hi == nullptr: fp32 (rows, ld).  hi != nullptr (ABI v12, ddpo_attention_fwd_*_po): the bf16 hi / lo planes
// of the SAME fp32 values (the split the fp32-fed GEMM loader applies) — the operand format of the plane-fed to_out projection, so that layer
// never reads an fp32 tensor; ld = plane row stride in elements (0: k-blocked planes (C / 32, rows, 32)), rows = B * Nq.
struct AttnOut {
  float* o;
  uint16_t* hi;
  uint16_t* lo;
  int ld;
  int64_t rows;
};
template <int D, int NDT>
__device__ __forceinline__ void attn_store_o(const AttnOut out, int64_t row, int col0, const f32x16 (&oacc)[NDT], float inv, int h) {
#pragma unroll
  for (int n = 0; n < NDT; ++n) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {            // registers 4g..4g+3 are rows 32n + 8g + 4h + {0,1,2,3} of O^T = 4 consecutive channels
      const int dc = 32 * n + 8 * g + 4 * h;
      if (dc >= D) continue;
      const float v0 = oacc[n][4 * g] * inv, v1 = oacc[n][4 * g + 1] * inv, v2 = oacc[n][4 * g + 2] * inv, v3 = oacc[n][4 * g + 3] * inv;
      if (out.hi) {
        uint32_t h0, l0, h1, l1;
        split2(v0, v1, h0, l0);
        split2(v2, v3, h1, l1);
        const int64_t off = plane_off(row, col0 + dc, out.ld, out.rows);
        *reinterpret_cast<uint2*>(out.hi + off) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(out.lo + off) = make_uint2(l0, l1);
      } else {
        *reinterpret_cast<float4*>(out.o + row * out.ld + col0 + dc) = make_float4(v0, v1, v2, v3);
      }
    }
  }
}

// (Round 5: an LDS-staged variant of the plane-emitting tail above — the wave's 32 x D block of each plane through a wave-private slice of the
// idle K / V LDS, leaving as 16-byte chunks, ~80 instead of 320 cache-line accesses per wave — produced the same values and the same step time:
// 3.905 vs 3.906 images/s interleaved on one box, profiles/r05_first_call.log.  Deleted.)

template <int D, int DKP, int DVP, bool F16P>     // head dim, padded to 16 (QK^T reduction) and to 32 (rows of O^T)
__global__ void __launch_bounds__(256) attn_fwd_bf16_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                            const float* __restrict__ v, int ldv, const AttnOut out,
                                                            float* __restrict__ lse, int heads, int Nq, int Nk, float scale_log2e) {
  constexpr int KT = 64;                  // keys per tile
  constexpr int LDK = DKP + 8;            // bf16 per K row: (DKP+8)*2 bytes = odd multiple of 16 B -> conflict-free b128 rows
  constexpr int LDVT = KT + 4;            // f16 per V^T row: 136 B -> 32 rows hit distinct even banks for ds_read_b64
  constexpr int NKS = DKP / 16;           // k-steps of S^T
  constexpr int NDT = DVP / 32;           // 32-row tiles of O^T
  constexpr bool ONES = F16P && DVP > D;  // V^T row D = 1: the softmax denominator is row D of O^T
  constexpr int K_BYTES = KT * LDK * 2, VT_BYTES = DVP * LDVT * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * K_BYTES + 2 * VT_BYTES];
  char* Khi = smem; char* Klo = smem + K_BYTES;
  char* Vhi = smem + 2 * K_BYTES; char* Vlo = Vhi + VT_BYTES;

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const int qrow = min(q0 + li, Nq - 1);

  // Q^T fragments (B operand): lane holds Q[q = li][dk = 16s + 8h .. +8], scaled, split
  bf16x8 qh[NKS], ql[NKS];
  {
    const float* qp = q + ((int64_t)b * Nq + qrow) * ldq + hd * D;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int dk = 16 * s + 8 * h + 2 * e;
        const float a = dk < D ? qp[dk] * scale_log2e : 0.f;
        const float c = dk + 1 < D ? qp[dk + 1] * scale_log2e : 0.f;
        split2(a, c, hi[e], lo[e]);
      }
      qh[s] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
      ql[s] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
  }
  // zero the padding of both LDS images once (K columns D..DKP, V^T rows D..DVP are never written again), then the ones row of V^T hi
  for (int i = t; i < (2 * K_BYTES + 2 * VT_BYTES) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
  if constexpr (ONES) {
    __syncthreads();
    if (t < KT) reinterpret_cast<uint16_t*>(Vhi)[D * LDVT + t] = (uint16_t)ATTN_F16_ONE;
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int n = 0; n < NDT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float* kb = k + (int64_t)b * Nk * ldk + hd * D;
  const float* vb = v + (int64_t)b * Nk * ldv + hd * D;

  for (int kt0 = 0; kt0 < Nk; kt0 += KT) {
    __syncthreads();
    // stage K (row-major, bf16 hi / lo) and V (transposed, f16 hi / lo)
    for (int i = t; i < KT * (D / 4); i += 256) {
      const int key = i / (D / 4), c4 = i - key * (D / 4);
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kt0 + key < Nk) {
        kv = *reinterpret_cast<const float4*>(kb + (int64_t)(kt0 + key) * ldk + c4 * 4);
        vv = *reinterpret_cast<const float4*>(vb + (int64_t)(kt0 + key) * ldv + c4 * 4);
      }
      uint32_t h0, l0, h1, l1;
      split2(kv.x, kv.y, h0, l0);
      split2(kv.z, kv.w, h1, l1);
      *reinterpret_cast<uint2*>(Khi + (key * LDK + c4 * 4) * 2) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(Klo + (key * LDK + c4 * 4) * 2) = make_uint2(l0, l1);
      if (F16P) { split2h(vv.x, vv.y, h0, l0); split2h(vv.z, vv.w, h1, l1); }
      else { split2(vv.x, vv.y, h0, l0); split2(vv.z, vv.w, h1, l1); }
      uint16_t* vh = reinterpret_cast<uint16_t*>(Vhi);
      uint16_t* vl = reinterpret_cast<uint16_t*>(Vlo);
      const int d0 = c4 * 4;
      vh[(d0 + 0) * LDVT + key] = (uint16_t)(h0 & 0xFFFFu); vh[(d0 + 1) * LDVT + key] = (uint16_t)(h0 >> 16);
      vh[(d0 + 2) * LDVT + key] = (uint16_t)(h1 & 0xFFFFu); vh[(d0 + 3) * LDVT + key] = (uint16_t)(h1 >> 16);
      vl[(d0 + 0) * LDVT + key] = (uint16_t)(l0 & 0xFFFFu); vl[(d0 + 1) * LDVT + key] = (uint16_t)(l0 >> 16);
      vl[(d0 + 2) * LDVT + key] = (uint16_t)(l1 & 0xFFFFu); vl[(d0 + 3) * LDVT + key] = (uint16_t)(l1 >> 16);
    }
    __syncthreads();

    // ---- S^T = K Q^T for the two 32-key sub-tiles
    f32x16 sacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = ((j * 32 + li) * LDK + 16 * s + 8 * h) * 2;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Khi + off);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Klo + off);
        sacc[j] = MFMA32(kl, qh[s], sacc[j]);
        sacc[j] = MFMA32(kh, ql[s], sacc[j]);
        sacc[j] = MFMA32(kh, qh[s], sacc[j]);
      }
    }
    // ---- online softmax: query = lane column; this lane holds keys 32j + (r&3) + 8(r>>2) + 4h
    if (kt0 + KT > Nk) {                   // only the last tile can hold padded keys (uniform branch)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h >= Nk) sacc[j][r] = -INFINITY;
    }
    // ---- online softmax + O^T += V^T P^T ; A fragment of step u: V^T[d = 32n + li][16u + 4h + {0..3, 8..11}]
    attn_tile_tail<F16P, ONES, NDT, LDVT>(sacc, m_run, l_run, oacc, Vhi, Vlo, li, h);
  }

  const float l_tot = attn_row_sum<D, NDT, ONES>(oacc, l_run, li);
  const float inv = 1.0f / l_tot;
  if (lse && h == 0 && q0 + li < Nq) lse[(int64_t)bh * Nq + q0 + li] = m_run + log2f(l_tot) - (F16P ? ATTN_P_SHIFT : 0.f);
  if (q0 + li < Nq) attn_store_o<D, NDT>(out, (int64_t)b * Nq + q0 + li, hd * D, oacc, inv, h);
}

// ------------------------------------------------------------------------------------------------
// Long key sequences (self-attention at 64x64 / 32x32 latents): every 128-query workgroup of the kernel above re-splits
// and re-transposes the same K / V tiles.  Here a pre-pass does that ONCE per (batch, head): it writes, per 64-key tile,
// the exact LDS image [K hi | K lo | V^T hi | V^T lo] (padded pitches, zero padding included) to a workspace, and the
// attention kernel streams images with 16-byte loads one tile ahead of the MFMAs (registers -> ds_write_b128): no
// conversion, no 2-byte transpose scatter and no exposed global-load latency in the key loop.
// ------------------------------------------------------------------------------------------------
template <int D, int DKP, int DVP>
struct AttnImg {
  static constexpr int KT = 64, LDK = DKP + 8, LDVT = KT + 4;
  static constexpr int K_BYTES = KT * LDK * 2, VT_BYTES = DVP * LDVT * 2;
  static constexpr int BYTES = 2 * K_BYTES + 2 * VT_BYTES;       // multiple of 16
  static constexpr int CHUNKS = BYTES / 16;
};

template <int D, int DKP, int DVP, bool F16P>
__global__ void __launch_bounds__(256) attn_pack_kv_kernel(const float* __restrict__ k, int ldk, const float* __restrict__ v, int ldv,
                                                           uint4* __restrict__ img, int heads, int Nk, int ntiles) {
  using I = AttnImg<D, DKP, DVP>;
  __shared__ __attribute__((aligned(16))) char smem[I::BYTES];
  char* Khi = smem; char* Klo = smem + I::K_BYTES;
  char* Vhi = smem + 2 * I::K_BYTES; char* Vlo = Vhi + I::VT_BYTES;
  const int t = threadIdx.x;
  const int tile = blockIdx.x, bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int kt0 = tile * I::KT;
  for (int i = t; i < I::CHUNKS; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  if (F16P && DVP > D && t < I::KT) reinterpret_cast<uint16_t*>(Vhi)[D * I::LDVT + t] = (uint16_t)ATTN_F16_ONE;      // the ones row (softmax denominator)
  const float* kb = k + (int64_t)b * Nk * ldk + hd * D;
  const float* vb = v + (int64_t)b * Nk * ldv + hd * D;
  for (int i = t; i < I::KT * (D / 4); i += 256) {
    const int key = i / (D / 4), c4 = i - key * (D / 4);
    if (kt0 + key >= Nk) continue;
    const float4 kv = *reinterpret_cast<const float4*>(kb + (int64_t)(kt0 + key) * ldk + c4 * 4);
    const float4 vv = *reinterpret_cast<const float4*>(vb + (int64_t)(kt0 + key) * ldv + c4 * 4);
    uint32_t h0, l0, h1, l1;
    split2(kv.x, kv.y, h0, l0);
    split2(kv.z, kv.w, h1, l1);
    *reinterpret_cast<uint2*>(Khi + (key * I::LDK + c4 * 4) * 2) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(Klo + (key * I::LDK + c4 * 4) * 2) = make_uint2(l0, l1);
    if (F16P) { split2h(vv.x, vv.y, h0, l0); split2h(vv.z, vv.w, h1, l1); }
    else { split2(vv.x, vv.y, h0, l0); split2(vv.z, vv.w, h1, l1); }
    uint16_t* vh = reinterpret_cast<uint16_t*>(Vhi);
    uint16_t* vl = reinterpret_cast<uint16_t*>(Vlo);
    const int d0 = c4 * 4;
    vh[(d0 + 0) * I::LDVT + key] = (uint16_t)(h0 & 0xFFFFu); vh[(d0 + 1) * I::LDVT + key] = (uint16_t)(h0 >> 16);
    vh[(d0 + 2) * I::LDVT + key] = (uint16_t)(h1 & 0xFFFFu); vh[(d0 + 3) * I::LDVT + key] = (uint16_t)(h1 >> 16);
    vl[(d0 + 0) * I::LDVT + key] = (uint16_t)(l0 & 0xFFFFu); vl[(d0 + 1) * I::LDVT + key] = (uint16_t)(l0 >> 16);
    vl[(d0 + 2) * I::LDVT + key] = (uint16_t)(l1 & 0xFFFFu); vl[(d0 + 3) * I::LDVT + key] = (uint16_t)(l1 >> 16);
  }
  __syncthreads();
  uint4* dst = img + ((int64_t)bh * ntiles + tile) * I::CHUNKS;
  for (int i = t; i < I::CHUNKS; i += 256) dst[i] = reinterpret_cast<const uint4*>(smem)[i];
}

template <int D, int DKP, int DVP, bool F16P>
__global__ void __launch_bounds__(256, (DVP <= 32 ? 4 : (DKP <= 48 ? 3 : 2))) attn_fwd_bf16_pk_kernel(const float* __restrict__ q, int ldq, const uint4* __restrict__ img,
                                                               const AttnOut out, float* __restrict__ lse, int heads,
                                                               int Nq, int Nk, int ntiles, float scale_log2e) {
  using I = AttnImg<D, DKP, DVP>;
  constexpr int KT = I::KT, LDK = I::LDK, LDVT = I::LDVT;
  constexpr int NKS = DKP / 16, NDT = DVP / 32;
  constexpr bool ONES = F16P && DVP > D;
  constexpr int NCH = (I::CHUNKS + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[I::BYTES];
  const char* Khi = smem; const char* Klo = smem + I::K_BYTES;
  const char* Vhi = smem + 2 * I::K_BYTES; const char* Vlo = Vhi + I::VT_BYTES;

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int q0 = blockIdx.x * 128 + wid * 32;
  const int qrow = min(q0 + li, Nq - 1);

  bf16x8 qh[NKS], ql[NKS];
  {
    const float* qp = q + ((int64_t)b * Nq + qrow) * ldq + hd * D;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int dk = 16 * s + 8 * h + 2 * e;
        const float a = dk < D ? qp[dk] * scale_log2e : 0.f;
        const float c = dk + 1 < D ? qp[dk + 1] * scale_log2e : 0.f;
        split2(a, c, hi[e], lo[e]);
      }
      qh[s] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
      ql[s] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
  }

  f32x16 oacc[NDT];
#pragma unroll
  for (int n = 0; n < NDT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[n][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const uint4* src = img + (int64_t)bh * ntiles * I::CHUNKS;
  uint4 pre[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = t + 256 * c;
    pre[c] = (NCH * 256 == I::CHUNKS || i < I::CHUNKS) ? src[i] : make_uint4(0u, 0u, 0u, 0u);
  }

  for (int tile = 0; tile < ntiles; ++tile) {
    const int kt0 = tile * KT;
    __syncthreads();                       // every wave is done reading the previous image
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = t + 256 * c;
      if (NCH * 256 == I::CHUNKS || i < I::CHUNKS) reinterpret_cast<uint4*>(smem)[i] = pre[c];
    }
    __syncthreads();
    {                                      // next image streams in behind the MFMAs of this one (last tile: re-reads itself)
      const uint4* nsrc = src + (int64_t)min(tile + 1, ntiles - 1) * I::CHUNKS;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int i = t + 256 * c;
        pre[c] = (NCH * 256 == I::CHUNKS || i < I::CHUNKS) ? nsrc[i] : make_uint4(0u, 0u, 0u, 0u);
      }
    }

    f32x16 sacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[j][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = ((j * 32 + li) * LDK + 16 * s + 8 * h) * 2;
        const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Khi + off);
        const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Klo + off);
        sacc[j] = MFMA32(kl, qh[s], sacc[j]);
        sacc[j] = MFMA32(kh, ql[s], sacc[j]);
        sacc[j] = MFMA32(kh, qh[s], sacc[j]);
      }
    }
    if (kt0 + KT > Nk) {                   // only the last tile can hold padded keys (uniform branch)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h >= Nk) sacc[j][r] = -INFINITY;
    }
    attn_tile_tail<F16P, ONES, NDT, LDVT>(sacc, m_run, l_run, oacc, Vhi, Vlo, li, h);
  }

  const float l_tot = attn_row_sum<D, NDT, ONES>(oacc, l_run, li);
  const float inv = 1.0f / l_tot;
  if (lse && h == 0 && q0 + li < Nq) lse[(int64_t)bh * Nq + q0 + li] = m_run + log2f(l_tot) - (F16P ? ATTN_P_SHIFT : 0.f);
  if (q0 + li < Nq) attn_store_o<D, NDT>(out, (int64_t)b * Nq + q0 + li, hd * D, oacc, inv, h);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant of the packed-image kernel (images whose K part and V part are whole KiB: d = 40, 64).
// The image of a key tile is already the exact LDS layout, lane-linear — so instead of 16-byte loads into registers one tile ahead
// and a ds_write_b128 pass behind two barriers, the K part and the V part go global -> LDS directly (buffer_load_dwordx4 ... lds,
// 1 KiB per wave instruction): no staging registers (32 VGPRs), no LDS write instructions.  Three LDS regions keep three workgroups
// per CU: K double-buffered (the next tile's K streams in during the whole current tile), V single (requested right after the
// barrier that retires the previous tile's P V reads, needed only after this tile's score + softmax phase).
//   per tile t:  wait all, BARRIER 1 (K(t) visible to everybody; everybody is done with V(t-1))
//                request V(t) -> V region, K(t+1) -> the other K region
//                S^T = K Q^T, softmax                       (K(t) region)
//                wait all, BARRIER 2 (V(t) visible)
//                O^T += V^T P^T                             (V region)
// Same arithmetic in the same order as attn_fwd_bf16_pk_kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4_a __attribute__((ext_vector_type(4)));

// QB: 32-query blocks per wave.  QB = 2: a workgroup covers 256 queries, every V^T fragment read from LDS feeds the MFMAs of two
// blocks and half as many workgroups stream the images; the score + softmax phases run one block after the other (their K fragments
// are re-read: both blocks' score accumulators at once do not fit two waves per SIMD).  Per-query arithmetic and order unchanged.
template <int D, int DKP, int DVP, int QB, bool F16P>
__global__ void __launch_bounds__(256, (QB == 2 ? 2 : (DVP <= 32 ? 4 : (DKP <= 48 ? 3 : 2)))) attn_fwd_bf16_dma_kernel(const float* __restrict__ q, int ldq, const uint4* __restrict__ img,
                                                               const AttnOut out, float* __restrict__ lse, int heads,
                                                               int Nq, int Nk, int ntiles, float scale_log2e) {
  using I = AttnImg<D, DKP, DVP>;
  constexpr int KT = I::KT, LDK = I::LDK, LDVT = I::LDVT;
  constexpr int NKS = DKP / 16, NDT = DVP / 32;
  constexpr bool ONES = F16P && DVP > D;
  constexpr int KP = 2 * I::K_BYTES, VP = 2 * I::VT_BYTES;          // K part (hi | lo) and V^T part (hi | lo) of an image
  static_assert(KP % 1024 == 0 && VP % 1024 == 0, "LDS-DMA moves whole KiB pieces");
  constexpr int NPK = KP / 1024, NPV = VP / 1024;
  __shared__ __attribute__((aligned(1024))) char smem[2 * KP + VP];  // [K region 0 | K region 1 | V region]

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(wid);
  const int li = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int q0 = blockIdx.x * (128 * QB) + wid * (32 * QB);          // this wave's queries: q0 + 32 * qb + li

  bf16x8 qh[QB][NKS], ql[QB][NKS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qrow = min(q0 + 32 * qb + li, Nq - 1);
    const float* qp = q + ((int64_t)b * Nq + qrow) * ldq + hd * D;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int dk = 16 * s + 8 * h + 2 * e;
        const float a = dk < D ? qp[dk] * scale_log2e : 0.f;
        const float c = dk + 1 < D ? qp[dk + 1] * scale_log2e : 0.f;
        split2(a, c, hi[e], lo[e]);
      }
      qh[qb][s] = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
      ql[qb][s] = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
  }

  f32x16 oacc[QB][NDT];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY; l_run[qb] = 0.f;
#pragma unroll
    for (int n = 0; n < NDT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][n][r] = 0.f;
  }

  // this (batch, head)'s images as a buffer resource; piece p of a part = 1 KiB = lane-linear 16 bytes per lane
  const uint64_t base = reinterpret_cast<uint64_t>(img + (int64_t)bh * ntiles * I::CHUNKS);
  const u32x4_a rs = {(uint32_t)base, (uint32_t)(base >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto fill = [&](uint32_t lds_region, uint32_t img_off, int npieces, int tile) {      // wave wv moves pieces wv, wv + 4, ...
    const uint32_t so = (uint32_t)tile * (uint32_t)I::BYTES + img_off;
    for (int p = wv; p < npieces; p += 4)
      asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                   :: "s"(lds0 + lds_region + (uint32_t)p * 1024u), "v"(lane16 + (uint32_t)p * 1024u), "s"(rs), "s"(so) : "memory");
  };

  fill(0, 0, NPK, 0);                      // K(0) -> K region 0
  for (int tile = 0; tile < ntiles; ++tile) {
    const int kt0 = tile * KT;
    const uint32_t kreg = (tile & 1) ? KP : 0;
    const char* Khi = smem + kreg; const char* Klo = Khi + I::K_BYTES;
    const char* Vhi = smem + 2 * KP; const char* Vlo = Vhi + I::VT_BYTES;
    // K(tile) has landed (requested a whole tile ago) and my P V reads of the previous tile have returned
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    fill(2 * KP, KP, NPV, tile);                                   // V(tile) -> V region (nobody reads V(tile - 1) any more)
    if (tile + 1 < ntiles) fill((tile & 1) ? 0 : KP, 0, NPK, tile + 1);   // K(tile + 1) -> the other K region

    f16x8 ph[QB][4];                       // F16P: the tile's probabilities, one f16 term
    bf16x8 xh[QB][4], xl[QB][4];           // bf16x3: bf16 hi / lo
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      f32x16 sacc[2];
  #pragma unroll
      for (int j = 0; j < 2; ++j)
  #pragma unroll
        for (int r = 0; r < 16; ++r) sacc[j][r] = 0.f;
  #pragma unroll
      for (int s = 0; s < NKS; ++s) {
  #pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int off = ((j * 32 + li) * LDK + 16 * s + 8 * h) * 2;
          const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Khi + off);
          const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Klo + off);
          sacc[j] = MFMA32(kl, qh[qb][s], sacc[j]);
          sacc[j] = MFMA32(kh, ql[qb][s], sacc[j]);
          sacc[j] = MFMA32(kh, qh[qb][s], sacc[j]);
        }
      }
      if (kt0 + KT > Nk) {                   // only the last tile can hold padded keys (uniform branch)
  #pragma unroll
        for (int j = 0; j < 2; ++j)
  #pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kt0 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * h >= Nk) sacc[j][r] = -INFINITY;
      }
      float alpha;
      bool grew;
      if constexpr (F16P) attn_softmax_tile<ONES>(sacc, m_run[qb], l_run[qb], ph[qb], alpha, grew);
      else attn_softmax_tile_x3(sacc, m_run[qb], l_run[qb], xh[qb], xl[qb], alpha, grew);
      if (__any(grew)) {
  #pragma unroll
        for (int n = 0; n < NDT; ++n)
  #pragma unroll
          for (int r = 0; r < 16; ++r) oacc[qb][n][r] *= alpha;
      }
    }
    // V(tile) (and K(tile + 1)) have landed; my K fragment reads have returned
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (F16P) attn_pv_tile<NDT, LDVT, QB>(Vhi, Vlo, li, h, ph, oacc);      // one V^T fragment feeds QB query blocks
    else attn_pv_tile_x3<NDT, LDVT, QB>(Vhi, Vlo, li, h, xh, xl, oacc);
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const int qi = q0 + 32 * qb + li;
    const float l_tot = attn_row_sum<D, NDT, ONES>(oacc[qb], l_run[qb], li);
    const float inv = 1.0f / l_tot;
    if (lse && h == 0 && qi < Nq) lse[(int64_t)bh * Nq + qi] = m_run[qb] + log2f(l_tot) - (F16P ? ATTN_P_SHIFT : 0.f);
    if (qi < Nq) attn_store_o<D, NDT>(out, (int64_t)b * Nq + qi, hd * D, oacc[qb], inv, h);
  }
}

#define ATTN_PK_MIN_NK 256      /* shorter key sequences (cross-attention over 77 tokens) stay on the self-staging kernel */

template <int D, int DKP, int DVP, bool F16P>
static int launch_pack_kv(const float* k, int ldk, const float* v, int ldv, uint4* img, int B, int heads, int Nk, hipStream_t st) {
  using I = AttnImg<D, DKP, DVP>;
  const int ntiles = (Nk + I::KT - 1) / I::KT;
  hipLaunchKernelGGL((attn_pack_kv_kernel<D, DKP, DVP, F16P>), dim3(ntiles, B * heads), dim3(256), 0, st, k, ldk, v, ldv, img, heads, Nk, ntiles);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// attention from pre-packed K / V^T images (one image per 64-key tile and (batch, head))
template <int D, int DKP, int DVP, bool F16P>
static int launch_attn_images(const float* q, int ldq, const uint4* img, const AttnOut& out, float* lse, int B, int heads, int Nq, int Nk,
                              float scale, hipStream_t st) {
  using I = AttnImg<D, DKP, DVP>;
  dim3 grid((Nq + 127) / 128, B * heads);
  const int ntiles = (Nk + I::KT - 1) / I::KT;
  // LDS-DMA image streaming wherever the image's K and V^T parts are whole KiB (d = 40, 64): 1.50 -> 1.42 ms on 4096^2, d = 40, batch 16
  // (profiles/r02_probe_attn_dma.log; 1.29 ms with the f16p second product, profiles/r04_probe_attn.log); the register-staged kernel below
  // takes the other head sizes (d = 80) and >= 2 GiB image sets.  (Two query blocks per wave on top measured 1.40 ms at 256 VGPRs with spills — not kept.)
  if constexpr ((2 * I::K_BYTES) % 1024 == 0 && (2 * I::VT_BYTES) % 1024 == 0) {
    if ((int64_t)ntiles * I::BYTES < 0x7FFFFFFF) {
      // (measured and not kept, round 4: two query blocks per wave, s_setprio around the MFMA phases, and a software-pipelined one-barrier loop
      // with K and V^T double-buffered — 1.23 / 1.23 / 1.31 ms against 1.22: profiles/r04_probe_attn_qb2_prio.log)
      hipLaunchKernelGGL((attn_fwd_bf16_dma_kernel<D, DKP, DVP, 1, F16P>), grid, dim3(256), 0, st, q, ldq, img, out, lse, heads, Nq, Nk, ntiles,
                         scale * 1.4426950408889634f);
      DDPO_LAUNCH_CHECK();
      return DDPO_OK;
    }
  }
  hipLaunchKernelGGL((attn_fwd_bf16_pk_kernel<D, DKP, DVP, F16P>), grid, dim3(256), 0, st, q, ldq, img, out, lse, heads, Nq, Nk, ntiles,
                     scale * 1.4426950408889634f);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

template <int D, int DKP, int DVP, bool F16P>
static int launch_attn_bf16(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const AttnOut& out, float* lse,
                            int B, int heads, int Nq, int Nk, float scale, void* ws, size_t ws_bytes, hipStream_t st) {
  using I = AttnImg<D, DKP, DVP>;
  dim3 grid((Nq + 127) / 128, B * heads);
  const int ntiles = (Nk + I::KT - 1) / I::KT;
  const size_t need = (size_t)B * heads * ntiles * I::BYTES;
  if (Nk >= ATTN_PK_MIN_NK && ws && ws_bytes >= need && !(reinterpret_cast<uintptr_t>(ws) & 15)) {
    uint4* img = reinterpret_cast<uint4*>(ws);
    const int rc = launch_pack_kv<D, DKP, DVP, F16P>(k, ldk, v, ldv, img, B, heads, Nk, st);
    if (rc != DDPO_OK) return rc;
    return launch_attn_images<D, DKP, DVP, F16P>(q, ldq, img, out, lse, B, heads, Nq, Nk, scale, st);
  }
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<D, DKP, DVP, F16P>), grid, dim3(256), 0, st, q, ldq, k, ldk, v, ldv, out, lse, heads, Nq, Nk,
                     scale * 1.4426950408889634f);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

template <int D, int DKP, int DVP>
static size_t attn_img_bytes(int B, int heads, int Nk) {
  using I = AttnImg<D, DKP, DVP>;
  return (size_t)B * heads * ((Nk + I::KT - 1) / I::KT) * I::BYTES;
}
template <int D, int DKP, int DVP>
static size_t attn_ws(int B, int heads, int Nk) {
  if (Nk < ATTN_PK_MIN_NK) return 0;
  return attn_img_bytes<D, DKP, DVP>(B, heads, Nk);
}

extern "C" size_t ddpo_attention_fwd_bf16x3_ws_bytes(int B, int heads, int Nk, int d) {
  if (B <= 0 || heads <= 0 || Nk <= 0) return 0;
  switch (d) {
    case 8:  return attn_ws<8, 16, 32>(B, heads, Nk);
    case 16: return attn_ws<16, 16, 32>(B, heads, Nk);
    case 40: return attn_ws<40, 48, 64>(B, heads, Nk);
    case 64: return attn_ws<64, 64, 64>(B, heads, Nk);
    case 80: return attn_ws<80, 80, 96>(B, heads, Nk);
    default: return 0;
  }
}

#define ATTN_BY_D(CALL)                                      \
  switch (d) {                                               \
    case 8:  return CALL(8, 16, 32);                         \
    case 16: return CALL(16, 16, 32);                        \
    case 40: return CALL(40, 48, 64);                        \
    case 64: return CALL(64, 64, 64);                        \
    case 80: return CALL(80, 80, 96);                        \
    default: return DDPO_EINVAL; /* other head dims stay on the exact-fp32 kernel */ \
  }

// output descriptor checks shared by the entry points: fp32 rows 16-byte aligned; planes 8-byte aligned, k-blocked (ld == 0) only with whole
// 32-channel blocks
static bool attn_out_ok(const AttnOut& out, int heads, int d) {
  if (out.hi) {
    if (!out.lo || out.o || (out.ld & 3) || ((reinterpret_cast<uintptr_t>(out.hi) | reinterpret_cast<uintptr_t>(out.lo)) & 7)) return false;
    return out.ld != 0 || ((heads * d) & 31) == 0;
  }
  return out.o && !(out.ld & 3) && !(reinterpret_cast<uintptr_t>(out.o) & 15);
}

template <bool F16P>
static int attention_fwd_impl(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const AttnOut& out,
                              float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* ws, size_t ws_bytes, void* stream) {
  if (!q || !k || !v || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || !attn_out_ok(out, heads, d)) return DDPO_EINVAL;
  if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (long)B * heads > 65535) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
#define CALL(DD, DK, DV) launch_attn_bf16<DD, DK, DV, F16P>(q, ldq, k, ldk, v, ldv, out, lse, B, heads, Nq, Nk, scale, ws, ws_bytes, st)
  ATTN_BY_D(CALL)
#undef CALL
}
extern "C" int ddpo_attention_fwd_bf16x3(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                         float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* ws, size_t ws_bytes,
                                         void* stream) {
  return attention_fwd_impl<false>(q, ldq, k, ldk, v, ldv, AttnOut{o, nullptr, nullptr, ldo, (int64_t)B * Nq}, lse, B, heads, Nq, Nk, d, scale, ws,
                                   ws_bytes, stream);
}
extern "C" int ddpo_attention_fwd_f16p(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                       float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* ws, size_t ws_bytes,
                                       void* stream) {
  return attention_fwd_impl<true>(q, ldq, k, ldk, v, ldv, AttnOut{o, nullptr, nullptr, ldo, (int64_t)B * Nq}, lse, B, heads, Nq, Nk, d, scale, ws,
                                  ws_bytes, stream);
}
/* Plane-emitting forms (ABI v12): the output leaves as bf16 hi / lo planes (o_hi / o_lo, row stride ld_planes elements; 0 = k-blocked
 * (heads * d / 32, B * Nq, 32)) of exactly the fp32 values the functions above write — no fp32 tensor is written. */
extern "C" int ddpo_attention_fwd_bf16x3_po(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* o_hi,
                                            uint16_t* o_lo, int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale,
                                            void* ws, size_t ws_bytes, void* stream) {
  if (!o_hi) return DDPO_EINVAL;
  return attention_fwd_impl<false>(q, ldq, k, ldk, v, ldv, AttnOut{nullptr, o_hi, o_lo, ld_planes, (int64_t)B * Nq}, lse, B, heads, Nq, Nk, d, scale,
                                   ws, ws_bytes, stream);
}
extern "C" int ddpo_attention_fwd_f16p_po(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* o_hi,
                                          uint16_t* o_lo, int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale,
                                          void* ws, size_t ws_bytes, void* stream) {
  if (!o_hi) return DDPO_EINVAL;
  return attention_fwd_impl<true>(q, ldq, k, ldk, v, ldv, AttnOut{nullptr, o_hi, o_lo, ld_planes, (int64_t)B * Nq}, lse, B, heads, Nq, Nk, d, scale,
                                  ws, ws_bytes, stream);
}

/* K / V of a (batch, head) set packed ONCE into the per-64-key-tile LDS images the attention kernels stream (any Nk), for callers whose
 * keys / values are constant over many attention calls — the text context of the cross-attention layers over the 50 DDIM steps of a
 * sampling call.  ddpo_attention_kv_images_bytes gives the image size (the same for both variants); ddpo_attention_fwd_{bf16x3,f16p}_images runs
 * the attention from the images of ITS variant's pack function (same kernels as the workspace form: identical results). */
extern "C" size_t ddpo_attention_kv_images_bytes(int B, int heads, int Nk, int d) {
  if (B <= 0 || heads <= 0 || Nk <= 0) return 0;
  switch (d) {
    case 8:  return attn_img_bytes<8, 16, 32>(B, heads, Nk);
    case 16: return attn_img_bytes<16, 16, 32>(B, heads, Nk);
    case 40: return attn_img_bytes<40, 48, 64>(B, heads, Nk);
    case 64: return attn_img_bytes<64, 64, 64>(B, heads, Nk);
    case 80: return attn_img_bytes<80, 80, 96>(B, heads, Nk);
    default: return 0;
  }
}

template <bool F16P>
static int pack_kv_impl(const float* k, int ldk, const float* v, int ldv, void* images, size_t images_bytes, int B, int heads, int Nk, int d,
                        void* stream) {
  if (!k || !v || !images || B <= 0 || heads <= 0 || Nk <= 0 || (ldk & 3) || (ldv & 3) || (long)B * heads > 65535) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(images)) & 15) return DDPO_EINVAL;
  const size_t need = ddpo_attention_kv_images_bytes(B, heads, Nk, d);
  if (need == 0 || images_bytes < need) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  uint4* img = reinterpret_cast<uint4*>(images);
#define CALL(DD, DK, DV) launch_pack_kv<DD, DK, DV, F16P>(k, ldk, v, ldv, img, B, heads, Nk, st)
  ATTN_BY_D(CALL)
#undef CALL
}
extern "C" int ddpo_attention_pack_kv_bf16x3(const float* k, int ldk, const float* v, int ldv, void* images, size_t images_bytes, int B, int heads,
                                             int Nk, int d, void* stream) {
  return pack_kv_impl<false>(k, ldk, v, ldv, images, images_bytes, B, heads, Nk, d, stream);
}
extern "C" int ddpo_attention_pack_kv_f16p(const float* k, int ldk, const float* v, int ldv, void* images, size_t images_bytes, int B, int heads,
                                           int Nk, int d, void* stream) {
  return pack_kv_impl<true>(k, ldk, v, ldv, images, images_bytes, B, heads, Nk, d, stream);
}

template <bool F16P>
static int attention_images_impl(const float* q, int ldq, const void* images, size_t images_bytes, const AttnOut& out, float* lse,
                                 int B, int heads, int Nq, int Nk, int d, float scale, void* stream) {
  if (!q || !images || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || d <= 0 || !attn_out_ok(out, heads, d)) return DDPO_EINVAL;
  if ((ldq & 3) || (long)B * heads > 65535) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(images)) & 15) return DDPO_EINVAL;
  const size_t need = ddpo_attention_kv_images_bytes(B, heads, Nk, d);
  if (need == 0 || images_bytes < need) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  const uint4* img = reinterpret_cast<const uint4*>(images);
#define CALL(DD, DK, DV) launch_attn_images<DD, DK, DV, F16P>(q, ldq, img, out, lse, B, heads, Nq, Nk, scale, st)
  ATTN_BY_D(CALL)
#undef CALL
}
extern "C" int ddpo_attention_fwd_bf16x3_images(const float* q, int ldq, const void* images, size_t images_bytes, float* o, int ldo, float* lse,
                                                int B, int heads, int Nq, int Nk, int d, float scale, void* stream) {
  return attention_images_impl<false>(q, ldq, images, images_bytes, AttnOut{o, nullptr, nullptr, ldo, (int64_t)B * Nq}, lse, B, heads, Nq, Nk, d,
                                      scale, stream);
}
extern "C" int ddpo_attention_fwd_f16p_images(const float* q, int ldq, const void* images, size_t images_bytes, float* o, int ldo, float* lse,
                                              int B, int heads, int Nq, int Nk, int d, float scale, void* stream) {
  return attention_images_impl<true>(q, ldq, images, images_bytes, AttnOut{o, nullptr, nullptr, ldo, (int64_t)B * Nq}, lse, B, heads, Nq, Nk, d,
                                     scale, stream);
}
extern "C" int ddpo_attention_fwd_bf16x3_images_po(const float* q, int ldq, const void* images, size_t images_bytes, uint16_t* o_hi, uint16_t* o_lo,
                                                   int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* stream) {
  if (!o_hi) return DDPO_EINVAL;
  return attention_images_impl<false>(q, ldq, images, images_bytes, AttnOut{nullptr, o_hi, o_lo, ld_planes, (int64_t)B * Nq}, lse, B, heads, Nq, Nk,
                                      d, scale, stream);
}
extern "C" int ddpo_attention_fwd_f16p_images_po(const float* q, int ldq, const void* images, size_t images_bytes, uint16_t* o_hi, uint16_t* o_lo,
                                                 int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* stream) {
  if (!o_hi) return DDPO_EINVAL;
  return attention_images_impl<true>(q, ldq, images, images_bytes, AttnOut{nullptr, o_hi, o_lo, ld_planes, (int64_t)B * Nq}, lse, B, heads, Nq, Nk,
                                     d, scale, stream);
}
