// Flash-style fused attention forward on the exact-fp32 MFMA datapath (v_mfma_f32_16x16x4_f32).
// Replaces the materialised einsum/softmax/einsum of diffusers' FlaxAttention (N x N scores: 512 MiB per
// sample at 64x64 latents) with an online-softmax sweep over LDS-staged K/V tiles.
//
// Work split: one workgroup = 4 waves = 64 queries of one (batch, head); each wave owns 16 queries.
// Both products are computed "transposed" so that a query is always a lane column and no register
// transposition is needed between them:
//   S^T (keys x queries) = K Q^T : A = K tile from LDS, B = Q^T held in registers for the whole sweep
//   O^T (d x queries)    = V^T P^T: A = V tile from LDS, B = P^T = exp2(S^T - m) straight from the S^T registers
// (the 16x16 C layout "row = 4*(lane>>4)+reg, col = lane&15" is exactly the B-operand layout "k = lane>>4,
//  col = lane&15" once the 4 MFMA k-slots of step r are assigned to keys 4*(lane>>4)+r).
// Row statistics (max / sum) are per lane column; the 4 lane groups are merged with two wave shuffles.
// LDS strides: K rows D+2 floats (2*odd -> the 32 (key, dk) pairs of a half-wave hit 32 banks),
//              V rows DPV+4 floats (== 4 mod 8 -> (key+4, col) pairs of a half-wave hit 32 banks).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int D, int DPV, int KT>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                       const float* __restrict__ v, int ldv, float* __restrict__ o, int ldo,
                                                       float* __restrict__ lse, int heads, int Nq, int Nk, float scale_log2e) {
  constexpr int LDK = D + 2;
  constexpr int LDV = DPV + 4;
  constexpr int NS = D / 4;         // k-steps of the S^T product
  constexpr int NJ = KT / 16;       // key sub-tiles
  constexpr int NN = DPV / 16;      // d sub-tiles of O^T
  __shared__ __attribute__((aligned(16))) float Ks[KT * LDK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * LDV];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int qi = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int q0 = blockIdx.x * 64 + wid * 16;
  const int qrow = min(q0 + qi, Nq - 1);

  float qr[NS];
  {
    const float* qp = q + ((int64_t)b * Nq + qrow) * ldq + h * D + g;
#pragma unroll
    for (int s = 0; s < NS; ++s) qr[s] = qp[4 * s] * scale_log2e;
  }
  if (DPV > D) {   // zero the pad columns of V once
    for (int i = t; i < KT * (DPV - D); i += 256) {
      const int key = i / (DPV - D), c = i - key * (DPV - D);
      Vs[key * LDV + D + c] = 0.f;
    }
  }

  f32x4 oacc[NN];
#pragma unroll
  for (int n = 0; n < NN; ++n) oacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const float* kb = k + (int64_t)b * Nk * ldk + h * D;
  const float* vb = v + (int64_t)b * Nk * ldv + h * D;

  for (int kt0 = 0; kt0 < Nk; kt0 += KT) {
    __syncthreads();
    for (int i = t; i < KT * (D / 4); i += 256) {
      const int key = i / (D / 4), c4 = i - key * (D / 4);
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kt0 + key < Nk) {
        kv = *reinterpret_cast<const float4*>(kb + (int64_t)(kt0 + key) * ldk + c4 * 4);
        vv = *reinterpret_cast<const float4*>(vb + (int64_t)(kt0 + key) * ldv + c4 * 4);
      }
      float2* kd = reinterpret_cast<float2*>(&Ks[key * LDK + c4 * 4]);
      kd[0] = make_float2(kv.x, kv.y);
      kd[1] = make_float2(kv.z, kv.w);
      *reinterpret_cast<float4*>(&Vs[key * LDV + c4 * 4]) = vv;
    }
    __syncthreads();

    // ---- S^T = K Q^T
    f32x4 sacc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float a = Ks[(j * 16 + qi) * LDK + 4 * s + g];
        sacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, qr[s], sacc[j], 0, 0, 0);
      }
    }
    // ---- online softmax (query = lane column; keys = 4 lane groups x 4 regs x NJ tiles)
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt0 + j * 16 + g * 4 + r >= Nk) sacc[j][r] = -INFINITY;
        mx = fmaxf(mx, sacc[j][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(sacc[j][r] - m_new);
        sacc[j][r] = p;
        ls += p;
      }
    }
    l_run = l_run * alpha + ls;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      oacc[n][0] *= alpha; oacc[n][1] *= alpha; oacc[n][2] *= alpha; oacc[n][3] *= alpha;
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vrow = &Vs[(j * 16 + g * 4 + r) * LDV + qi];
        const float p = sacc[j][r];
#pragma unroll
        for (int n = 0; n < NN; ++n) oacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[n * 16], p, oacc[n], 0, 0, 0);
      }
    }
  }

  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  const float inv = 1.0f / l_tot;
  if (lse && g == 0 && q0 + qi < Nq) lse[(int64_t)bh * Nq + q0 + qi] = m_run + log2f(l_tot);   // log2-domain logsumexp
  if (q0 + qi < Nq) {
    float* op = o + ((int64_t)b * Nq + q0 + qi) * ldo + h * D;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      const int dcol = n * 16 + g * 4;
      if (dcol < D) *reinterpret_cast<float4*>(op + dcol) = make_float4(oacc[n][0] * inv, oacc[n][1] * inv, oacc[n][2] * inv, oacc[n][3] * inv);
    }
  }
}

template <int D, int DPV, int KT>
static int launch_attn(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo, float* lse,
                       int B, int heads, int Nq, int Nk, float scale, hipStream_t st) {
  dim3 grid((Nq + 63) / 64, B * heads);
  hipLaunchKernelGGL((attn_fwd_kernel<D, DPV, KT>), grid, dim3(256), 0, st, q, ldq, k, ldk, v, ldv, o, ldo, lse, heads, Nq, Nk,
                     scale * 1.4426950408889634f);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, float* o, int ldo,
                                  float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* stream) {
  if (!q || !k || !v || !o || B <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0) return DDPO_EINVAL;
  if ((ldq & 3) || (ldk & 3) || (ldv & 3) || (ldo & 3)) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(o)) & 15) return DDPO_EINVAL;
  if ((long)B * heads > 65535) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  switch (d) {
    case 4:   return launch_attn<4, 16, 64>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nk, scale, st);
    case 8:   return launch_attn<8, 16, 64>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nk, scale, st);
    case 16:  return launch_attn<16, 16, 64>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nk, scale, st);
    case 40:  return launch_attn<40, 48, 64>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nk, scale, st);
    case 64:  return launch_attn<64, 64, 64>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nk, scale, st);
    case 80:  return launch_attn<80, 80, 64>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nk, scale, st);
    case 160: return launch_attn<160, 160, 32>(q, ldq, k, ldk, v, ldv, o, ldo, lse, B, heads, Nq, Nk, scale, st);
    default:  return DDPO_EINVAL;
  }
}
