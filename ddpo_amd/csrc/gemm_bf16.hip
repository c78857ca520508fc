// Implicit-GEMM convolution / dense GEMM on the bf16 MFMA datapath of gfx950 (v_mfma_f32_32x32x16_bf16, ~2.5 PFLOP/s
// dense) with fp32 operands emulated by a bf16 split:   x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi)
//   NPASS = 3:  a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi   (XLA's "bf16_3x" / HIGH precision, ~2^-16 relative per product)
//   NPASS = 1:  a*b ~= a_hi*b_hi                            (XLA's TPU DEFAULT precision, what the reference ran with)
// accumulated in fp32.  Same contract as gemm_conv_kernel (gemm.hip):
//   out[m][n] = alpha * sum_k A(m,k) W(k,n) + bias[n] + rowbias[m / rows_per_batch][n] + residual[m][n]
// Activations stay fp32 in HBM and are split while they are staged into LDS (v_cvt_pk_bf16_f32); weights are
// pre-split once per optimizer update into bf16 hi/lo planes, k-contiguous per output column ([N][Kp] for the forward
// pass, the original [K][N] order for data gradients), so a B fragment is one 16-byte load.
// LDS tiles are [row][32 k] bf16 (64 B per row) with the 16-byte chunk index XOR-swizzled by (row>>2)&3: every
// ds_read_b128 / ds_write of a 16-lane group touches 16 distinct 16-byte slots (conflict-free).
#include "common.h"
#include <cstdlib>
#include <type_traits>

// Phase timing of the k-loop kernels (tools/native/kernel_probe_timing only: the macro is never defined for libddpo_hip.so).
// Thread 0 of every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) at: entry, first barrier of the k-loop,
// end of the k-loop, end of the output stage.
#ifdef DDPO_KLOOP_TIMING
__device__ unsigned long long ddpo_dbg_t[2 * 16384 * 8];
#define DBG_T(i)                                                                                      \
  do {                                                                                                \
    if (threadIdx.x == 0) {                                                                           \
      const int w_ = (blockIdx.x + gridDim.x * blockIdx.y) & 16383;                                   \
      ddpo_dbg_t[w_ * 8 + (i)] = __builtin_amdgcn_s_memtime();                                        \
      ddpo_dbg_t[w_ * 8 + 4 + (i)] = __builtin_amdgcn_s_memrealtime();                                \
    }                                                                                                 \
  } while (0)
// exposed wait of wave 0 at the k-tile boundaries: DBG_W0 before the s_waitcnt in front of the barrier, DBG_W1 behind the barrier,
// DBG_WSTORE once after the loop (slot 3 of the realtime half is overwritten: the probe reads slot 7 as "wait cycles")
#define DBG_ABL(bit) ((d.splits >> 4) & (bit))      /* timing build only: 1 = no LDS-DMA inside the loop, 2 = no MFMAs */
#define DBG_WDECL unsigned long long dbg_w0_ = 0, dbg_wacc_ = 0, dbg_bacc_ = 0
#define DBG_W0() do { dbg_w0_ = __builtin_amdgcn_s_memtime(); } while (0)
#define DBG_WMID() do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dbg_wacc_ += n_ - dbg_w0_; dbg_w0_ = n_; } while (0)
#define DBG_W1() do { dbg_bacc_ += __builtin_amdgcn_s_memtime() - dbg_w0_; } while (0)
#define DBG_WSTORE()                                                                 \
  do {                                                                               \
    if (threadIdx.x == 0) {                                                          \
      const int w_ = (blockIdx.x + gridDim.x * blockIdx.y) & 16383;                  \
      ddpo_dbg_t[(16384 + w_) * 8 + 0] = dbg_wacc_;                                  \
      ddpo_dbg_t[(16384 + w_) * 8 + 1] = dbg_bacc_;                                  \
    }                                                                                \
  } while (0)
extern "C" int ddpo_debug_kloop_times(unsigned long long* host, int n_wg) {
  (void)n_wg;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(ddpo_dbg_t), (size_t)2 * 16384 * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define DBG_T(i) do { } while (0)
#define DBG_ABL(bit) 0
#define DBG_WDECL do { } while (0)
#define DBG_W0() do { } while (0)
#define DBG_WMID() do { } while (0)
#define DBG_W1() do { } while (0)
#define DBG_WSTORE() do { } while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define BF_BK 32
#define BF_THREADS 256

// two floats -> packed bf16 hi pair and packed bf16 lo pair
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = cvt_pk_bf16(a, b);
  lo = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}

// the f16mx cross-term MFMA: A = e5m2 (cbsz 1), B = e4m3 (blgp 0); the weight scale is byte `opb` of `sb` (op_sel is an immediate: the
// switch folds away in unrolled callers)
__device__ __forceinline__ f32x16 mx_mfma(const i32x8 a, const i32x8 b, const f32x16 c, int sa, int opb, int sb) {
  switch (opb) {
    case 0: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 0, 0, sa, 0, sb);
    case 1: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 0, 0, sa, 1, sb);
    case 2: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 0, 0, sa, 2, sb);
    default: return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 0, 0, sa, 3, sb);
  }
}

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }   // bytes

__device__ __forceinline__ void st_out4(float* p, const float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}

// plane-emitting output stage: 4 consecutive output values -> 4 bf16 hi + 4 bf16 lo (the split the fp32-fed loader would apply)
__device__ __forceinline__ void store_planes4(const ddpo_gemm_desc& d, int64_t row, int col, const float4 v) {
  if (d.planes_fmt == 1) {                                            // f16mx planes (common.h)
    mx_store4(d.out_hi, d.out_lo, row, col, d.ld_planes, d.M, v);
    return;
  }
  uint2 h, l;
  split4(v, h, l);
  const int64_t o = plane_off(row, col, d.ld_planes, d.M);           // ld_planes == 0: k-blocked planes (ncols / 32, M, 32)
  *reinterpret_cast<uint2*>(d.out_hi + o) = h;
  *reinterpret_cast<uint2*>(d.out_lo + o) = l;
}

// Vector output stage of the buffer-addressed kernels: one wave moves NIT x 64 float4 of its sub-tile (rows of WTN columns, LPR = WTN / 4
// float4 per row) from its LDS slice `cw` to the output, 512 B .. 1 KiB contiguous per row.  Round 4: TWO PHASES.  The first form of this
// loop loaded bias / row bias / residual inside each of its 16-40 iterations, behind runtime flags — every iteration its own basic blocks with
// an s_waitcnt vmcnt(0) in front of the add, i.e. one exposed L2 / HBM round trip per float4, AND (vmcnt counts stores too, in order) a drain of
// the previous iteration's stores: the residual read of a 64x64-level projection ran at 2.8 TB/s and cost 30 us of a 92 us launch
// (profiles/r04_timeline_sampling_step_before_handover.txt).  Now
//   combine: the operands of NB iterations are requested back to back (clamped addresses: no per-lane branches; the bias float4s of a lane's
//            PER distinct column positions once per tile), the caller's LDS transposition runs under the first batch's latency, and
//            alpha * acc + bias (+ row bias) (+ residual) — same arithmetic, same order per element — is written BACK to the same LDS slot
//            (same lane reads and writes it: wave-private, in-order LDS access, no barrier).  No store is in flight in this phase, so its
//            waits only ever cover loads.  Skipped when there is nothing to combine (alpha == 1, no bias / row bias / residual: q, k, v).
//   emit:    LDS -> fp32 rows and / or planes; no loads, so no vmcnt wait: the stores of all iterations stream.
constexpr int epi_gcd(int a, int b) { return b == 0 ? a : epi_gcd(b, a % b); }
template <int NIT, int LPR, int WTN, int NBMAX = 10>
struct EpiRows {
  static constexpr int PER = LPR / epi_gcd(64, LPR);          // the column of iteration `it` depends on it % PER only
  // iterations per batch (NBMAX: the tall tile, whose second half of the accumulators is still live during its first pass, takes 5)
  static constexpr int NB = (NIT % 10 == 0 && NBMAX >= 10) ? 10 : (NIT >= 16 && NIT % 8 == 0 && NBMAX >= 8 ? 8 : (NIT % 5 == 0 && NBMAX >= 5 ? 5 : (NIT % 4 == 0 ? 4 : 1)));
  static constexpr int LB = NIT % 5 == 0 ? 5 : (NIT % 4 == 0 ? 4 : 1);     // LDS reads in flight in the emit phase
  float4 bv[PER];                                             // bias of this lane's PER column positions
  float4 ex[NB];                                              // the batch's residual (or, without a residual, row-bias) operands
  bool any;                                                   // there is something to combine
  // bias values, once per tile (zeros without a bias: alpha * acc + 0, as the scalar form does)
  __device__ __forceinline__ void init(const ddpo_gemm_desc& d, int col_base, int lane) {
    any = d.bias || d.rowbias || d.residual || d.alpha != 1.0f;
    if (!any) return;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int e = j * 64 + lane, rr = e / LPR;
      const int colc = min(col_base + (e - rr * LPR) * 4, d.N - 4);
      bv[j] = d.bias ? *reinterpret_cast<const float4*>(d.bias + colc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // request the operands of iterations b0 .. b0 + NB - 1
  // (`lane` is laundered through an empty asm in every phase: the index arithmetic of an iteration is otherwise recognised as common to all of
  // them and to every pass of the tall tile, computed for all NIT iterations up front and spilled — 1.4 KB of scratch per lane)
  __device__ __forceinline__ void fetch(const ddpo_gemm_desc& d, int b0, int row_base, int col_base, int lane) {
    if (!d.residual && !d.rowbias) return;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int e = (b0 + i) * 64 + lane, rr = e / LPR;
      const int rowc = min(row_base + rr, d.M - 1), colc = min(col_base + (e - rr * LPR) * 4, d.N - 4);
      ex[i] = d.residual ? *reinterpret_cast<const float4*>(d.residual + (int64_t)rowc * d.ld_res + colc)
                         : *reinterpret_cast<const float4*>(d.rowbias + (int64_t)(rowc / d.rows_per_batch) * d.ld_rowbias + colc);
    }
  }
  // alpha * acc + bias (+ row bias) (+ residual), iterations b0 .. b0 + NB - 1, back into the LDS slot
  __device__ __forceinline__ void combine(const ddpo_gemm_desc& d, float* cw, int b0, int row_base, int col_base, int lane) {
    const bool both = d.residual && d.rowbias;                // never in the U-Net (time-embedding bias: conv1; residual: conv2): loaded in place
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int it = b0 + i;
      const int e = it * 64 + lane, rr = e / LPR, lcol = (e - rr * LPR) * 4;        // float4 index inside the rows: consecutive lanes, consecutive 16 bytes
      float4 v = *reinterpret_cast<const float4*>(cw + rr * WTN + lcol);
      const float4 b4 = bv[it % PER];
      v.x = d.alpha * v.x + b4.x; v.y = d.alpha * v.y + b4.y; v.z = d.alpha * v.z + b4.z; v.w = d.alpha * v.w + b4.w;
      if (both) {
        const int rowc = min(row_base + rr, d.M - 1), colc = min(col_base + lcol, d.N - 4);
        const float4 rb = *reinterpret_cast<const float4*>(d.rowbias + (int64_t)(rowc / d.rows_per_batch) * d.ld_rowbias + colc);
        v.x += rb.x; v.y += rb.y; v.z += rb.z; v.w += rb.w;
      }
      if (d.residual || d.rowbias) { v.x += ex[i].x; v.y += ex[i].y; v.z += ex[i].z; v.w += ex[i].w; }
      *reinterpret_cast<float4*>(cw + rr * WTN + lcol) = v;
    }
  }
  // the whole combine phase of one set of staged rows (the first batch was requested by the caller in front of its LDS transposition)
  __device__ __forceinline__ void combine_all(const ddpo_gemm_desc& d, float* cw, int row_base, int col_base, int lane) {
    if (!any) return;
#pragma unroll
    for (int b0 = 0; b0 < NIT; b0 += NB) {
      if (b0) fetch(d, b0, row_base, col_base, lane);
      combine(d, cw, b0, row_base, col_base, lane);
    }
  }
  // LDS -> fp32 rows and / or planes
  __device__ __forceinline__ static void emit(const ddpo_gemm_desc& d, const float* cw, int row_base, int col_base, int lane) {
#pragma unroll
    for (int b0 = 0; b0 < NIT; b0 += LB) {
      asm volatile("" : "+v"(lane));
      float4 v[LB];
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int e = (b0 + i) * 64 + lane, rr = e / LPR;
        v[i] = *reinterpret_cast<const float4*>(cw + rr * WTN + (e - rr * LPR) * 4);
      }
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int e = (b0 + i) * 64 + lane, rr = e / LPR;
        const int row = row_base + rr, col = col_base + (e - rr * LPR) * 4;
        if (row >= d.M || col >= d.N) continue;
        if (d.out) st_out4(d.out + (int64_t)row * d.ld_out + col, v[i]);
        if (d.out_hi) store_planes4(d, row, col, v[i]);
      }
    }
  }
  // split-K: raw partial sums of the staged rows
  __device__ __forceinline__ static void partials(const ddpo_gemm_desc& d, const float* cw, float* pp, int row_base, int col_base, int lane) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e = it * 64 + lane, rr = e / LPR, lcol = (e - rr * LPR) * 4;
      const int row = row_base + rr, col = col_base + lcol;
      if (row >= d.M || col >= d.N) continue;
      *reinterpret_cast<float4*>(pp + (int64_t)row * d.N + col) = *reinterpret_cast<const float4*>(cw + rr * WTN + lcol);
    }
  }
};

// AFFINE: no upsampling / zero-insert in the gather, so the source address of tap (ky,kx) is rowptr + (ky*W + kx)*ld + ci
// and all per-k-tile work is a mask test and one 64-bit add per row (the generic path recomputes coordinates).
template <int BM, int BN, int NPASS, bool AFFINE>
__global__ void __launch_bounds__(BF_THREADS) gemm_conv_bf16_kernel(const ddpo_gemm_desc d, const uint16_t* __restrict__ w_hi,
                                                                   const uint16_t* __restrict__ w_lo, int ldw, int tiles_m,
                                                                   int tiles_n, int nblk, int kt_per_split,
                                                                   float* __restrict__ part) {
  constexpr int BK = BF_BK;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AROWS = BM / 32;                 // float4 chunks per thread (A tile)
  constexpr int BCH = BN / 64;                   // 16-byte chunks per thread per plane (W tile)
  constexpr int NPL = (NPASS == 3) ? 2 : 1;      // planes per operand
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = NPL * (A_BYTES + B_BYTES);      // per stage: A_hi | A_lo | B_hi | B_lo
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // block -> tile: XCD-contiguous chunks (bijective remap), then GROUP_M x tiles_n super-rows swept m-fastest, so the
  // ~64 blocks resident on one XCD cover a compact 2-D patch and share both A and W panels in that XCD's L2
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int group = bid / per_group, in_group = bid - group * per_group;
  const int gm0 = group * GROUP_M;
  const int gsz = min(tiles_m - gm0, GROUP_M);
  const int tile_n = in_group / gsz, tile_m = gm0 + (in_group - tile_n * gsz);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bool conv = d.ksize > 0;
  const int VH = d.upsample ? d.H * 2 : d.H, VW = d.upsample ? d.W * 2 : d.W;
  const bool zins = d.upsample == 2;

  // ---- A loader: thread owns float4 index kq (k = 4*kq..) of rows (t>>3) + 32*i
  const int kq = t & 7;
  const float* arow_ptr[AROWS];     // AFFINE conv: &src[pixel(oy*s-pad, ox*s-pad)][0] (may point outside; masked); dense: &src[m][0]
  uint32_t amask[AROWS];            // bit tap = that tap is inside the image (and the row is a real row)
  int aiy0[AROWS], aix0[AROWS];     // generic path
  int64_t abase[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + (t >> 3) + 32 * i;
    const bool valid = m < d.M;
    amask[i] = 0;
    if (conv) {
      const int ohw = d.OH * d.OW;
      const int mm = valid ? m : 0;
      const int b = mm / ohw, rem = mm - b * ohw;
      const int oy = rem / d.OW, ox = rem - oy * d.OW;
      const int iy0 = oy * d.stride - d.pad, ix0 = ox * d.stride - d.pad;
      abase[i] = (int64_t)b * d.H * d.W;
      aiy0[i] = iy0; aix0[i] = ix0;
      arow_ptr[i] = d.src + (abase[i] + (int64_t)iy0 * d.W + ix0) * d.ld_src;
      if (AFFINE && valid) {
        for (int ky = 0; ky < d.ksize; ++ky)
          for (int kx = 0; kx < d.ksize; ++kx)
            if (iy0 + ky >= 0 && iy0 + ky < d.H && ix0 + kx >= 0 && ix0 + kx < d.W) amask[i] |= 1u << (ky * d.ksize + kx);
      }
    } else {
      abase[i] = 0; aiy0[i] = aix0[i] = 0;
      arow_ptr[i] = d.src + (int64_t)m * d.ld_src;
      amask[i] = valid ? 1u : 0u;
    }
  }
  // split-K: blockIdx.y owns k-tiles [kt0, kt0 + nk) and writes raw partial sums to part[split] (reduced afterwards)
  const int nk_total = (d.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * kt_per_split;
  const int nk = min(kt_per_split, nk_total - kt0);
  // running (tap, ci) of this thread's k quad; advanced by BK per k-tile without divisions
  int a_tap = 0, a_ci = kt0 * BK + kq * 4;
  if (conv) { a_tap = a_ci / d.Cin; a_ci -= a_tap * d.Cin; }
  // ---- W loader: thread owns 16-byte chunk bc (8 bf16 of k) of rows (t>>2) + 64*i
  const int bc = t & 3;
  int b_tap = 0, b_co = kt0 * BK + bc * 8;
  if (d.w_dgrad) { b_tap = b_co / d.Cin; b_co -= b_tap * d.Cin; }
  const int ntaps = conv ? d.ksize * d.ksize : 1;

  struct Stage { float4 a[AROWS]; uint4 bh[BCH], bl[BCH]; };
  Stage s0, s1;       // two register stages: global loads run two k-tiles ahead of the MFMAs that consume them

  auto load_tile = [&](int ktr, Stage& sg) {
    const int kt = kt0 + ktr;                  // absolute k-tile; tiles at or beyond this split's end load as zeros
    const bool in_split = ktr < nk;
    // ---- A
    const bool kval = in_split && (kt * BK + kq * 4) < d.K;
    if (conv) {
      const int ky = (a_tap * 11) >> 5;                 // a_tap / 3 for a_tap < 32 (ksize 3); ksize 1 -> tap 0
      const int kyy = d.ksize == 3 ? ky : 0;
      const int kxx = d.ksize == 3 ? a_tap - ky * 3 : 0;
      if (AFFINE) {
        const int toff = (kyy * d.W + kxx) * d.ld_src + a_ci;
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kval && ((amask[i] >> a_tap) & 1u)) v = *reinterpret_cast<const float4*>(arow_ptr[i] + toff);
          sg.a[i] = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          const int m = m0 + (t >> 3) + 32 * i;
          const int iy = aiy0[i] + kyy, ix = aix0[i] + kxx;
          if (m < d.M && kval && iy >= 0 && iy < VH && ix >= 0 && ix < VW && !(zins && ((iy | ix) & 1))) {
            const int sy = d.upsample ? (iy >> 1) : iy, sx = d.upsample ? (ix >> 1) : ix;
            v = *reinterpret_cast<const float4*>(d.src + (abase[i] + (int64_t)sy * d.W + sx) * d.ld_src + a_ci);
          }
          sg.a[i] = v;
        }
      }
      a_ci += BK;
      while (a_ci >= d.Cin) { a_ci -= d.Cin; ++a_tap; }
    } else {
      const int kg = kt * BK + kq * 4;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kval && amask[i]) v = *reinterpret_cast<const float4*>(arow_ptr[i] + kg);
        sg.a[i] = v;
      }
    }
    // ---- W
    const int kb = kt * BK + bc * 8;
    const bool bval = in_split && kb < d.K;
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int n = n0 + (t >> 2) + 64 * i;
      uint4 h = make_uint4(0u, 0u, 0u, 0u), l = h;
      if (n < d.N && bval) {
        int64_t off;
        if (d.w_dgrad) off = ((int64_t)(ntaps - 1 - b_tap) * d.N + n) * d.Cin + b_co;   // forward [tap][ci=n][co] order, flipped tap
        else if (d.w_layout == 1) off = ((int64_t)(kb >> 5) * d.N + n) * 32 + (kb & 31);   // k-blocked (Kb, N, 32)
        else off = (int64_t)n * ldw + kb;
        h = *reinterpret_cast<const uint4*>(w_hi + off);
        if (NPASS == 3) l = *reinterpret_cast<const uint4*>(w_lo + off);
      }
      sg.bh[i] = h;
      sg.bl[i] = l;
    }
    if (d.w_dgrad) {
      b_co += BK;
      while (b_co >= d.Cin) { b_co -= d.Cin; ++b_tap; }
    }
  };

  // LDS offsets are loop invariant
  int a_st[AROWS], b_st[BCH];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) a_st[i] = swz_off((t >> 3) + 32 * i, kq >> 1) + (kq & 1) * 8;
#pragma unroll
  for (int i = 0; i < BCH; ++i) b_st[i] = swz_off((t >> 2) + 64 * i, bc);

  auto store_tile = [&](int buf, const Stage& sg) {
    char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      uint2 hi, lo;
      split4(sg.a[i], hi, lo);
      *reinterpret_cast<uint2*>(st + a_st[i]) = hi;
      if (NPASS == 3) *reinterpret_cast<uint2*>(st + A_BYTES + a_st[i]) = lo;
    }
    char* sb = st + NPL * A_BYTES;
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      *reinterpret_cast<uint4*>(sb + b_st[i]) = sg.bh[i];
      if (NPASS == 3) *reinterpret_cast<uint4*>(sb + B_BYTES + b_st[i]) = sg.bl[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int khalf = lane >> 5;
  int a_ld[BK / 16][TM], b_ld[BK / 16][TN];
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a_ld[ks][i] = swz_off(wm * (BM / 2) + i * 32 + (lane & 31), ks * 2 + khalf);
#pragma unroll
    for (int j = 0; j < TN; ++j) b_ld[ks][j] = swz_off(wn * (BN / 2) + j * 32 + (lane & 31), ks * 2 + khalf);
  }

  auto compute = [&](int cur) {
    const char* sa = smem + cur * STAGE;
    const char* sb = sa + NPL * A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8*>(sa + a_ld[ks][i]);
        if (NPASS == 3) al[i] = *reinterpret_cast<const bf16x8*>(sa + A_BYTES + a_ld[ks][i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(sb + b_ld[ks][j]);
        if (NPASS == 3) bl[j] = *reinterpret_cast<const bf16x8*>(sb + B_BYTES + b_ld[ks][j]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (NPASS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // prologue: tile 0 -> LDS[0]; tile 1 in flight in s0.  Tiles past the end load as zeros (kval / bval guards), so the
  // loop runs an even number of steps without a mid-body exit (keeps ONE copy of the accumulators live).
  load_tile(0, s0);
  store_tile(0, s0);
  load_tile(1, s0);
  __syncthreads();
  const int nk2 = (nk + 1) & ~1;
#pragma unroll 1
  for (int kt = 0; kt < nk2; kt += 2) {
    load_tile(kt + 2, s1);          // even step: MFMAs on LDS[0]; s0 holds tile kt+1, tile kt+2 starts loading into s1
    compute(0);
    store_tile(1, s0);
    __syncthreads();
    load_tile(kt + 3, s0);          // odd step: MFMAs on LDS[1]; s1 holds tile kt+2, tile kt+3 starts loading into s0
    compute(1);
    store_tile(0, s1);
    __syncthreads();
  }

  if (part) {      // split-K: raw partial sums, epilogue applied by splitk_reduce_kernel
    float* pp = part + (int64_t)blockIdx.y * d.M * d.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        if (col >= d.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          if (row < d.M) pp[(int64_t)row * d.N + col] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      if (col >= d.N) continue;
      const float bv = d.bias ? d.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row >= d.M) continue;
        float v = d.alpha * acc[i][j][r] + bv;
        if (d.rowbias) v += d.rowbias[(int64_t)(row / d.rows_per_batch) * d.ld_rowbias + col];
        if (d.residual) v += d.residual[(int64_t)row * d.ld_res + col];
        d.out[(int64_t)row * d.ld_out + col] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fast variant for the regular layers (conv: Cin % 32 == 0; dense: K % 32 == 0; every byte offset < 2^31): operands are
// fetched with raw BUFFER loads.  A k-tile of 32 never straddles a filter tap, so the tap is wave-uniform: the per-row
// byte offsets of the current tap live in VGPRs and are recomputed only when the tap changes (every Cin/32 k-tiles,
// covering padding, stride, nearest-2x upsampling and zero-insertion alike); masked rows carry an out-of-range offset
// and the buffer unit returns zeros for them.  The per-k-tile advance is one scalar add on the instruction's soffset:
// no per-load branches, no 64-bit address arithmetic, no zero-fill moves (the generic kernel above spends ~8 VALU +
// 1 branch per MFMA on those; here the only VALU work left in the k-loop is the fp32 -> bf16 hi/lo split).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define BUF_OOB 0x80000000u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7FFFFFFF, 0x00020000);
}

// WM x WN waves of (BM / WM) x (BN / WN) each: 2 x 2 waves for the 128x128 / 128x64 tiles (256 threads, 2-3 workgroups per
// CU); 4 x 2 waves of 32 x 160 for the 128x320 tile (512 threads, one workgroup per CU), which moves 233 B from L2 per
// MFMA instead of 341 (128x128) / 512 (128x64) and makes N = 320 / 640 / 1280 tile counts multiples of the 256 CUs.
// DEEP: global loads run two k-tiles ahead of the MFMAs (two register stages) instead of one.  Measured on the 8-wave tile:
// one-ahead frees 26 VGPRs (no spills) but is 4-18 % slower than two-ahead with its 14 spilled dwords, so DEEP stays on.
//
// APL ("A planes"): the activation operand arrives ALREADY split into bf16 hi / lo planes ([rows][ld] bf16, k contiguous;
// d.src = hi plane, d.w = lo plane, d.ld_src = row stride in ELEMENTS) written by the producing kernel (GroupNorm / LayerNorm
// apply, ddpo_split_planes_bf16).  Both operands then go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging
// VGPRs, no v_cvt / v_sub split, no ds_write.  A wave instruction fills 16 rows x 64 B lane-linearly, so the XOR swizzle of
// swz_off() is applied to the SOURCE chunk each lane fetches.  Same tiles, same k order, same three MFMA passes as the
// register-staged path: results are bit-identical to it.
template <int BM, int BN, int NPASS, int ABL = 0, int WM = 2, int WN = 2, bool DEEP = true, int APL = 0>     // ABL: timing ablations (tools/ablate_gemm.py; wrong results)
__global__ void __launch_bounds__(64 * WM * WN, (BM * BN <= 128 * 64 ? 3 : 1)) gemm_conv_bf16_buf_kernel(const ddpo_gemm_desc d, const uint16_t* __restrict__ w_hi,
                                                                       const uint16_t* __restrict__ w_lo, int ldw, int tiles_m,
                                                                       int tiles_n, int nblk, int kt_per_split,
                                                                       float* __restrict__ part) {
  constexpr int BK = BF_BK;
  constexpr int THREADS = 64 * WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;           // wave sub-tile
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AR = THREADS / 8, BR = THREADS / 4;     // rows covered by one pass of the A / W loaders
  constexpr int AROWS = BM / AR;
  constexpr int BCH = (BN + BR - 1) / BR;
  constexpr bool BFULL = (BN % BR) == 0;                // else the last W pass covers only part of the threads
  static_assert(BM % AR == 0 && WTM % 32 == 0 && WTN % 32 == 0, "tile / wave-grid mismatch");
  constexpr bool MX = NPASS == 4;                       // f16mx datapath (plane-fed only): f16 plane + 8-bit plane per operand
  static_assert(!MX || APL == 3 || APL == 7, "the f16mx datapath exists on the plane-fed 128-row tiles (APL 3) and on the tall tile (APL 7)");
  constexpr int NPL = (NPASS == 3 || NPASS == 4) ? 2 : 1;          // NPASS = 5: single-pass f16 (plane-fed only; the f16mx planes' 16-bit plane alone)
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = NPL * (A_BYTES + B_BYTES);
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid / WN, wn = wid % WN;
  DBG_T(0);
  DBG_WDECL;

  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int group = bid / per_group, in_group = bid - group * per_group;
  const int gm0 = group * GROUP_M;
  const int gsz = min(tiles_m - gm0, GROUP_M);
  const int tile_n = in_group / gsz, tile_m = gm0 + (in_group - tile_n * gsz);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bool conv = d.ksize > 0;
  const int VH = d.upsample ? d.H * 2 : d.H, VW = d.upsample ? d.W * 2 : d.W;
  const bool zins = d.upsample == 2;
  const int cin = conv ? d.Cin : d.K;            // reduction channels per tap (dense: one "tap" spanning K)
  const int ntaps = conv ? d.ksize * d.ksize : 1;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int khalf = lane >> 5;
  // Fragment read offsets.  A 32-row block further down adds 32 * 64 B and leaves the swizzle term ((row >> 2) & 3) unchanged, and
  // the ks = 1 chunk index (2 + khalf) differs from the ks = 0 one (khalf) in bit 1 only, i.e. the byte offset in bit 5: all offsets
  // of a wave derive from TWO registers (a_ld0, b_ld0) by compile-time additions that fold into the ds_read offset field and one
  // XOR for ks = 1 — 4 address VGPRs instead of 2 * (TM + TN) (14 on the 256x320 tile, which has none to spare).
  static_assert((WTM % 32) == 0 && (WTN % 32) == 0, "wave sub-tiles are whole 32-row blocks");
  const int a_ld0 = swz_off(wm * WTM + (lane & 31), khalf), b_ld0 = swz_off(wn * WTN + (lane & 31), khalf);
  const int a_ld1 = a_ld0 ^ 32, b_ld1 = b_ld0 ^ 32;

  // Fragment loads are software-pipelined by hand across the barrier: the ks=0 fragments of the NEXT k-tile are requested
  // right after the barrier that publishes it and the second half of the current tile's ks=1 MFMAs is issued behind them,
  // so the LDS latency is covered by matrix work instead of stalling the wave at the top of every k-tile.
  // f16mx: the 16-bit fragments are f16 (same bytes, same offsets), and at ks = 1 each operand's 32 bytes of the 8-bit plane are read as
  // one fragment from the SAME two offsets (the plane's chunks are [h8 | l8 | h8 | l8] for activations, [l8 | h8 | l8 | h8] for weights: lane half
  // 0 gets a_h8 and w_l8 of k 0..31, lane half 1 a_l8 and w_h8) — the two cross terms of the k-tile ride in the two lane halves of ONE 32x32x64 MFMA.
  struct Frag { bf16x8 ah[TM], al[TM], bh[TN], bl[TN]; i32x8 a8[TM], b8[TN]; };
  auto ldfrag_at = [&](const char* sa, const char* sb, int ks, Frag& f) {
    const char* pa = sa + (ks ? a_ld1 : a_ld0);
    const char* pb = sb + (ks ? b_ld1 : b_ld0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      f.ah[i] = *reinterpret_cast<const bf16x8*>(pa + i * 2048);
      if (NPASS == 3) f.al[i] = *reinterpret_cast<const bf16x8*>(pa + A_BYTES + i * 2048);
      if (MX && ks) {
        const i32x4 x = *reinterpret_cast<const i32x4*>(sa + a_ld0 + A_BYTES + i * 2048), y = *reinterpret_cast<const i32x4*>(sa + a_ld1 + A_BYTES + i * 2048);
        f.a8[i] = i32x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f.bh[j] = *reinterpret_cast<const bf16x8*>(pb + j * 2048);
      if (NPASS == 3) f.bl[j] = *reinterpret_cast<const bf16x8*>(pb + B_BYTES + j * 2048);
      if (MX && ks) {
        const i32x4 x = *reinterpret_cast<const i32x4*>(sb + b_ld0 + B_BYTES + j * 2048), y = *reinterpret_cast<const i32x4*>(sb + b_ld1 + B_BYTES + j * 2048);
        f.b8[j] = i32x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
      }
    }
  };
  auto ldfrag = [&](int cur, int ks, Frag& f) {
    const char* sa = smem + cur * STAGE;
    ldfrag_at(sa, sa + NPL * A_BYTES, ks, f);
  };
  // f16mx block scales (E8M0 bytes, 2^(byte - 127)): activations are stored at scale 1 (h8) and 2^-11 (l8 = l * 2^11); a weight column
  // at its own scale s_n (h8) and s_n * 2^-11 (l8).  Lane half 0 multiplies a_h8 * w_l8, lane half 1 a_l8 * w_h8.
  int mx_sa = 0, mx_sbp[(TN + 3) / 4];          // weight scales: byte (j & 3) of register j >> 2 (the MFMA's op_sel picks the byte)
  if constexpr (MX) {
    mx_sa = khalf ? 127 - 11 : 127;
#pragma unroll
    for (int q = 0; q < (TN + 3) / 4; ++q) mx_sbp[q] = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WTN + j * 32 + (lane & 31);
      const int sc = n < d.N ? (int)d.w_scale[n] : 127;
      mx_sbp[j >> 2] |= (khalf ? sc : sc - 11) << (8 * (j & 3));
    }
  }
  auto mma = [&](const Frag& f, int tbeg, int tend, int ks = 0) {       // 32x32 blocks [tbeg, tend) of the wave tile, row-major
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (i * TN + j < tbeg || i * TN + j >= tend) continue;
        if constexpr (MX) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.ah[i]), __builtin_bit_cast(f16x8, f.bh[j]), acc[i][j], 0, 0, 0);
          if (ks) acc[i][j] = mx_mfma(f.a8[i], f.b8[j], acc[i][j], mx_sa, j & 3, mx_sbp[j >> 2]);
          continue;
        }
        if (NPASS == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[i], f.bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bl[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (NPASS == 5) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.ah[i]), __builtin_bit_cast(f16x8, f.bh[j]), acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[i], f.bh[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  if constexpr (APL != 0) {
    // ---------------- LDS-DMA path: A planes + W planes straight into the swizzled LDS image ----------------
    static_assert(NPASS == 1 || NPASS == 3 || NPASS == 4 || NPASS == 5, "the plane-fed path: bf16x3 / f16mx (two planes per operand) or single-pass bf16 / f16 (one)");
    // two planes per operand: even waves move hi planes, odd waves lo planes (PAIRS loader groups per plane); ONE plane (NPASS = 1, round 6):
    // every wave is a loader group of the only plane.  NB is rounded up: with 8 waves the 20 weight pieces of a 320-column tile are 3 per wave,
    // the four surplus ones carry an out-of-range offset (they arrive as zeros, no memory traffic) and land in rows 320 .. 383 of a weight
    // stage padded to B_LDS bytes, which no fragment read touches — every wave issues the same number of pieces, so ONE counted vmcnt serves all.
    constexpr int NW = WM * WN, PAIRS = NPL == 2 ? NW / 2 : NW;
    constexpr int GA = BM / 16, GB = BN / 16;            // 16-row groups = 1 KiB LDS-DMA pieces per plane
    static_assert(NW % 2 == 0 && GA % PAIRS == 0 && (NPL == 1 || GB % PAIRS == 0), "pieces must divide evenly over the loader groups");
    constexpr int NA = GA / PAIRS, NB = (GB + PAIRS - 1) / PAIRS;
    constexpr int B_LDS = NB * PAIRS * 1024;             // == B_BYTES unless padded (NPASS = 1, 320 columns on 8 waves: 24 KB for 20)
    const int wv = __builtin_amdgcn_readfirstlane(wid);
    const int plane = NPL == 2 ? (wv & 1) : 0, pr = NPL == 2 ? (wv >> 1) : wv;
    const int lr = lane >> 2;                            // row inside the 16-row piece
    const uint32_t lc16 = (uint32_t)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;     // source chunk whose lane-linear slot equals swz_off()
    const uint64_t a_ptr = reinterpret_cast<uint64_t>(plane ? reinterpret_cast<const void*>(d.w) : reinterpret_cast<const void*>(d.src));
    const uint64_t w_ptr = reinterpret_cast<uint64_t>(plane ? w_lo : w_hi);
    const u32x4 rs_a = {(uint32_t)a_ptr, (uint32_t)(a_ptr >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
    const u32x4 rs_w = {(uint32_t)w_ptr, (uint32_t)(w_ptr >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const uint32_t lds_a = lds0 + plane * A_BYTES + pr * 1024;                       // + stage * STAGE + i * PAIRS * 1024
    const uint32_t lds_w = lds0 + NPL * A_BYTES + plane * B_BYTES + pr * 1024;

    // activation planes: row-major (rows, ld_src) -> row stride ld_src * 2 B, k-tile advance 64 B; k-blocked (ld_src == 0: (C / 32, rows, 32))
    // -> row stride 64 B, k-tile advance rows * 64 B (16 consecutive pixels of a piece = 1 KiB of consecutive memory)
    const uint32_t a_row_b = d.ld_src == 0 ? 64u : (uint32_t)d.ld_src * 2u;
    const uint32_t a_kt_b = d.ld_src == 0 ? (uint32_t)(conv ? d.B * d.H * d.W : d.M) * 64u : (uint32_t)(BK * 2);
    int aiy0[NA], aix0[NA], apix[NA];
    uint32_t avoff[NA], bvoff[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = m0 + 16 * (pr + PAIRS * i) + lr;
      const bool valid = m < d.M;
      if (conv) {
        const int ohw = d.OH * d.OW;
        const int mm = valid ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / d.OW, ox = rem - oy * d.OW;
        aiy0[i] = valid ? oy * d.stride - d.pad : -(1 << 24);
        aix0[i] = ox * d.stride - d.pad;
        apix[i] = b * d.H * d.W;
        avoff[i] = BUF_OOB;
      } else {
        aiy0[i] = aix0[i] = apix[i] = 0;
        avoff[i] = valid ? (uint32_t)m * a_row_b + lc16 : BUF_OOB;
      }
    }
    auto set_tap = [&](int tap) {
      const int ky = d.ksize == 3 ? (tap * 11) >> 5 : 0;
      const int kx = d.ksize == 3 ? tap - ky * 3 : 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int iy = aiy0[i] + ky, ix = aix0[i] + kx;
        const bool ok = (unsigned)iy < (unsigned)VH && (unsigned)ix < (unsigned)VW && !(zins && ((iy | ix) & 1));
        const int sy = d.upsample ? (iy >> 1) : iy, sx = d.upsample ? (ix >> 1) : ix;
        const uint32_t off = (uint32_t)(apix[i] + sy * d.W + sx) * a_row_b + lc16;
        avoff[i] = ok ? off : BUF_OOB;
      }
    };
    // weight planes: row-major (N, ldw) -> row stride ldw * 2 B, k-tile advance 64 B; k-blocked (Kb, N, 32) -> row stride 64 B,
    // k-tile advance N * 64 B (a piece = 16 consecutive columns = 1 KiB of consecutive memory)
    const uint32_t w_row_b = d.w_layout == 1 ? 64u : (uint32_t)ldw * 2u;
    const uint32_t w_kt_b = d.w_layout == 1 ? (uint32_t)d.N * 64u : (uint32_t)(BK * 2);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + 16 * (pr + PAIRS * i) + lr;
      bool in_tile = true;
      if constexpr (GB % PAIRS != 0) in_tile = pr + PAIRS * i < GB;          // surplus pieces of a rounded-up NB (single plane, 320 columns)
      bvoff[i] = (n < d.N && in_tile) ? (uint32_t)n * w_row_b + lc16 : BUF_OOB;
    }
    const int nk_total = d.K / BK;
    const int kt0 = blockIdx.y * kt_per_split;
    const int nk = min(kt_per_split, nk_total - kt0);
    int tap = (kt0 * BK) / cin, cib = kt0 * BK - tap * cin;        // position of the NEXT k-tile to request
    if (conv) set_tap(tap);
    int kt_next = kt0;

    // one k-tile = NA + NB LDS-DMA pieces per wave (issued back to back; nothing of it touches a VGPR besides the offsets)
    auto fill = [&](int stage) {
      const uint32_t so_a = (uint32_t)(cib >> 5) * a_kt_b, so_w = (uint32_t)kt_next * w_kt_b;
      const uint32_t la = lds_a + stage * STAGE, lw = lds_w + stage * STAGE;
      if (!DBG_ABL(4) || (kt_next - kt0) % 9 < 2) {         // timing ablation 4: the activation bytes of a halo loader (2 of 9 k-tiles)
#pragma unroll
      for (int i = 0; i < NA; ++i)
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(la + i * (PAIRS * 1024)), "v"(avoff[i]), "s"(rs_a), "s"(so_a) : "memory");
      }
#pragma unroll
      for (int i = 0; i < NB; ++i)
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(lw + i * (PAIRS * 1024)), "v"(bvoff[i]), "s"(rs_w), "s"(so_w) : "memory");
      ++kt_next;
      cib += BK;
      if (cib >= cin) {
        cib = 0; ++tap;
        if (conv && tap < ntaps) set_tap(tap);
      }
    };
    if constexpr (APL == 5 || APL == 6) {
      // Schedule of the TALL 256x320 tile (64 x 160 per wave: 160 accumulator registers leave room for ONE fragment set;
      // the second wave of the SIMD covers the LDS latency).  28 fragment reads and 9 LDS-DMA pieces feed 60 MFMAs per wave and
      // k-tile, against 24 + 7 for 30 MFMAs on the 128x320 tile: 36 % fewer L2 and 42 % fewer LDS bytes per MFMA.
      const bool late = (d.splits & 1) != 0 && wv >= NW / 2;
      auto step4 = [&](int kt, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        DBG_W0();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        DBG_WMID();
        __builtin_amdgcn_s_barrier();
        DBG_W1();
        if (kt == 0) DBG_T(1);
        if (kt + 1 < nk && !late && !DBG_ABL(1)) fill(cur ^ 1);
        Frag g;
        ldfrag(cur, 0, g);
        if (!DBG_ABL(2)) mma(g, 0, TM * TN);
        if (kt + 1 < nk && late && !DBG_ABL(1)) fill(cur ^ 1);
        ldfrag(cur, 1, g);
        if (!DBG_ABL(2)) mma(g, 0, TM * TN);
        if (DBG_ABL(2)) asm volatile("" :: "v"(g.ah[0]), "v"(g.bl[TN - 1]), "v"(g.al[TM - 1]), "v"(g.bh[0]));
      };
      {
        // ROTATED schedule (round 3).  A plain loop (every wave: wait, barrier, request, read ks = 0, multiply, read ks = 1, multiply) makes both waves of a SIMD do the same thing at the same time: after the
        // barrier both request fragments (nobody computes), then both compute.  Here the upper half of the waves runs the SAME
        // per-k-tile work shifted by half a phase: it carries the ks = 1 fragments of tile kt - 1 ACROSS the barrier and multiplies them
        // while the lower half issues its LDS-DMA pieces and reads the ks = 0 fragments of tile kt; from then on one wave of every SIMD
        // reads while the other multiplies.  The barrier contract is unchanged — every wave has (a) waited for its own pieces of tile kt
        // and (b) received all of its fragment READS of tile kt - 1 (lgkmcnt(0)) before it arrives; only the register-only MFMAs of
        // those fragments are issued after it.  Per-element accumulation order is the plain loop's: bit-identical results.
        const bool rot = wv >= NW / 2;
        fill(0);
        Frag g;
        if (!rot) {
          int kt = 0;
#pragma unroll 1
          for (; kt + 1 < nk; kt += 2) {
            step4(kt, std::integral_constant<int, 0>{});
            step4(kt + 1, std::integral_constant<int, 1>{});
          }
          if (kt < nk) step4(kt, std::integral_constant<int, 0>{});
        } else {
          auto step5 = [&](int kt, auto cur_c) {
            constexpr int cur = decltype(cur_c)::value;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt > 0) mma(g, 0, TM * TN);              // ks = 1 of tile kt - 1 (fragments read before the barrier)
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) fill(cur ^ 1);
            ldfrag(cur, 0, g);
            mma(g, 0, TM * TN);
            ldfrag(cur, 1, g);
          };
          int kt = 0;
#pragma unroll 1
          for (; kt + 1 < nk; kt += 2) {
            step5(kt, std::integral_constant<int, 0>{});
            step5(kt + 1, std::integral_constant<int, 1>{});
          }
          if (kt < nk) step5(kt, std::integral_constant<int, 0>{});
          mma(g, 0, TM * TN);                            // ks = 1 of the last tile
        }
      }
    } else if constexpr (APL == 3) {
      // Mode 2's shape with the WEIGHT operand three LDS stages deep: [A s0 | A s1 | W s0 | W s1 | W s2] (128x320: 2 x 16 KB +
      // 3 x 40 KB = 152 KB).  At the barrier of k-tile s the activation pieces of tile s + 2 and the weight pieces of tile s + 3
      // are requested, in that order; the wait in front of the next barrier is a COUNTED vmcnt(NB): everything but the newest NB
      // pieces (the weight tile that is not needed for another whole k-tile) has landed.  Weights are the cold operand in the
      // model — every launch streams them from HBM — and now have two k-tiles of latency tolerance like the register-staged loop.
      constexpr int A_STAGE = NPL * A_BYTES, W_STAGE = NPL * B_LDS;
      const uint32_t lds_a3 = lds0 + plane * A_BYTES + pr * 1024;
      const uint32_t lds_w3 = lds0 + 2 * A_STAGE + plane * B_LDS + pr * 1024;
      int kw_next = kt0;                                   // tap / cib / set_tap follow the ACTIVATION tiles
      int ka_next = 0;                                     // (timing ablation 4 only)
      auto fill_a = [&](int stage) {
        const uint32_t so_a = (uint32_t)(cib >> 5) * a_kt_b, la = lds_a3 + stage * A_STAGE;
        if (!DBG_ABL(4) || ka_next % 9 < 2) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
          asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                       :: "s"(la + i * (PAIRS * 1024)), "v"(avoff[i]), "s"(rs_a), "s"(so_a) : "memory");
        }
        ++ka_next;
        cib += BK;
        if (cib >= cin) {
          cib = 0; ++tap;
          if (conv && tap < ntaps) set_tap(tap);
        }
      };
      auto fill_w = [&](int stage) {
        const uint32_t so_w = (uint32_t)kw_next * w_kt_b, lw = lds_w3 + stage * W_STAGE;
#pragma unroll
        for (int i = 0; i < NB; ++i)
          asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                       :: "s"(lw + i * (PAIRS * 1024)), "v"(bvoff[i]), "s"(rs_w), "s"(so_w) : "memory");
        ++kw_next;
      };
      // stage offsets are RUNTIME scalars (one v_add per fragment read): with compile-time stages the 152 KB image exceeds the
      // 64 KB reach of the ds_read offset field, the compiler keeps one address register per (stage, fragment) and spills
      auto ldfrag3 = [&](uint32_t a_off, uint32_t w_off, int ks, Frag& f) {
        ldfrag_at(smem + a_off, smem + 2 * A_STAGE + w_off, ks, f);
      };
      const bool late = (d.splits & 1) != 0 && wv >= NW / 2;
      Frag g0, g1;
      fill_a(0); fill_w(0);
      if (nk > 1) { fill_a(1); fill_w(1); }
      if (nk > 2) fill_w(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      DBG_T(1);
      ldfrag3(0, 0, 0, g0);
      uint32_t as = 0, ws = 0;                             // stage INDICES of the current k-tile (kt & 1, kt % 3)
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) {
        const uint32_t a_off = as * A_STAGE, w_off = ws * W_STAGE;
        const uint32_t as_n = as ^ 1, ws_n = ws == 2 ? 0 : ws + 1;
        ldfrag3(a_off, w_off, 1, g1);
        __builtin_amdgcn_sched_barrier(0);
        if (!DBG_ABL(2)) mma(g0, 0, TM * TN);
        else asm volatile("" :: "v"(g0.ah[0]), "v"(g0.bl[TN - 1]), "v"(g0.al[TM - 1]), "v"(g0.bh[0]));
        __builtin_amdgcn_sched_barrier(0);
        // tile kt + 1 (A requested one barrier ago, W two barriers ago) must have landed; the weight tile kt + 2 requested one
        // barrier ago — the newest NB pieces of this wave — may stay in flight.  lgkmcnt(0): my reads of tile kt's stages returned.
        DBG_W0();
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        DBG_WMID();
        __builtin_amdgcn_s_barrier();
        DBG_W1();
        if (!late && !DBG_ABL(1)) {                        // activation tile kt + 2, then weight tile kt + 3, into the stages just freed
          if (kt + 2 < nk) fill_a(as);
          if (kt + 3 < nk) fill_w(ws);
        }
        if (kt + 1 < nk) ldfrag3(as_n * A_STAGE, ws_n * W_STAGE, 0, g0);
        __builtin_amdgcn_sched_barrier(0);
        if (!DBG_ABL(2)) mma(g1, 0, (TM * TN) / 2, 1);
        else asm volatile("" :: "v"(g1.ah[0]), "v"(g1.bl[TN - 1]), "v"(g1.al[TM - 1]), "v"(g1.bh[0]));
        __builtin_amdgcn_sched_barrier(0);
        if (late && !DBG_ABL(1)) {
          if (kt + 2 < nk) fill_a(as);
          if (kt + 3 < nk) fill_w(ws);
        }
        if (!DBG_ABL(2)) mma(g1, (TM * TN) / 2, TM * TN, 1);
        __builtin_amdgcn_sched_barrier(0);
        as = as_n; ws = ws_n;
      }
    } else if constexpr (APL == 7) {
      // f16mx on the TALL 256 x 320 tile (round 5): eight waves of 64 x 160, two per SIMD (256 registers per lane: 160 accumulators + 96).
      // Why a tall tile: the f16mx kernels are bound by the chip's L2 -> LDS stream (8.4 - 11.4 TB/s, tools/native/dma_bench), not by the
      // matrix pipe — the 128 x 320 tile fetches 57 KB per k-tile for 1.31 M MAC (23 MAC / B), this one 73.7 KB for 2.62 M (35.5 MAC / B):
      // 36 % fewer operand bytes per product.  Round 3 built this tile twice and lost 44 - 58 values to scratch around the 8-register operands
      // of the scaled MFMA; this form keeps the fragment set at 64 registers by construction: the activation fragments of the k-tile
      // (2 x [f16 ks 0 | f16 ks 1 | 8-bit] = 32 registers) stay resident, the weight fragments of ONE column block (16 registers) are read in
      // front of that block's six MFMAs into one of two buffers, and scheduling barriers between the column blocks keep the compiler from
      // hoisting later blocks' reads (a four-wave form — 128 x 160 per wave, 320 accumulators — does not compile to anything usable: the
      // MFMAs take the AGPR form, 64 accumulators do not fit the 256 AGPRs and travel through v_accvgpr moves and 1.8 KB of scratch).
      // Per accumulator the order is f16 ks 0, f16 ks 1, MX — the 128-row kernel's — so the results are bit-identical to it.
      // Two LDS stages (2 x 72 KB); one barrier per k-tile: behind it stage cur ^ 1 is free (every wave's reads of k-tile kt - 1 returned:
      // lgkmcnt(0) in front of the barrier) and k-tile kt is published (every wave waited for its own pieces: vmcnt(0)).  The upper half of
      // the waves requests its pieces of k-tile kt + 1 two column blocks later than the lower half (the 128-row loop's stagger).
      static_assert(MX && BM == 256 && BN == 320 && WM == 4 && WN == 2, "the f16mx tall tile");
      // fragment reads (offsets as in ldfrag_at: the 8-bit plane of an operand sits A_BYTES / B_BYTES behind its 16-bit plane)
      const bool late = (d.splits & 1) != 0 && wv >= NW / 2;
      fill(0);
      auto ktile = [&](int kt, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        const char* sa = smem + cur * STAGE;
        const char* sb = sa + NPL * A_BYTES;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt == 0) DBG_T(1);
        if (kt + 1 < nk && !late && !DBG_ABL(1)) fill(cur ^ 1);
        // phase 1: the two f16 products of every accumulator (k halves 0 and 1 of the k-tile)
        {
          bf16x8 ah0[TM], ah1[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            ah0[i] = *reinterpret_cast<const bf16x8*>(sa + a_ld0 + i * 2048);
            ah1[i] = *reinterpret_cast<const bf16x8*>(sa + a_ld1 + i * 2048);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(sb + b_ld0 + j * 2048);
            const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(sb + b_ld1 + j * 2048);
            if (DBG_ABL(2)) { asm volatile("" :: "v"(ah0[0]), "v"(ah1[TM - 1]), "v"(bh0), "v"(bh1)); continue; }
#pragma unroll
            for (int i = 0; i < TM; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah0[i]), __builtin_bit_cast(f16x8, bh0), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah1[i]), __builtin_bit_cast(f16x8, bh1), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (kt + 1 < nk && late && !DBG_ABL(1)) fill(cur ^ 1);
        // phase 2: the MX product (both cross terms of the whole k-tile in one 32x32x64 MFMA per accumulator)
        {
          i32x8 a8[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const i32x4 x = *reinterpret_cast<const i32x4*>(sa + a_ld0 + A_BYTES + i * 2048), y = *reinterpret_cast<const i32x4*>(sa + a_ld1 + A_BYTES + i * 2048);
            a8[i] = i32x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const i32x4 x = *reinterpret_cast<const i32x4*>(sb + b_ld0 + B_BYTES + j * 2048), y = *reinterpret_cast<const i32x4*>(sb + b_ld1 + B_BYTES + j * 2048);
            const i32x8 b8 = i32x8{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
            if (DBG_ABL(2)) { asm volatile("" :: "v"(a8[0]), "v"(a8[TM - 1]), "v"(b8)); continue; }
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = mx_mfma(a8[i], b8, acc[i][j], mx_sa, j & 3, mx_sbp[j >> 2]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      int kt = 0;
#pragma unroll 1
      for (; kt + 1 < nk; kt += 2) {
        ktile(kt, std::integral_constant<int, 0>{});
        ktile(kt + 1, std::integral_constant<int, 1>{});
      }
      if (kt < nk) ktile(kt, std::integral_constant<int, 0>{});
    } else if constexpr (APL == 8) {
      // SINGLE-PASS bf16 on the TALL 256 x 320 tile (round 6; BASELINE configs[4] names bf16: one v_mfma_f32_32x32x16_bf16 per product, XLA's TPU
      // default precision).  One plane per operand halves a k-tile's bytes (A 16 KB + W 20 KB, padded to 24), so the 160 KB of LDS hold a RING of
      // S = 4 stages instead of two: S - 1 k-tiles are requested ahead and the wait in front of a k-tile's barrier is a COUNTED vmcnt that leaves the
      // S - 2 youngest tiles (10 pieces of this wave) in flight — the stream never drains inside the loop (tools/native/dma_bench2: the same request
      // pattern sustains 13.5 TB/s with 80 KB in flight per CU against 10.8 with one 40 KB tile and a drain per tile).  Single pass has a third of
      // bf16x3's MFMAs per byte, i.e. this is the instantiation where the loader structure, not the matrix pipe, sets the rate.
      // One barrier per k-tile: behind it tile kt is published (every wave waited for its own pieces) and the stage of tile kt - 1 is free
      // (every wave's fragment reads of it returned: lgkmcnt(0)), so tile kt + S - 1 is requested into it.  Stage offsets are runtime values
      // (one v_add per fragment base: the 160 KB image exceeds the 64 KB reach of the ds_read offset field).  Per accumulator the order is
      // k half 0, k half 1 of consecutive k-tiles — the fp32-fed single-pass kernel's — so the two agree bit for bit.
      static_assert((NPASS == 1 || NPASS == 5) && BM == 256 && BN == 320 && WM == 4 && WN == 2, "the single-pass tall tile");
      constexpr int S = 4, ST = A_BYTES + B_LDS, P = NA + NB;
      static_assert(S * ST <= 160 * 1024 && P * (S - 2) < 64, "stage ring must fit the LDS and the vmcnt field");
      const uint32_t lds_a8 = lds0 + pr * 1024, lds_w8 = lds0 + A_BYTES + pr * 1024;
      auto fill8 = [&](uint32_t st_off) {
        const uint32_t so_a = (uint32_t)(cib >> 5) * a_kt_b, so_w = (uint32_t)kt_next * w_kt_b;
        const uint32_t la = lds_a8 + st_off, lw = lds_w8 + st_off;
#pragma unroll
        for (int i = 0; i < NA; ++i)
          asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                       :: "s"(la + i * (PAIRS * 1024)), "v"(avoff[i]), "s"(rs_a), "s"(so_a) : "memory");
#pragma unroll
        for (int i = 0; i < NB; ++i)
          asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                       :: "s"(lw + i * (PAIRS * 1024)), "v"(bvoff[i]), "s"(rs_w), "s"(so_w) : "memory");
        ++kt_next;
        cib += BK;
        if (cib >= cin) {
          cib = 0; ++tap;
          if (conv && tap < ntaps) set_tap(tap);
        }
      };
      const bool late = (d.splits & 1) != 0 && wv >= NW / 2;
#pragma unroll
      for (int s_ = 0; s_ < S - 1; ++s_)
        if (s_ < nk) fill8(s_ * ST);
      uint32_t st_c = 0, st_f = (S - 1) * ST;              // byte offsets of the stage computed on / the stage requested into
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed; the S - 2 younger tiles (if the reduction still has them) stay in flight
        if (kt + S - 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(P * (S - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt == 0) DBG_T(1);
        if (kt + S - 1 < nk && !late && !DBG_ABL(1)) fill8(st_f);
        const char* sa = smem + st_c;
        const char* sb = sa + A_BYTES;
        bf16x8 ah0[TM], ah1[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah0[i] = *reinterpret_cast<const bf16x8*>(sa + a_ld0 + i * 2048);
          ah1[i] = *reinterpret_cast<const bf16x8*>(sa + a_ld1 + i * 2048);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(sb + b_ld0 + j * 2048);
          const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(sb + b_ld1 + j * 2048);
          if (DBG_ABL(2)) { asm volatile("" :: "v"(ah0[0]), "v"(ah1[TM - 1]), "v"(bh0), "v"(bh1)); continue; }
          if constexpr (NPASS == 5) {      // single-pass f16 (opt-in: the f16mx operator without its cross terms)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah0[i]), __builtin_bit_cast(f16x8, bh0), acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah1[i]), __builtin_bit_cast(f16x8, bh1), acc[i][j], 0, 0, 0);
          } else {
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah0[i], bh0, acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah1[i], bh1, acc[i][j], 0, 0, 0);
          }
          if (j == 1 && kt + S - 1 < nk && late && !DBG_ABL(1)) fill8(st_f);      // the upper half of the waves requests two column blocks later
        }
        st_c = st_c + ST == S * ST ? 0 : st_c + ST;
        st_f = st_f + ST == S * ST ? 0 : st_f + ST;
      }
    } else {
      static_assert(APL == 3 || APL == 5 || APL == 6, "plane-fed k-loops: 3 = three weight stages (128-row tiles), 5 = tall tile, 6 = tall tile with the GEGLU output stage, 7 = f16mx tall tile, 8 = single-pass tall tile");
    }
  } else {
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(d.src);
    const __amdgpu_buffer_rsrc_t rs_wh = make_rsrc(w_hi);
    const __amdgpu_buffer_rsrc_t rs_wl = make_rsrc(NPASS == 3 ? w_lo : w_hi);

    // ---- A rows of this thread: (t>>3) + 32*i, float4 index kq inside the 32-wide k-tile
    const int kq = t & 7;
    int aiy0[AROWS], aix0[AROWS], apix[AROWS];    // top-left input coordinate and batch pixel base; invalid rows get iy0 << 0
    uint32_t avoff[AROWS];                        // byte offset of the CURRENT tap's pixel (+ kq*16), or BUF_OOB
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int m = m0 + (t >> 3) + AR * i;
      const bool valid = m < d.M;
      if (conv) {
        const int ohw = d.OH * d.OW;
        const int mm = valid ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oy = rem / d.OW, ox = rem - oy * d.OW;
        aiy0[i] = valid ? oy * d.stride - d.pad : -(1 << 24);
        aix0[i] = ox * d.stride - d.pad;
        apix[i] = b * d.H * d.W;
        avoff[i] = BUF_OOB;
      } else {
        aiy0[i] = aix0[i] = apix[i] = 0;
        avoff[i] = valid ? (uint32_t)m * (uint32_t)d.ld_src * 4u + kq * 16u : BUF_OOB;
      }
    }
    auto set_tap = [&](int tap) {                 // conv only; wave-uniform tap
      const int ky = d.ksize == 3 ? (tap * 11) >> 5 : 0;
      const int kx = d.ksize == 3 ? tap - ky * 3 : 0;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int iy = aiy0[i] + ky, ix = aix0[i] + kx;
        const bool ok = (unsigned)iy < (unsigned)VH && (unsigned)ix < (unsigned)VW && !(zins && ((iy | ix) & 1));
        const int sy = d.upsample ? (iy >> 1) : iy, sx = d.upsample ? (ix >> 1) : ix;
        const uint32_t off = (uint32_t)(apix[i] + sy * d.W + sx) * (uint32_t)d.ld_src * 4u + kq * 16u;
        avoff[i] = ok ? off : BUF_OOB;
      }
    };

    // ---- W rows of this thread: (t>>2) + 64*i, 16-byte chunk bc of the k-tile
    const int bc = t & 3;
    uint32_t bvoff[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const int br = (t >> 2) + BR * i;
      const int n = n0 + br;
      const uint32_t row_bytes = d.w_dgrad ? (uint32_t)d.Cin * 2u : (d.w_layout == 1 ? 64u : (uint32_t)ldw * 2u);
      bvoff[i] = (n < d.N && (BFULL || br < BN)) ? (uint32_t)n * row_bytes + bc * 16u : BUF_OOB;
    }

    const int nk_total = d.K / BK;
    const int kt0 = blockIdx.y * kt_per_split;
    const int nk = min(kt_per_split, nk_total - kt0);
    // wave-uniform running position of the NEXT k-tile to load: tap index and channel base inside the tap
    int tap = (kt0 * BK) / cin, cib = kt0 * BK - tap * cin;
    if (conv) set_tap(tap);

    struct Stage { float4 a[AROWS]; uint4 bh[BCH], bl[BCH]; };
    Stage s0, s1;

    // Loads are unconditional (a branch around them would make the compiler's s_waitcnt placement conservative and
    // collapse the prefetch distance): requests past this split's last k-tile re-fetch the last tile and are never consumed.
    auto load_tile = [&](int ktr, Stage& sg) {
      const int so_a = cib * 4;
      const int so_w = d.w_dgrad ? ((ntaps - 1 - tap) * d.N * d.Cin + cib) * 2 : (kt0 + min(ktr, nk - 1)) * (d.w_layout == 1 ? d.N * 64 : BK * 2);
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_a, avoff[i], so_a, 0);
        sg.a[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
#pragma unroll
      for (int i = 0; i < BCH; ++i) {
        const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(rs_wh, bvoff[i], so_w, 0);
        sg.bh[i] = make_uint4(h.x, h.y, h.z, h.w);
        if (NPASS == 3) {
          const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(rs_wl, bvoff[i], so_w, 0);
          sg.bl[i] = make_uint4(l.x, l.y, l.z, l.w);
        }
      }
      if (ktr < nk - 1) {                         // uniform; no memory operations inside
        cib += BK;
        if (cib >= cin) {                         // next k-tile starts a new tap
          cib = 0; ++tap;
          if (conv) set_tap(tap);
        }
      }
    };

    int a_st[AROWS], b_st[BCH];
#pragma unroll
    for (int i = 0; i < AROWS; ++i) a_st[i] = swz_off((t >> 3) + AR * i, kq >> 1) + (kq & 1) * 8;
#pragma unroll
    for (int i = 0; i < BCH; ++i) b_st[i] = swz_off((t >> 2) + BR * i, bc);

    auto store_tile = [&](int buf, const Stage& sg) {
      char* st = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        uint2 hi, lo;
        split4(sg.a[i], hi, lo);
        *reinterpret_cast<uint2*>(st + a_st[i]) = hi;
        if (NPASS == 3) *reinterpret_cast<uint2*>(st + A_BYTES + a_st[i]) = lo;
      }
      char* sb = st + NPL * A_BYTES;
#pragma unroll
      for (int i = 0; i < BCH; ++i) {
        if (!BFULL && (t >> 2) + BR * i >= BN) continue;
        *reinterpret_cast<uint4*>(sb + b_st[i]) = sg.bh[i];
        if (NPASS == 3) *reinterpret_cast<uint4*>(sb + B_BYTES + b_st[i]) = sg.bl[i];
      }
    };

    Frag f0, f1;
    constexpr int TT = TM * TN, TH = TT / 2;          // all blocks / the part issued in front of the barrier

    // prologue: tile 0 -> LDS[0]; tile 1 in flight in s0.  The loop consumes k-tiles in pairs; an odd last tile is
    // computed after it (it already sits in LDS[0] with its ks=0 fragments in f0).
    if (ABL & 32) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) s0.a[i] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
      for (int i = 0; i < BCH; ++i) s0.bh[i] = s0.bl[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
      store_tile(0, s0);
    } else {
      load_tile(0, s0);
      store_tile(0, s0);
      if (DEEP) load_tile(1, s0);
    }
    __syncthreads();
    DBG_T(1);
    ldfrag(0, 0, f0);
    if (ABL & 8) ldfrag(0, 1, f1);
    if (ABL & 1) s1 = s0;
    const int nk2 = nk & ~1;
#pragma unroll 1
    for (int kt = 0; kt < nk2; kt += 2) {
      // even step: MFMAs on LDS[0]; DEEP: s0 holds tile kt+1 and tile kt+2 starts loading into s1; else tile kt+1 loads into s0 now
      if (!(ABL & 1)) { if (DEEP) load_tile(kt + 2, s1); else load_tile(kt + 1, s0); }
      if (!(ABL & 8)) ldfrag(0, 1, f1);
      mma(f0, 0, TT);
      if (!(ABL & 2)) store_tile(1, s0);
      __builtin_amdgcn_sched_barrier(0);          // LDS stores retire under the next MFMAs, not in front of the barrier
      mma(f1, 0, TH);
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 4)) __syncthreads();
      if (!(ABL & 8)) ldfrag(1, 0, f0);
      __builtin_amdgcn_sched_barrier(0);
      mma(f1, TH, TT);
      // odd step: MFMAs on LDS[1]
      if (!(ABL & 1)) { if (DEEP) load_tile(kt + 3, s0); else load_tile(kt + 2, s0); }
      if (!(ABL & 8)) ldfrag(1, 1, f1);
      mma(f0, 0, TT);
      if (!(ABL & 2)) { if (DEEP) store_tile(0, s1); else store_tile(0, s0); }
      __builtin_amdgcn_sched_barrier(0);
      mma(f1, 0, TH);
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 4)) __syncthreads();
      if (!(ABL & 8)) ldfrag(0, 0, f0);
      __builtin_amdgcn_sched_barrier(0);
      mma(f1, TH, TT);
    }
    if (nk & 1) {
      ldfrag(0, 1, f1);
      mma(f0, 0, TT);
      mma(f1, 0, TT);
    }

  }

  DBG_T(2);
  DBG_WSTORE();
  // ---- epilogue.  The C fragment gives a lane one column and 16 scattered rows (dword stores, 2 x 128 B per wave
  // instruction); instead each wave transposes its 64 x (BN/2) sub-tile through its own slice of the (now idle) LDS and
  // writes whole rows with 16-byte stores: 4x fewer store / residual-load instructions, 512 B..1 KiB contiguous each.
  const bool vec_ok = ((d.N | d.ld_out) & 3) == 0 && (reinterpret_cast<uintptr_t>(d.out) & 15) == 0 &&
                      (!d.residual || ((d.ld_res & 3) == 0 && (reinterpret_cast<uintptr_t>(d.residual) & 15) == 0)) &&
                      (!d.rowbias || ((d.ld_rowbias & 3) == 0 && (reinterpret_cast<uintptr_t>(d.rowbias) & 15) == 0)) &&
                      (!d.bias || (reinterpret_cast<uintptr_t>(d.bias) & 15) == 0);
  if constexpr (BM == 256) {
    // tall tile: the wave sub-tile (64 x 160 fp32 = 40 KB) does not fit an eighth of the LDS, so it is transposed and stored in
    // TM passes of 32 rows through a 20 KB wave-private slice (in-order LDS access within the wave: no barrier between passes)
    if (vec_ok) {
      constexpr int LPR = WTN / 4;
      constexpr int NIT = 32 * LPR / 64;
      static_assert((32 * LPR) % 64 == 0, "32-row pass must be a whole number of wave instructions");
      __syncthreads();
      float* cw = reinterpret_cast<float*>(smem) + wid * (32 * WTN);
      float* pp = part ? part + (int64_t)blockIdx.y * d.M * d.N : nullptr;
      using Epi = EpiRows<NIT, LPR, WTN, 5>;
      Epi ep;
      const int colb = n0 + wn * WTN;
      static_assert(TM == 2, "the tall tile's output stage is written for two 32-row passes per wave");
      if constexpr (APL == 6) {
        // GEGLU on the tall tile (round 5; its own instantiation, APL = 6, so that the plain tall tile's code does not change by an instruction).  The weight columns of tile t come as [a (160) | gate (160)] of output columns 160 t .. 160 t + 159
        // (ddpo_gemm_desc.epilogue == 2), so the wave pair (wm, 0) / (wm, 1) holds the value and the gate accumulators of the SAME 64 x 160
        // outputs.  Per 32-row pass: both waves add their bias IN REGISTERS (a lane's column is fixed per accumulator block) and transpose
        // the pre-activations into their LDS slices (adjacent: wid = 2 wm + wn); one barrier; then each wave of the pair takes 16 of the 32
        // rows (40 float4 per row, 10 wave instructions): gelu_tanh of the gate, the product, the stores — and, for the training forward,
        // the pre-activation rows in the original [a | gate] column order (aux_out) — so the gelus and the stores are spread over all eight
        // waves.  Same arithmetic per element as the 128 x 128 GEGLU tile — (acc_a + b_a) * gelu_tanh(acc_g + b_g) on the same accumulation
        // order — so the two tiles agree bit for bit (tests/test_gpu_bf16.py::test_linear_geglu_tall_tile_is_bit_identical).
        static_assert(WN == 2 && WTN == 160, "value / gate wave pairs of 160 columns");
        const int oc0 = (n0 / 320) * 160;                  // first OUTPUT column of this tile
        float bj[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bj[j] = d.bias ? d.bias[colb + j * 32 + (lane & 31)] : 0.f;
        const float* ca = reinterpret_cast<const float*>(smem) + (wid & ~1) * (32 * WTN);
        const float* cg = ca + 32 * WTN;
        auto gpass = [&](auto IH) {
          constexpr int ih = decltype(IH)::value;
          if (ih > 0) __syncthreads();                     // every wave has read the slices of the previous pass
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              cw[((r & 3) + 8 * (r >> 2) + 4 * khalf) * WTN + j * 32 + (lane & 31)] = acc[ih][j][r] + bj[j];
            }
          __syncthreads();
          constexpr int GL = WTN / 4;                      // float4 per output row
          int ln = lane;
          asm volatile("" : "+v"(ln));                     // (as in EpiRows: keeps the index arithmetic of the iterations from being hoisted and spilled)
#pragma unroll
          for (int it = 0; it < 16 * GL / 64; ++it) {
            const int e = it * 64 + ln, rr = wn * 16 + e / GL, c4 = (e - (e / GL) * GL) * 4;
            const float4 a = *reinterpret_cast<const float4*>(ca + rr * WTN + c4);
            const float4 g = *reinterpret_cast<const float4*>(cg + rr * WTN + c4);
            const float4 o = make_float4(a.x * gelu_tanh_f(g.x), a.y * gelu_tanh_f(g.y), a.z * gelu_tanh_f(g.z), a.w * gelu_tanh_f(g.w));
            const int row = m0 + wm * WTM + ih * 32 + rr, col = oc0 + c4;
            if (row >= d.M) continue;
            if (d.aux_out) {                               // pre-activation in the original [a | gate] column order (training forward)
              float* pa = d.aux_out + (int64_t)row * d.N + col;
              *reinterpret_cast<float4*>(pa) = a;
              *reinterpret_cast<float4*>(pa + (d.N >> 1)) = g;
            }
            if (d.out) st_out4(d.out + (int64_t)row * d.ld_out + col, o);
            if (d.out_hi) store_planes4(d, row, col, o);
          }
        };
        gpass(std::integral_constant<int, 0>{});
        gpass(std::integral_constant<int, 1>{});
        DBG_T(3);
        return;
      }
      // (the passes are spelled out: left as a loop the optimizer declined to unroll it, and the dynamically indexed accumulators went to scratch)
      auto pass = [&](auto IH) {
        constexpr int ih = decltype(IH)::value;
        const int rowb = m0 + wm * WTM + ih * 32;
        // the first pass stages BEFORE it requests anything (all 160 accumulators are still live: no room for the operands); the second
        // requests first, into the registers the first freed, and stages under that latency
        if (ih > 0 && !pp) ep.fetch(d, 0, rowb, colb, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            cw[((r & 3) + 8 * (r >> 2) + 4 * khalf) * WTN + j * 32 + (lane & 31)] = acc[ih][j][r];
        if (pp) {
          Epi::partials(d, cw, pp, rowb, colb, lane);
        } else {
          if (ih == 0) { ep.init(d, colb, lane); ep.fetch(d, 0, rowb, colb, lane); }
          ep.combine_all(d, cw, rowb, colb, lane);
          Epi::emit(d, cw, rowb, colb, lane);
        }
      };
      pass(std::integral_constant<int, 0>{});
      pass(std::integral_constant<int, 1>{});
      DBG_T(3);
      return;
    }
  }
  if (vec_ok && !(ABL & 16) && BM != 256) {
    constexpr int LPR = WTN / 4;                 // float4 per row of the wave sub-tile (WTM rows x WTN columns)
    constexpr int NIT = WTM * LPR / 64;          // wave instructions to move it
    static_assert((WTM * LPR) % 64 == 0, "wave sub-tile must be a whole number of 1 KiB rows");
    __syncthreads();                             // all waves are done with the operand tiles
    float* cw = reinterpret_cast<float*>(smem) + wid * (WTM * WTN);
    using Epi = EpiRows<NIT, LPR, WTN>;
    Epi ep;
    const int rowb = m0 + wm * WTM, colb = n0 + wn * WTN;
    const bool plain = !part && !(BN == 128 && WN == 2 && WM == 2 && d.epilogue == 1);
    if (plain) { ep.init(d, colb, lane); ep.fetch(d, 0, rowb, colb, lane); }           // in flight under the transposition below
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          cw[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf) * WTN + j * 32 + (lane & 31)] = acc[i][j][r];
    // (wave-private slice: program order + lgkmcnt is all the synchronisation needed)
    if (BN == 128 && WN == 2 && WM == 2 && d.epilogue == 1) {          // GEGLU: this wave's 64 columns are [a (32) | gate (32)] of output chunk q
      const int q = (n0 + wn * 64) >> 6;
      const int gc = (lane & 7) * 4, grow = lane >> 3;          // 8 lanes x float4 = 32 output columns; 8 rows per instruction
      const float4 ba = d.bias ? *reinterpret_cast<const float4*>(d.bias + q * 64 + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 bg = d.bias ? *reinterpret_cast<const float4*>(d.bias + q * 64 + 32 + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int rr = it * 8 + grow;
        const int row = m0 + wm * 64 + rr;
        if (row >= d.M) continue;
        const float4 a = *reinterpret_cast<const float4*>(cw + rr * 64 + gc);
        const float4 g = *reinterpret_cast<const float4*>(cw + rr * 64 + 32 + gc);
        float4 o;
        o.x = (a.x + ba.x) * gelu_tanh_f(g.x + bg.x); o.y = (a.y + ba.y) * gelu_tanh_f(g.y + bg.y);
        o.z = (a.z + ba.z) * gelu_tanh_f(g.z + bg.z); o.w = (a.w + ba.w) * gelu_tanh_f(g.w + bg.w);
        if (d.aux_out) {                          // pre-activation in the original [a | gate] column order (training forward)
          float* pa = d.aux_out + (int64_t)row * d.N + q * 32 + gc;
          *reinterpret_cast<float4*>(pa) = make_float4(a.x + ba.x, a.y + ba.y, a.z + ba.z, a.w + ba.w);
          *reinterpret_cast<float4*>(pa + (d.N >> 1)) = make_float4(g.x + bg.x, g.y + bg.y, g.z + bg.z, g.w + bg.w);
        }
        if (d.out) st_out4(d.out + (int64_t)row * d.ld_out + q * 32 + gc, o);
        if (d.out_hi) store_planes4(d, row, q * 32 + gc, o);
      }
      DBG_T(3);
      return;
    }
    if (part) {
      Epi::partials(d, cw, part + (int64_t)blockIdx.y * d.M * d.N, rowb, colb, lane);
    } else {
      ep.combine_all(d, cw, rowb, colb, lane);
      Epi::emit(d, cw, rowb, colb, lane);
    }
    DBG_T(3);
    return;
  }
  if (part) {
    float* pp = part + (int64_t)blockIdx.y * d.M * d.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + (lane & 31);
        if (col >= d.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          if (row < d.M) pp[(int64_t)row * d.N + col] = acc[i][j][r];
        }
      }
    DBG_T(3);
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + (lane & 31);
      if (col >= d.N) continue;
      const float bv = d.bias ? d.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row >= d.M) continue;
        if ((ABL & 16) && (r | i | j)) { if (acc[i][j][r] == 123.456f) d.out[0] = 1.f; continue; }
        float v = d.alpha * acc[i][j][r] + bv;
        if (d.rowbias) v += d.rowbias[(int64_t)(row / d.rows_per_batch) * d.ld_rowbias + col];
        if (d.residual) v += d.residual[(int64_t)row * d.ld_res + col];
        d.out[(int64_t)row * d.ld_out + col] = v;
      }
    }
  }
  DBG_T(3);
}

// fixed-order reduction of the split-K partials + the fused epilogue (bit-reproducible: no atomics)
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const ddpo_gemm_desc d, const float* __restrict__ part, int splits) {
  const int n4 = d.N >> 2;
  const int64_t total = (int64_t)d.M * n4, mn = (int64_t)d.M * d.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / n4), col = (int)(i - (int64_t)row * n4) << 2;
    float4 a = *reinterpret_cast<const float4*>(part + (int64_t)row * d.N + col);
    for (int s = 1; s < splits; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(part + s * mn + (int64_t)row * d.N + col);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = d.alpha * v[e] + (d.bias ? d.bias[col + e] : 0.f);
      if (d.rowbias) x += d.rowbias[(int64_t)(row / d.rows_per_batch) * d.ld_rowbias + col + e];
      if (d.residual) x += d.residual[(int64_t)row * d.ld_res + col + e];
      v[e] = x;
      if (d.out) d.out[(int64_t)row * d.ld_out + col + e] = x;
    }
    if (d.out_hi) store_planes4(d, row, col, make_float4(v[0], v[1], v[2], v[3]));
  }
}

// plane-emitting output stage: only the vector output stage of the buffer-addressed kernels (and the split-K reduce) writes planes
static bool planes_out_ok(const ddpo_gemm_desc& d) {
  if (!d.out_hi) return d.out != nullptr && !d.out_lo;
  if (!d.out_lo || d.ld_planes < 0 || (d.ld_planes & 3)) return false;
  if ((reinterpret_cast<uintptr_t>(d.out_hi) | reinterpret_cast<uintptr_t>(d.out_lo)) & 7) return false;
  const int ncols = d.epilogue != 0 ? d.N / 2 : d.N;
  if (d.ld_planes == 0 ? (ncols & 31) != 0 : d.ld_planes < ncols) return false;      // 0: k-blocked (ncols / 32, M, 32)
  if (d.planes_fmt != 0 && (d.planes_fmt != 1 || (ncols & 31) || (d.ld_planes & 31))) return false;      // f16mx planes: whole 32-column blocks
  if (d.N & 3) return false;
  if (d.out && ((d.ld_out & 3) || (reinterpret_cast<uintptr_t>(d.out) & 15))) return false;
  if (d.residual && ((d.ld_res & 3) || (reinterpret_cast<uintptr_t>(d.residual) & 15))) return false;
  if (d.rowbias && ((d.ld_rowbias & 3) || (reinterpret_cast<uintptr_t>(d.rowbias) & 15))) return false;
  if (d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15)) return false;
  return true;
}

// Host-side launch counters per tile class (ABI v10, ddpo_gemm_tile_launch_counts): which instantiation a layer geometry was routed to is
// otherwise invisible to a caller — tests/test_gpu_headline_geometry.py asserts that the bench geometry really ran the tall tile.
enum { TC_TALL = 0, TC_WIDE = 1, TC_128 = 2, TC_64 = 3, TC_GENERIC = 4, TC_SPLITK_REDUCE = 5, TC_MX = 6, TC_COUNT = 8 };
static unsigned long long g_tile_launches[TC_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" int ddpo_gemm_tile_launch_counts(unsigned long long* out_host, int n) {
  if (!out_host || n < TC_COUNT) return DDPO_EINVAL;
  for (int i = 0; i < TC_COUNT; ++i) out_host[i] = __atomic_load_n(&g_tile_launches[i], __ATOMIC_RELAXED);
  return DDPO_OK;
}
static inline void count_tile(int cls) { __atomic_fetch_add(&g_tile_launches[cls], 1ull, __ATOMIC_RELAXED); }

// buffer-addressed fast path: k-tiles never straddle a tap and every byte offset fits the 31-bit buffer range
static bool g_force_generic = false;          // only ever set through the debug hook below (probe builds)
static bool buf_path_ok(const ddpo_gemm_desc& d, int ldw) {
  if (g_force_generic) return false;
  const int64_t lim = 0x7FFFFFFF;
  if (d.w_layout == 1) ldw = (d.K + 31) / 32 * 32;          // k-blocked planes: Kb * N * 32 elements
  if (d.ksize > 0) {
    if (d.Cin % BF_BK) return false;
    if ((int64_t)d.B * d.H * d.W * (d.ld_src ? d.ld_src : d.Cin) * 4 >= lim) return false;       // ld_src == 0: k-blocked activation planes
    const int64_t wbytes = d.w_dgrad ? (int64_t)d.K * d.N * 2 : (int64_t)d.N * ldw * 2;
    if (wbytes >= lim) return false;
  } else {
    if (d.K % BF_BK) return false;
    if ((int64_t)d.M * (d.ld_src ? d.ld_src : d.K) * 4 >= lim || (int64_t)d.N * ldw * 2 >= lim) return false;
  }
  return true;
}
#ifdef DDPO_DEBUG_HOOKS      // tools/native builds only: not part of libddpo_hip.so's ABI
extern "C" void ddpo_debug_force_generic_gemm(int on) { g_force_generic = on != 0; }
#endif

template <int BM, int BN, int NPASS, int WM = 2, int WN = 2, int APL = 0>
static int launch_bf16(const ddpo_gemm_desc& d, const uint16_t* w_hi, const uint16_t* w_lo, int ldw, float* ws, size_t ws_bytes,
                       hipStream_t st) {
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  // split-K when the tile grid under-fills the 256 CUs and the reduction is long (8x8 / 16x16 latent levels)
  const int nk_total = (d.K + BF_BK - 1) / BF_BK;
  int splits = 1;
  if (ws && nblk < 192 && nk_total >= 32 && (d.N & 3) == 0 && d.epilogue == 0) {
    splits = (384 + nblk - 1) / nblk;
    if (splits > 8) splits = 8;
    if (splits > nk_total / 8) splits = nk_total / 8;
    while (splits > 1 && (size_t)splits * d.M * d.N * sizeof(float) > ws_bytes) --splits;
  }
  int ktps = (nk_total + splits - 1) / splits;
  ktps = (ktps + 1) & ~1;                                  // the pipelined loop consumes k-tiles in pairs
  splits = (nk_total + ktps - 1) / ktps;
  float* part = splits > 1 ? ws : nullptr;
  constexpr int NPL = (NPASS == 3 || NPASS == 4) ? 2 : 1;
  size_t lds = (APL == 3) ? NPL * (size_t)(2 * BM + 3 * BN) * 64 : 2 * NPL * (size_t)(BM + BN) * 64;     // APL 3: three weight stages
  if (lds < (size_t)BM * BN * 4) lds = (size_t)BM * BN * 4;     // the epilogue transposes the C tile through LDS
  static bool attr_set = false;
  if (!attr_set) {
    if constexpr (APL == 0) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_bf16_kernel<BM, BN, NPASS, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_bf16_kernel<BM, BN, NPASS, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_bf16_buf_kernel<BM, BN, NPASS, 0, WM, WN, true, APL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  count_tile(BN == 128 ? TC_128 : TC_64);
  if constexpr (NPASS == 4) count_tile(TC_MX);
  if constexpr (APL != 0) {                // the plane-fed entry points have already checked buf_path_ok
    hipLaunchKernelGGL((gemm_conv_bf16_buf_kernel<BM, BN, NPASS, 0, WM, WN, true, APL>), dim3(nblk, splits), dim3(64 * WM * WN), lds, st, d, w_hi,
                       w_lo, ldw, tiles_m, tiles_n, nblk, ktps, part);
  } else {
    if (!buf_path_ok(d, ldw)) count_tile(TC_GENERIC);
    if (buf_path_ok(d, ldw))
      hipLaunchKernelGGL((gemm_conv_bf16_buf_kernel<BM, BN, NPASS, 0, WM, WN, true, APL>), dim3(nblk, splits), dim3(64 * WM * WN), lds, st, d, w_hi,
                         w_lo, ldw, tiles_m, tiles_n, nblk, ktps, part);
    else if (d.upsample == 0)
      hipLaunchKernelGGL((gemm_conv_bf16_kernel<BM, BN, NPASS, true>), dim3(nblk, splits), dim3(BF_THREADS), lds, st, d, w_hi, w_lo, ldw,
                         tiles_m, tiles_n, nblk, ktps, part);
    else
      hipLaunchKernelGGL((gemm_conv_bf16_kernel<BM, BN, NPASS, false>), dim3(nblk, splits), dim3(BF_THREADS), lds, st, d, w_hi, w_lo, ldw,
                         tiles_m, tiles_n, nblk, ktps, part);
  }
  DDPO_LAUNCH_CHECK();
  if (splits > 1) {
    int64_t blocks = ((int64_t)d.M * (d.N >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    count_tile(TC_SPLITK_REDUCE);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, d, part, splits);
    DDPO_LAUNCH_CHECK();
  }
  return DDPO_OK;
}


// 128x320 tiles, 8 waves (4 x 2), one workgroup per CU: buffer-addressed kernel only (caller checked buf_path_ok).
// Tile counts of the U-Net layers are multiples of the 256 CUs at the 64x64 and 32x32 levels; below that the reduction is
// split so that ~256 workgroups exist.
static int wide_splits(const ddpo_gemm_desc& d, bool have_ws, size_t ws_bytes) {
  const int nblk = ((d.M + 127) / 128) * ((d.N + 319) / 320);
  const int nk_total = d.K / BF_BK;
  int splits = 1;
  if (have_ws && nblk <= 192 && nk_total >= 16) {
    splits = (256 + nblk / 2) / nblk;
    if (splits > 8) splits = 8;
    if (splits > nk_total / 8) splits = nk_total / 8;
    if (splits < 1) splits = 1;
    while (splits > 1 && (size_t)splits * d.M * d.N * sizeof(float) > ws_bytes) --splits;
  }
  return splits;
}

template <int NPASS, int APL = 0>
static int launch_bf16_wide(const ddpo_gemm_desc& d, const uint16_t* w_hi, const uint16_t* w_lo, int ldw, float* ws, size_t ws_bytes,
                            hipStream_t st) {
  constexpr int BM = 128, BN = 320, WM = 4, WN = 2;      // 8 waves of 32x160, two per SIMD
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  const int nk_total = d.K / BF_BK;
  int splits = wide_splits(d, ws != nullptr, ws_bytes);
  int ktps = (nk_total + splits - 1) / splits;
  ktps = (ktps + 1) & ~1;
  splits = (nk_total + ktps - 1) / ktps;
  float* part = splits > 1 ? ws : nullptr;
  const size_t lds = (size_t)BM * BN * 4;                  // epilogue image (160 KB) > 2 stages of operand tiles (112 KB)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_bf16_buf_kernel<BM, BN, NPASS, 0, WM, WN, true, APL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  count_tile(TC_WIDE);
  if constexpr (NPASS == 4) count_tile(TC_MX);
  hipLaunchKernelGGL((gemm_conv_bf16_buf_kernel<BM, BN, NPASS, 0, WM, WN, true, APL>), dim3(nblk, splits), dim3(64 * WM * WN), lds, st, d, w_hi,
                     w_lo, ldw, tiles_m, tiles_n, nblk, ktps, part);
  DDPO_LAUNCH_CHECK();
  if (splits > 1) {
    int64_t blocks = ((int64_t)d.M * (d.N >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    count_tile(TC_SPLITK_REDUCE);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, d, part, splits);
    DDPO_LAUNCH_CHECK();
  }
  return DDPO_OK;
}

// Tall 256x320 tiles (plane-fed path only, 8 waves of 64x160, one workgroup per CU, plain k-loop APL = 4): for layers whose tile
// grid still covers the chip — the 64x64-latent level of the U-Net (M = 65536: 256 tiles per 320 columns).  No split-K.
template <int APL, int NPASS = 3>
static int launch_bf16_tall(const ddpo_gemm_desc& d, const uint16_t* w_hi, const uint16_t* w_lo, int ldw, hipStream_t st) {
  constexpr int BM = 256, BN = 320, WM = 4, WN = 2;
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  const int nk_total = d.K / BF_BK;
  const size_t lds = 8 * 32 * 160 * 4;                     // epilogue slices (160 KB) > 2 stages of operand tiles (144 KB)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_bf16_buf_kernel<BM, BN, NPASS, 0, WM, WN, true, APL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  count_tile(TC_TALL);
  hipLaunchKernelGGL((gemm_conv_bf16_buf_kernel<BM, BN, NPASS, 0, WM, WN, true, APL>), dim3(nblk, 1), dim3(64 * WM * WN), lds, st, d, w_hi, w_lo, ldw,
                     tiles_m, tiles_n, nblk, nk_total, (float*)nullptr);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// f16mx on the tall tile: eight waves of 64x160 (APL = 7), 144 KB of operand stages, no split-K (launched only where the grid fills the chip).
static int launch_mx_tall(const ddpo_gemm_desc& d, const uint16_t* w16, const uint16_t* w8, hipStream_t st) {
  constexpr int BM = 256, BN = 320, WM = 4, WN = 2;
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  const int nk_total = d.K / BF_BK;
  const size_t lds = 8 * 32 * 160 * 4;                      // the output stage's slices (160 KB) > two stages of [A16 | A8 | W16 | W8] (144 KB)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv_bf16_buf_kernel<BM, BN, 4, 0, WM, WN, true, 7>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  count_tile(TC_TALL);
  count_tile(TC_MX);
  hipLaunchKernelGGL((gemm_conv_bf16_buf_kernel<BM, BN, 4, 0, WM, WN, true, 7>), dim3(nblk, 1), dim3(64 * WM * WN), lds, st, d, w16, w8, 0,
                     tiles_m, tiles_n, nblk, nk_total, (float*)nullptr);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// Tile-shape / split-K selection shared by the fp32-fed and the plane-fed entry points: the SAME rules, so both produce
// bit-identical results for the same layer (APL is only instantiated for npass == 3).
template <int APL>
static int dispatch_bf16(const ddpo_gemm_desc& d, const uint16_t* w_hi, const uint16_t* w_lo, int ldw, int npass, void* ws, size_t ws_bytes,
                         hipStream_t st) {
  if (d.epilogue != 0) {       // GEGLU output stage: 128-wide tiles of the buffer-addressed kernel, vector epilogue only
    if ((d.epilogue != 1 && d.epilogue != 2) || (d.N & 127) || !buf_path_ok(d, ldw) || d.rowbias || d.residual || d.alpha != 1.0f || d.w_dgrad) return DDPO_EINVAL;
    if ((d.ld_out & 3) || (reinterpret_cast<uintptr_t>(d.out) & 15) || (d.bias && (reinterpret_cast<uintptr_t>(d.bias) & 15))) return DDPO_EINVAL;
    if (reinterpret_cast<uintptr_t>(d.aux_out) & 15) return DDPO_EINVAL;
    if (d.epilogue == 2) {     // the tall 256 x 320 tile with value / gate wave pairs: plane-fed bf16x3 only, columns in [a (160) | gate (160)] blocks
      if constexpr (APL == 3) {
        if (npass != 3 || d.N % 320 || ((d.N >> 1) & 3)) return DDPO_EINVAL;
        return launch_bf16_tall<6>(d, w_hi, w_lo, ldw, st);
      }
      return DDPO_EINVAL;
    }
    if constexpr (APL == 3) {
      if (npass == 4) return launch_bf16<128, 128, 4, 2, 2, 3>(d, w_hi, w_lo, ldw, nullptr, 0, st);
      if (npass == 1) return launch_bf16<128, 128, 1, 2, 2, 3>(d, w_hi, w_lo, ldw, nullptr, 0, st);
      if (npass == 5) return launch_bf16<128, 128, 5, 2, 2, 3>(d, w_hi, w_lo, ldw, nullptr, 0, st);
    }
    return npass == 3 ? launch_bf16<128, 128, 3, 2, 2, APL>(d, w_hi, w_lo, ldw, nullptr, 0, st) : launch_bf16<128, 128, 1>(d, w_hi, w_lo, ldw, nullptr, 0, st);
  }
  float* wsf = (ws && !(reinterpret_cast<uintptr_t>(ws) & 15)) ? reinterpret_cast<float*>(ws) : nullptr;
  constexpr int wide_mode = 1;
  // 128x320 tiles (one workgroup per CU) when they, times the split of the reduction, give every CU a workgroup; a
  // many-column GEMM with a very short reduction is better on 128x128 (measured: K=320, N=2560; the 160 KB epilogue image)
  if constexpr (APL != 0) {
    // 256x320 tiles where they still give every CU a workgroup (the 64x64-latent level): 345 / 428 TF against 320 / 390 for the
    // 128x320 tile on conv 320->320 / 960->320 (bit-identical; round-2 probe).
    constexpr int tall_mode = 1;
    // ... and where their last round of 256 is not much emptier than the 128x320 grid's: 256 tall tiles (SD-1.5, 64x64 latents at batch 16)
    // are exactly one round; 576 (SD-2.1, 96x96) are 2.25 rounds = 3 rounds of time, where 1152 wide tiles waste half a round of five
    const long ntall = (long)((d.M + 255) / 256) * (d.N / 320), nwide = (long)((d.M + 127) / 128) * (d.N / 320);
    const double eff_tall = (double)ntall / (double)(((ntall + 255) / 256) * 256), eff_wide = (double)nwide / (double)(((nwide + 255) / 256) * 256);
    // (bf16x3 only.  An f16mx tall loop — the weight operand's ks = 1 fragments streamed per column block beside the 160 accumulators — was
    // built twice in round 3: correct, but ~20 registers short at two waves per SIMD.  The compiler spills 44-58 values around the 8-register
    // operands of the scaled MFMA (0 without the 8-bit weight fragments, 16 with one 8-bit activation fragment, the same 58 with the convolution
    // bookkeeping compiled out — it is the fragments, not the addressing), and the reloads sit behind the LDS-DMA requests on the in-order
    // vmcnt counter, so every k-tile waits for its own prefetch: 0.326 ms against 0.321 (bf16x3 tall) / 0.317 (f16mx 128x320) on conv 320->320
    // @64^2, profiles/r03_probe_mx_tall.log.  The f16mx tall tile wants FOUR waves of 128 x 160 (one per SIMD, accumulators in AGPRs).)
    if (tall_mode && npass == 3 && d.N % 320 == 0 && d.epilogue == 0 && ntall >= 200 && eff_tall * 1.08 >= eff_wide)
      return launch_bf16_tall<5>(d, w_hi, w_lo, ldw, st);
    if constexpr (APL == 3) {
      // single-pass bf16, plane-fed (round 6): the tall tile with the four-stage ring (APL = 8) under the same rule
      if (npass == 1 && d.N % 320 == 0 && d.epilogue == 0 && ntall >= 200 && eff_tall * 1.08 >= eff_wide)
        return launch_bf16_tall<8, 1>(d, w_hi, w_lo, ldw, st);
      if (npass == 5 && d.N % 320 == 0 && d.epilogue == 0 && ntall >= 200 && eff_tall * 1.08 >= eff_wide)
        return launch_bf16_tall<8, 5>(d, w_hi, w_lo, ldw, st);
      // f16mx layers on the tall tile (APL = 7) under the bf16x3 tall tile's rule: grids that fill whole rounds of the chip unsplit — the 64x64
      // level at batch 16 and the up-sampled 32x32 -> 64x64 convolution.  Measured round 5 (profiles/r05_probe_mx_tall.log, r05_ab_mx_tall.log):
      // conv 320->320 @64^2 0.259 -> 0.236 ms, 960->320 0.930 -> 0.709 ms, up-conv 640->640 1.084 -> 0.878 ms, bit-identical; sampling +1.4 %.
      // A split-K grid of tall tiles for the 32x32 / 16x16 levels (128 / 64 tiles x 2 / 4 splits) was built and measured with it: no gain over
      // the 128x320 tile there (4.09 vs 4.09 images/s) — deleted.  DDPO_MX_TALL=0 (read per launch) keeps every f16mx layer on the 128-row tiles:
      // the probe's and the tests' bit-identity comparison.
      const char* mx_tall_env = getenv("DDPO_MX_TALL");
      if (npass == 4 && !(mx_tall_env && mx_tall_env[0] == '0') && d.N % 320 == 0 && d.epilogue == 0 && ntall >= 200 && eff_tall * 1.08 >= eff_wide)
        return launch_mx_tall(d, w_hi, w_lo, st);
    }
  }
  const int wsplits = wide_splits(d, wsf != nullptr, ws_bytes);
  if (wide_mode && d.N % 320 == 0 && d.M >= 512 && buf_path_ok(d, ldw) && !(d.K / BF_BK < 16 && d.N > 1280) &&
      (long)((d.M + 127) / 128) * (d.N / 320) * wsplits >= 200 &&
      !(wsplits > 1 && d.K / BF_BK < 64)) {     // a split short reduction only adds the reduce pass (measured equal to 128x128 unsplit)
    if constexpr (APL == 3) {
      if (npass == 4) return launch_bf16_wide<4, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
      if (npass == 1) return launch_bf16_wide<1, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
      if (npass == 5) return launch_bf16_wide<5, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
    }
    return npass == 3 ? launch_bf16_wide<3, APL>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st) : launch_bf16_wide<1>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
  }
  const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
  constexpr long big_min = 256;
  // 128x128 tiles need >= 512 of them to fill both workgroup slots of every CU; a short reduction on 256..511 of them (the 16x16
  // level's q / k / v / out projections: M = 4096, N = K = 1280 -> 320 tiles) runs ~15 % faster on 640 tiles of 128x64, three per CU
  // (probe, cold weights: 0.084 -> 0.070 ms fp32-fed; profiles/r02_probe_tiles_small.log).
  constexpr int mid64 = 1;
  const bool mid_short = mid64 && t128 < 512 && d.K / BF_BK <= 64;
  const bool big = (d.N % 128 == 0) && t128 >= big_min && !mid_short;
  if constexpr (APL == 3) {
    if (npass == 4)
      return big ? launch_bf16<128, 128, 4, 2, 2, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st) : launch_bf16<128, 64, 4, 2, 2, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
    if (npass == 1)
      return big ? launch_bf16<128, 128, 1, 2, 2, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st) : launch_bf16<128, 64, 1, 2, 2, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
    if (npass == 5)
      return big ? launch_bf16<128, 128, 5, 2, 2, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st) : launch_bf16<128, 64, 5, 2, 2, 3>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
  }
  if (npass == 3)
    return big ? launch_bf16<128, 128, 3, 2, 2, APL>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st) : launch_bf16<128, 64, 3, 2, 2, APL>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
  return big ? launch_bf16<128, 128, 1>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st) : launch_bf16<128, 64, 1>(d, w_hi, w_lo, ldw, wsf, ws_bytes, st);
}

extern "C" int ddpo_gemm_conv_fwd_bf16(const ddpo_gemm_desc* dp, const uint16_t* w_hi, const uint16_t* w_lo, int ldw, int npass,
                                       void* ws, size_t ws_bytes, void* stream) {
  if (!dp || !w_hi) return DDPO_EINVAL;
  const ddpo_gemm_desc& d = *dp;
  if (npass != 1 && npass != 3) return DDPO_EINVAL;
  if (npass == 3 && !w_lo) return DDPO_EINVAL;
  if (!d.src || d.M <= 0 || d.N <= 0 || d.K <= 0 || d.ld_src <= 0 || (d.ld_src & 3) || (reinterpret_cast<uintptr_t>(d.src) & 15)) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(w_hi) & 15) || (w_lo && (reinterpret_cast<uintptr_t>(w_lo) & 15))) return DDPO_EINVAL;
  if (!planes_out_ok(d) || (d.out_hi && !buf_path_ok(d, ldw))) return DDPO_EINVAL;
  if (d.ksize > 0) {
    if (d.ksize != 1 && d.ksize != 3) return DDPO_EINVAL;
    if ((d.Cin & 7) || d.K != d.ksize * d.ksize * d.Cin || d.M != d.B * d.OH * d.OW) return DDPO_EINVAL;
    if (d.upsample < 0 || d.upsample > 2) return DDPO_EINVAL;
  } else if (d.K & 7) {
    return DDPO_EINVAL;
  }
  if (d.w_layout != 0 && (d.w_layout != 1 || d.w_dgrad)) return DDPO_EINVAL;
  if (d.w_dgrad) {
    if (d.ksize <= 0 || (d.Cin & 7)) return DDPO_EINVAL;       // W planes in forward [K][N] order; co chunks of 8 stay inside a tap
  } else if (d.w_layout == 0 && (ldw < d.K || (ldw & 7))) {
    return DDPO_EINVAL;
  }
  return dispatch_bf16<0>(d, w_hi, w_lo, ldw, npass, ws, ws_bytes, as_stream(stream));
}

/* Plane-fed variant: the activation operand comes as bf16 hi / lo planes (see the APL note on gemm_conv_bf16_buf_kernel). */
extern "C" int ddpo_gemm_conv_fwd_bf16_planes(const ddpo_gemm_desc* dp, const uint16_t* a_hi, const uint16_t* a_lo, int lda,
                                              const uint16_t* w_hi, const uint16_t* w_lo, int ldw, void* ws, size_t ws_bytes,
                                              void* stream) {
  if (!dp || !a_hi || !w_hi || (a_lo == nullptr) != (w_lo == nullptr)) return DDPO_EINVAL;
  const int npass = a_lo ? 3 : 1;          // ABI v14: BOTH lo planes NULL = single-pass bf16 (a_hi * w_hi only: XLA's TPU default precision)
  ddpo_gemm_desc d = *dp;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || lda < 0 || (lda & 7) || d.w_dgrad || !planes_out_ok(d)) return DDPO_EINVAL;      // lda == 0: k-blocked planes
  if (npass == 1 && d.epilogue == 2) return DDPO_EINVAL;      // the tall GEGLU tile exists on the three-pass datapath only
  if ((reinterpret_cast<uintptr_t>(a_hi) | reinterpret_cast<uintptr_t>(a_lo) | reinterpret_cast<uintptr_t>(w_hi) |
       reinterpret_cast<uintptr_t>(w_lo)) & 15) return DDPO_EINVAL;
  if (d.w_layout != 0 && d.w_layout != 1) return DDPO_EINVAL;
  if (d.w_layout == 0 && (ldw < d.K || (ldw & 7))) return DDPO_EINVAL;
  if (d.ksize > 0) {
    if (d.ksize != 1 && d.ksize != 3) return DDPO_EINVAL;
    if (d.K != d.ksize * d.ksize * d.Cin || d.M != d.B * d.OH * d.OW || d.upsample < 0 || d.upsample > 2 || (lda && lda < d.Cin)) return DDPO_EINVAL;
  } else if (lda && lda < d.K) {
    return DDPO_EINVAL;
  }
  d.src = reinterpret_cast<const float*>(a_hi);      // the kernel reads d.src / d.w as the two planes and d.ld_src in elements
  d.w = reinterpret_cast<const float*>(a_lo);
  d.ld_src = lda;
  if (!buf_path_ok(d, ldw)) return DDPO_EINVAL;      // Cin (K) % 32 == 0 and 31-bit byte offsets: callers keep such layers on the fp32-fed entry
  // k-loop of the 128-row tiles: weights three LDS stages deep (requested two k-tiles ahead, counted vmcnt), activations two; the upper half
  // of the waves requests its pieces half a k-tile later than the lower half (d.splits bit 0), so the two waves of a SIMD alternate between
  // DMA issue and MFMAs.  Measured and removed (rounds 1-3): the plain wait / barrier / request / compute loop, two weight stages, no stagger,
  // s_setprio around the MFMA clusters, four-wave 128x320 and 128x160 tiles, requests spread one per accumulator block (profiles/r03_probe_kloop.log).
  d.splits = 1;                                      // `splits` is a wgrad-only field: the forward kernel reads bit 0 as the stagger flag
#ifdef DDPO_KLOOP_TIMING
  { const char* e = getenv("DDPO_DBG_ABL"); if (e) d.splits |= atoi(e) << 4; }
#endif
  return dispatch_bf16<3>(d, w_hi, w_lo, ldw, npass, ws, ws_bytes, as_stream(stream));
}

/* f16mx plane-fed variant (include/ddpo_hip.h): same kernel family, NPASS = 4 */
extern "C" int ddpo_gemm_conv_fwd_f16mx_planes(const ddpo_gemm_desc* dp, const uint16_t* a16, const uint16_t* a8, int lda,
                                               const uint16_t* w16, const uint16_t* w8, void* ws, size_t ws_bytes, void* stream) {
  if (!dp || !a16 || !w16 || (a8 == nullptr) != (w8 == nullptr) || (a8 && !dp->w_scale)) return DDPO_EINVAL;
  const int npass = a8 ? 4 : 5;            // ABI v14: BOTH 8-bit planes NULL = single-pass f16 (a_h * w_h only: the operator without its cross terms; opt-in)
  ddpo_gemm_desc d = *dp;
  if (npass == 5 && d.epilogue == 2) return DDPO_EINVAL;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || lda < 0 || (lda & 31) || d.w_dgrad || d.w_layout != 1 || !planes_out_ok(d)) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a16) | reinterpret_cast<uintptr_t>(a8) | reinterpret_cast<uintptr_t>(w16) | reinterpret_cast<uintptr_t>(w8)) & 15)
    return DDPO_EINVAL;
  if (d.ksize > 0) {
    if (d.ksize != 1 && d.ksize != 3) return DDPO_EINVAL;
    if (d.K != d.ksize * d.ksize * d.Cin || d.M != d.B * d.OH * d.OW || d.upsample < 0 || d.upsample > 2 || (lda && lda < d.Cin)) return DDPO_EINVAL;
  } else if (lda && lda < d.K) {
    return DDPO_EINVAL;
  }
  d.src = reinterpret_cast<const float*>(a16);
  d.w = reinterpret_cast<const float*>(a8);
  d.ld_src = lda;
  if (!buf_path_ok(d, 0)) return DDPO_EINVAL;
  d.splits = 1;
#ifdef DDPO_KLOOP_TIMING
  { const char* e = getenv("DDPO_DBG_ABL"); if (e) d.splits |= atoi(e) << 4; }
#endif
  return dispatch_bf16<3>(d, w16, w8, 0, npass, ws, ws_bytes, as_stream(stream));
}

// ------------------------------------------------------------------------------------------------
// weight packing: fp32 W[K][N] -> bf16 hi/lo planes, forward order [N][Kp] (k contiguous, Kp = K rounded up to 8,
// pad zero) and backward order [K][N] (the original order).  32x32 LDS-tiled transpose.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ w, int K, int N, int Kp,
                                                           uint16_t* __restrict__ fwd_hi, uint16_t* __restrict__ fwd_lo,
                                                           uint16_t* __restrict__ bwd_hi, uint16_t* __restrict__ bwd_lo) {
  __shared__ uint32_t tile[32][33];      // hi | lo<<16
  const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    uint32_t packed = 0;
    if (k < K && n < N) {
      const float x = w[(int64_t)k * N + n];
      const uint32_t h = cvt_pk_bf16(x, 0.f) & 0xFFFFu;
      const float rem = x - __uint_as_float(h << 16);
      const uint32_t l = cvt_pk_bf16(rem, 0.f) & 0xFFFFu;
      packed = h | (l << 16);
      if (bwd_hi) { bwd_hi[(int64_t)k * N + n] = (uint16_t)h; bwd_lo[(int64_t)k * N + n] = (uint16_t)l; }
    }
    tile[r][tx] = packed;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, k = k0 + tx;
    if (n < N && k < Kp) {
      const uint32_t p = tile[tx][r];
      fwd_hi[(int64_t)n * Kp + k] = (uint16_t)(p & 0xFFFFu);
      fwd_lo[(int64_t)n * Kp + k] = (uint16_t)(p >> 16);
    }
  }
}

// k-blocked forward planes (Kb = ceil(K / 32), N, 32): same 32x32 tile transpose, the tile of k-block kb lands at [kb][n0 .. n0+31][0..31]
__global__ void __launch_bounds__(256) pack_weights_kblocked_kernel(const float* __restrict__ w, int K, int N, uint16_t* __restrict__ fwd_hi,
                                                                    uint16_t* __restrict__ fwd_lo) {
  __shared__ uint32_t tile[32][33];
  const int kb = blockIdx.y, k0 = kb * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    uint32_t packed = 0;
    if (k < K && n < N) {
      const float x = w[(int64_t)k * N + n];
      const uint32_t h = cvt_pk_bf16(x, 0.f) & 0xFFFFu;
      const float rem = x - __uint_as_float(h << 16);
      packed = h | ((cvt_pk_bf16(rem, 0.f) & 0xFFFFu) << 16);
    }
    tile[r][tx] = packed;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r;
    if (n < N) {
      const uint32_t p = tile[tx][r];
      const int64_t o = ((int64_t)kb * N + n) * 32 + tx;
      fwd_hi[o] = (uint16_t)(p & 0xFFFFu);
      fwd_lo[o] = (uint16_t)(p >> 16);
    }
  }
}

extern "C" int ddpo_pack_weights_bf16_kblocked(const float* w, int K, int N, uint16_t* fwd_hi, uint16_t* fwd_lo, void* stream) {
  if (!w || !fwd_hi || !fwd_lo || K <= 0 || N <= 0) return DDPO_EINVAL;
  dim3 grid((N + 31) / 32, (K + 31) / 32);
  hipLaunchKernelGGL(pack_weights_kblocked_kernel, grid, dim3(256), 0, as_stream(stream), w, K, N, fwd_hi, fwd_lo);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// Data-gradient planes straight from the forward kernel w (taps, Cin, Cout) (HWIO flattened; dense: taps = 1): the forward-style operand of
// dX = dY * W' is W'[k'][n'] = w[taps - 1 - tap'][n'][co] with k' = tap' * Cout + co (taps flipped, channels transposed), K' = taps * Cout rows,
// N' = Cin columns.  For a fixed column n' consecutive k' are consecutive co: the reads are contiguous along the same index the k-blocked
// layout stores contiguously, so no LDS transpose is needed (one thread per element: lanes along k').  Replaces the flip / permute /
// contiguous copy torch made of every contraction weight after every optimizer update.
__global__ void __launch_bounds__(256) pack_weights_kblocked_dgrad_kernel(const float* __restrict__ w, int taps, int Cin, int Cout,
                                                                          uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
  const int kb = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int Kd = taps * Cout;
  const int k = kb * 32 + tx;
  const bool kok = k < Kd;
  const int tap = kok ? k / Cout : 0, co = k - tap * Cout;
  const float* src = w + ((int64_t)(taps - 1 - tap) * Cin) * Cout + co;
  for (int r = ty; r < 32; r += 8) {
    const int n = blockIdx.x * 32 + r;
    if (n >= Cin) continue;
    uint32_t h = 0, l = 0;
    if (kok) {
      const float x = src[(int64_t)n * Cout];
      h = cvt_pk_bf16(x, 0.f) & 0xFFFFu;
      l = cvt_pk_bf16(x - __uint_as_float(h << 16), 0.f) & 0xFFFFu;
    }
    const int64_t o = ((int64_t)kb * Cin + n) * 32 + tx;
    hi[o] = (uint16_t)h;
    lo[o] = (uint16_t)l;
  }
}
extern "C" int ddpo_pack_weights_bf16_kblocked_dgrad(const float* w, int taps, int Cin, int Cout, uint16_t* hi, uint16_t* lo, void* stream) {
  if (!w || !hi || !lo || taps <= 0 || Cin <= 0 || Cout <= 0) return DDPO_EINVAL;
  dim3 grid((Cin + 31) / 32, (taps * Cout + 31) / 32);
  hipLaunchKernelGGL(pack_weights_kblocked_dgrad_kernel, grid, dim3(256), 0, as_stream(stream), w, taps, Cin, Cout, hi, lo);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_pack_weights_bf16(const float* w, int K, int N, int Kp, uint16_t* fwd_hi, uint16_t* fwd_lo, uint16_t* bwd_hi,
                                      uint16_t* bwd_lo, void* stream) {
  if (!w || !fwd_hi || !fwd_lo || K <= 0 || N <= 0 || Kp < K || (Kp & 7) || (bwd_hi && !bwd_lo)) return DDPO_EINVAL;
  dim3 grid((N + 31) / 32, (Kp + 31) / 32);
  hipLaunchKernelGGL(pack_weights_kernel, grid, dim3(256), 0, as_stream(stream), w, K, N, Kp, fwd_hi, fwd_lo, bwd_hi, bwd_lo);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// ------------------------------------------------------------------------------------------------
// fp32 activations (rows, cols) with row stride ldx -> bf16 hi / lo planes (rows, ld_out): the operand format of
// ddpo_gemm_conv_fwd_bf16_planes.  Stand-alone form of what the normalisation kernels do in their output stage
// (same v_cvt_pk_bf16_f32 split as the GEMM loader: the planes hold exactly the bits the loader would have produced).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ hi,
                                                           uint16_t* __restrict__ lo, int ld_out, int64_t rows, int cols4) {
  const int64_t total = rows * cols4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols4;
    const int c = (int)(i - r * cols4) << 2;
    uint2 h, l;
    split4(*reinterpret_cast<const float4*>(x + r * ldx + c), h, l);
    const int64_t o = plane_off(r, c, ld_out, rows);
    *reinterpret_cast<uint2*>(hi + o) = h;
    *reinterpret_cast<uint2*>(lo + o) = l;
  }
}

extern "C" int ddpo_split_planes_bf16(const float* x, int ldx, uint16_t* hi, uint16_t* lo, int ld_out, int64_t rows, int cols, void* stream) {
  if (!x || !hi || !lo || rows <= 0 || cols <= 0 || (cols & 3) || (ldx & 3) || (ld_out & 3) || ldx < cols) return DDPO_EINVAL;
  if (ld_out == 0 ? (cols & 31) != 0 : ld_out < cols) return DDPO_EINVAL;          // ld_out == 0: k-blocked planes (cols / 32, rows, 32)
  if ((reinterpret_cast<uintptr_t>(x) & 15) || ((reinterpret_cast<uintptr_t>(hi) | reinterpret_cast<uintptr_t>(lo)) & 7)) return DDPO_EINVAL;
  int64_t blocks = (rows * (cols >> 2) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_planes_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, ldx, hi, lo, ld_out, rows, cols >> 2);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

__global__ void __launch_bounds__(256) split_planes_mx_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ p16,
                                                              uint16_t* __restrict__ p8, int ld_out, int64_t rows, int cols4) {
  const int64_t total = rows * cols4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols4;
    const int c = (int)(i - r * cols4) << 2;
    mx_store4(p16, p8, r, c, ld_out, rows, *reinterpret_cast<const float4*>(x + r * ldx + c));
  }
}

extern "C" int ddpo_split_planes_f16mx(const float* x, int ldx, uint16_t* p16, uint16_t* p8, int ld_out, int64_t rows, int cols, void* stream) {
  if (!x || !p16 || !p8 || rows <= 0 || cols <= 0 || (cols & 31) || (ldx & 3) || (ld_out & 31) || ldx < cols) return DDPO_EINVAL;
  if (ld_out != 0 && ld_out < cols) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || ((reinterpret_cast<uintptr_t>(p16) | reinterpret_cast<uintptr_t>(p8)) & 7)) return DDPO_EINVAL;
  int64_t blocks = (rows * (cols >> 2) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_planes_mx_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, ldx, p16, p8, ld_out, rows, cols >> 2);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// f16mx weight planes (include/ddpo_hip.h).  Pass 1 (a, b): biased exponent of every column's largest |w| -> scale byte; pass 2: the 32x32
// tile transpose of the k-blocked packer, writing the f16 plane and the [l8 | h8] byte plane.
// pass 1a: biased exponent of every column's largest |w|.  64 columns x 4 k-lanes per workgroup, the reduction split over blockIdx.y chunks of k
// and combined with atomicMax on the (non-negative) fp32 bit patterns in a caller-provided uint32 scratch (N words, zeroed).
__global__ void __launch_bounds__(256) mx_colmax_kernel(const float* __restrict__ w, int K, int N, int k_per_block, uint32_t* __restrict__ colmax) {
  __shared__ uint32_t red[4][64];
  const int c = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const int k0 = blockIdx.y * k_per_block, k1 = min(K, k0 + k_per_block);
  uint32_t m = 0;
  if (n < N)
    for (int k = k0 + kl; k < k1; k += 4) m = max(m, __float_as_uint(w[(int64_t)k * N + n]) & 0x7FFFFFFFu);
  red[kl][c] = m;
  __syncthreads();
  if (kl == 0 && n < N) {
    m = max(max(red[0][c], red[1][c]), max(red[2][c], red[3][c]));
    if (m) atomicMax(colmax + n, m);
  }
}
// pass 1b: scale byte
__global__ void __launch_bounds__(256) mx_colscale_kernel(const uint32_t* __restrict__ colmax, int N, uint8_t* __restrict__ scale) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int e = (int)(colmax[n] >> 23);         // 2^(e - 127) <= max|w| < 2^(e - 126)
  e = min(max(e, 32), 240);               // all-zero / denormal columns: any valid scale; keeps (byte - 11) and 2^(261 - e) in range
  scale[n] = (uint8_t)(e - 7);            // max|w| / 2^(e - 7 - 127) in [128, 256) <= 448 (e4m3 range)
}
__global__ void __launch_bounds__(256) pack_weights_mx_kernel(const float* __restrict__ w, int K, int N, const uint8_t* __restrict__ scale,
                                                              uint16_t* __restrict__ w16, uint8_t* __restrict__ w8) {
  __shared__ uint32_t tile[32][33];       // f16 bits | h8 << 16 | l8 << 24
  const int kb = blockIdx.y, k0 = kb * 32, n0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, n = n0 + tx;
    uint32_t packed = 0;
    if (k < K && n < N) {
      const float x = fminf(fmaxf(w[(int64_t)k * N + n], -65504.f), 65504.f);
      const _Float16 h = (_Float16)x;
      const float hf = (float)h, inv = __uint_as_float((uint32_t)(254 - (int)scale[n]) << 23);       // 2^-(byte - 127)
      const int p = __builtin_amdgcn_cvt_pk_fp8_f32(hf * inv, (x - hf) * 2048.f * inv, 0, false);
      packed = (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)p << 16);
    }
    tile[r][tx] = packed;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r;
    if (n < N) {
      const uint32_t p = tile[tx][r];
      const int64_t row = (int64_t)kb * N + n;
      w16[row * 32 + tx] = (uint16_t)(p & 0xFFFFu);
      const int o = 2 * (tx & 16) + (tx & 15);              // chunks [l8 k0-15 | h8 k0-15 | l8 k16-31 | h8 k16-31]
      w8[row * 64 + o] = (uint8_t)(p >> 24);                // l8 first: lane half 0 of the MFMA pairs it with the activations' h8
      w8[row * 64 + o + 16] = (uint8_t)((p >> 16) & 0xFFu);
    }
  }
}
extern "C" int ddpo_pack_weights_f16mx(const float* w, int K, int N, uint16_t* w16, uint16_t* w8, uint8_t* scale, void* stream) {
  if (!w || !w16 || !w8 || !scale || K <= 0 || N <= 0) return DDPO_EINVAL;
  // the column maxima are gathered in the first N words of the f16 plane (>= 64 bytes per column; overwritten by the pack pass below)
  uint32_t* colmax = reinterpret_cast<uint32_t*>(w16);
  if (hipMemsetAsync(colmax, 0, (size_t)N * sizeof(uint32_t), as_stream(stream)) != hipSuccess) return DDPO_ELAUNCH;
  int ky = (K + 255) / 256;
  if (ky > 128) ky = 128;
  const int kpb = ((K + ky - 1) / ky + 3) / 4 * 4;
  hipLaunchKernelGGL(mx_colmax_kernel, dim3((N + 63) / 64, (K + kpb - 1) / kpb), dim3(256), 0, as_stream(stream), w, K, N, kpb, colmax);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL(mx_colscale_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), colmax, N, scale);
  DDPO_LAUNCH_CHECK();
  hipLaunchKernelGGL(pack_weights_mx_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, as_stream(stream), w, K, N, scale, w16,
                     reinterpret_cast<uint8_t*>(w8));
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on the bf16x3 MFMA datapath:  dW[k][n] += sum_m A(m,k) * dY[m][n]   (k = (ky,kx,ci); m = pixels)
// Both operands have the reduction index m as their SLOW memory dimension, so each is transposed while it is staged:
// a thread loads float4s of two consecutive pixels and writes, per channel, the packed (pixel m, pixel m+1) bf16 pair
// as one dword of the [row][m] LDS image.  Row pitch is 18 dwords (72 B): the pair writes of a half-wave hit 32
// distinct banks (x2, free) and the two ds_read_b64 of a fragment are conflict free.  Fast path only: stride 1,
// no upsampling (output pixel m == input pixel m, source address linear in m); other layers use gemm_wgrad_kernel.
// ------------------------------------------------------------------------------------------------
#define WG_PITCH 18     // dwords per LDS row (16 dwords = 32 pixels of the k-tile, +2 pad)

__device__ __forceinline__ bf16x8 lds_frag(const uint32_t* base, int row, int dw) {
  const uint2 a = *reinterpret_cast<const uint2*>(base + row * WG_PITCH + dw);
  const uint2 b = *reinterpret_cast<const uint2*>(base + row * WG_PITCH + dw + 2);
  return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}

// APLN / BPLN: that operand arrives ALREADY split into bf16 hi / lo planes ((rows, ld) bf16, same element offsets as the fp32
// tensor: the activation planes a forward GroupNorm / LayerNorm wrote, or dY planes from a plane-emitting output stage) — the
// loader then only has to pair pixels m / m+1 of a channel into a dword (one v_perm_b32 per plane dword) instead of running the
// fp32 -> bf16 split (2 v_cvt_pk + 2 v_sub + 2 mask / shift per pair): the split was ~2/3 of this kernel's VALU work, which
// looked like its bound (measured in round 2: +0.5 % on the train step, so it is not).  Same values reach the MFMAs as in the fp32-fed form.
// Also measured and rejected in round 2: unconditional loads + a second register stage (two k-tiles of prefetch): 256 VGPRs with
// 12-24 spilled at two waves per SIMD, train step 9 % SLOWER (profiles/r02_ab_wgrad_deep.log).
// ROWL ("row loader", round 2): the SQ counters of the loader below showed ~10 VALU + 4 SALU per MFMA at 31 % MFMA-pipe busy: 16 bytes per
// fetch, ~25 VALU per fetch of 64-bit address arithmetic, per-pixel (batch, y, x) bookkeeping with loops and divergent branches around every
// load.  For the regular layers (dense, or stride-1 "same" convolutions whose rows tile into the 32-pixel k-tiles: OW % 32 == 0 or
// 32 % OW == 0; M % 32 == 0; operand tensors < 2 GiB) all of that collapses: a k-tile is 32 consecutive pixels starting at an image-row
// boundary that is the SAME for the whole workgroup, so (oy, ox) of the tile live in scalars, every thread's four byte offsets relative
// to the tile are CONSTANTS, the per-tile advance is one scalar soffset, and a masked element is an out-of-range buffer offset that reads
// zeros (raw buffer loads) — ~6 VALU per activation fetch, none per dY fetch, no branches.
template <bool APLN, bool BPLN, bool ROWL = false>
__global__ void __launch_bounds__(BF_THREADS) gemm_wgrad_bf16_kernel(const ddpo_gemm_desc d, int tiles_n, int m_per_split,
                                                                   const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                                                   const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo) {
  constexpr int BM = 128, BN = 128, BK = 32;
  constexpr int PLANE = BM * WG_PITCH;                 // dwords per plane (BM == BN)
  __shared__ __attribute__((aligned(16))) uint32_t smem[2][4 * PLANE];     // per stage: A_hi | A_lo | B_hi | B_lo
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int k0 = tile_m * BM, n0 = tile_n * BN;
  const int m_begin = blockIdx.y * m_per_split;
  const int m_end = min(m_begin + m_per_split, d.M);
  if (m_begin >= m_end) return;

  const bool conv = d.ksize > 0;
  // loader geometry: quad q = lane&7 (4 consecutive k or n), pixel pair pp = lane>>3, wave w covers rows 32w..32w+31
  const int q = lane & 7, pp = lane >> 3;
  const int arow = 32 * wid + 4 * q;                   // first of this thread's 4 LDS rows (same for A and B tiles)
  const int kg = k0 + arow;                            // global k of those rows
  const bool kvalid = kg < d.K;
  int dky = 0, dkx = 0, ci = kg;
  if (conv) {
    const int tap = kg / d.Cin;
    ci = kg - tap * d.Cin;
    const int ky = tap / d.ksize;
    dky = ky - d.pad;
    dkx = tap - ky * d.ksize - d.pad;
  }
  const int ng = n0 + arow;
  const bool nvalid = ng < d.N;
  // the 4 OUTPUT pixels this thread stages per k-tile: m = m_begin + kt*32 + 16*p + 2*pp + e ; (batch, oy, ox) are tracked
  // incrementally.  Stride-1 "same" convolutions read input pixel m + a constant tap offset (`simple`); strided and
  // nearest-2x-upsampled ones compute the source pixel of the tap from (oy, ox).
  const bool simple = !conv || (d.stride == 1 && d.upsample == 0 && d.OH == d.H && d.OW == d.W);
  const int VH = d.upsample ? 2 * d.H : d.H, VW = d.upsample ? 2 * d.W : d.W;
  int pb[2][2], poy[2][2], pox[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int m = m_begin + 16 * p + 2 * pp + e;
      pox[p][e] = conv ? m % d.OW : 0;
      poy[p][e] = conv ? (m / d.OW) % d.OH : 0;
      pb[p][e] = conv ? m / (d.OW * d.OH) : 0;
    }
  // element strides / channel terms of the two operands.  A k-blocked PLANE operand (ld == 0: (C / 32, rows, 32), ABI v6) has pixel
  // stride 32 and the channel quad's block base + offset inside the block as its "channel term"; everything below is written on these.
  const bool kbA = APLN && d.ld_src == 0, kbB = BPLN && d.ld_w == 0;
  const int64_t rowsA = conv ? (int64_t)d.B * d.H * d.W : (int64_t)d.M;
  const int lda_e = kbA ? 32 : d.ld_src, ldb_e = kbB ? 32 : d.ld_w;
  const int64_t a_c0 = kbA ? (int64_t)(ci >> 5) * rowsA * 32 + (ci & 31) : (int64_t)ci;          // dense: ci == kg
  const int64_t b_c0 = kbB ? (int64_t)(ng >> 5) * (int64_t)d.M * 32 + (ng & 31) : (int64_t)ng;
  const int64_t tap_off = conv ? ((int64_t)dky * d.W + dkx) * lda_e + a_c0 : a_c0;

  float4 ra[2][2], rb[2][2];          // fp32 operands; a plane operand keeps (hi.x, hi.y, lo.x, lo.y) raw bits in the same registers
  auto as_f4 = [](const uint2 h, const uint2 l) {
    return make_float4(__uint_as_float(h.x), __uint_as_float(h.y), __uint_as_float(l.x), __uint_as_float(l.y));
  };
  // ---- ROWL state (see the note above the kernel)
  constexpr uint32_t ESA = APLN ? 2u : 4u, ESB = BPLN ? 2u : 4u;     // bytes per element of the operands as stored
  uint32_t rl_va[2][2], rl_vb[2][2];   // byte offsets of this thread's elements for k-tile 0 (BUF_OOB: never valid)
  int rl_cy[2][2], rl_cx[2][2];        // iy = oy_t + cy, ix = ox_t + cx of the element's tap
  int rl_oy = 0, rl_ox = 0;            // image row / column of the CURRENT k-tile's first pixel (uniform)
  // a tap above / left of the tile has a NEGATIVE offset relative to its pixel: the activation descriptors start `rl_guard` bytes in front of
  // the tensor so that every offset is non-negative (such elements are only ever fetched when their tap is inside the image, i.e. in range)
  const int64_t rl_guard = conv ? (int64_t)(d.W + 1) * lda_e * (int64_t)ESA : 0;
  __amdgpu_buffer_rsrc_t rl_ra0 = make_rsrc(reinterpret_cast<const char*>(APLN ? (const void*)a_hi : (const void*)d.src) - rl_guard),
                         rl_ra1 = make_rsrc(reinterpret_cast<const char*>(APLN ? (const void*)a_lo : (const void*)d.src) - rl_guard);
  __amdgpu_buffer_rsrc_t rl_rb0 = make_rsrc(BPLN ? (const void*)b_hi : (const void*)d.w), rl_rb1 = make_rsrc(BPLN ? (const void*)b_lo : (const void*)d.w);
  if constexpr (ROWL) {
    const int rem = conv ? m_begin % (d.OH * d.OW) : 0;
    rl_oy = conv ? rem / d.OW : 0;
    rl_ox = conv ? rem - rl_oy * d.OW : 0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int eoff = 16 * p + 2 * pp + e;
        const int dy_e = (conv && d.OW < BK) ? eoff / d.OW : 0;
        const int x_e = (conv && d.OW < BK) ? eoff - dy_e * d.OW : eoff;
        rl_cy[p][e] = dy_e + dky;
        rl_cx[p][e] = x_e + dkx;
        const int64_t ao = ((int64_t)(m_begin + eoff) * lda_e + tap_off) * (int64_t)ESA + rl_guard;
        const int64_t bo = ((int64_t)(m_begin + eoff) * ldb_e + b_c0) * (int64_t)ESB;
        rl_va[p][e] = (kvalid && ao >= 0 && ao < 0x7FFFFFF0ll) ? (uint32_t)ao : BUF_OOB;
        rl_vb[p][e] = (nvalid && bo >= 0 && bo < 0x7FFFFFF0ll) ? (uint32_t)bo : BUF_OOB;
      }
  }
  auto load_tile_rows = [&](int kt) {
    const uint32_t so_a = (uint32_t)kt * (uint32_t)(BK * lda_e) * ESA, so_b = (uint32_t)kt * (uint32_t)(BK * ldb_e) * ESB;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        uint32_t voa = rl_va[p][e];
        if (conv) {
          const bool ok = (unsigned)(rl_oy + rl_cy[p][e]) < (unsigned)d.H && (unsigned)(rl_ox + rl_cx[p][e]) < (unsigned)d.W;
          voa = ok ? voa : BUF_OOB;
        }
        if (APLN) {
          const u32x2 h2 = __builtin_amdgcn_raw_buffer_load_b64(rl_ra0, voa, so_a, 0), l2 = __builtin_amdgcn_raw_buffer_load_b64(rl_ra1, voa, so_a, 0);
          ra[p][e] = make_float4(__uint_as_float(h2.x), __uint_as_float(h2.y), __uint_as_float(l2.x), __uint_as_float(l2.y));
        } else {
          const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(rl_ra0, voa, so_a, 0);
          ra[p][e] = make_float4(__uint_as_float(v4.x), __uint_as_float(v4.y), __uint_as_float(v4.z), __uint_as_float(v4.w));
        }
        if (BPLN) {
          const u32x2 h2 = __builtin_amdgcn_raw_buffer_load_b64(rl_rb0, rl_vb[p][e], so_b, 0), l2 = __builtin_amdgcn_raw_buffer_load_b64(rl_rb1, rl_vb[p][e], so_b, 0);
          rb[p][e] = make_float4(__uint_as_float(h2.x), __uint_as_float(h2.y), __uint_as_float(l2.x), __uint_as_float(l2.y));
        } else {
          const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(rl_rb0, rl_vb[p][e], so_b, 0);
          rb[p][e] = make_float4(__uint_as_float(v4.x), __uint_as_float(v4.y), __uint_as_float(v4.z), __uint_as_float(v4.w));
        }
      }
    if (conv) {                           // next k-tile: 32 pixels on (uniform)
      if (d.OW >= BK) {
        rl_ox += BK;
        if (rl_ox >= d.OW) { rl_ox = 0; rl_oy = rl_oy + 1 >= d.OH ? 0 : rl_oy + 1; }
      } else {
        rl_oy += BK / d.OW;
        if (rl_oy >= d.OH) rl_oy -= d.OH;
      }
    }
  };
  auto load_tile_px = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int m = m_begin + kt * BK + 16 * p + 2 * pp + e;
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        if (m < m_end) {
          bool ok = kvalid;
          int64_t aoff = (int64_t)m * lda_e + tap_off;
          if (conv) {
            const int iy = poy[p][e] * d.stride + dky, ix = pox[p][e] * d.stride + dkx;       // virtual (upsampled) coordinates
            ok = ok && iy >= 0 && iy < VH && ix >= 0 && ix < VW;
            if (!simple) {
              const int sy = d.upsample ? (iy >> 1) : iy, sx = d.upsample ? (ix >> 1) : ix;
              aoff = ((int64_t)(pb[p][e] * d.H + sy) * d.W + sx) * lda_e + a_c0;
            }
          }
          if (ok) {
            if (APLN) va = as_f4(*reinterpret_cast<const uint2*>(a_hi + aoff), *reinterpret_cast<const uint2*>(a_lo + aoff));
            else va = *reinterpret_cast<const float4*>(d.src + aoff);
          }
          if (nvalid) {
            const int64_t boff = (int64_t)m * ldb_e + b_c0;
            if (BPLN) vb = as_f4(*reinterpret_cast<const uint2*>(b_hi + boff), *reinterpret_cast<const uint2*>(b_lo + boff));
            else vb = *reinterpret_cast<const float4*>(d.w + boff);
          }
        }
        ra[p][e] = va;
        rb[p][e] = vb;
        if (conv) {          // advance this pixel by BK
          pox[p][e] += BK;
          while (pox[p][e] >= d.OW) { pox[p][e] -= d.OW; ++poy[p][e]; }
          while (poy[p][e] >= d.OH) { poy[p][e] -= d.OH; ++pb[p][e]; }
        }
      }
  };
  auto load_tile = [&](int kt) {
    if constexpr (ROWL) load_tile_rows(kt); else load_tile_px(kt);
  };
  // bias gradient (d.colsum, fp32 dY only): the k = 0 row of workgroups also sums the dY values it stages, per channel
  const bool do_cs = !BPLN && d.colsum != nullptr && tile_m == 0;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  auto store_tile = [&](int buf) {
    uint32_t* st = smem[buf];
    if (do_cs) {
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[j] += ((&rb[0][0].x)[j] + (&rb[0][1].x)[j]) + ((&rb[1][0].x)[j] + (&rb[1][1].x)[j]);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int dw = pp + 8 * p;                        // dword (= pixel pair) index within the row
      const float* a0 = &ra[p][0].x; const float* a1 = &ra[p][1].x;
      const float* b0 = &rb[p][0].x; const float* b1 = &rb[p][1].x;
      // plane operand: registers hold [ch0|ch1, ch2|ch3] (hi) and the same for lo, per pixel; pair channel j of pixels m, m+1
      auto pair = [](const float* p0, const float* p1, int j, int plane) {
        const uint32_t w0 = __float_as_uint(p0[2 * plane + (j >> 1)]), w1 = __float_as_uint(p1[2 * plane + (j >> 1)]);
        return __builtin_amdgcn_perm(w1, w0, (j & 1) ? 0x07060302u : 0x05040100u);
      };
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t hi, lo;
        if (APLN) { hi = pair(a0, a1, j, 0); lo = pair(a0, a1, j, 1); }
        else split2(a0[j], a1[j], hi, lo);              // (pixel m, pixel m+1) of channel k+j
        st[(arow + j) * WG_PITCH + dw] = hi;
        st[PLANE + (arow + j) * WG_PITCH + dw] = lo;
        if (BPLN) { hi = pair(b0, b1, j, 0); lo = pair(b0, b1, j, 1); }
        else split2(b0[j], b1[j], hi, lo);
        st[2 * PLANE + (arow + j) * WG_PITCH + dw] = hi;
        st[3 * PLANE + (arow + j) * WG_PITCH + dw] = lo;
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (m_end - m_begin + BK - 1) / BK;
  const int li = lane & 31, h = lane >> 5;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const uint32_t* st = smem[cur];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int dw = 8 * ms + 4 * h;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = lds_frag(st, wm * 64 + i * 32 + li, dw);
        al[i] = lds_frag(st + PLANE, wm * 64 + i * 32 + li, dw);
        bh[i] = lds_frag(st + 2 * PLANE, wn * 64 + i * 32 + li, dw);
        bl[i] = lds_frag(st + 3 * PLANE, wn * 64 + i * 32 + li, dw);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + li;
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = k0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= d.K) continue;
        atomicAdd(d.out + (int64_t)row * d.ld_out + col, d.alpha * acc[i][j][r]);
      }
    }
  if (do_cs) {                                  // lanes q + 8 * pp of a wave hold the same 4 channels: fold the 8 pixel-pair lanes, lane pp == 0 adds
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = cs[j];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if (pp == 0 && ng + j < d.N) atomicAdd(d.colsum + ng + j, v);
    }
  }
}

// WIDE weight-gradient tile (round 3): 128 (k) x 320 (n) per workgroup, 8 waves of 32 x 160, one workgroup per CU.  The 128 x 128 kernel above
// moves 32 KB of operands per 0.52 M multiply-adds (61 B / kMAC) and sits at the CU's ~20 B / clk fetch rate with the matrix pipe 42 % busy
// (236 TF); this tile moves 56 KB per 1.31 M (43 B / kMAC) — the forward 128x320 tile's ratio.  Row loader only (the regular layers: dense, and
// stride-1 "same" convolutions whose rows tile into the 32-pixel k-tiles), N % 320 == 0; same staging (pixel pairs of a channel packed into one
// dword of the [row][m] LDS image), same MFMA order per element as the 128 x 128 kernel.
//   loader tasks (4 channels x 2 pixels, 8 quads x 8 pixel pairs per wave): A = 128 rows x 16 pairs = 8 wave tasks, one per wave;
//   dY = 320 rows x 16 pairs = 20 wave tasks: waves 0-3 take three, waves 4-7 two.
template <bool APLN, bool BPLN>
__global__ void __launch_bounds__(512) gemm_wgrad_bf16_wide_kernel(const ddpo_gemm_desc d, int tiles_n, int m_per_split,
                                                                   const uint16_t* __restrict__ a_hi, const uint16_t* __restrict__ a_lo,
                                                                   const uint16_t* __restrict__ b_hi, const uint16_t* __restrict__ b_lo) {
  constexpr int BM = 128, BN = 320, BK = 32, TN = 5, NBT = 3;
  constexpr int PA = BM * WG_PITCH, PB = BN * WG_PITCH;          // dwords per plane
  constexpr int STAGE = 2 * PA + 2 * PB;                         // A_hi | A_lo | B_hi | B_lo
  extern __shared__ __attribute__((aligned(16))) uint32_t wsm[];
  const int t = threadIdx.x, lane = t & 63, wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int k0 = tile_m * BM, n0 = tile_n * BN;
  const int m_begin = blockIdx.y * m_per_split;
  const int m_end = min(m_begin + m_per_split, d.M);
  if (m_begin >= m_end) return;
  const bool conv = d.ksize > 0;
  const int q = lane & 7, pp = lane >> 3;
  // ---- this thread's A task: 4 channel rows, one pixel pair
  const int a_row = 32 * (wid & 3) + 4 * q, a_dw = 8 * (wid >> 2) + pp;
  const int kg = k0 + a_row;
  const bool kvalid = kg < d.K;
  int dky = 0, dkx = 0, ci = kg;
  if (conv) {
    const int tap = kg / d.Cin;
    ci = kg - tap * d.Cin;
    const int ky = tap / d.ksize;
    dky = ky - d.pad;
    dkx = tap - ky * d.ksize - d.pad;
  }
  const bool kbA = APLN && d.ld_src == 0, kbB = BPLN && d.ld_w == 0;
  const int64_t rowsA = conv ? (int64_t)d.B * d.H * d.W : (int64_t)d.M;
  const int lda_e = kbA ? 32 : d.ld_src, ldb_e = kbB ? 32 : d.ld_w;
  const int64_t a_c0 = kbA ? (int64_t)(ci >> 5) * rowsA * 32 + (ci & 31) : (int64_t)ci;
  const int64_t tap_off = conv ? ((int64_t)dky * d.W + dkx) * lda_e + a_c0 : a_c0;
  constexpr uint32_t ESA = APLN ? 2u : 4u, ESB = BPLN ? 2u : 4u;
  const int64_t rl_guard = conv ? (int64_t)(d.W + 1) * lda_e * (int64_t)ESA : 0;
  const __amdgpu_buffer_rsrc_t rs_a0 = make_rsrc(reinterpret_cast<const char*>(APLN ? (const void*)a_hi : (const void*)d.src) - rl_guard),
                               rs_a1 = make_rsrc(reinterpret_cast<const char*>(APLN ? (const void*)a_lo : (const void*)d.src) - rl_guard);
  const __amdgpu_buffer_rsrc_t rs_b0 = make_rsrc(BPLN ? (const void*)b_hi : (const void*)d.w), rs_b1 = make_rsrc(BPLN ? (const void*)b_lo : (const void*)d.w);
  uint32_t va[2], vb[NBT][2];
  int cy[2], cx[2];
  const int rem0 = conv ? m_begin % (d.OH * d.OW) : 0;
  int t_oy = conv ? rem0 / d.OW : 0, t_ox = conv ? rem0 - (rem0 / d.OW) * d.OW : 0;      // image position of the current k-tile's first pixel (uniform)
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int eoff = 2 * a_dw + e;
    const int dy_e = (conv && d.OW < BK) ? eoff / d.OW : 0;
    const int x_e = (conv && d.OW < BK) ? eoff - dy_e * d.OW : eoff;
    cy[e] = dy_e + dky;
    cx[e] = x_e + dkx;
    const int64_t ao = ((int64_t)(m_begin + eoff) * lda_e + tap_off) * (int64_t)ESA + rl_guard;
    va[e] = (kvalid && ao >= 0 && ao < 0x7FFFFFF0ll) ? (uint32_t)ao : BUF_OOB;
  }
  int b_row[NBT], b_dw[NBT];
#pragma unroll
  for (int i = 0; i < NBT; ++i) {
    const int T = wid + 8 * i;                         // wave task: channel block T >> 1 (of 10), pixel-pair half T & 1
    b_row[i] = 32 * (T >> 1) + 4 * q;
    b_dw[i] = 8 * (T & 1) + pp;
    const int ng = n0 + b_row[i];
    const int64_t b_c0 = kbB ? (int64_t)(ng >> 5) * (int64_t)d.M * 32 + (ng & 31) : (int64_t)ng;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t bo = ((int64_t)(m_begin + 2 * b_dw[i] + e) * ldb_e + b_c0) * (int64_t)ESB;
      vb[i][e] = (T < 20 && ng < d.N && bo >= 0 && bo < 0x7FFFFFF0ll) ? (uint32_t)bo : BUF_OOB;
    }
  }
  const bool third = wid < 4;                          // wave-uniform: this wave stages a third dY task
  // TWO register sets: tile T travels in set T & 1 and is requested two k-tiles before it is written to LDS (one workgroup per CU: nothing else
  // covers the fetch latency; with one set the loop measured no faster than the 128 x 128 kernel's two workgroups per CU)
  float4 ra[2][2], rb[2][NBT][2];
  auto ld4 = [&](const __amdgpu_buffer_rsrc_t r0, const __amdgpu_buffer_rsrc_t r1, uint32_t vo, uint32_t so, bool pl) {
    if (pl) {
      const u32x2 h2 = __builtin_amdgcn_raw_buffer_load_b64(r0, vo, so, 0), l2 = __builtin_amdgcn_raw_buffer_load_b64(r1, vo, so, 0);
      return make_float4(__uint_as_float(h2.x), __uint_as_float(h2.y), __uint_as_float(l2.x), __uint_as_float(l2.y));
    }
    const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(r0, vo, so, 0);
    return make_float4(__uint_as_float(v4.x), __uint_as_float(v4.y), __uint_as_float(v4.z), __uint_as_float(v4.w));
  };
  const int nk = (m_end - m_begin + BK - 1) / BK;
  auto load_tile = [&](int kt, auto sc) {              // requests past the last k-tile re-fetch it (unconditional loads: counted waits stay exact)
    constexpr int S = decltype(sc)::value;
    const int ktc = min(kt, nk - 1);
    const uint32_t so_a = (uint32_t)ktc * (uint32_t)(BK * lda_e) * ESA, so_b = (uint32_t)ktc * (uint32_t)(BK * ldb_e) * ESB;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      uint32_t vo = va[e];
      if (conv) vo = ((unsigned)(t_oy + cy[e]) < (unsigned)d.H && (unsigned)(t_ox + cx[e]) < (unsigned)d.W) ? vo : BUF_OOB;
      ra[S][e] = ld4(rs_a0, rs_a1, vo, so_a, APLN);
    }
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
      if (i == 2 && !third) continue;
#pragma unroll
      for (int e = 0; e < 2; ++e) rb[S][i][e] = ld4(rs_b0, rs_b1, vb[i][e], so_b, BPLN);
    }
    if (conv && kt < nk - 1) {            // next k-tile: 32 pixels on (uniform)
      if (d.OW >= BK) {
        t_ox += BK;
        if (t_ox >= d.OW) { t_ox = 0; t_oy = t_oy + 1 >= d.OH ? 0 : t_oy + 1; }
      } else {
        t_oy += BK / d.OW;
        if (t_oy >= d.OH) t_oy -= d.OH;
      }
    }
  };
  auto pair = [](const float* p0, const float* p1, int j, int plane) {
    const uint32_t w0 = __float_as_uint(p0[2 * plane + (j >> 1)]), w1 = __float_as_uint(p1[2 * plane + (j >> 1)]);
    return __builtin_amdgcn_perm(w1, w0, (j & 1) ? 0x07060302u : 0x05040100u);
  };
  const bool do_cs = !BPLN && d.colsum != nullptr && tile_m == 0;      // bias gradient: see the 128 x 128 kernel
  float cs[NBT][4];
#pragma unroll
  for (int i = 0; i < NBT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[i][j] = 0.f;
  auto store_tile = [&](int buf, auto sc) {
    constexpr int S = decltype(sc)::value;
    uint32_t* st = wsm + buf * STAGE;
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < NBT; ++i) {
        if (i == 2 && !third) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[i][j] += (&rb[S][i][0].x)[j] + (&rb[S][i][1].x)[j];
      }
    }
    {
      const float* a0 = &ra[S][0].x; const float* a1 = &ra[S][1].x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t hi, lo;
        if (APLN) { hi = pair(a0, a1, j, 0); lo = pair(a0, a1, j, 1); }
        else split2(a0[j], a1[j], hi, lo);
        st[(a_row + j) * WG_PITCH + a_dw] = hi;
        st[PA + (a_row + j) * WG_PITCH + a_dw] = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
      if (i == 2 && !third) continue;
      const float* b0 = &rb[S][i][0].x; const float* b1 = &rb[S][i][1].x;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t hi, lo;
        if (BPLN) { hi = pair(b0, b1, j, 0); lo = pair(b0, b1, j, 1); }
        else split2(b0[j], b1[j], hi, lo);
        st[2 * PA + (b_row[i] + j) * WG_PITCH + b_dw[i]] = hi;
        st[2 * PA + PB + (b_row[i] + j) * WG_PITCH + b_dw[i]] = lo;
      }
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int li = lane & 31, h = lane >> 5;
  auto compute = [&](int cur) {
    const uint32_t* st = wsm + cur * STAGE;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int dw = 8 * ms + 4 * h;
      const bf16x8 ah = lds_frag(st, wm * 32 + li, dw), al = lds_frag(st + PA, wm * 32 + li, dw);
      bf16x8 bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = lds_frag(st + 2 * PA, wn * 160 + j * 32 + li, dw);
        bl[j] = lds_frag(st + 2 * PA + PB, wn * 160 + j * 32 + li, dw);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[j], 0, 0, 0);
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  load_tile(0, S0{});
  store_tile(0, S0{});
  load_tile(1, S1{});
  load_tile(2, S0{});
  __syncthreads();
  // iteration kt: tile kt + 1 (requested two iterations ago) -> the LDS stage everybody left at the last barrier; request tile kt + 3 into
  // the registers just freed; multiply tile kt
  auto step = [&](int kt, auto sc) {
    constexpr int S = decltype(sc)::value;             // == (kt + 1) & 1
    if (kt + 1 < nk) store_tile(S, sc);
    load_tile(kt + 3, sc);
    compute(S ^ 1);
    __syncthreads();
  };
  int kt = 0;
#pragma unroll 1
  for (; kt + 1 < nk; kt += 2) {
    step(kt, S1{});
    step(kt + 1, S0{});
  }
  if (kt < nk) step(kt, S1{});
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * 160 + j * 32 + li;
    if (col >= d.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = k0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row >= d.K) continue;
      atomicAdd(d.out + (int64_t)row * d.ld_out + col, d.alpha * acc[j][r]);
    }
  }
  if (do_cs) {
#pragma unroll
    for (int i = 0; i < NBT; ++i) {
      if (i == 2 && !third) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = cs[i][j];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        const int n = n0 + b_row[i] + j;
        if (pp == 0 && vb[i][0] != BUF_OOB && n < d.N) atomicAdd(d.colsum + n, v);
      }
    }
  }
}

static int wgrad_bf16x3(const ddpo_gemm_desc* dp, const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* b_hi, const uint16_t* b_lo,
                       void* stream) {
  if (!dp) return DDPO_EINVAL;
  ddpo_gemm_desc d = *dp;
  if ((!d.src && !a_hi) || (!d.w && !b_hi) || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0) return DDPO_EINVAL;
  if ((a_hi && !a_lo) || (b_hi && !b_lo)) return DDPO_EINVAL;
  if (d.colsum && b_hi) return DDPO_EINVAL;          // the fused bias gradient sums the fp32 dY registers
  if ((d.ld_src & 3) || (d.ld_w & 3) || (d.N & 3) || (d.K & 3)) return DDPO_EINVAL;
  if (!a_hi && (reinterpret_cast<uintptr_t>(d.src) & 15)) return DDPO_EINVAL;
  if (!b_hi && (reinterpret_cast<uintptr_t>(d.w) & 15)) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(a_hi) | reinterpret_cast<uintptr_t>(a_lo) | reinterpret_cast<uintptr_t>(b_hi) | reinterpret_cast<uintptr_t>(b_lo)) & 7)
    return DDPO_EINVAL;
  // ld == 0 marks a k-blocked PLANE operand (channels / 32, rows, 32): planes only, whole 32-channel blocks
  if (d.ld_src < 0 || d.ld_w < 0) return DDPO_EINVAL;
  if (d.ld_src == 0 && (!a_hi || ((d.ksize > 0 ? d.Cin : d.K) & 31))) return DDPO_EINVAL;
  if (d.ld_w == 0 && (!b_hi || (d.N & 31))) return DDPO_EINVAL;
  if (d.ksize > 0) {
    if (d.ksize != 1 && d.ksize != 3) return DDPO_EINVAL;
    if ((d.Cin & 3) || d.K != d.ksize * d.ksize * d.Cin || d.M != d.B * d.OH * d.OW) return DDPO_EINVAL;
    if (d.stride < 1 || d.stride > 2 || d.upsample < 0 || d.upsample > 1 || d.pad != d.ksize / 2) return DDPO_EINVAL;
    if ((int64_t)d.B * d.H * d.W * d.ld_src >= ((int64_t)1 << 40)) return DDPO_EINVAL;
  }
  const int tiles_m = (d.K + 127) / 128, tiles_n = (d.N + 127) / 128, tiles = tiles_m * tiles_n;
  int splits = d.splits;
  if (splits <= 0) {
    // two workgroups fit a CU (74 KB LDS): pick the split of the pixel reduction whose tiles x splits fills whole rounds
    // of the 512 slots best (1035 workgroups cost three rounds, 966 two); among near-equal fills prefer fewer splits
    // (fewer atomic adds, longer k-loops).
    const int max_splits = (d.M + 255) / 256;
    int best = 1;
    double best_eff = 0.0;
    for (int r = 1; r <= 4; ++r) {
      int cand = (512 * r) / tiles;
      if (cand > max_splits) cand = max_splits;
      if (cand < 1) cand = 1;
      const long wgs = (long)tiles * cand;
      const double eff = (double)wgs / (double)(((wgs + 511) / 512) * 512);
      if (eff > best_eff + 0.03 || (best_eff == 0.0)) { best_eff = eff; best = cand; }
    }
    splits = best;
  }
  int mps = (d.M + splits - 1) / splits;
  mps = (mps + 31) / 32 * 32;
  splits = (d.M + mps - 1) / mps;
  hipStream_t st = as_stream(stream);
  const dim3 grid(tiles, splits), blk(BF_THREADS);
  // row loader for the regular layers (the per-pixel loader takes the rest): conv 320->320 @ 64^2, U-Net batch 64:
  // 2.54 -> 2.05 ms (190 -> 236 TF), profiles/r02_ab_wgrad_rows.log
  constexpr int rows_mode = 1;
  const bool conv_ = d.ksize > 0;
  const bool simple_ = !conv_ || (d.stride == 1 && d.upsample == 0 && d.OH == d.H && d.OW == d.W);
  const int64_t lda_b = d.ld_src ? d.ld_src : (conv_ ? d.Cin : d.K), ldb_b = d.ld_w ? d.ld_w : d.N;      // k-blocked planes: the same bytes in all
  const int64_t a_bytes = (int64_t)d.M * lda_b * (a_hi ? 2 : 4), b_bytes = (int64_t)d.M * ldb_b * (b_hi ? 2 : 4);
  const bool rows_ok = rows_mode && simple_ && (d.M % 32) == 0 && a_bytes + (conv_ ? (int64_t)(d.W + 1) * lda_b * 4 : 0) < 0x7FFFFFF0ll && b_bytes < 0x7FFFFFF0ll &&
                       (!conv_ || (((d.OW % 32) == 0 || (32 % d.OW) == 0) && ((d.OH * d.OW) % 32) == 0));
  // wide 128 x 320 tile (one workgroup per CU) where it measured faster (tools/native/kernel_probe wgrad, profiles/r03_probe_wgrad.log): the
  // 320-column layers with a long k and many pixels — the 3x3 convolutions of the 64x64 level: 320->320 0.67 -> 0.48 ms (181 -> 254 TF) from
  // fp32 operands, 0.57 -> 0.45 from planes; 960->320 1.62 -> 1.23 / 1.44 -> 1.29; train step +0.6 % (profiles/r03_ab_wgrad_wide.log).  With two or more 320-column tiles, short reductions or
  // few pixels it ties or loses against two 128 x 128 workgroups per CU (LDS read bytes per MFMA of a 32 x 160 wave tile), so those stay there.
  const bool wide_ok = rows_ok && d.N == 320 && d.K >= 2560 && d.M >= 16384 && dp->splits <= 0;
  if (wide_ok) {
    const int wt = ((d.K + 127) / 128) * (d.N / 320);
    const int max_splits = (d.M + 255) / 256;
    int ws_ = 1;                                         // split of the pixel reduction whose tiles x splits fills whole rounds of the 256 CUs best
    double best_eff = 0.0;
    for (int r = 1; r <= 3; ++r) {
      int cand = (256 * r) / wt;
      if (cand > max_splits) cand = max_splits;
      if (cand < 1) cand = 1;
      const long wgs = (long)wt * cand;
      const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
      if (eff > best_eff + 0.04 || best_eff == 0.0) { best_eff = eff; ws_ = cand; }
    }
    int wmps = (d.M + ws_ - 1) / ws_;
    wmps = (wmps + 31) / 32 * 32;
    ws_ = (d.M + wmps - 1) / wmps;
    const size_t lds = (size_t)2 * (2 * 128 + 2 * 320) * WG_PITCH * 4;
    const dim3 wgrid(wt, ws_), wblk(512);
#define WGW_LAUNCH(A, B)                                                                                                           \
  do {                                                                                                                             \
    static bool attr_ = false;                                                                                                     \
    if (!attr_) {                                                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wgrad_bf16_wide_kernel<A, B>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                             \
      attr_ = true;                                                                                                                \
    }                                                                                                                              \
    hipLaunchKernelGGL((gemm_wgrad_bf16_wide_kernel<A, B>), wgrid, wblk, lds, st, d, d.N / 320, wmps, a_hi, a_lo, b_hi, b_lo);     \
  } while (0)
    if (a_hi && b_hi) WGW_LAUNCH(true, true);
    else if (a_hi) WGW_LAUNCH(true, false);
    else if (b_hi) WGW_LAUNCH(false, true);
    else WGW_LAUNCH(false, false);
#undef WGW_LAUNCH
    DDPO_LAUNCH_CHECK();
    return DDPO_OK;
  }
#define WG_LAUNCH(A, B)                                                                                                                        \
  do {                                                                                                                                         \
    if (rows_ok) hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<A, B, true>), grid, blk, 0, st, d, tiles_n, mps, a_hi, a_lo, b_hi, b_lo);          \
    else hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<A, B, false>), grid, blk, 0, st, d, tiles_n, mps, a_hi, a_lo, b_hi, b_lo);                 \
  } while (0)
  if (a_hi && b_hi) WG_LAUNCH(true, true);
  else if (a_hi) WG_LAUNCH(true, false);
  else if (b_hi) WG_LAUNCH(false, true);
  else WG_LAUNCH(false, false);
#undef WG_LAUNCH
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_gemm_conv_wgrad_bf16x3(const ddpo_gemm_desc* dp, void* stream) {
  return wgrad_bf16x3(dp, nullptr, nullptr, nullptr, nullptr, stream);
}

/* Same contraction with one or both operands pre-split into bf16 hi / lo planes (NULL pair = that operand is fp32 in the descriptor):
 * a_* replace d->src (row stride d->ld_src ELEMENTS), dy_* replace d->w (row stride d->ld_w elements). */
extern "C" int ddpo_gemm_conv_wgrad_bf16x3_planes(const ddpo_gemm_desc* dp, const uint16_t* a_hi, const uint16_t* a_lo,
                                                  const uint16_t* dy_hi, const uint16_t* dy_lo, void* stream) {
  if (!a_hi && !dy_hi) return DDPO_EINVAL;
  return wgrad_bf16x3(dp, a_hi, a_lo, dy_hi, dy_lo, stream);
}
