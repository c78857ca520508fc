// Implicit-GEMM convolution / dense GEMM on the exact-fp32 MFMA datapath of gfx950
// (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-equal to an fmaf chain; 157 TFLOP/s chip peak).
//
//   out[m][n] = alpha * sum_k A(m,k) W(k,n) + bias[n] + rowbias[m / rows_per_batch][n] + residual[m][n]
//
// A(m,k) is either a dense row-major matrix or the im2col view of an NHWC tensor gathered on the fly
// (k = (ky,kx,ci); zero padding, stride 1|2 and nearest-2x upsampling folded into the gather).
// Tiling: BMxBNx16 block tile, 4 waves (2x2), each wave (BM/2)x(BN/2) as 32x32 MFMA tiles; operands are staged
// through LDS k-major ([k][m], [k][n]) so every ds_read_b32 of a fragment is bank-conflict free; global loads
// are float4 along the contiguous dimension (ci for activations, n for HWIO weights) and are issued for tile
// t+1 before the MFMAs of tile t (register-staged double buffering, one barrier per k-tile).
// The 1-D grid is remapped so that the tiles sharing an A panel run on the same XCD (private L2).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GEMM_BK 16
#define GEMM_THREADS 256

struct RowInfo {          // per (thread, A-row) gather state
  int64_t base;           // dense: m*ld ; conv: b*H*W (pixel index base)
  int iy0, ix0;           // conv: oy*stride - pad, ox*stride - pad
  bool valid;
};

template <int BM, int BN, bool WTRANS>
__global__ void __launch_bounds__(GEMM_THREADS) gemm_conv_kernel(const ddpo_gemm_desc d, int tiles_n, int nblk) {
  constexpr int BK = GEMM_BK;
  constexpr int LDA = BM + 2;
  constexpr int LDB = WTRANS ? (BN + 2) : (BN + 4);
  constexpr int TM = BM / 64;            // 32x32 tiles per wave along m
  constexpr int TN = BN / 64;
  constexpr int AROWS = BM / 64;         // float4 chunks per thread for the A tile
  constexpr int BCH = BN / 64;           // float4 chunks per thread for the W tile
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware bijective remap: consecutive logical tiles (same A panel) land on one XCD
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const bool conv = d.ksize > 0;
  const int VH = d.upsample ? d.H * 2 : d.H, VW = d.upsample ? d.W * 2 : d.W;
  const bool zins = d.upsample == 2;     // zero-insert source (transposed conv = dgrad of a stride-2 conv)

  // ---- A loader state: thread owns k-quad kq and rows (t>>2) + 64*i
  const int kq = t & 3;
  RowInfo ri[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + (t >> 2) + 64 * i;
    ri[i].valid = m < d.M;
    if (conv) {
      const int ohw = d.OH * d.OW;
      const int mm = ri[i].valid ? m : 0;
      const int b = mm / ohw, rem = mm - b * ohw;
      const int oy = rem / d.OW, ox = rem - oy * d.OW;
      ri[i].base = (int64_t)b * d.H * d.W;
      ri[i].iy0 = oy * d.stride - d.pad;
      ri[i].ix0 = ox * d.stride - d.pad;
    } else {
      ri[i].base = (int64_t)m * d.ld_src;
      ri[i].iy0 = ri[i].ix0 = 0;
    }
  }
  // ---- W loader state
  const int bn4 = WTRANS ? 0 : (t % (BN / 4));           // direct: column quad
  const int bk0 = WTRANS ? 0 : (t / (BN / 4));           // direct: first k row (step 256/(BN/4))
  constexpr int BKSTEP = GEMM_THREADS / (BN / 4);

  float4 ra[AROWS], rb[BCH];

  auto load_tile = [&](int kt) {
    const int kg = kt * BK + kq * 4;                     // this thread's k offset for A (and W^T)
    int ky = 0, kx = 0, ci = kg;
    if (conv) {
      const int tap = kg / d.Cin;
      ci = kg - tap * d.Cin;
      ky = tap / d.ksize;
      kx = tap - ky * d.ksize;
    }
    const bool kval = kg < d.K;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (conv) {
        const int iy = ri[i].iy0 + ky, ix = ri[i].ix0 + kx;
        if (ri[i].valid && kval && iy >= 0 && iy < VH && ix >= 0 && ix < VW && !(zins && ((iy | ix) & 1))) {
          const int sy = d.upsample ? (iy >> 1) : iy, sx = d.upsample ? (ix >> 1) : ix;
          v = *reinterpret_cast<const float4*>(d.src + (ri[i].base + (int64_t)sy * d.W + sx) * d.ld_src + ci);
        }
      } else if (ri[i].valid && kval) {
        v = *reinterpret_cast<const float4*>(d.src + ri[i].base + kg);
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (WTRANS) {
        const int n = n0 + (t >> 2) + 64 * i;
        if (n < d.N && kval) {
          if (d.w_dgrad) {   // W is the forward HWIO kernel (taps, N=Cin_fwd, Cin=Cout_fwd): flipped tap, (ci,co) swapped
            const int tapf = d.ksize * d.ksize - 1 - (ky * d.ksize + kx);
            v = *reinterpret_cast<const float4*>(d.w + ((int64_t)tapf * d.N + n) * d.Cin + ci);
          } else {
            v = *reinterpret_cast<const float4*>(d.w + (int64_t)n * d.K + kg);
          }
        }
      } else {
        const int k = kt * BK + bk0 + BKSTEP * i;
        const int n = n0 + bn4 * 4;
        if (k < d.K && n < d.N) v = *reinterpret_cast<const float4*>(d.w + (int64_t)k * d.N + n);
      }
      rb[i] = v;
    }
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      float* p = &As[buf][(kq * 4) * LDA + (t >> 2) + 64 * i];
      p[0] = ra[i].x; p[LDA] = ra[i].y; p[2 * LDA] = ra[i].z; p[3 * LDA] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      if (WTRANS) {
        float* p = &Bs[buf][(kq * 4) * LDB + (t >> 2) + 64 * i];
        p[0] = rb[i].x; p[LDB] = rb[i].y; p[2 * LDB] = rb[i].z; p[3 * LDB] = rb[i].w;
      } else {
        *reinterpret_cast<float4*>(&Bs[buf][(bk0 + BKSTEP * i) * LDB + bn4 * 4]) = rb[i];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (d.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int a_off = wm * (BM / 2) + (lane & 31);
  const int b_off = wn * (BN / 2) + (lane & 31);
  const int khalf = lane >> 5;

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* as = As[cur];
    const float* bs = Bs[cur];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = kk * 2 + khalf;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[k * LDA + a_off + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[k * LDB + b_off + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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
// Weight gradient:  dW[k][n] (+)= sum_m A(m,k) * dY[m][n]   (k = (ky,kx,ci) for convs, reduction over the
// B*OH*OW output pixels).  The A tile is gathered transposed — float4 along ci (the OUTPUT row dimension here),
// one reduction index (pixel) per LDS row — and dY streams in directly; the long reduction is split across
// gridDim.y and combined with fp32 atomic adds, which also implements the gradient accumulation
// (AccumulatingTrainState: grad_acc += g) in place.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ void __launch_bounds__(GEMM_THREADS) gemm_wgrad_kernel(const ddpo_gemm_desc d, int tiles_n, int m_per_split) {
  constexpr int BK = GEMM_BK;
  constexpr int LDA = BM + 4;
  constexpr int LDB = BN + 4;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int APASS = (BK * BM / 4) / GEMM_THREADS;     // float4 chunks per thread (A tile)
  constexpr int BPASS = (BK * BN / 4) / GEMM_THREADS;
  constexpr int ARSTEP = GEMM_THREADS / (BM / 4);
  constexpr int BRSTEP = GEMM_THREADS / (BN / 4);
  __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int k0 = tile_m * BM, n0 = tile_n * BN;          // output rows are k (K = taps*Cin), columns n
  const int m_begin = blockIdx.y * m_per_split;
  const int m_end = min(m_begin + m_per_split, d.M);
  if (m_begin >= m_end) return;

  const bool conv = d.ksize > 0;
  const int VH = d.upsample ? d.H * 2 : d.H, VW = d.upsample ? d.W * 2 : d.W;
  const int ohw = conv ? d.OH * d.OW : 1;

  // A loader: fixed output-row quad per thread
  const int arow4 = t % (BM / 4), ar0 = t / (BM / 4);
  const int kg = k0 + arow4 * 4;
  const bool kvalid = kg < d.K;
  int ky = 0, kx = 0, ci = kg;
  if (conv) {
    const int tap = kg / d.Cin;
    ci = kg - tap * d.Cin;
    ky = tap / d.ksize;
    kx = tap - ky * d.ksize;
  }
  const int bn4 = t % (BN / 4), br0 = t / (BN / 4);

  float4 ra[APASS], rb[BPASS];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int m = m_begin + kt * BK + ar0 + ARSTEP * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kvalid && m < m_end) {
        if (conv) {
          const int b = m / ohw, rem = m - b * ohw;
          const int oy = rem / d.OW, ox = rem - oy * d.OW;
          const int iy = oy * d.stride - d.pad + ky, ix = ox * d.stride - d.pad + kx;
          if (iy >= 0 && iy < VH && ix >= 0 && ix < VW) {
            const int sy = d.upsample ? (iy >> 1) : iy, sx = d.upsample ? (ix >> 1) : ix;
            v = *reinterpret_cast<const float4*>(d.src + (((int64_t)b * d.H + sy) * d.W + sx) * d.ld_src + ci);
          }
        } else {
          v = *reinterpret_cast<const float4*>(d.src + (int64_t)m * d.ld_src + kg);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int m = m_begin + kt * BK + br0 + BRSTEP * i;
      const int n = n0 + bn4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < m_end && n < d.N) v = *reinterpret_cast<const float4*>(d.w + (int64_t)m * d.ld_w + n);
      rb[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) *reinterpret_cast<float4*>(&As[buf][(ar0 + ARSTEP * i) * LDA + arow4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *reinterpret_cast<float4*>(&Bs[buf][(br0 + BRSTEP * i) * LDB + bn4 * 4]) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (m_end - m_begin + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int a_off = wm * (BM / 2) + (lane & 31);
  const int b_off = wn * (BN / 2) + (lane & 31);
  const int khalf = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const float* as = As[cur];
    const float* bs = Bs[cur];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = kk * 2 + khalf;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[k * LDA + a_off + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bs[k * LDB + b_off + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      if (col >= d.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = k0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (row >= d.K) continue;
        float* o = d.out + (int64_t)row * d.ld_out + col;
        if (d.accumulate) atomicAdd(o, d.alpha * acc[i][j][r]);
        else *o = d.alpha * acc[i][j][r];
      }
    }
  }
}

template <int BM, int BN>
static int launch_wgrad(const ddpo_gemm_desc& d0, hipStream_t st) {
  ddpo_gemm_desc d = d0;
  const int tiles_m = (d.K + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int tiles = tiles_m * tiles_n;
  int splits = d.splits;
  if (splits <= 0) {
    splits = (1024 + tiles - 1) / tiles;
    const int max_splits = (d.M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  if (splits > 1) d.accumulate = 1;      // caller must have zero-initialised (or be accumulating into) out
  int mps = (d.M + splits - 1) / splits;
  mps = (mps + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
  splits = (d.M + mps - 1) / mps;
  hipLaunchKernelGGL((gemm_wgrad_kernel<BM, BN>), dim3(tiles, splits), dim3(GEMM_THREADS), 0, st, d, tiles_n, mps);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_gemm_conv_wgrad(const ddpo_gemm_desc* dp, void* stream) {
  if (!dp) return DDPO_EINVAL;
  const ddpo_gemm_desc& d = *dp;
  if (!d.src || !d.w || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0) return DDPO_EINVAL;
  if ((d.ld_src & 3) || (d.ld_w & 3) || (d.N & 3) || (d.K & 3)) return DDPO_EINVAL;
  if ((reinterpret_cast<uintptr_t>(d.src) | reinterpret_cast<uintptr_t>(d.w)) & 15) return DDPO_EINVAL;
  if (d.ksize > 0) {
    if (d.ksize != 1 && d.ksize != 3) return DDPO_EINVAL;
    if ((d.Cin & 3) || d.K != d.ksize * d.ksize * d.Cin || d.M != d.B * d.OH * d.OW || d.upsample > 1) return DDPO_EINVAL;
  }
  if (d.splits != 1 && !d.accumulate && d.splits != 0) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  const long t128 = (long)((d.K + 127) / 128) * ((d.N + 127) / 128);
  if (d.N % 128 == 0 && d.K >= 128 && t128 >= 16) return launch_wgrad<128, 128>(d, st);
  if (d.N > 32 && d.K >= 128) return launch_wgrad<128, 64>(d, st);
  return launch_wgrad<64, 64>(d, st);
}

template <int BM, int BN>
static int launch_cfg(const ddpo_gemm_desc& d, hipStream_t st) {
  const int tiles_m = (d.M + BM - 1) / BM, tiles_n = (d.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  if (d.w_trans)
    hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, true>), dim3(nblk), dim3(GEMM_THREADS), 0, st, d, tiles_n, nblk);
  else
    hipLaunchKernelGGL((gemm_conv_kernel<BM, BN, false>), dim3(nblk), dim3(GEMM_THREADS), 0, st, d, tiles_n, nblk);
  DDPO_LAUNCH_CHECK();
  return DDPO_OK;
}

extern "C" int ddpo_gemm_conv_fwd(const ddpo_gemm_desc* dp, void* stream) {
  if (!dp) return DDPO_EINVAL;
  const ddpo_gemm_desc& d = *dp;
  if (!d.src || !d.w || !d.out || d.M <= 0 || d.N <= 0 || d.K <= 0) return DDPO_EINVAL;
  if (d.epilogue != 0 || d.out_hi || d.out_lo) return DDPO_EINVAL;     // fused / plane-emitting output stages exist on the bf16 datapath only
  if ((d.ld_src & 3) || (reinterpret_cast<uintptr_t>(d.src) & 15) || (reinterpret_cast<uintptr_t>(d.w) & 15)) return DDPO_EINVAL;
  if (d.ksize > 0) {
    if (d.ksize != 1 && d.ksize != 3) return DDPO_EINVAL;
    if ((d.Cin & 3) || d.K != d.ksize * d.ksize * d.Cin || d.M != d.B * d.OH * d.OW) return DDPO_EINVAL;
    if (d.stride < 1 || d.pad < 0 || d.B <= 0 || d.H <= 0 || d.W <= 0) return DDPO_EINVAL;
  } else if (d.K & 3) {
    return DDPO_EINVAL;
  }
  if (d.w_trans ? (d.K & 3) : (d.N & 3)) return DDPO_EINVAL;
  if (d.w_dgrad && (!d.w_trans || d.ksize <= 0)) return DDPO_EINVAL;
  if (d.upsample < 0 || d.upsample > 2) return DDPO_EINVAL;
  if (d.rowbias && d.rows_per_batch <= 0) return DDPO_EINVAL;
  hipStream_t st = as_stream(stream);
  // tile choice: big tiles when they still fill the chip (256 CUs), smaller ones for small problems
  const long t128 = (long)((d.M + 127) / 128) * ((d.N + 127) / 128);
  const long t12864 = (long)((d.M + 127) / 128) * ((d.N + 63) / 64);
  if (d.N % 128 == 0 && t128 >= 512) return launch_cfg<128, 128>(d, st);
  if (d.N > 32 && t12864 >= 512) return launch_cfg<128, 64>(d, st);
  if (d.N % 128 == 0 && t128 >= 256) return launch_cfg<128, 128>(d, st);
  return launch_cfg<64, 64>(d, st);
}
