// Depthwise Conv1d over time on the dense (B, T, C) layout, optionally fused with the GLU that
// precedes it and with the per-channel sum / sum-of-squares BatchNorm needs after it.
// Replaces: torchaudio Conformer conv module's GLU + depthwise Conv1d(k) (+ the statistics pass of
// BatchNorm1d) (call sites fs2/model.py:193, :241) and the depthwise half of
// DepthwiseSeparableConv1d (fs2/blocks.py:8-13).  HBM-bound: each input element is read once
// (+ halo from L2), lanes run along channels so every row access is a contiguous 256-byte line.
#include "common.h"

namespace {

constexpr int RUN = 16;  // consecutive time steps per thread

// element loads / stores of a tensor that is fp32 (XB = false) or bf16 (XB = true) in memory
__device__ __forceinline__ unsigned short dw_bf16(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);  // round to nearest even, NaN stays NaN
}
template <bool XB>
__device__ __forceinline__ float dw_ld(const void* p, long long i) {
  if constexpr (XB) return __builtin_bit_cast(float, (unsigned)((const unsigned short*)p)[i] << 16);
  else return ((const float*)p)[i];
}

// x: [B*T][ldx]; value at column c, gate (GLU) at column C + c.  XB: x and y are bf16 tensors (precision "bf16-mixed"
// with bf16 activation storage: what autocast hands a convolution); the statistics are those of the ROUNDED outputs,
// which is what BatchNorm then normalises.
template <int K, bool GLU, bool STATS, bool XB = false, bool YB = XB>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const void* __restrict__ x, int ldx,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          void* __restrict__ y, float* __restrict__ partial, int B,
                                                          int T, int C) {
  constexpr int PAD = (K - 1) / 2, WIN = RUN + K - 1;
  __shared__ float red[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int t0 = blockIdx.y * (4 * RUN) + wave * RUN;
  const int b = blockIdx.z;
  const bool cok = c < C;
  float wk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wk[k] = cok ? w[k * C + c] : 0.f;
  const float bs = (cok && bias) ? bias[c] : 0.f;
  float a[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) {
    int t = t0 - PAD + i;
    float v = 0.f;
    if (cok && t >= 0 && t < T) {
      const long long row = ((long long)b * T + t) * ldx;
      v = dw_ld<XB>(x, row + c);
      if (GLU) v *= fs2_sigmoid(dw_ld<XB>(x, row + C + c));
    }
    a[i] = v;
  }
  // BatchNorm statistics of this thread's run as (n, mean, M2): sums of (y - pivot), pivot = the run's first output
  // (bn.hip explains why not sum / sum of squares)
  float s1 = 0.f, s2 = 0.f, pivot = 0.f;
#pragma unroll
  for (int o = 0; o < RUN; ++o) {
    int t = t0 + o;
    if (cok && t < T) {
      float acc = bs;
#pragma unroll
      for (int k = 0; k < K; ++k) acc = fmaf(wk[k], a[o + k], acc);
      if constexpr (YB) {
        const unsigned short r = dw_bf16(acc);
        ((unsigned short*)y)[((long long)b * T + t) * C + c] = r;
        acc = __builtin_bit_cast(float, (unsigned)r << 16);
      } else {
        ((float*)y)[((long long)b * T + t) * C + c] = acc;
      }
      if (STATS) {
        if (o == 0) pivot = acc;
        const float d = acc - pivot;
        s1 += d;
        s2 += d * d;
      }
    }
  }
  if (STATS) {
    const int nw = max(0, min(RUN, T - t0));  // rows of this wavefront's run (uniform over its lanes)
    const float inv = nw > 0 ? 1.f / (float)nw : 0.f;
    red[wave][0][lane] = pivot + s1 * inv;
    red[wave][1][lane] = fmaxf(s2 - s1 * s1 * inv, 0.f);
    __syncthreads();
    if (wave == 0 && cok) {
      float n = 0.f, mean = 0.f, m2 = 0.f;  // Chan merge of the four runs
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float nr = (float)max(0, min(RUN, T - (t0 + w * RUN)));
        if (nr > 0.f) {
          const float delta = red[w][0][lane] - mean, nt = n + nr;
          mean += delta * (nr / nt);
          m2 += red[w][1][lane] + delta * delta * (n * nr / nt);
          n = nt;
        }
      }
      long long blk = (long long)b * gridDim.y + blockIdx.y;
      partial[(blk * 2 + 0) * C + c] = mean;
      partial[(blk * 2 + 1) * C + c] = m2;
    }
  }
}

// dy [B*T][C]; x as in the forward; dx has the layout of x.  partial [blk][K+1][C]: dw taps, dbias.
// DXB: dx is written as bf16 (the operand of the pointwise convolution's data- and weight-gradient GEMMs)
// XB: dy and x are bf16 tensors
template <int K, bool GLU, bool DXB = false, bool XB = false>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                          int ldx, const float* __restrict__ w, void* __restrict__ dxv,
                                                          float* __restrict__ partial, int B, int T, int C) {
  constexpr int PAD = (K - 1) / 2, WIN = RUN + K - 1;
  __shared__ float red[4][K + 1][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int t0 = blockIdx.y * (4 * RUN) + wave * RUN;
  const int b = blockIdx.z;
  const bool cok = c < C;
  float wk[K], dwk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    wk[k] = cok ? w[k * C + c] : 0.f;
    dwk[k] = 0.f;
  }
  float a[WIN], g[WIN], vc[RUN], sc[RUN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) {
    int t = t0 - PAD + i;
    float v = 0.f, d = 0.f, vv = 0.f, sg = 0.f;
    if (cok && t >= 0 && t < T) {
      const long long row = ((long long)b * T + t) * ldx;
      vv = dw_ld<XB>(x, row + c);
      v = vv;
      if (GLU) {
        sg = fs2_sigmoid(dw_ld<XB>(x, row + C + c));
        v = vv * sg;
      }
      d = dw_ld<XB>(dy, ((long long)b * T + t) * C + c);
    }
    a[i] = v;
    g[i] = d;
    if (i >= PAD && i < PAD + RUN) {
      vc[i - PAD] = vv;
      sc[i - PAD] = sg;
    }
  }
  float db = 0.f;
#pragma unroll
  for (int o = 0; o < RUN; ++o) {
    int t = t0 + o;
    // da[t] = sum_k w[k] * dy[t - k + PAD]
    float da = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) da = fmaf(wk[k], g[o + (K - 1 - k)], da);
    const float gy = g[o + PAD];
    db += gy;
#pragma unroll
    for (int k = 0; k < K; ++k) dwk[k] = fmaf(gy, a[o + k], dwk[k]);
    if (cok && t < T) {
      if constexpr (DXB) {
        unsigned short* row = (unsigned short*)dxv + ((long long)b * T + t) * ldx;
        if (GLU) {
          row[c] = dw_bf16(da * sc[o]);
          row[C + c] = dw_bf16(da * vc[o] * sc[o] * (1.f - sc[o]));
        } else {
          row[c] = dw_bf16(da);
        }
      } else {
        float* row = (float*)dxv + ((long long)b * T + t) * ldx;
        if (GLU) {
          row[c] = da * sc[o];
          row[C + c] = da * vc[o] * sc[o] * (1.f - sc[o]);
        } else {
          row[c] = da;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) red[wave][k][lane] = dwk[k];
  red[wave][K][lane] = db;
  __syncthreads();
  if (wave == 0 && cok) {
    long long blk = (long long)b * gridDim.y + blockIdx.y;
#pragma unroll
    for (int k = 0; k <= K; ++k)
      partial[(blk * (K + 1) + k) * C + c] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
  }
}

// ---- tiled GLU forms ------------------------------------------------------------------------------------------------
// The kernels above give every thread its own window of rows: one 2- or 4-byte element per lane and load instruction,
// and the GLU (a sigmoid) of the K - 1 halo rows is recomputed by each of the four wavefronts of a workgroup.  Measured
// at the decoder's shape they run at 2 TB/s whether the tensors are fp32 or bf16 -- bound by load instructions, not
// bytes.  Here a workgroup first brings its (4 RUN + K - 1) rows x 64 channels through LDS: 16 bytes per lane and load
// (eight channels of one row), the GLU applied once per element, a = value * sigmoid(gate) kept as fp32; then every
// thread runs the same per-channel loop as above with its window read from LDS.  Same arithmetic in the same order:
// the results are the bits of the kernels above.  Needs C % 64 == 0 and ldx % 8 == 0.
template <bool XB>
__device__ __forceinline__ void dw_ld8(const void* p, long long e, float (&v)[8]) {
  if constexpr (XB) {
    const uint4 u = *reinterpret_cast<const uint4*>((const unsigned short*)p + e);
    const unsigned q[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __builtin_bit_cast(float, q[j] << 16);
      v[2 * j + 1] = __builtin_bit_cast(float, q[j] & 0xffff0000u);
    }
  } else {
    const float4 a = *reinterpret_cast<const float4*>((const float*)p + e);
    const float4 b = *reinterpret_cast<const float4*>((const float*)p + e + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
__device__ __forceinline__ void dw_st8(float* dst, const float (&v)[8]) {
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// eight consecutive channels of a staged result row -> global memory, as fp32 (32 bytes) or bf16 (16 bytes)
template <bool OB>
__device__ __forceinline__ void dw_st8_global(void* dst, long long e, const float* src) {
  const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
  if constexpr (OB) {
    uint4 u;
    u.x = (unsigned)dw_bf16(a.x) | ((unsigned)dw_bf16(a.y) << 16);
    u.y = (unsigned)dw_bf16(a.z) | ((unsigned)dw_bf16(a.w) << 16);
    u.z = (unsigned)dw_bf16(b.x) | ((unsigned)dw_bf16(b.y) << 16);
    u.w = (unsigned)dw_bf16(b.z) | ((unsigned)dw_bf16(b.w) << 16);
    *reinterpret_cast<uint4*>((unsigned short*)dst + e) = u;
  } else {
    *reinterpret_cast<float4*>((float*)dst + e) = a;
    *reinterpret_cast<float4*>((float*)dst + e + 4) = b;
  }
}

template <int K, bool STATS, bool XB>
__global__ __launch_bounds__(256) void dwconv_glu_fwd_tile_kernel(const void* __restrict__ x, int ldx,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ bias, void* __restrict__ y,
                                                                   float* __restrict__ partial, int B, int T, int C) {
  constexpr int PAD = (K - 1) / 2, WIN = RUN + K - 1, TT = 4 * RUN, ROWS = TT + K - 1;
  __shared__ __attribute__((aligned(16))) float as[ROWS][64];
  __shared__ __attribute__((aligned(16))) float ys[TT][64];  // results, staged for whole-row stores
  __shared__ float red[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 64, c = c0 + lane;
  const int tbase = blockIdx.y * TT, t0 = tbase + wave * RUN;
  const int b = blockIdx.z;
  {
    const int sub = tid & 7;
#pragma unroll
    for (int r = tid >> 3; r < ROWS; r += 32) {
      const int t = tbase - PAD + r;
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (t >= 0 && t < T) {
        const long long row = ((long long)b * T + t) * ldx + c0 + 8 * sub;
        float v8[8], g8[8];
        dw_ld8<XB>(x, row, v8);
        dw_ld8<XB>(x, row + C, g8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a8[j] = v8[j] * fs2_sigmoid(g8[j]);
      }
      dw_st8(&as[r][8 * sub], a8);
    }
  }
  float wk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wk[k] = w[k * C + c];
  const float bs = bias ? bias[c] : 0.f;
  __syncthreads();
  float a[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) a[i] = as[wave * RUN + i][lane];
  float s1 = 0.f, s2 = 0.f, pivot = 0.f;
#pragma unroll
  for (int o = 0; o < RUN; ++o) {
    int t = t0 + o;
    if (t < T) {
      float acc = bs;
#pragma unroll
      for (int k = 0; k < K; ++k) acc = fmaf(wk[k], a[o + k], acc);
      if constexpr (XB) acc = __builtin_bit_cast(float, (unsigned)dw_bf16(acc) << 16);  // (the stored value)
      ys[wave * RUN + o][lane] = acc;
      if (STATS) {
        if (o == 0) pivot = acc;
        const float d = acc - pivot;
        s1 += d;
        s2 += d * d;
      }
    }
  }
  if (STATS) {
    const int nw = max(0, min(RUN, T - t0));
    const float inv = nw > 0 ? 1.f / (float)nw : 0.f;
    red[wave][0][lane] = pivot + s1 * inv;
    red[wave][1][lane] = fmaxf(s2 - s1 * s1 * inv, 0.f);
  }
  __syncthreads();
  {
    const int sub = tid & 7;
#pragma unroll
    for (int r = tid >> 3; r < TT; r += 32) {
      const int t = tbase + r;
      if (t < T) dw_st8_global<XB>(y, ((long long)b * T + t) * C + c0 + 8 * sub, &ys[r][8 * sub]);
    }
  }
  if (STATS) {
    if (wave == 0) {
      float n = 0.f, mean = 0.f, m2 = 0.f;  // Chan merge of the four runs (as above)
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        const float nr = (float)max(0, min(RUN, T - (tbase + w4 * RUN)));
        if (nr > 0.f) {
          const float delta = red[w4][0][lane] - mean, nt = n + nr;
          mean += delta * (nr / nt);
          m2 += red[w4][1][lane] + delta * delta * (n * nr / nt);
          n = nt;
        }
      }
      long long blk = (long long)b * gridDim.y + blockIdx.y;
      partial[(blk * 2 + 0) * C + c] = mean;
      partial[(blk * 2 + 1) * C + c] = m2;
    }
  }
}

template <int K, bool XB>
__global__ __launch_bounds__(256) void dwconv_glu_bwd_tile_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                                   int ldx, const float* __restrict__ w,
                                                                   void* __restrict__ dxv, float* __restrict__ partial,
                                                                   int B, int T, int C, int dx_b) {
  constexpr int PAD = (K - 1) / 2, WIN = RUN + K - 1, TT = 4 * RUN, ROWS = TT + K - 1;
  __shared__ __attribute__((aligned(16))) float as[ROWS][64];  // value * sigmoid(gate)
  __shared__ __attribute__((aligned(16))) float gs[ROWS][64];  // dy
  __shared__ __attribute__((aligned(16))) float vs[TT][64];    // value, centre rows
  __shared__ __attribute__((aligned(16))) float ss[TT][64];    // sigmoid(gate), centre rows
  __shared__ float red[4][K + 1][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 64, c = c0 + lane;
  const int tbase = blockIdx.y * TT, t0 = tbase + wave * RUN;
  const int b = blockIdx.z;
  {
    const int sub = tid & 7;
#pragma unroll
    for (int r = tid >> 3; r < ROWS; r += 32) {
      const int t = tbase - PAD + r;
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float v8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (t >= 0 && t < T) {
        const long long row = ((long long)b * T + t) * ldx + c0 + 8 * sub;
        float g8[8];
        dw_ld8<XB>(x, row, v8);
        dw_ld8<XB>(x, row + C, g8);
        dw_ld8<XB>(dy, ((long long)b * T + t) * C + c0 + 8 * sub, d8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s8[j] = fs2_sigmoid(g8[j]);
          a8[j] = v8[j] * s8[j];
        }
      }
      dw_st8(&as[r][8 * sub], a8);
      dw_st8(&gs[r][8 * sub], d8);
      if (r >= PAD && r < PAD + TT) {
        dw_st8(&vs[r - PAD][8 * sub], v8);
        dw_st8(&ss[r - PAD][8 * sub], s8);
      }
    }
  }
  float wk[K], dwk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    wk[k] = w[k * C + c];
    dwk[k] = 0.f;
  }
  __syncthreads();
  float a[WIN], g[WIN];
#pragma unroll
  for (int i = 0; i < WIN; ++i) {
    a[i] = as[wave * RUN + i][lane];
    g[i] = gs[wave * RUN + i][lane];
  }
  float db = 0.f;
#pragma unroll
  for (int o = 0; o < RUN; ++o) {
    int t = t0 + o;
    float da = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) da = fmaf(wk[k], g[o + (K - 1 - k)], da);
    const float gy = g[o + PAD];
    db += gy;
#pragma unroll
    for (int k = 0; k < K; ++k) dwk[k] = fmaf(gy, a[o + k], dwk[k]);
    {  // the two halves of dx replace the value / sigmoid this thread has just read (its own rows, its own column)
      const float vc = vs[wave * RUN + o][lane], sc = ss[wave * RUN + o][lane];
      vs[wave * RUN + o][lane] = da * sc;
      ss[wave * RUN + o][lane] = da * vc * sc * (1.f - sc);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) red[wave][k][lane] = dwk[k];
  red[wave][K][lane] = db;
  __syncthreads();
  {
    const int sub = tid & 7;
#pragma unroll
    for (int r = tid >> 3; r < TT; r += 32) {
      const int t = tbase + r;
      if (t < T) {
        const long long row = ((long long)b * T + t) * ldx + c0 + 8 * sub;
        if (dx_b) {
          dw_st8_global<true>(dxv, row, &vs[r][8 * sub]);
          dw_st8_global<true>(dxv, row + C, &ss[r][8 * sub]);
        } else {
          dw_st8_global<false>(dxv, row, &vs[r][8 * sub]);
          dw_st8_global<false>(dxv, row + C, &ss[r][8 * sub]);
        }
      }
    }
  }
  if (wave == 0) {
    long long blk = (long long)b * gridDim.y + blockIdx.y;
#pragma unroll
    for (int k = 0; k <= K; ++k)
      partial[(blk * (K + 1) + k) * C + c] = red[0][k][lane] + red[1][k][lane] + red[2][k][lane] + red[3][k][lane];
  }
}

}  // namespace

extern "C" int fs2hip_dwconv_blocks(int B, int T) { return B * ((T + 4 * RUN - 1) / (4 * RUN)); }
extern "C" int fs2hip_dwconv_part_rows(void) { return 4 * RUN; }

#define DW_FWD(KK)                                                                                              \
  if (io_bf16 == 2) { /* fp32 in, bf16 out: the variance predictors' plain depthwise layer feeding a bf16-storage GEMM */ \
    if (glu || stats) return FS2HIP_EINVAL;                                                                     \
    dwconv_fwd_kernel<KK, false, false, false, true><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C); \
  } else if (io_bf16) {                                                                                         \
    if (!glu) return FS2HIP_EINVAL; /* bf16 tensors: the Conformer convolution module's GLU form only */         \
    if (stats) dwconv_fwd_kernel<KK, true, true, true><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C); \
    else dwconv_fwd_kernel<KK, true, false, true><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C);      \
  } else if (glu && stats) dwconv_fwd_kernel<KK, true, true><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C); \
  else if (glu) dwconv_fwd_kernel<KK, true, false><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C);   \
  else if (stats) dwconv_fwd_kernel<KK, false, true><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C); \
  else dwconv_fwd_kernel<KK, false, false><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C);

extern "C" int fs2hip_dwconv_fwd_b(const void* x, int ldx, const float* w, const float* bias, void* y, float* partial,
                                   int B, int T, int C, int K, int glu, int stats, int io_bf16, void* stream);
extern "C" int fs2hip_dwconv_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, float* partial,
                                 int B, int T, int C, int K, int glu, int stats, void* stream) {
  return fs2hip_dwconv_fwd_b(x, ldx, w, bias, y, partial, B, T, C, K, glu, stats, 0, stream);
}

// io_bf16 = 1: x and y are bf16 tensors (GLU form only); 2: x fp32, y bf16 (plain form, no statistics)
extern "C" int fs2hip_dwconv_fwd_b(const void* x, int ldx, const float* w, const float* bias, void* y, float* partial,
                                   int B, int T, int C, int K, int glu, int stats, int io_bf16, void* stream) {
  if (B <= 0 || T <= 0 || C <= 0 || ldx < (glu ? 2 * C : C)) return FS2HIP_EINVAL;
  if (stats && !partial) return FS2HIP_EINVAL;
  dim3 grid((C + 63) / 64, (T + 4 * RUN - 1) / (4 * RUN), B);
  hipStream_t s = (hipStream_t)stream;
  const char* tile_env = getenv("FS2_DWCONV_TILE");  // "0": the per-thread-window kernels everywhere (measurement aid, tests)
  const bool tiles_off = tile_env && atoi(tile_env) == 0;
  if (io_bf16 != 0 && io_bf16 != 1 && io_bf16 != 2) return FS2HIP_EINVAL;
  if (io_bf16 == 2 && (glu || stats)) return FS2HIP_EINVAL;
  if (glu && !tiles_off && (C % 64) == 0 && (ldx % 8) == 0 && ((uintptr_t)x % 16) == 0) {
#define DW_FWD_T(KK)                                                                                              \
  if (io_bf16 && stats) dwconv_glu_fwd_tile_kernel<KK, true, true><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C);        \
  else if (io_bf16) dwconv_glu_fwd_tile_kernel<KK, false, true><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C);          \
  else if (stats) dwconv_glu_fwd_tile_kernel<KK, true, false><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C);            \
  else dwconv_glu_fwd_tile_kernel<KK, false, false><<<grid, dim3(256), 0, s>>>(x, ldx, w, bias, y, partial, B, T, C);
    switch (K) {
      case 3: DW_FWD_T(3) break;
      case 5: DW_FWD_T(5) break;
      case 7: DW_FWD_T(7) break;
      case 9: DW_FWD_T(9) break;
      case 15: DW_FWD_T(15) break;
      case 31: DW_FWD_T(31) break;
      default: return FS2HIP_EINVAL;
    }
#undef DW_FWD_T
    FS2_LAUNCH_CHECK();
    return 0;
  }
  switch (K) {
    case 3: DW_FWD(3) break;
    case 5: DW_FWD(5) break;
    case 7: DW_FWD(7) break;
    case 9: DW_FWD(9) break;
    case 15: DW_FWD(15) break;
    case 31: DW_FWD(31) break;
    default: return FS2HIP_EINVAL;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

#define DW_BWD(KK)                                                                                                \
  if (dx_bf16 & 2) {                                                                                              \
    if (!glu || !(dx_bf16 & 1)) return FS2HIP_EINVAL; /* bf16 inputs: GLU form with a bf16 result only */          \
    dwconv_bwd_kernel<KK, true, true, true><<<grid, dim3(256), 0, s>>>(dy, x, ldx, w, dx, partial, B, T, C);       \
  } else if (dx_bf16) {                                                                                           \
    if (glu) dwconv_bwd_kernel<KK, true, true><<<grid, dim3(256), 0, s>>>(dy, x, ldx, w, dx, partial, B, T, C);    \
    else dwconv_bwd_kernel<KK, false, true><<<grid, dim3(256), 0, s>>>(dy, x, ldx, w, dx, partial, B, T, C);       \
  } else {                                                                                                        \
    if (glu) dwconv_bwd_kernel<KK, true><<<grid, dim3(256), 0, s>>>(dy, x, ldx, w, dx, partial, B, T, C);          \
    else dwconv_bwd_kernel<KK, false><<<grid, dim3(256), 0, s>>>(dy, x, ldx, w, dx, partial, B, T, C);             \
  }

// partial: [fs2hip_dwconv_blocks(B,T)][K+1][C]; dw [K][C], dbias [C] are finished here
extern "C" int fs2hip_dwconv_bwd_b(const void* dy, const void* x, int ldx, const float* w, void* dx, int dx_bf16,
                                   float* partial, float* dw, float* dbias, int B, int T, int C, int K, int glu,
                                   void* stream);
extern "C" int fs2hip_dwconv_bwd(const float* dy, const float* x, int ldx, const float* w, float* dx, float* partial,
                                 float* dw, float* dbias, int B, int T, int C, int K, int glu, void* stream) {
  return fs2hip_dwconv_bwd_b(dy, x, ldx, w, dx, 0, partial, dw, dbias, B, T, C, K, glu, stream);
}

// dx_bf16 bit 0: dx (layout of x, leading dimension ldx) is written as bf16; bit 1: dy and x are bf16 tensors
extern "C" int fs2hip_dwconv_bwd_b(const void* dy, const void* x, int ldx, const float* w, void* dx, int dx_bf16,
                                   float* partial, float* dw, float* dbias, int B, int T, int C, int K, int glu,
                                   void* stream) {
  if (B <= 0 || T <= 0 || C <= 0 || ldx < (glu ? 2 * C : C) || !partial) return FS2HIP_EINVAL;
  dim3 grid((C + 63) / 64, (T + 4 * RUN - 1) / (4 * RUN), B);
  hipStream_t s = (hipStream_t)stream;
  const char* tile_env = getenv("FS2_DWCONV_TILE");  // "0": the per-thread-window kernels everywhere (measurement aid, tests)
  const bool tiles_off = tile_env && atoi(tile_env) == 0;
  if (glu && !tiles_off && (C % 64) == 0 && (ldx % 8) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0) {
    if ((dx_bf16 & 2) && !(dx_bf16 & 1)) return FS2HIP_EINVAL;
#define DW_BWD_T(KK)                                                                                              \
  if (dx_bf16 & 2) dwconv_glu_bwd_tile_kernel<KK, true><<<grid, dim3(256), 0, s>>>(dy, x, ldx, w, dx, partial, B, T, C, 1); \
  else dwconv_glu_bwd_tile_kernel<KK, false><<<grid, dim3(256), 0, s>>>(dy, x, ldx, w, dx, partial, B, T, C, dx_bf16 & 1);
    switch (K) {
      case 3: DW_BWD_T(3) break;
      case 5: DW_BWD_T(5) break;
      case 7: DW_BWD_T(7) break;
      case 9: DW_BWD_T(9) break;
      case 15: DW_BWD_T(15) break;
      case 31: DW_BWD_T(31) break;
      default: return FS2HIP_EINVAL;
    }
#undef DW_BWD_T
  } else {
    switch (K) {
      case 3: DW_BWD(3) break;
      case 5: DW_BWD(5) break;
      case 7: DW_BWD(7) break;
      case 9: DW_BWD(9) break;
      case 15: DW_BWD(15) break;
      case 31: DW_BWD(31) break;
      default: return FS2HIP_EINVAL;
    }
  }
  FS2_LAUNCH_CHECK();
  if (!dw) return 0;  // partial sums only: the caller finishes them with fs2hip_reduce_rows_multi (same sums, same order)
  const int nblk = fs2hip_dwconv_blocks(B, T);
  const long long stride = (long long)(K + 1) * C;
  // (one launch for both when they take the row-parallel path of fs2hip_reduce_slabs anyway: same sums, same order)
  if (dbias && stride <= 16384 && nblk >= 8)
    return fs2_reduce_rows(partial, nblk, (int)stride, stride, dw, K * C, dbias, (hipStream_t)stream);
  int rc = fs2hip_reduce_slabs(partial, dw, (long long)K * C, nblk, stride, stream);
  if (rc) return rc;
  if (dbias) rc = fs2hip_reduce_slabs(partial + (long long)K * C, dbias, C, nblk, stride, stream);
  return rc;
}
