// LayerNorm forward/backward: one 64-lane wavefront per row, float4 loads, wavefront-shuffle
// reductions (no LDS on the row statistics).  HBM-bound: forward reads x once and writes y once.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int LN_MAX_CH = 4;           // C <= 1024: at most 4 float4 chunks per lane
constexpr int LN_BWD_ROWS = 32;        // rows per workgroup in the backward (4 waves x 8 rows) -- of LONG matrices:
// a wavefront walks its rows one after the other (load, two wave reductions, store: a dependent chain of ~1.5 us per row),
// so with 32 rows per workgroup the 4 096 rows of the encoder / the variance predictors are 128 workgroups on 256 CUs,
// each wavefront 8 rows deep: 13 us for a kernel whose bytes take 3.  Short matrices take fewer rows per workgroup
// (more workgroups, shallower chains); the price is more partial rows for the batched second stage, bounded at 512.
// FS2_LN_BWD_ROWS=8|16|32 (measurement aid): one value for every row count
static inline int ln_bwd_rows(int M) {
  static const int forced = getenv("FS2_LN_BWD_ROWS") ? atoi(getenv("FS2_LN_BWD_ROWS")) : 0;
  if (forced == 8 || forced == 16 || forced == 32) return forced;
  return M > 16384 ? LN_BWD_ROWS : (M > 4096 ? 16 : 8);
}

__device__ __forceinline__ uint2 ln_pack_bf16(float4 o) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  typedef float f32v4 __attribute__((ext_vector_type(4)));
  const f32v4 v = {o.x, o.y, o.z, o.w};
  return __builtin_bit_cast(uint2, __builtin_convertvector(v, bf16x4));  // round to nearest even
}
__device__ __forceinline__ float4 ln_unpack_bf16(uint2 u) {
  return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
}

// YB: y is bf16 (the operand a bf16-storage GEMM reads: the normalised activations exist only in that form)
// DROP: y = dropout(LayerNorm(x)) -- the variance predictors' Conv -> ReLU -> LayerNorm -> Dropout layers
// (fs2/layers.py:30-48): the mask is drawn at the element index row * C + c, as the separate pass over y drew it
template <int NCH, bool YB, bool DROP = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, void* __restrict__ y,
                                                      float* __restrict__ mean, float* __restrict__ rstd, int M,
                                                      int C, float eps, Fs2Drop drop_in = Fs2Drop{}) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  Fs2Drop drop = drop_in;
  if constexpr (DROP) drop = fs2_resolve_drop(drop_in);
  const int c4 = C >> 2;
  float4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int i = lane + 64 * j;
    v[j] = i < c4 ? reinterpret_cast<const float4*>(x + (long long)row * C)[i] : make_float4(0, 0, 0, 0);
    s += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  const float mu = fs2_wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int i = lane + 64 * j;
    if (i < c4) {
      float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, d = v[j].w - mu;
      q += a * a + b * b + c * c + d * d;
    }
  }
  const float rs = rsqrtf(fs2_wave_sum(q) / C + eps);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int i = lane + 64 * j;
    if (i < c4) {
      float4 g = reinterpret_cast<const float4*>(gamma)[i];
      float4 b = reinterpret_cast<const float4*>(beta)[i];
      float4 o;
      o.x = (v[j].x - mu) * rs * g.x + b.x;
      o.y = (v[j].y - mu) * rs * g.y + b.y;
      o.z = (v[j].z - mu) * rs * g.z + b.z;
      o.w = (v[j].w - mu) * rs * g.w + b.w;
      if constexpr (DROP) {
        float f[4];  // (C is a multiple of 4: row * C + 4 i is even, the quad is two whole hash pairs)
        fs2_drop_quad(drop, (unsigned)(row * C + 4 * i), f);
        o.x *= f[0]; o.y *= f[1]; o.z *= f[2]; o.w *= f[3];
      }
      if constexpr (YB) reinterpret_cast<uint2*>((unsigned short*)y + (long long)row * C)[i] = ln_pack_bf16(o);
      else reinterpret_cast<float4*>((float*)y + (long long)row * C)[i] = o;
    }
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) (+ dx_add);
// partial[blk][0][C] = sum_rows dy*xhat, partial[blk][1][C] = sum_rows dy
// DZ: a second output dz = dz_scale * dropmask * dx (the gradient entering the sub-module below through its output
// dropout and residual scale: what would otherwise be an axpby launch over dx) and partial[blk][2][C] = sum_rows dz
// (the gradient of that sub-module's last bias: what would otherwise be a column-sum launch over dz)
// DYB: dy is bf16 (the result of a bf16-storage data-gradient GEMM); ZB: dz is written as bf16 (the operand of the next
// data-gradient and weight-gradient GEMMs; its column sums are those of the rounded values)
// PRED: the backward of such a predictor layer in one pass -- dy is first multiplied by the layer's dropout mask
// (`drop_in`), and dx by relu'(pre-activation) = (x > 0), x being the ReLU's output that the LayerNorm normalised
// (what were an axpby launch in front of this kernel and a dact_mul launch behind it)
// XOB (with PRED): dx is written as bf16 -- the operand of the predictor layer's bf16-storage weight- and data-gradient GEMMs
template <int NCH, bool DZ, bool DYB = false, bool ZB = false, bool PRED = false, bool XOB = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ dx_add,
                                                      void* __restrict__ dx, float* __restrict__ partial, int M, int C,
                                                      void* __restrict__ dz, float dz_scale, Fs2Drop drop_in,
                                                      const int rows_per_wg) {
  constexpr int NP = DZ ? 3 : 2;
  __shared__ float red[4][NP][LN_MAX_CH * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4 = C >> 2;
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  float4 g[NCH], dg[NCH], db[NCH], dzs[DZ ? NCH : 1];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int i = lane + 64 * j;
    g[j] = i < c4 ? reinterpret_cast<const float4*>(gamma)[i] : make_float4(0, 0, 0, 0);
    dg[j] = make_float4(0, 0, 0, 0);
    db[j] = make_float4(0, 0, 0, 0);
    if constexpr (DZ) dzs[j] = make_float4(0, 0, 0, 0);
  }
  const int row0 = blockIdx.x * rows_per_wg;
  for (int rr = wave; rr < rows_per_wg; rr += 4) {
    const int row = row0 + rr;
    if (row >= M) break;
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NCH], d[NCH];
    unsigned pos[PRED ? NCH : 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int i = lane + 64 * j;
      if (i < c4) {
        float4 xv = reinterpret_cast<const float4*>(x + (long long)row * C)[i];
        if constexpr (DYB) d[j] = ln_unpack_bf16(reinterpret_cast<const uint2*>((const unsigned short*)dy + (long long)row * C)[i]);
        else d[j] = reinterpret_cast<const float4*>((const float*)dy + (long long)row * C)[i];
        xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        if constexpr (PRED) {
          float f[4];
          fs2_drop_quad(drop, (unsigned)(row * C + 4 * i), f);
          d[j].x *= f[0]; d[j].y *= f[1]; d[j].z *= f[2]; d[j].w *= f[3];
          pos[j] = (xv.x > 0.f ? 1u : 0u) | (xv.y > 0.f ? 2u : 0u) | (xv.z > 0.f ? 4u : 0u) | (xv.w > 0.f ? 8u : 0u);
        }
      } else {
        d[j] = make_float4(0, 0, 0, 0);
        xh[j] = make_float4(0, 0, 0, 0);
      }
      dg[j].x += d[j].x * xh[j].x; dg[j].y += d[j].y * xh[j].y; dg[j].z += d[j].z * xh[j].z; dg[j].w += d[j].w * xh[j].w;
      db[j].x += d[j].x; db[j].y += d[j].y; db[j].z += d[j].z; db[j].w += d[j].w;
      d[j].x *= g[j].x; d[j].y *= g[j].y; d[j].z *= g[j].z; d[j].w *= g[j].w;
      s1 += d[j].x + d[j].y + d[j].z + d[j].w;
      s2 += d[j].x * xh[j].x + d[j].y * xh[j].y + d[j].z * xh[j].z + d[j].w * xh[j].w;
    }
    s1 = fs2_wave_sum(s1) / C;
    s2 = fs2_wave_sum(s2) / C;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      int i = lane + 64 * j;
      if (i < c4) {
        float4 o;
        o.x = rs * (d[j].x - s1 - xh[j].x * s2);
        o.y = rs * (d[j].y - s1 - xh[j].y * s2);
        o.z = rs * (d[j].z - s1 - xh[j].z * s2);
        o.w = rs * (d[j].w - s1 - xh[j].w * s2);
        if (dx_add) {
          float4 r = reinterpret_cast<const float4*>(dx_add + (long long)row * C)[i];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if constexpr (PRED) {
          if (!(pos[j] & 1u)) o.x = 0.f;
          if (!(pos[j] & 2u)) o.y = 0.f;
          if (!(pos[j] & 4u)) o.z = 0.f;
          if (!(pos[j] & 8u)) o.w = 0.f;
        }
        if constexpr (XOB) reinterpret_cast<uint2*>((unsigned short*)dx + (long long)row * C)[i] = ln_pack_bf16(o);
        else reinterpret_cast<float4*>((float*)dx + (long long)row * C)[i] = o;
        if constexpr (DZ) {
          const unsigned long long e = ((unsigned long long)row * C) + 4ull * i;  // element index in the [M, C] tensor
          float4 z;
          z.x = o.x * dz_scale * fs2_drop_factor(drop, e + 0);
          z.y = o.y * dz_scale * fs2_drop_factor(drop, e + 1);
          z.z = o.z * dz_scale * fs2_drop_factor(drop, e + 2);
          z.w = o.w * dz_scale * fs2_drop_factor(drop, e + 3);
          if constexpr (ZB) {
            const uint2 zb = ln_pack_bf16(z);
            reinterpret_cast<uint2*>((unsigned short*)dz + (long long)row * C)[i] = zb;
            z = ln_unpack_bf16(zb);
          } else {
            reinterpret_cast<float4*>((float*)dz + (long long)row * C)[i] = z;
          }
          dzs[j].x += z.x; dzs[j].y += z.y; dzs[j].z += z.z; dzs[j].w += z.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    int i = lane + 64 * j;
    reinterpret_cast<float4*>(&red[wave][0][0])[i] = dg[j];
    reinterpret_cast<float4*>(&red[wave][1][0])[i] = db[j];
    if constexpr (DZ) reinterpret_cast<float4*>(&red[wave][2][0])[i] = dzs[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
#pragma unroll
    for (int k = 0; k < NP; ++k)
      partial[((long long)blockIdx.x * NP + k) * C + c] = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
  }
}

}  // namespace

extern "C" int fs2hip_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                                    float* mean, float* rstd, int M, int C, float eps, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || C > LN_MAX_CH * 256) return FS2HIP_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)y % 16) || ((uintptr_t)gamma % 16) || ((uintptr_t)beta % 16)) return FS2HIP_EINVAL;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nch = (C / 4 + 63) / 64;
  switch (nch) {
    case 1: ln_fwd_kernel<1, false><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, M, C, eps); break;
    case 2: ln_fwd_kernel<2, false><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, M, C, eps); break;
    default: ln_fwd_kernel<4, false><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, M, C, eps); break;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_layernorm_fwd_drop(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                         float* rstd, int M, int C, float eps, float drop_p, unsigned long long drop_seed,
                                         const unsigned long long* drop_step, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || C > LN_MAX_CH * 256 || (long long)M * C >= 0x7fffffffLL) return FS2HIP_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)y % 16) || ((uintptr_t)gamma % 16) || ((uintptr_t)beta % 16)) return FS2HIP_EINVAL;
  if (!(drop_p > 0.f)) return fs2hip_layernorm_fwd(x, gamma, beta, y, mean, rstd, M, C, eps, stream);
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const Fs2Drop drop = fs2_make_drop(drop_p, drop_seed, drop_step);
  const int nch = (C / 4 + 63) / 64;
  switch (nch) {
    case 1: ln_fwd_kernel<1, false, true><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, M, C, eps, drop); break;
    case 2: ln_fwd_kernel<2, false, true><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, M, C, eps, drop); break;
    default: ln_fwd_kernel<4, false, true><<<grid, block, 0, s>>>(x, gamma, beta, y, mean, rstd, M, C, eps, drop); break;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_layernorm_bwd_pred(const float* dy, const float* x, const float* gamma, const float* mean,
                                         const float* rstd, void* dx, int dx_bf16, float* partial, int M, int C, float drop_p,
                                         unsigned long long drop_seed, const unsigned long long* drop_step, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || C > LN_MAX_CH * 256 || !partial || (long long)M * C >= 0x7fffffffLL) return FS2HIP_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)gamma % 16)) return FS2HIP_EINVAL;
  const int nblk = fs2hip_layernorm_bwd_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  const Fs2Drop drop = fs2_make_drop(drop_p, drop_seed, drop_step);  // (p = 0: every factor is 1)
  const int nch = (C / 4 + 63) / 64;
#define LN_PRED(N_, XOB_) ln_bwd_kernel<N_, false, false, false, true, XOB_><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, nullptr, dx, partial, M, C, nullptr, 0.f, drop, ln_bwd_rows(M))
  if (dx_bf16) {
    switch (nch) {
      case 1: LN_PRED(1, true); break;
      case 2: LN_PRED(2, true); break;
      default: LN_PRED(4, true); break;
    }
  } else {
    switch (nch) {
      case 1: LN_PRED(1, false); break;
      case 2: LN_PRED(2, false); break;
      default: LN_PRED(4, false); break;
    }
  }
#undef LN_PRED
  FS2_LAUNCH_CHECK();
  return 0;  // partial is [nblk][2][C]: dgamma | dbeta partial sums, finished by fs2hip_reduce_rows_multi
}

extern "C" int fs2hip_layernorm_fwd_b(const float* x, const float* gamma, const float* beta, void* y_bf16,
                                      float* mean, float* rstd, int M, int C, float eps, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || C > LN_MAX_CH * 256) return FS2HIP_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)y_bf16 % 8) || ((uintptr_t)gamma % 16) || ((uintptr_t)beta % 16)) return FS2HIP_EINVAL;
  dim3 grid((M + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int nch = (C / 4 + 63) / 64;
  switch (nch) {
    case 1: ln_fwd_kernel<1, true><<<grid, block, 0, s>>>(x, gamma, beta, y_bf16, mean, rstd, M, C, eps); break;
    case 2: ln_fwd_kernel<2, true><<<grid, block, 0, s>>>(x, gamma, beta, y_bf16, mean, rstd, M, C, eps); break;
    default: ln_fwd_kernel<4, true><<<grid, block, 0, s>>>(x, gamma, beta, y_bf16, mean, rstd, M, C, eps); break;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_layernorm_bwd_blocks(int M) {
  const int rows = ln_bwd_rows(M);
  return (M + rows - 1) / rows;
}

extern "C" int fs2hip_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                    const float* rstd, const float* dx_add, float* dx, float* partial,
                                    float* dgamma, float* dbeta, int M, int C, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || C > LN_MAX_CH * 256) return FS2HIP_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)gamma % 16)) return FS2HIP_EINVAL;
  if (dx_add && ((uintptr_t)dx_add % 16)) return FS2HIP_EINVAL;
  const int nblk = fs2hip_layernorm_bwd_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  const int nch = (C / 4 + 63) / 64;
  const Fs2Drop nodrop = fs2_make_drop(0.f, 0);
  switch (nch) {
    case 1: ln_bwd_kernel<1, false><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, dx_add, dx, partial, M, C, nullptr, 0.f, nodrop, ln_bwd_rows(M)); break;
    case 2: ln_bwd_kernel<2, false><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, dx_add, dx, partial, M, C, nullptr, 0.f, nodrop, ln_bwd_rows(M)); break;
    default: ln_bwd_kernel<4, false><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, dx_add, dx, partial, M, C, nullptr, 0.f, nodrop, ln_bwd_rows(M)); break;
  }
  FS2_LAUNCH_CHECK();
  // partial is [nblk][2][C]: columns [0, C) -> dgamma, [C, 2C) -> dbeta
  if (!dgamma && !dbeta) return 0;  // the caller finishes the partial sums with fs2hip_reduce_rows_multi
  return fs2_reduce_rows(partial, nblk, 2 * C, 2LL * C, dgamma, C, dbeta, s);
}

extern "C" int fs2hip_layernorm_bwd_dz(const float* dy, const float* x, const float* gamma, const float* mean,
                                       const float* rstd, const float* dx_add, float* dx, float* dz, float dz_scale,
                                       float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                       float* partial, int M, int C, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || C > LN_MAX_CH * 256 || !dz || !partial) return FS2HIP_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)dx % 16) || ((uintptr_t)dz % 16) || ((uintptr_t)gamma % 16))
    return FS2HIP_EINVAL;
  if (dx_add && ((uintptr_t)dx_add % 16)) return FS2HIP_EINVAL;
  const int nblk = fs2hip_layernorm_bwd_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  const Fs2Drop drop = fs2_make_drop(drop_p, drop_seed, drop_step);
  const int nch = (C / 4 + 63) / 64;
  switch (nch) {
    case 1: ln_bwd_kernel<1, true><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, dx_add, dx, partial, M, C, dz, dz_scale, drop, ln_bwd_rows(M)); break;
    case 2: ln_bwd_kernel<2, true><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, dx_add, dx, partial, M, C, dz, dz_scale, drop, ln_bwd_rows(M)); break;
    default: ln_bwd_kernel<4, true><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, dx_add, dx, partial, M, C, dz, dz_scale, drop, ln_bwd_rows(M)); break;
  }
  FS2_LAUNCH_CHECK();
  return 0;  // partial is [nblk][3][C]; the caller finishes it with fs2hip_reduce_rows_multi
}

// The same backward with bf16 on either side of it.  flags bit 0: dy is bf16; bit 1: dz is bf16.  dz == NULL: no
// second output (partial is [nblk][2][C]), else partial is [nblk][3][C].  The caller finishes the partial sums.
extern "C" int fs2hip_layernorm_bwd_x(const void* dy, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, const float* dx_add, float* dx, void* dz, float dz_scale,
                                      float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                      float* partial, int M, int C, int flags, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || C > LN_MAX_CH * 256 || !partial) return FS2HIP_EINVAL;
  const bool dyb = flags & 1, zb = flags & 2;
  if (((uintptr_t)x % 16) || ((uintptr_t)dy % (dyb ? 8 : 16)) || ((uintptr_t)dx % 16) || ((uintptr_t)gamma % 16)) return FS2HIP_EINVAL;
  if (dz && ((uintptr_t)dz % (zb ? 8 : 16))) return FS2HIP_EINVAL;
  if (dx_add && ((uintptr_t)dx_add % 16)) return FS2HIP_EINVAL;
  const int nblk = fs2hip_layernorm_bwd_blocks(M);
  hipStream_t s = (hipStream_t)stream;
  const Fs2Drop drop = fs2_make_drop(dz ? drop_p : 0.f, drop_seed, drop_step);
  const int nch = (C / 4 + 63) / 64;
#define FS2_LNB(N_, DZ_, DYB_, ZB_) \
  ln_bwd_kernel<N_, DZ_, DYB_, ZB_><<<dim3(nblk), dim3(256), 0, s>>>(dy, x, gamma, mean, rstd, dx_add, dx, partial, M, C, dz, dz_scale, drop, ln_bwd_rows(M))
#define FS2_LNB_N(N_)                                      \
  do {                                                     \
    if (!dz) {                                             \
      if (dyb) FS2_LNB(N_, false, true, false);            \
      else FS2_LNB(N_, false, false, false);               \
    } else if (dyb) {                                      \
      if (zb) FS2_LNB(N_, true, true, true);               \
      else FS2_LNB(N_, true, true, false);                 \
    } else {                                               \
      if (zb) FS2_LNB(N_, true, false, true);              \
      else FS2_LNB(N_, true, false, false);                \
    }                                                      \
  } while (0)
  switch (nch) {
    case 1: FS2_LNB_N(1); break;
    case 2: FS2_LNB_N(2); break;
    default: FS2_LNB_N(4); break;
  }
#undef FS2_LNB_N
#undef FS2_LNB
  FS2_LAUNCH_CHECK();
  return 0;
}
