// BatchNorm1d over the channel (last) dimension of a dense [M][C] matrix (M = B*T rows, padding
// rows included -- the reference normalises over them too), training and evaluation mode, fused
// with the activation (+dropout) that follows it.
// Replaces nn.BatchNorm1d (+SiLU) in torchaudio's Conformer conv module (call sites
// fs2/model.py:193, :241) and nn.BatchNorm1d + tanh + F.dropout in PostNet (fs2/layers.py:204-212).
//
//   statistics : colstats (or the depthwise-conv kernel's fused partials)  -> partial[nparts][2][C] = per-part
//                (mean, sum of squared deviations), computed on pivot-shifted values
//   finalize   : Chan merge of the parts in fp64 (64 channels x 16 part lanes per workgroup),
//                running-stat update, per-channel scale/shift
//   apply      : out = dropout(act(y*scale + shift))                         (one pass)
//   backward   : reduce (sum dz, sum dz*xhat) -> finalize (dgamma, dbeta, means) -> apply
// The statistics passes sweep whole rows (contiguous >= 1 KiB per wavefront instruction).
#include "common.h"

namespace {

constexpr int CS_ROWS = 32;   // rows per workgroup in the column-statistics passes (more for very tall matrices,
                              // so that the fp64 finish never walks more than ~2048 partial rows: cs_rows())
__host__ __device__ inline int cs_rows(int M) { return CS_ROWS * ((M + CS_ROWS * 2048 - 1) / (CS_ROWS * 2048)); }

// thread -> (row lane rl, float4 column c4): tpr = C/4 threads per row, rpi = 256/tpr rows per pass
struct WideMap {
  int tpr, rpi;
};

// four consecutive elements starting at element index e (a multiple of 4) of a tensor that is fp32 (IB = false) or
// bf16 (IB = true) in memory
template <bool IB>
__device__ __forceinline__ float4 bn_ld4(const void* p, long long e) {
  if constexpr (IB) {
    const uint2 u = *reinterpret_cast<const uint2*>((const unsigned short*)p + e);
    return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                       __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
  } else {
    return *reinterpret_cast<const float4*>((const float*)p + e);
  }
}

// Per-stripe (mean, M2 = sum of squared deviations) per channel.  Sums are taken of d = y - pivot with the stripe's
// first row as the pivot, so M2 = sum d^2 - (sum d)^2 / n subtracts two numbers of the size of the variance, not of
// the squared mean (a plain E[y^2] - E[y]^2 in fp32 loses the variance of a channel whose |mean| >> std; torch's
// BatchNorm is Welford).  bn_finalize merges the stripes with Chan's formula in fp64.
template <bool IB>
__global__ __launch_bounds__(256) void colstats_wide_kernel(const void* __restrict__ y, int M, int C,
                                                             float* __restrict__ partial, WideMap wm) {
  __shared__ float4 red[2][256];
  const int tid = threadIdx.x;
  const int rl = tid / wm.tpr, c4 = tid - rl * wm.tpr;
  const int rows = cs_rows(M);
  const int r0 = blockIdx.x * rows, r1 = min(M, r0 + rows);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0), pv = make_float4(0, 0, 0, 0);
  if (rl < wm.rpi) {
    pv = bn_ld4<IB>(y, (long long)r0 * C + c4 * 4);
    auto add = [&](float4 v) {
      v.x -= pv.x; v.y -= pv.y; v.z -= pv.z; v.w -= pv.w;
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
    };
    int r = r0 + rl;
    for (; r + 3 * wm.rpi < r1; r += 4 * wm.rpi) {  // four rows' loads in flight per thread, added in row order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = bn_ld4<IB>(y, (long long)(r + u * wm.rpi) * C + c4 * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) add(v[u]);
    }
    for (; r < r1; r += wm.rpi) add(bn_ld4<IB>(y, (long long)r * C + c4 * 4));
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < wm.rpi; ++l) {
      float4 a = red[0][l * wm.tpr + c4], b = red[1][l * wm.tpr + c4];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    const float inv = 1.f / (float)(r1 - r0);
    float4 mean, m2;
    mean.x = pv.x + s1.x * inv; m2.x = fmaxf(s2.x - s1.x * s1.x * inv, 0.f);
    mean.y = pv.y + s1.y * inv; m2.y = fmaxf(s2.y - s1.y * s1.y * inv, 0.f);
    mean.z = pv.z + s1.z * inv; m2.z = fmaxf(s2.z - s1.z * s1.z * inv, 0.f);
    mean.w = pv.w + s1.w * inv; m2.w = fmaxf(s2.w - s1.w * s1.w * inv, 0.f);
    *reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2 + 0) * C + c4 * 4) = mean;
    *reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2 + 1) * C + c4 * 4) = m2;
  }
}

// The finalize kernels: one 1024-thread workgroup per 16 channels, thread = (channel tid & 15, part lane tid >> 4):
// 64 part lanes with four independent loads in flight each (these kernels are pure latency: a few hundred parts per
// channel), fp64 sums merged in a fixed order -- the lane pairs of a wavefront, then the 16 wavefronts through LDS.
constexpr int FIN_CH = 16, FIN_LANES = 64;
// every thread gets the block's sum of v for its channel
__device__ __forceinline__ double fin_sum(double v, double (&red)[16][FIN_CH]) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  const int ch = threadIdx.x & 15, w = threadIdx.x >> 6;
  __syncthreads();  // (the previous call's readers are done with red)
  if ((threadIdx.x & 63) < 16) red[w][ch] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int l = 0; l < 16; ++l) s += red[l][ch];
  return s;
}
// sum over this thread's parts of f(p, partial[p][0][c], partial[p][1][c]) -> (a, b)
template <class F>
__device__ __forceinline__ void fin_parts(const float* __restrict__ partial, int nparts, int C, int c, F&& f, double& a,
                                          double& b) {
  a = 0.0;
  b = 0.0;
  if (c >= C) return;
  int p = threadIdx.x >> 4;
  for (; p + 3 * FIN_LANES < nparts; p += 4 * FIN_LANES) {
    float u[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u[k] = partial[((long long)(p + k * FIN_LANES) * 2 + 0) * C + c];
      v[k] = partial[((long long)(p + k * FIN_LANES) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) f(p + k * FIN_LANES, u[k], v[k], a, b);
  }
  float u[3], v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (p + k * FIN_LANES < nparts) {
      u[k] = partial[((long long)(p + k * FIN_LANES) * 2 + 0) * C + c];
      v[k] = partial[((long long)(p + k * FIN_LANES) * 2 + 1) * C + c];
    }
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (p + k * FIN_LANES < nparts) f(p + k * FIN_LANES, u[k], v[k], a, b);
}

// rows of part p (bn_finalize): the parts tile groups of group_rows rows (one group = the whole matrix for colstats,
// one utterance for the depthwise conv's fused statistics) in stripes of part_rows

// stats layout (per channel): [0]=scale (gamma*invstd) [1]=shift (beta-mean*scale) [2]=mean [3]=invstd
// training: Chan's parallel merge of the per-part (n, mean, M2) in fp64 -- mean = sum n_p mean_p / N, then
// M2 = sum [M2_p + n_p (mean_p - mean)^2]
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                            long long count, int part_rows, int group_rows,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, float momentum, float eps,
                                                            int training, float* __restrict__ stats, int C) {
  __shared__ double red[16][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & 15);
  const bool writer = threadIdx.x < FIN_CH && c < C;
  float mean = 0.f, invstd = 0.f;
  if (training) {
    const int ppg = (group_rows + part_rows - 1) / part_rows;
    auto cnt = [&](int p) { return (double)min(part_rows, group_rows - (p % ppg) * part_rows); };
    double a, b;
    fin_parts(partial, nparts, C, c, [&](int p, float u, float, double& x, double&) { x += cnt(p) * (double)u; }, a, b);
    const double mu = fin_sum(a, red) / (double)count;
    fin_parts(partial, nparts, C, c, [&](int p, float u, float v, double& x, double&) {
      const double d = (double)u - mu;
      x += (double)v + cnt(p) * d * d;
    }, a, b);
    const double m2 = fin_sum(a, red);
    if (!writer) return;
    double var = m2 / (double)count;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
  } else {
    if (!writer) return;
    mean = rmean[c];
    invstd = 1.f / sqrtf(rvar[c] + eps);
  }
  const float sc = gamma[c] * invstd;
  stats[0 * C + c] = sc;
  stats[1 * C + c] = beta[c] - mean * sc;
  stats[2 * C + c] = mean;
  stats[3 * C + c] = invstd;
}

// four fp32 -> four bf16 (round to nearest even), 8 bytes
typedef __bf16 bn_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_bf16x4(void* dst, long long i4, float4 v) {
  const f32x4 f = {v.x, v.y, v.z, v.w};
  reinterpret_cast<bn_bf16x4*>(dst)[i4] = __builtin_convertvector(f, bn_bf16x4);
}

// out_b (may be null): a bf16 copy of the output for a GEMM that takes bf16 operands from memory (operand_bf16 == 3)
// IB: y is a bf16 tensor (the depthwise convolution's bf16 result)
template <bool IB>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const void* __restrict__ y, const float* __restrict__ stats,
                                                          float* __restrict__ out, void* __restrict__ out_b, long long n4,
                                                          int C, int act, Fs2Drop drop_in) {
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    float4 v = bn_ld4<IB>(y, i * 4);
    float4 sc = reinterpret_cast<const float4*>(stats)[c4];
    float4 sh = reinterpret_cast<const float4*>(stats + C)[c4];
    float4 o;
    o.x = fs2_act(act, fmaf(v.x, sc.x, sh.x)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 0));
    o.y = fs2_act(act, fmaf(v.y, sc.y, sh.y)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 1));
    o.z = fs2_act(act, fmaf(v.z, sc.z, sh.z)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 2));
    o.w = fs2_act(act, fmaf(v.w, sc.w, sh.w)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 3));
    if (out) reinterpret_cast<float4*>(out)[i] = o;   // (null: the activations exist only as the bf16 operand)
    if (out_b) store_bf16x4(out_b, i, o);
  }
}

__device__ __forceinline__ float dz_of(float dout, float v, float sc, float sh, int act, const Fs2Drop& drop,
                                       unsigned long long idx) {
  return dout * fs2_drop_factor(drop, idx) * fs2_dact(act, fmaf(v, sc, sh));
}

// dz = dout * dropmask * act'(y*scale+shift) ; partial[blk][0][C] = sum dz, [1] = sum dz * xhat
template <bool IB, bool DB>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const void* __restrict__ dout, const void* __restrict__ y,
                                                             const float* __restrict__ stats, int M, int C, int act,
                                                             Fs2Drop drop_in, float* __restrict__ partial, WideMap wm) {
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  __shared__ float4 red[2][256];
  const int tid = threadIdx.x;
  const int rl = tid / wm.tpr, c4 = tid - rl * wm.tpr;
  const int rows = cs_rows(M);
  const int r0 = blockIdx.x * rows, r1 = min(M, r0 + rows);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  if (rl < wm.rpi) {
    const float4 sc = reinterpret_cast<const float4*>(stats)[c4], sh = reinterpret_cast<const float4*>(stats + C)[c4];
    const float4 mu = reinterpret_cast<const float4*>(stats + 2 * C)[c4], is = reinterpret_cast<const float4*>(stats + 3 * C)[c4];
    auto add = [&](long long idx, const float4& v, const float4& d) {
      float dz;
      dz = dz_of(d.x, v.x, sc.x, sh.x, act, drop, idx + 0); s1.x += dz; s2.x += dz * (v.x - mu.x) * is.x;
      dz = dz_of(d.y, v.y, sc.y, sh.y, act, drop, idx + 1); s1.y += dz; s2.y += dz * (v.y - mu.y) * is.y;
      dz = dz_of(d.z, v.z, sc.z, sh.z, act, drop, idx + 2); s1.z += dz; s2.z += dz * (v.z - mu.z) * is.z;
      dz = dz_of(d.w, v.w, sc.w, sh.w, act, drop, idx + 3); s1.w += dz; s2.w += dz * (v.w - mu.w) * is.w;
    };
    int r = r0 + rl;
    // a part is one workgroup per 64 rows, i.e. about five wavefronts per CU: four rows' loads in flight per thread
    // (the sums are added in the same row order as before)
    for (; r + 3 * wm.rpi < r1; r += 4 * wm.rpi) {
      long long idx[4];
      float4 v[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        idx[u] = (long long)(r + u * wm.rpi) * C + c4 * 4;
        v[u] = bn_ld4<IB>(y, idx[u]);
        d[u] = bn_ld4<DB>(dout, idx[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(idx[u], v[u], d[u]);
    }
    for (; r < r1; r += wm.rpi) {
      const long long idx = (long long)r * C + c4 * 4;
      add(idx, bn_ld4<IB>(y, idx), bn_ld4<DB>(dout, idx));
    }
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < wm.rpi; ++l) {
      float4 a = red[0][l * wm.tpr + c4], b = red[1][l * wm.tpr + c4];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
    }
    *reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2 + 0) * C + c4 * 4) = s1;
    *reinterpret_cast<float4*>(partial + ((long long)blockIdx.x * 2 + 1) * C + c4 * 4) = s2;
  }
}

// dgamma = sum dz*xhat, dbeta = sum dz; coef[0][c] = mean(dz), coef[1][c] = mean(dz*xhat)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                                long long count, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float* __restrict__ coef, int C) {
  __shared__ double red[16][FIN_CH];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & 15);
  double a, b;
  fin_parts(partial, nparts, C, c, [&](int, float u, float v, double& x, double& y) { x += (double)u; y += (double)v; }, a, b);
  const double s1 = fin_sum(a, red), s2 = fin_sum(b, red);
  if (threadIdx.x >= FIN_CH || c >= C) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  coef[c] = (float)(s1 / (double)count);
  coef[C + c] = (float)(s2 / (double)count);
}

// dy = scale * (dz - mean(dz) - xhat * mean(dz*xhat))   [training]   or scale * dz   [eval]
template <bool IB, bool DB>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* __restrict__ dout, const void* __restrict__ y,
                                                            const float* __restrict__ stats, const float* __restrict__ coef,
                                                            float* __restrict__ dy, void* __restrict__ dy_b, long long n4,
                                                            int C, int act, Fs2Drop drop_in, int training) {
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const float4 sc = reinterpret_cast<const float4*>(stats)[c4], sh = reinterpret_cast<const float4*>(stats + C)[c4];
    const float4 v = bn_ld4<IB>(y, i * 4);
    const float4 d = bn_ld4<DB>(dout, i * 4);
    float4 dz;
    dz.x = dz_of(d.x, v.x, sc.x, sh.x, act, drop, (unsigned long long)(i * 4 + 0));
    dz.y = dz_of(d.y, v.y, sc.y, sh.y, act, drop, (unsigned long long)(i * 4 + 1));
    dz.z = dz_of(d.z, v.z, sc.z, sh.z, act, drop, (unsigned long long)(i * 4 + 2));
    dz.w = dz_of(d.w, v.w, sc.w, sh.w, act, drop, (unsigned long long)(i * 4 + 3));
    if (training) {
      const float4 mu = reinterpret_cast<const float4*>(stats + 2 * C)[c4], is = reinterpret_cast<const float4*>(stats + 3 * C)[c4];
      const float4 k1 = reinterpret_cast<const float4*>(coef)[c4], k2 = reinterpret_cast<const float4*>(coef + C)[c4];
      dz.x = dz.x - k1.x - (v.x - mu.x) * is.x * k2.x;
      dz.y = dz.y - k1.y - (v.y - mu.y) * is.y * k2.y;
      dz.z = dz.z - k1.z - (v.z - mu.z) * is.z * k2.z;
      dz.w = dz.w - k1.w - (v.w - mu.w) * is.w * k2.w;
    }
    const float4 o = make_float4(sc.x * dz.x, sc.y * dz.y, sc.z * dz.z, sc.w * dz.w);
    if (dy) reinterpret_cast<float4*>(dy)[i] = o;
    if (dy_b) store_bf16x4(dy_b, i, o);
  }
}

bool wide_ok(int C, WideMap& wm) {
  if ((C % 4) || C > 1024) return false;
  wm.tpr = C / 4;
  wm.rpi = 256 / wm.tpr;
  return true;
}

}  // namespace

extern "C" int fs2hip_colstats_parts(int M) { return M > 0 ? (M + cs_rows(M) - 1) / cs_rows(M) : 0; }
extern "C" int fs2hip_colstats_part_rows(int M) { return M > 0 ? cs_rows(M) : 0; }

extern "C" int fs2hip_colstats_b(const void* y, int M, int C, float* partial, int in_bf16, void* stream);
extern "C" int fs2hip_colstats(const float* y, int M, int C, float* partial, void* stream) {
  return fs2hip_colstats_b(y, M, C, partial, 0, stream);
}

// in_bf16: y is a bf16 tensor
extern "C" int fs2hip_colstats_b(const void* y, int M, int C, float* partial, int in_bf16, void* stream) {
  WideMap wm;
  if (M <= 0 || C <= 0 || !wide_ok(C, wm) || ((uintptr_t)y % (in_bf16 ? 8 : 16)) || ((uintptr_t)partial % 16)) return FS2HIP_EINVAL;
  const dim3 grid(fs2hip_colstats_parts(M));
  if (in_bf16) colstats_wide_kernel<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(y, M, C, partial, wm);
  else colstats_wide_kernel<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(y, M, C, partial, wm);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_bn_finalize(const float* partial, int nparts, long long count, int part_rows, int group_rows,
                                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  float momentum, float eps, int training, float* stats, int C, void* stream) {
  if (C <= 0 || (training && (nparts <= 0 || count <= 0 || !partial)) || (!training && (!running_mean || !running_var)))
    return FS2HIP_EINVAL;
  if (training) {  // the parts must tile `count` rows exactly: groups of group_rows rows in stripes of part_rows
    if (part_rows <= 0 || group_rows <= 0 || count % group_rows) return FS2HIP_EINVAL;
    const long long ppg = (group_rows + part_rows - 1) / part_rows;
    if (ppg * (count / group_rows) != nparts) return FS2HIP_EINVAL;
  }
  bn_finalize_kernel<<<dim3((C + FIN_CH - 1) / FIN_CH), dim3(1024), 0, (hipStream_t)stream>>>(
      partial, nparts, count, part_rows, group_rows, gamma, beta, running_mean, running_var, momentum, eps, training,
      stats, C);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_bn_act_fwd_b(const void* y, const float* stats, float* out, void* out_bf16, int M, int C, int act,
                                   float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                   int in_bf16, void* stream);
extern "C" int fs2hip_bn_act_fwd(const float* y, const float* stats, float* out, int M, int C, int act, float drop_p,
                                 unsigned long long drop_seed, const unsigned long long* drop_step, void* stream) {
  return fs2hip_bn_act_fwd_b(y, stats, out, nullptr, M, C, act, drop_p, drop_seed, drop_step, 0, stream);
}

// in_bf16: y is a bf16 tensor
extern "C" int fs2hip_bn_act_fwd_b(const void* y, const float* stats, float* out, void* out_bf16, int M, int C, int act,
                                   float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                   int in_bf16, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || ((uintptr_t)y % (in_bf16 ? 8 : 16)) || ((uintptr_t)out % 16) ||
      ((uintptr_t)stats % 16) || ((uintptr_t)out_bf16 % 8) || (!out && !out_bf16))
    return FS2HIP_EINVAL;
  const long long n4 = (long long)M * C / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const Fs2Drop drop = fs2_make_drop(drop_p, drop_seed, drop_step);
  if (in_bf16)
    bn_act_fwd_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(y, stats, out, out_bf16, n4, C, act, drop);
  else
    bn_act_fwd_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(y, stats, out, out_bf16, n4, C, act, drop);
  FS2_LAUNCH_CHECK();
  return 0;
}

// partial: [fs2hip_colstats_parts(M)][2][C]; coef: [2][C] scratch (16-byte aligned)
extern "C" int fs2hip_bn_act_bwd_b(const void* dout, const void* y, const float* stats, float* partial, float* coef,
                                   float* dgamma, float* dbeta, float* dy, void* dy_bf16, int M, int C, int act,
                                   float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                   int training, int in_bf16, void* stream);
extern "C" int fs2hip_bn_act_bwd(const float* dout, const float* y, const float* stats, float* partial, float* coef,
                                 float* dgamma, float* dbeta, float* dy, int M, int C, int act, float drop_p,
                                 unsigned long long drop_seed, const unsigned long long* drop_step, int training,
                                 void* stream) {
  return fs2hip_bn_act_bwd_b(dout, y, stats, partial, coef, dgamma, dbeta, dy, nullptr, M, C, act, drop_p, drop_seed,
                             drop_step, training, 0, stream);
}

// in_bf16 bit 0: y is a bf16 tensor; bit 1: dout is
extern "C" int fs2hip_bn_act_bwd_b(const void* dout, const void* y, const float* stats, float* partial, float* coef,
                                   float* dgamma, float* dbeta, float* dy, void* dy_bf16, int M, int C, int act,
                                   float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                   int training, int in_bf16, void* stream) {
  if (((uintptr_t)dy_bf16 % 8) || (!dy && !dy_bf16)) return FS2HIP_EINVAL;
  WideMap wm;
  if (M <= 0 || C <= 0 || !wide_ok(C, wm)) return FS2HIP_EINVAL;
  if (((uintptr_t)dout % ((in_bf16 & 2) ? 8 : 16)) || ((uintptr_t)y % ((in_bf16 & 1) ? 8 : 16)) || ((uintptr_t)stats % 16) || ((uintptr_t)partial % 16) ||
      ((uintptr_t)coef % 16) || ((uintptr_t)dy % 16))
    return FS2HIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const Fs2Drop drop = fs2_make_drop(drop_p, drop_seed, drop_step);
  const int nparts = fs2hip_colstats_parts(M);
#define FS2_BN_RED(IB_, DB_) bn_bwd_reduce_kernel<IB_, DB_><<<dim3(nparts), dim3(256), 0, s>>>(dout, y, stats, M, C, act, drop, partial, wm)
  switch (in_bf16 & 3) {
    case 0: FS2_BN_RED(false, false); break;
    case 1: FS2_BN_RED(true, false); break;
    case 2: FS2_BN_RED(false, true); break;
    default: FS2_BN_RED(true, true); break;
  }
#undef FS2_BN_RED
  FS2_LAUNCH_CHECK();
  bn_bwd_finalize_kernel<<<dim3((C + FIN_CH - 1) / FIN_CH), dim3(1024), 0, s>>>(partial, nparts, (long long)M, dgamma, dbeta, coef, C);
  FS2_LAUNCH_CHECK();
  const long long n4 = (long long)M * C / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
#define FS2_BN_APP(IB_, DB_) bn_bwd_apply_kernel<IB_, DB_><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dout, y, stats, coef, dy, dy_bf16, n4, C, act, drop, training)
  switch (in_bf16 & 3) {
    case 0: FS2_BN_APP(false, false); break;
    case 1: FS2_BN_APP(true, false); break;
    case 2: FS2_BN_APP(false, true); break;
    default: FS2_BN_APP(true, true); break;
  }
#undef FS2_BN_APP
  FS2_LAUNCH_CHECK();
  return 0;
}
