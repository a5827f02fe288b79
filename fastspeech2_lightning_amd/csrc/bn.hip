// BatchNorm1d over the channel (last) dimension of a dense [M][C] matrix (M = B*T rows, padding
// rows included -- the reference normalises over them too), training and evaluation mode, fused
// with the activation (+dropout) that follows it.
// Replaces nn.BatchNorm1d (+SiLU) in torchaudio's Conformer conv module (call sites
// fs2/model.py:193, :241) and nn.BatchNorm1d + tanh + F.dropout in PostNet (fs2/layers.py:204-212).
//
//   statistics : colstats (or the depthwise-conv kernel's fused partials)  -> partial[nparts][2][C]
//   finalize   : fp64 finish of the partials, running-stat update, per-channel scale/shift
//   apply      : out = dropout(act(y*scale + shift))                         (one pass)
//   backward   : reduce (sum dz, sum dz*xhat) -> finalize (dgamma, dbeta, means) -> apply
#include "common.h"

namespace {

constexpr int CS_ROWS = 256;  // rows per workgroup in the column-statistics passes

__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ y, int M, int C,
                                                        float* __restrict__ partial) {
  __shared__ float red[4][2][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) {
      float v = y[(long long)r * C + c];
      s1 += v;
      s2 += v * v;
    }
  red[rl][0][lane] = s1;
  red[rl][1][lane] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    partial[((long long)blockIdx.y * 2 + 0) * C + c] = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
    partial[((long long)blockIdx.y * 2 + 1) * C + c] = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
  }
}

// stats layout (per channel): [0]=scale (gamma*invstd) [1]=shift (beta-mean*scale) [2]=mean [3]=invstd
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int nparts, long long count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                   int training, float* __restrict__ stats, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, invstd;
  if (training) {
    double s1 = 0.0, s2 = 0.0;
    for (int p = 0; p < nparts; ++p) {
      s1 += (double)partial[((long long)p * 2 + 0) * C + c];
      s2 += (double)partial[((long long)p * 2 + 1) * C + c];
    }
    double mu = s1 / (double)count;
    double var = s2 / (double)count - mu * mu;
    if (var < 0.0) var = 0.0;
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
  } else {
    mean = rmean[c];
    invstd = 1.f / sqrtf(rvar[c] + eps);
  }
  const float sc = gamma[c] * invstd;
  stats[0 * C + c] = sc;
  stats[1 * C + c] = beta[c] - mean * sc;
  stats[2 * C + c] = mean;
  stats[3 * C + c] = invstd;
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ y, const float* __restrict__ stats,
                                                          float* __restrict__ out, long long n4, int C, int act,
                                                          Fs2Drop drop_in) {
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 sc = reinterpret_cast<const float4*>(stats)[c4];
    float4 sh = reinterpret_cast<const float4*>(stats + C)[c4];
    float4 o;
    o.x = fs2_act(act, fmaf(v.x, sc.x, sh.x)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 0));
    o.y = fs2_act(act, fmaf(v.y, sc.y, sh.y)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 1));
    o.z = fs2_act(act, fmaf(v.z, sc.z, sh.z)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 2));
    o.w = fs2_act(act, fmaf(v.w, sc.w, sh.w)) * fs2_drop_factor(drop, (unsigned long long)(i * 4 + 3));
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

// dz = dout * dropmask * act'(y*scale+shift) ; partial[blk][0][C] = sum dz, [1] = sum dz * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                             const float* __restrict__ stats, int M, int C, int act,
                                                             Fs2Drop drop_in, float* __restrict__ partial) {
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  __shared__ float red[4][2][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float sc = stats[c], sh = stats[C + c], mean = stats[2 * C + c], invstd = stats[3 * C + c];
    for (int r = r0 + rl; r < r1; r += 4) {
      const long long idx = (long long)r * C + c;
      const float v = y[idx];
      const float dz = dout[idx] * fs2_drop_factor(drop, (unsigned long long)idx) * fs2_dact(act, fmaf(v, sc, sh));
      s1 += dz;
      s2 += dz * (v - mean) * invstd;
    }
  }
  red[rl][0][lane] = s1;
  red[rl][1][lane] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    partial[((long long)blockIdx.y * 2 + 0) * C + c] = red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
    partial[((long long)blockIdx.y * 2 + 1) * C + c] = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
  }
}

// dgamma = sum dz*xhat, dbeta = sum dz; coef[0][c] = mean(dz), coef[1][c] = mean(dz*xhat)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nparts, long long count,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ coef, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int p = 0; p < nparts; ++p) {
    s1 += (double)partial[((long long)p * 2 + 0) * C + c];
    s2 += (double)partial[((long long)p * 2 + 1) * C + c];
  }
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  coef[c] = (float)(s1 / (double)count);
  coef[C + c] = (float)(s2 / (double)count);
}

// dy = scale * (dz - mean(dz) - xhat * mean(dz*xhat))   [training]   or scale * dz   [eval]
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            const float* __restrict__ stats, const float* __restrict__ coef,
                                                            float* __restrict__ dy, long long n, int C, int act,
                                                            Fs2Drop drop_in, int training) {
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float sc = stats[c], sh = stats[C + c];
    const float v = y[i];
    float dz = dout[i] * fs2_drop_factor(drop, (unsigned long long)i) * fs2_dact(act, fmaf(v, sc, sh));
    if (training) {
      const float xhat = (v - stats[2 * C + c]) * stats[3 * C + c];
      dz = dz - coef[c] - xhat * coef[C + c];
    }
    dy[i] = sc * dz;
  }
}

}  // namespace

extern "C" int fs2hip_colstats_parts(int M) { return (M + CS_ROWS - 1) / CS_ROWS; }

extern "C" int fs2hip_colstats(const float* y, int M, int C, float* partial, void* stream) {
  if (M <= 0 || C <= 0) return FS2HIP_EINVAL;
  colstats_kernel<<<dim3((C + 63) / 64, fs2hip_colstats_parts(M)), dim3(256), 0, (hipStream_t)stream>>>(y, M, C, partial);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_bn_finalize(const float* partial, int nparts, long long count, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum,
                                  float eps, int training, float* stats, int C, void* stream) {
  if (C <= 0 || (training && (nparts <= 0 || count <= 0 || !partial)) || (!training && (!running_mean || !running_var)))
    return FS2HIP_EINVAL;
  bn_finalize_kernel<<<dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(
      partial, nparts, count, gamma, beta, running_mean, running_var, momentum, eps, training, stats, C);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_bn_act_fwd(const float* y, const float* stats, float* out, int M, int C, int act, float drop_p,
                                 unsigned long long drop_seed, const unsigned long long* drop_step, void* stream) {
  if (M <= 0 || C <= 0 || (C % 4) || ((uintptr_t)y % 16) || ((uintptr_t)out % 16) || ((uintptr_t)stats % 16))
    return FS2HIP_EINVAL;
  const long long n4 = (long long)M * C / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  bn_act_fwd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(y, stats, out, n4, C, act,
                                                                                     fs2_make_drop(drop_p, drop_seed, drop_step));
  FS2_LAUNCH_CHECK();
  return 0;
}

// partial: [fs2hip_colstats_parts(M)][2][C]; coef: [2][C] scratch
extern "C" int fs2hip_bn_act_bwd(const float* dout, const float* y, const float* stats, float* partial, float* coef,
                                 float* dgamma, float* dbeta, float* dy, int M, int C, int act, float drop_p,
                                 unsigned long long drop_seed, const unsigned long long* drop_step, int training,
                                 void* stream) {
  if (M <= 0 || C <= 0) return FS2HIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const Fs2Drop drop = fs2_make_drop(drop_p, drop_seed, drop_step);
  const int nparts = fs2hip_colstats_parts(M);
  bn_bwd_reduce_kernel<<<dim3((C + 63) / 64, nparts), dim3(256), 0, s>>>(dout, y, stats, M, C, act, drop, partial);
  FS2_LAUNCH_CHECK();
  bn_bwd_finalize_kernel<<<dim3((C + 63) / 64), dim3(64), 0, s>>>(partial, nparts, (long long)M, dgamma, dbeta, coef, C);
  FS2_LAUNCH_CHECK();
  const long long n = (long long)M * C;
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  bn_bwd_apply_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dout, y, stats, coef, dy, n, C, act, drop, training);
  FS2_LAUNCH_CHECK();
  return 0;
}
