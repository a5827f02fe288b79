// Gather / scatter / integer-index kernels and small reductions of the path: positional table,
// text embedding, variance bucketize+embedding, LengthRegulator, predictor head, masked losses,
// fused AdamW + global-norm clip, elementwise axpby.  All HBM/latency-bound; lanes run along the
// channel dimension (float4, coalesced) and integer results are exact.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// fs2/layers.py:123-140  PositionalEmbedding: table[t] = [sin(t*f) | cos(t*f)]
// ---------------------------------------------------------------------------------------------
__global__ void posenc_table_kernel(const float* __restrict__ inv_freq, float* __restrict__ table, int T, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * D) return;
  const int t = i / D, c = i % D, half = D / 2;
  const float ang = (float)t * inv_freq[c % half];
  table[i] = c < half ? sinf(ang) : cosf(ang);
}

// out[b,t,:] = x[b,t,:] + table[t,:] * (t < lens[b])      (fs2/model.py:186-193, :233-241)
__global__ __launch_bounds__(256) void add_posenc_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                          const int* __restrict__ lens, float* __restrict__ out, int B,
                                                          int T, int D) {
  const int d4 = D >> 2;
  const long long n4 = (long long)B * T * d4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4);
    const long long row = i / d4;
    const int t = (int)(row % T), b = (int)(row / T);
    float4 v = reinterpret_cast<const float4*>(x)[i];
    if (t < lens[b]) {
      float4 p = reinterpret_cast<const float4*>(table)[(long long)t * d4 + c4];
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// fs2/model.py:183  nn.Embedding gather: out[m,:] = W[idx[m],:]   (one wavefront per row)
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int* __restrict__ idx, const float* __restrict__ W,
                                                             float* __restrict__ out, int M, int V, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  int v = idx[row];
  v = v < 0 ? 0 : (v >= V ? V - 1 : v);
  const float4* src = reinterpret_cast<const float4*>(W + (long long)v * D);
  float4* dst = reinterpret_cast<float4*>(out + (long long)row * D);
  for (int i = lane; i < (D >> 2); i += 64) dst[i] = src[i];
}

// one-hot rows for the embedding backward: dW = onehot^T @ dy runs as a weight-gradient GEMM on the
// matrix cores (deterministic).  out is [M][Vp]; the padding row never gets a 1 (its gradient is 0).
__global__ __launch_bounds__(256) void onehot_kernel(const int* __restrict__ idx, float* __restrict__ out, int M,
                                                      int Vp, int padding_idx) {
  const long long n = (long long)M * Vp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / Vp), v = (int)(i % Vp);
    const int k = idx[m];
    out[i] = (k == v && k != padding_idx) ? 1.f : 0.f;
  }
}

// fs2/variance_adaptor.py:197-205, :322, :343  torch.bucketize (right=False) + embedding + add
__global__ __launch_bounds__(256) void bucket_embed_add_kernel(const float* __restrict__ val, float control,
                                                                const float* __restrict__ bins, int NB,
                                                                const float* __restrict__ W, const float* __restrict__ x,
                                                                float* __restrict__ out, int* __restrict__ idx_out,
                                                                int M, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float v = val[row] * control;
  int lo = 0, hi = NB;  // lower_bound: number of edges strictly below v
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (bins[mid] < v) lo = mid + 1; else hi = mid;
  }
  if (lane == 0 && idx_out) idx_out[row] = lo;
  const float4* e = reinterpret_cast<const float4*>(W + (long long)lo * D);
  const float4* xi = reinterpret_cast<const float4*>(x + (long long)row * D);
  float4* o = reinterpret_cast<float4*>(out + (long long)row * D);
  for (int i = lane; i < (D >> 2); i += 64) {
    float4 a = xi[i], b = e[i];
    o[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// ---------------------------------------------------------------------------------------------
// LengthRegulator  fs2/variance_adaptor.py:65-81
// ---------------------------------------------------------------------------------------------
// expect / mismatch / bad_count (all optional): the reference's consistency check of the aligner's durations
// (fs2/variance_adaptor.py:289-305): mismatch[b] = sum_j dur[b][j] != expect[b]; every mismatch also bumps the
// persistent counter, so that the host can poll ONE word and read the per-utterance flags only when it is non-zero.
__global__ void lr_cumsum_kernel(const int* __restrict__ dur, int* __restrict__ cum, int* __restrict__ out_lens,
                                 const int* __restrict__ expect, int* __restrict__ mismatch,
                                 int* __restrict__ bad_count, int B, int Ts, int Tm) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int s = 0;
  for (int j = 0; j < Ts; ++j) {
    int d = dur[b * Ts + j];
    s += d > 0 ? d : 0;
    cum[b * Ts + j] = s;
  }
  out_lens[b] = s < Tm ? s : Tm;
  if (expect && mismatch) {
    const int bad = s != expect[b];
    mismatch[b] = bad;
    if (bad && bad_count) atomicAdd(bad_count, 1);
  }
}

// one wavefront per output frame: j = first token with cum[j] > t
__global__ __launch_bounds__(256) void lr_gather_kernel(const float* __restrict__ x, const int* __restrict__ cum,
                                                         const float* __restrict__ table, float* __restrict__ out,
                                                         int* __restrict__ src_idx, int B, int Ts, int Tm, int D) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * Tm) return;
  const int b = (int)(row / Tm), t = (int)(row % Tm);
  const int* cb = cum + b * Ts;
  int lo = 0, hi = Ts;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (cb[mid] > t) hi = mid; else lo = mid + 1;
  }
  const bool valid = lo < Ts;
  if (lane == 0 && src_idx) src_idx[row] = valid ? lo : -1;
  float4* o = reinterpret_cast<float4*>(out + row * D);
  const float4* s = reinterpret_cast<const float4*>(x + ((long long)b * Ts + (valid ? lo : 0)) * D);
  const float4* p = table ? reinterpret_cast<const float4*>(table + (long long)t * D) : nullptr;
  for (int i = lane; i < (D >> 2); i += 64) {
    float4 v = make_float4(0, 0, 0, 0);
    if (valid) {
      v = s[i];
      if (p) { float4 q = p[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    }
    o[i] = v;
  }
}

// dx[b,j,:] = sum of dy over the token's contiguous frame segment (no atomics)
__global__ __launch_bounds__(256) void lr_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ cum,
                                                      float* __restrict__ dx, int B, int Ts, int Tm, int D) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * Ts) return;
  const int b = (int)(row / Ts), j = (int)(row % Ts);
  const int start = j > 0 ? cum[b * Ts + j - 1] : 0;
  const int end = min(cum[b * Ts + j], Tm);
  for (int i = lane; i < (D >> 2); i += 64) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (int t = start; t < end; ++t) {
      float4 v = reinterpret_cast<const float4*>(dy + ((long long)b * Tm + t) * D)[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(dx + row * D)[i] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// predictor head  fs2/variance_adaptor.py:53-62: out[m] = (x[m,:] . w + b) * (t < lens[b])
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const int* __restrict__ lens,
                                                          float* __restrict__ out, int M, int T, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float s = 0.f;
  for (int i = lane; i < (C >> 2); i += 64) {
    float4 a = reinterpret_cast<const float4*>(x + (long long)row * C)[i];
    float4 b = reinterpret_cast<const float4*>(w)[i];
    s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  s = fs2_wave_sum(s);
  if (lane == 0) {
    const bool on = !lens || (row % T) < lens[row / T];
    out[row] = on ? s + bias[0] : 0.f;
  }
}

constexpr int RD_ROWS = 16;
// dx[m,:] = g[m]*w ; partial[blk][C+1] = (sum_m g[m]*x[m,:], sum_m g[m]),  g = dout * mask
// (16 rows per workgroup: 256 workgroups at the encoder's 4 096 rows; the masked gradients of the rows are staged once)
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                          const float* __restrict__ w, const int* __restrict__ lens,
                                                          float* __restrict__ dx, float* __restrict__ partial, int M,
                                                          int T, int C) {
  __shared__ float gs[RD_ROWS];
  const int r0 = blockIdx.x * RD_ROWS, nr = min(M - r0, RD_ROWS);
  if (threadIdx.x < RD_ROWS) {
    const int r = r0 + threadIdx.x;
    float g = 0.f;
    if (r < M) {
      const bool on = !lens || (r % T) < lens[r / T];
      g = on ? dout[r] : 0.f;
    }
    gs[threadIdx.x] = g;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float wc = w[c];
    float acc = 0.f;
#pragma unroll 4
    for (int i = 0; i < nr; ++i) {
      const float g = gs[i];
      dx[(long long)(r0 + i) * C + c] = g * wc;
      acc += g * x[(long long)(r0 + i) * C + c];
    }
    partial[(long long)blockIdx.x * (C + 1) + c] = acc;
  }
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int i = 0; i < nr; ++i) acc += gs[i];
    partial[(long long)blockIdx.x * (C + 1) + C] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// masked losses  fs2/loss.py:44-106:  weight * mean_{all B*T*C}( f((pred - tgt) * mask) )
//   kind 0 = MSE, 1 = MAE;  tgt_int != null: target = log(int + 1)   (duration loss, :81)
// forward value and d(loss)/d(pred) in one pass
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_loss_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                           const int* __restrict__ tgt_int, const int* __restrict__ lens,
                                                           int B, int T, int C, int kind, float weight,
                                                           float* __restrict__ dpred, float* __restrict__ partial) {
  __shared__ float red[4];
  const long long n = (long long)B * T * C;
  const float inv = weight / (float)n;
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / C;
    const int t = (int)(row % T), b = (int)(row / T);
    float d = 0.f;
    if (t < lens[b]) {
      const float tg = tgt_int ? logf((float)tgt_int[i] + 1.f) : tgt[i];
      d = pred[i] - tg;
    }
    float g;
    if (kind == 0) { s += d * d; g = 2.f * d; }
    else { s += fabsf(d); g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
    if (dpred) dpred[i] = g * inv;
  }
  s = fs2_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void loss_finish_kernel(const float* __restrict__ partial, int nparts, float scale, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) s += (double)partial[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[0] = (float)(s * (double)scale);
}

// ---------------------------------------------------------------------------------------------
// optimizer  fs2/model.py:530-549 (AdamW), fs2/noam.py:20-26, gradient_clip_val fs2/cli/train.py:38
// device-resident step state so that a captured hipGraph advances without new kernel arguments
// ---------------------------------------------------------------------------------------------
struct StepState {  // mirrors the layout documented in fs2hip.h
  unsigned long long step;
  float lr, bc1, bc2, clip_coef, grad_norm, pad;
};
__global__ void step_advance_kernel(StepState* st, float base_lr, float warmup, float beta1, float beta2) {
  const unsigned long long k = st->step + 1;  // this optimizer step (1-based)
  st->step = k;
  const double s = k > 1 ? (double)(k - 1) : 1.0;  // NoamLR's last_epoch = max(1, k-1)
  const double w = (double)warmup;
  const double scale = sqrt(w) * fmin(1.0 / sqrt(s), s / (w * sqrt(w)));
  st->lr = (float)((double)base_lr * scale);
  st->bc1 = (float)(1.0 - pow((double)beta1, (double)k));
  st->bc2 = (float)(1.0 - pow((double)beta2, (double)k));
}
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 << 2; i < n; ++i) s += g[i] * g[i];
  s = fs2_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void clip_finish_kernel(const float* __restrict__ partial, int nparts, float max_norm, float extra_scale,
                                   StepState* st) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) s += (double)partial[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) {
    const double norm = sqrt(s) * (double)extra_scale;
    st->grad_norm = (float)norm;
    double c = max_norm > 0.f ? (double)max_norm / (norm + 1e-6) : 1.0;
    st->clip_coef = (float)((c < 1.0 ? c : 1.0) * (double)extra_scale);
  }
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, long long n,
                                                     const StepState* __restrict__ st, float beta1, float beta2,
                                                     float eps, float wd) {
  const float lr = st->lr, coef = st->clip_coef;
  const float step_size = lr / st->bc1, inv_bc2_sqrt = rsqrtf(st->bc2), decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] * decay - step_size * mi / (sqrtf(vi) * inv_bc2_sqrt + eps);
  }
}

// out = a * x * dropmask(idx) + b * y     (y may be null)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     float* __restrict__ out, long long n, float a, float b,
                                                     Fs2Drop drop_in) {
  const Fs2Drop drop = fs2_resolve_drop(drop_in);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = a * x[i] * fs2_drop_factor(drop, (unsigned long long)i);
    if (y) v += b * y[i];
    out[i] = v;
  }
}

// x *= *scalar, the scalar in device memory (the upstream gradient autograd hands to the step's backward node:
// exactly 1 unless the caller scaled the loss, and then the pass is skipped)
__global__ __launch_bounds__(256) void scale_dev_kernel(float* __restrict__ x, long long n, const float* __restrict__ scalar) {
  const float s = *scalar;
  if (s == 1.0f) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    x[i] *= s;
}

// out[b,t,:] = x[b,t,:] + e[b,:]   (speaker / language / style vectors, fs2/model.py:203-213)
__global__ __launch_bounds__(256) void add_rowvec_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                                          float* __restrict__ out, int B, int T, int D) {
  const long long n = (long long)B * T * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long long)T * D));
    out[i] = x[i] + e[(long long)b * D + (int)(i % D)];
  }
}

// out = dy * act'(aux)  (aux = the activation's OUTPUT for ReLU, its input otherwise)
__global__ __launch_bounds__(256) void dact_mul_kernel(const float* __restrict__ dy, const float* __restrict__ aux,
                                                        float* __restrict__ out, long long n, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = dy[i] * fs2_dact(act, aux[i]);
}

// fs2/utils/heavy.py:11-15  mask[b][t] = t < lens[b]   (bool as uint8)
__global__ void mask_from_lens_kernel(const int* __restrict__ lens, unsigned char* __restrict__ mask, int B, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * T) mask[i] = (i % T) < lens[i / T] ? 1 : 0;
}

__global__ void sum_slots_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += x[i];
  out[0] = s;
}

// fs2/variance_adaptor.py:360-366: clamp(round(exp(logd) - 1) * control, min=0).int(); torch.round is
// round-half-to-even = rintf; .int() truncates.  The integer result flips where exp(x) - 1 lands on k + 0.5, i.e.
// where the fp32 exponential itself is exactly k + 1.5: a 1-ulp difference between two fp32 exp implementations
// (this chip's expf and the host's vectorised one are both "<= 1 ulp", not the same function) would change a
// duration there.  The exponential is therefore taken in double and rounded once -- the correctly rounded fp32
// value, which is what the host's exp returns everywhere it is not off by its allowed ulp.
__global__ void duration_round_kernel(const float* __restrict__ logd, float control, int* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float e = (float)exp((double)logd[i]);
  const float v = rintf(e - 1.f) * control;
  out[i] = (int)fmaxf(v, 0.f);
}

inline unsigned grid_for(long long n, int per_block = 256, long long cap = 4096) {
  long long b = (n + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int fs2hip_posenc_table(const float* inv_freq, float* table, int T, int D, void* stream) {
  if (T <= 0 || D <= 0 || (D % 2)) return FS2HIP_EINVAL;
  posenc_table_kernel<<<dim3((T * D + 255) / 256), dim3(256), 0, S_>>>(inv_freq, table, T, D);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_add_posenc(const float* x, const float* table, const int* lens, float* out, int B, int T, int D,
                                 void* stream) {
  if (B <= 0 || T <= 0 || D <= 0 || (D % 4)) return FS2HIP_EINVAL;
  add_posenc_kernel<<<dim3(grid_for((long long)B * T * D / 4)), dim3(256), 0, S_>>>(x, table, lens, out, B, T, D);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_embedding_fwd(const int* idx, const float* W, float* out, int M, int V, int D, void* stream) {
  if (M <= 0 || V <= 0 || D <= 0 || (D % 4)) return FS2HIP_EINVAL;
  embedding_fwd_kernel<<<dim3((M + 3) / 4), dim3(256), 0, S_>>>(idx, W, out, M, V, D);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_onehot(const int* idx, float* out, int M, int Vp, int padding_idx, void* stream) {
  if (M <= 0 || Vp <= 0 || (Vp % 4)) return FS2HIP_EINVAL;
  onehot_kernel<<<dim3(grid_for((long long)M * Vp, 256, 8192)), dim3(256), 0, S_>>>(idx, out, M, Vp, padding_idx);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_bucket_embed_add(const float* val, float control, const float* bins, int NB, const float* W,
                                       const float* x, float* out, int* idx_out, int M, int D, void* stream) {
  if (M <= 0 || NB <= 0 || D <= 0 || (D % 4)) return FS2HIP_EINVAL;
  bucket_embed_add_kernel<<<dim3((M + 3) / 4), dim3(256), 0, S_>>>(val, control, bins, NB, W, x, out, idx_out, M, D);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_length_regulate_fwd(const float* x, const int* dur, const float* posenc_table, float* out,
                                          int* cum, int* out_lens, int* src_idx, int B, int Ts, int Tm, int D,
                                          void* stream) {
  if (B <= 0 || Ts <= 0 || Tm <= 0 || D <= 0 || (D % 4)) return FS2HIP_EINVAL;
  lr_cumsum_kernel<<<dim3((B + 63) / 64), dim3(64), 0, S_>>>(dur, cum, out_lens, nullptr, nullptr, nullptr, B, Ts, Tm);
  FS2_LAUNCH_CHECK();
  const long long rows = (long long)B * Tm;
  lr_gather_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_>>>(x, cum, posenc_table, out, src_idx, B, Ts, Tm, D);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_duration_cumsum(const int* dur, int* cum, int* out_lens, const int* expect_lens, int* mismatch,
                                      int* bad_count, int B, int Ts, int Tm, void* stream) {
  if (B <= 0 || Ts <= 0 || ((expect_lens == nullptr) != (mismatch == nullptr))) return FS2HIP_EINVAL;
  lr_cumsum_kernel<<<dim3((B + 63) / 64), dim3(64), 0, S_>>>(dur, cum, out_lens, expect_lens, mismatch, bad_count, B, Ts, Tm);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_length_regulate_bwd(const float* dy, const int* cum, float* dx, int B, int Ts, int Tm, int D,
                                          void* stream) {
  if (B <= 0 || Ts <= 0 || Tm <= 0 || D <= 0 || (D % 4)) return FS2HIP_EINVAL;
  const long long rows = (long long)B * Ts;
  lr_bwd_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_>>>(dy, cum, dx, B, Ts, Tm, D);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_rowdot_fwd(const float* x, const float* w, const float* bias, const int* lens, float* out, int M,
                                 int T, int C, void* stream) {
  if (M <= 0 || T <= 0 || C <= 0 || (C % 4)) return FS2HIP_EINVAL;
  rowdot_fwd_kernel<<<dim3((M + 3) / 4), dim3(256), 0, S_>>>(x, w, bias, lens, out, M, T, C);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_rowdot_blocks(int M) { return (M + RD_ROWS - 1) / RD_ROWS; }

extern "C" int fs2hip_rowdot_bwd(const float* dout, const float* x, const float* w, const int* lens, float* dx,
                                 float* partial, float* dw, float* dbias, int M, int T, int C, void* stream) {
  if (M <= 0 || T <= 0 || C <= 0) return FS2HIP_EINVAL;
  const int nblk = fs2hip_rowdot_blocks(M);
  rowdot_bwd_kernel<<<dim3(nblk), dim3(256), 0, S_>>>(dout, x, w, lens, dx, partial, M, T, C);
  FS2_LAUNCH_CHECK();
  if (C + 1 <= 16384 && nblk >= 8)  // (both sums in one launch: the path fs2hip_reduce_slabs would take for each)
    return fs2_reduce_rows(partial, nblk, C + 1, C + 1, dw, C, dbias, S_);
  int rc = fs2hip_reduce_slabs(partial, dw, C, nblk, C + 1, stream);
  if (rc) return rc;
  return fs2hip_reduce_slabs(partial + C, dbias, 1, nblk, C + 1, stream);
}

extern "C" int fs2hip_masked_loss(const float* pred, const float* tgt, const int* tgt_int, const int* lens, int B, int T,
                                  int C, int kind, float weight, float* dpred, float* partial, float* loss_out,
                                  void* stream) {
  if (B <= 0 || T <= 0 || C <= 0 || (!tgt && !tgt_int) || !lens) return FS2HIP_EINVAL;
  const long long n = (long long)B * T * C;
  const unsigned nb = grid_for(n, 256, 1024);
  masked_loss_kernel<<<dim3(nb), dim3(256), 0, S_>>>(pred, tgt, tgt_int, lens, B, T, C, kind, weight, dpred, partial);
  FS2_LAUNCH_CHECK();
  loss_finish_kernel<<<dim3(1), dim3(64), 0, S_>>>(partial, (int)nb, weight / (float)n, loss_out);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_step_advance(void* state, float base_lr, float warmup, float beta1, float beta2, void* stream) {
  step_advance_kernel<<<dim3(1), dim3(1), 0, S_>>>((StepState*)state, base_lr, warmup, beta1, beta2);
  FS2_LAUNCH_CHECK();
  return 0;
}

// partial: >= 1024 floats.  grad_scale multiplies every gradient first (1/world_size after a sum all-reduce).
extern "C" int fs2hip_grad_clip_coef(const float* grad, long long n, float max_norm, float grad_scale, float* partial,
                                     void* state, void* stream) {
  if (n <= 0 || ((uintptr_t)grad % 16)) return FS2HIP_EINVAL;
  const unsigned nb = grid_for(n / 4, 256, 1024);
  sumsq_kernel<<<dim3(nb), dim3(256), 0, S_>>>(grad, n, partial);
  FS2_LAUNCH_CHECK();
  clip_finish_kernel<<<dim3(1), dim3(64), 0, S_>>>(partial, (int)nb, max_norm, grad_scale, (StepState*)state);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_adamw_step(float* p, const float* g, float* m, float* v, long long n, const void* state,
                                 float beta1, float beta2, float eps, float weight_decay, void* stream) {
  if (n <= 0) return FS2HIP_EINVAL;
  adamw_kernel<<<dim3(grid_for(n, 256, 8192)), dim3(256), 0, S_>>>(p, g, m, v, n, (const StepState*)state, beta1, beta2,
                                                                    eps, weight_decay);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_axpby(const float* x, const float* y, float* out, long long n, float a, float b, float drop_p,
                            unsigned long long drop_seed, const unsigned long long* drop_step, void* stream) {
  if (n <= 0) return FS2HIP_EINVAL;
  axpby_kernel<<<dim3(grid_for(n, 256, 8192)), dim3(256), 0, S_>>>(x, y, out, n, a, b,
                                                                    fs2_make_drop(drop_p, drop_seed, drop_step));
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_scale_dev(float* x, long long n, const float* scalar, void* stream) {
  if (n <= 0 || !scalar) return FS2HIP_EINVAL;
  scale_dev_kernel<<<dim3(grid_for(n, 256, 8192)), dim3(256), 0, S_>>>(x, n, scalar);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_dact_mul(const float* dy, const float* aux, float* out, long long n, int act, void* stream) {
  if (n <= 0) return FS2HIP_EINVAL;
  dact_mul_kernel<<<dim3(grid_for(n, 256, 8192)), dim3(256), 0, S_>>>(dy, aux, out, n, act);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_mask_from_lens(const int* lens, unsigned char* mask, int B, int T, void* stream) {
  if (B <= 0 || T <= 0) return FS2HIP_EINVAL;
  mask_from_lens_kernel<<<dim3((B * T + 255) / 256), dim3(256), 0, S_>>>(lens, mask, B, T);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_sum_slots(const float* x, int n, float* out, void* stream) {
  if (n <= 0) return FS2HIP_EINVAL;
  sum_slots_kernel<<<dim3(1), dim3(1), 0, S_>>>(x, n, out);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_duration_round(const float* logd, float control, int* out, int n, void* stream) {
  if (n <= 0) return FS2HIP_EINVAL;
  duration_round_kernel<<<dim3((n + 255) / 256), dim3(256), 0, S_>>>(logd, control, out, n);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_add_rowvec(const float* x, const float* e, float* out, int B, int T, int D, void* stream) {
  if (B <= 0 || T <= 0 || D <= 0) return FS2HIP_EINVAL;
  add_rowvec_kernel<<<dim3(grid_for((long long)B * T * D, 256, 8192)), dim3(256), 0, S_>>>(x, e, out, B, T, D);
  FS2_LAUNCH_CHECK();
  return 0;
}
