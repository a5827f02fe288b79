// GST style encoder kernels (reference fs2/gst/model.py:103-257, fs2/gst/attn.py:48-194) -- BASELINE config 5.
// The whole branch is ~0.3 % of the step's FLOPs (6 small stride-2 convs over the mel, an 11-step GRU(128),
// a 10-token attention), so these are simple VALU kernels laid out for coalescing, not MFMA work:
//   conv2d_s2       : 3x3, stride 2, pad 1, no bias, channels-last [B][H][W][C]; weights [kh][kw][ci][co]
//   gru_gate        : r, z, n gates + state update of one time step (the two matmuls run on the GEMM kernel)
//   gst_attn        : softmax(q k^T / sqrt(d_k)) v of one query per utterance against the style tokens
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void conv2d_s2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, int B, int H, int W, int Cin,
                                                             int Ho, int Wo, int Cout) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Ho * Wo * Cout;
  if (idx >= total) return;
  const int co = (int)(idx % Cout);
  long long row = idx / Cout;
  const int wo = (int)(row % Wo);
  row /= Wo;
  const int ho = (int)(row % Ho), b = (int)(row / Ho);
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho + kh - 1;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo + kw - 1;
      if (wi < 0 || wi >= W) continue;
      const float* xp = x + (((long long)b * H + hi) * W + wi) * Cin;
      const float* wp = w + (long long)(kh * 3 + kw) * Cin * Cout + co;
      for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], wp[(long long)ci * Cout], acc);
    }
  }
  y[idx] = acc;
}

// The reference encoder's FIRST layer (fs2/gst/model.py:103-139: Conv2d(1, 32, 3, stride 2) over the mel "image"): one
// input channel, so a thread of the generic kernel above loaded 9 + 9 scalars for ONE output and the launch -- 53 M
// threads for the configs[4] batch -- took 419 us to write 211 MB (round 5 profile; the style encoder's forward pass is
// what the main stream waits for at its join).  Here a thread makes four neighbouring output channels of a pixel: the
// nine taps once, nine 16-byte weight loads, one 16-byte store; the same fmaf sequence per output (taps outside the
// image skipped, in the same order), so the results are the generic kernel's bit for bit.
__global__ __launch_bounds__(256) void conv2d_s2_fwd_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ y, int B, int H, int W, int Ho, int Wo,
                                                                int Cout) {
  const int c4n = Cout >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Ho * Wo * c4n;
  if (idx >= total) return;
  const int c4 = (int)(idx % c4n);
  long long row = idx / c4n;
  const int wo = (int)(row % Wo);
  row /= Wo;
  const int ho = (int)(row % Ho), b = (int)(row / Ho);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho + kh - 1;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo + kw - 1;
      if (wi < 0 || wi >= W) continue;
      const float xv = x[((long long)b * H + hi) * W + wi];
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (kh * 3 + kw) * Cout + 4 * c4);
      acc[0] = fmaf(xv, wv[0], acc[0]);
      acc[1] = fmaf(xv, wv[1], acc[1]);
      acc[2] = fmaf(xv, wv[2], acc[2]);
      acc[3] = fmaf(xv, wv[3], acc[3]);
    }
  }
  *reinterpret_cast<f32x4*>(y + idx * 4) = acc;
}

// dx[b,hi,wi,ci] = sum over (kh,kw) with hi = 2ho+kh-1, wi = 2wo+kw-1 of sum_co dy[b,ho,wo,co] * w[kh][kw][ci][co]
__global__ __launch_bounds__(256) void conv2d_s2_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                                  float* __restrict__ dx, int B, int H, int W, int Cin,
                                                                  int Ho, int Wo, int Cout) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * H * W * Cin;
  if (idx >= total) return;
  const int ci = (int)(idx % Cin);
  long long row = idx / Cin;
  const int wi = (int)(row % W);
  row /= W;
  const int hi = (int)(row % H), b = (int)(row / H);
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int h2 = hi + 1 - kh;
    if (h2 < 0 || (h2 & 1) || (h2 >> 1) >= Ho) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int w2 = wi + 1 - kw;
      if (w2 < 0 || (w2 & 1) || (w2 >> 1) >= Wo) continue;
      const float* dp = dy + (((long long)b * Ho + (h2 >> 1)) * Wo + (w2 >> 1)) * Cout;
      const float* wp = w + ((long long)(kh * 3 + kw) * Cin + ci) * Cout;
      for (int co = 0; co < Cout; ++co) acc = fmaf(dp[co], wp[co], acc);
    }
  }
  dx[idx] = acc;
}

constexpr int CW_ROWS = 2048;  // output rows per partial of the weight gradient
// partial[chunk][kh][kw][ci][co] = sum over the chunk's output rows of x[...] * dy[row][co]
__global__ __launch_bounds__(256) void conv2d_s2_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ partial, int B, int H, int W,
                                                                    int Cin, int Ho, int Wo, int Cout) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int nw = 9 * Cin * Cout;
  if (e >= nw) return;
  const int co = e % Cout, ci = (e / Cout) % Cin, k = e / (Cout * Cin);
  const int kh = k / 3, kw = k % 3;
  const long long rows = (long long)B * Ho * Wo;
  const long long r0 = (long long)blockIdx.y * CW_ROWS, r1 = min(rows, r0 + CW_ROWS);
  float acc = 0.f;
  for (long long r = r0; r < r1; ++r) {
    const int wo = (int)(r % Wo);
    const long long t = r / Wo;
    const int ho = (int)(t % Ho), b = (int)(t / Ho);
    const int hi = 2 * ho + kh - 1, wi = 2 * wo + kw - 1;
    if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
    acc = fmaf(x[(((long long)b * H + hi) * W + wi) * Cin + ci], dy[r * Cout + co], acc);
  }
  partial[(long long)blockIdx.y * nw + e] = acc;
}

// ---- GRU (nn.GRU gate order r, z, n) ----------------------------------------------------------------
// gi: rows of stride gi_stride (input projections incl. b_ih of step t), gh [B][3U] (recurrent incl. b_hh)
// ---- 3x3 stride-2 convolution as GEMM (Cin % 4 == 0): gather the 9 taps of every output pixel into a row of
// col[B*Ho*Wo][9*Cin] (zero where the window leaves the image), multiply by the weight seen as [9*Cin][Cout] on the
// MFMA GEMM; backward data is the GEMM dcol = dy W^T followed by the inverse gather (each input pixel belongs to at
// most four windows).  The direct kernels above reached ~1 TFLOP/s, which made the style encoder 30 % of a
// configs[4]-shaped step; only the first layer (Cin = 1, 9 MACs per output) still uses them.
__global__ __launch_bounds__(256) void im2col_s2_kernel(const float* __restrict__ x, float* __restrict__ col, int B,
                                                         int H, int W, int Cin, int Ho, int Wo) {
  const int c4n = Cin >> 2;
  const long long total = (long long)B * Ho * Wo * 9 * c4n;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c4 = (int)(idx % c4n);
    long long r = idx / c4n;
    const int tap = (int)(r % 9);
    r /= 9;  // output pixel (b, ho, wo)
    const int wo = (int)(r % Wo);
    const long long bh = r / Wo;
    const int ho = (int)(bh % Ho), b = (int)(bh / Ho);
    const int hi = 2 * ho + tap / 3 - 1, wi = 2 * wo + tap % 3 - 1;
    float4 v = make_float4(0, 0, 0, 0);
    if (hi >= 0 && hi < H && wi >= 0 && wi < W)
      v = *reinterpret_cast<const float4*>(x + (((long long)b * H + hi) * W + wi) * Cin + c4 * 4);
    *reinterpret_cast<float4*>(col + r * (9LL * Cin) + (long long)tap * Cin + c4 * 4) = v;
  }
}

// single input channel (the mel itself): rows of 12 floats = 9 taps + 3 zeros, for the weight-gradient GEMM
__global__ __launch_bounds__(256) void im2col_s2_c1_kernel(const float* __restrict__ x, float* __restrict__ col, int B,
                                                            int H, int W, int Ho, int Wo) {
  const long long total = (long long)B * Ho * Wo;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < total; r += (long long)gridDim.x * 256) {
    const int wo = (int)(r % Wo);
    const long long bh = r / Wo;
    const int ho = (int)(bh % Ho), b = (int)(bh / Ho);
    float v[12];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hi = 2 * ho + tap / 3 - 1, wi = 2 * wo + tap % 3 - 1;
      v[tap] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((long long)b * H + hi) * W + wi] : 0.f;
    }
    v[9] = v[10] = v[11] = 0.f;
    float4* dst = reinterpret_cast<float4*>(col + r * 12);
    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
    dst[2] = make_float4(v[8], v[9], v[10], v[11]);
  }
}

__global__ __launch_bounds__(256) void col2im_s2_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B,
                                                         int H, int W, int Cin, int Ho, int Wo) {
  const int c4n = Cin >> 2;
  const long long total = (long long)B * H * W * c4n;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c4 = (int)(idx % c4n);
    long long r = idx / c4n;
    const int wi = (int)(r % W);
    const long long bh = r / W;
    const int hi = (int)(bh % H), b = (int)(bh / H);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h2 = hi + 1 - kh;
      if (h2 < 0 || (h2 & 1) || (h2 >> 1) >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w2 = wi + 1 - kw;
        if (w2 < 0 || (w2 & 1) || (w2 >> 1) >= Wo) continue;
        const long long m = ((long long)b * Ho + (h2 >> 1)) * Wo + (w2 >> 1);
        const float4 v = *reinterpret_cast<const float4*>(dcol + m * (9LL * Cin) + (long long)(kh * 3 + kw) * Cin + c4 * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dx + r * Cin + c4 * 4) = acc;
  }
}

__global__ void gru_gate_fwd_kernel(const float* __restrict__ gi, long long gi_stride, const float* __restrict__ gh,
                                    const float* __restrict__ hprev, float* __restrict__ hnew, float* __restrict__ gates,
                                    int B, int U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * U) return;
  const int b = i / U, j = i % U;
  const float* gib = gi + (long long)b * gi_stride;
  const float* ghb = gh + (long long)b * 3 * U;
  const float r = fs2_sigmoid(gib[j] + ghb[j]);
  const float z = fs2_sigmoid(gib[U + j] + ghb[U + j]);
  const float hn = ghb[2 * U + j];
  const float n = tanhf(gib[2 * U + j] + r * hn);
  hnew[i] = (1.f - z) * n + z * hprev[i];
  float* g = gates + (long long)b * 4 * U;  // saved for the backward: r, z, n, gh_n
  g[j] = r; g[U + j] = z; g[2 * U + j] = n; g[3 * U + j] = hn;
}
// dh: gradient of h_t; outputs dgi (row stride dgi_stride), dgh [B][3U], dhprev = dh * z (the recurrent matmul's
// contribution is added by the caller)
__global__ void gru_gate_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ gates,
                                    const float* __restrict__ hprev, float* __restrict__ dgi, long long dgi_stride,
                                    float* __restrict__ dgh, float* __restrict__ dhprev, int B, int U) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * U) return;
  const int b = i / U, j = i % U;
  const float* g = gates + (long long)b * 4 * U;
  const float r = g[j], z = g[U + j], n = g[2 * U + j], hn = g[3 * U + j];
  const float d = dh[i];
  const float dn = d * (1.f - z) * (1.f - n * n);
  const float dz = d * (hprev[i] - n) * z * (1.f - z);
  const float dr = dn * hn * r * (1.f - r);
  float* a = dgi + (long long)b * dgi_stride;
  float* c = dgh + (long long)b * 3 * U;
  a[j] = dr; a[U + j] = dz; a[2 * U + j] = dn;
  c[j] = dr; c[U + j] = dz; c[2 * U + j] = dn * r;
  dhprev[i] = d * z;
}

// ---- style-token attention: one query per utterance, NT tokens, HEADS x 64 dims (one wavefront per head) ----
constexpr int GST_MAX_TOKENS = 32;
__global__ __launch_bounds__(256) void gst_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ p,
                                                            float* __restrict__ ctx, int B, int NT, int heads) {
  const int b = blockIdx.x, h = threadIdx.x >> 6, d = threadIdx.x & 63;
  if (h >= heads) return;
  const int F = heads * 64;
  const float qv = q[(long long)b * F + h * 64 + d];
  float s[GST_MAX_TOKENS];
  float m = -INFINITY;
  for (int j = 0; j < NT; ++j) {
    s[j] = fs2_wave_sum(qv * k[(long long)j * F + h * 64 + d]) * 0.125f;  // 1/sqrt(64)
    m = fmaxf(m, s[j]);
  }
  float den = 0.f;
  for (int j = 0; j < NT; ++j) { s[j] = expf(s[j] - m); den += s[j]; }
  float acc = 0.f;
  for (int j = 0; j < NT; ++j) {
    const float pj = s[j] / den;
    if (d == 0) p[((long long)b * heads + h) * NT + j] = pj;
    acc = fmaf(pj, v[(long long)j * F + h * 64 + d], acc);
  }
  ctx[(long long)b * F + h * 64 + d] = acc;
}
// dq [B][F]; dk_part, dv_part [B][NT][F] (summed over the batch by the caller)
__global__ __launch_bounds__(256) void gst_attn_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ q,
                                                            const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ p, float* __restrict__ dq,
                                                            float* __restrict__ dk_part, float* __restrict__ dv_part,
                                                            int B, int NT, int heads) {
  const int b = blockIdx.x, h = threadIdx.x >> 6, d = threadIdx.x & 63;
  if (h >= heads) return;
  const int F = heads * 64;
  const long long o = (long long)b * F + h * 64 + d;
  const float dc = dctx[o], qv = q[o];
  const float* pb = p + ((long long)b * heads + h) * NT;
  float dp[GST_MAX_TOKENS];
  float dot = 0.f;
  for (int j = 0; j < NT; ++j) {
    dp[j] = fs2_wave_sum(dc * v[(long long)j * F + h * 64 + d]);
    dot += pb[j] * dp[j];
  }
  float acc = 0.f;
  for (int j = 0; j < NT; ++j) {
    const float ds = pb[j] * (dp[j] - dot) * 0.125f;
    acc = fmaf(ds, k[(long long)j * F + h * 64 + d], acc);
    dk_part[((long long)b * NT + j) * F + h * 64 + d] = ds * qv;
    dv_part[((long long)b * NT + j) * F + h * 64 + d] = pb[j] * dc;
  }
  dq[o] = acc;
}

// out = act(x)
__global__ void act_apply_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, int act) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = fs2_act(act, x[i]);
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int fs2hip_conv2d_s2_fwd(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout,
                                    void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return FS2HIP_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * Ho * Wo * Cout;
  if (Cin == 1 && (Cout % 4) == 0 && !((uintptr_t)w % 16) && !((uintptr_t)y % 16) && total / 4 < 0x7fffffffLL * 256) {
    conv2d_s2_fwd_c1_kernel<<<dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, S_>>>(x, w, y, B, H, W, Ho, Wo, Cout);
    FS2_LAUNCH_CHECK();
    return 0;
  }
  conv2d_s2_fwd_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S_>>>(x, w, y, B, H, W, Cin, Ho, Wo, Cout);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_conv2d_s2_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout,
                                         void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return FS2HIP_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * H * W * Cin;
  conv2d_s2_bwd_data_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S_>>>(dy, w, dx, B, H, W, Cin, Ho, Wo, Cout);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_im2col_s2(const float* x, float* col, int B, int H, int W, int Cin, void* stream) {
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  if (Cin == 1 && B > 0 && H > 0 && W > 0 && !((uintptr_t)col % 16)) {  // col[M][12]
    long long nb = ((long long)B * Ho * Wo + 255) / 256;
    if (nb > 65535 * 16) nb = 65535 * 16;
    im2col_s2_c1_kernel<<<dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream>>>(x, col, B, H, W, Ho, Wo);
    FS2_LAUNCH_CHECK();
    return 0;
  }
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || ((uintptr_t)x % 16) || ((uintptr_t)col % 16)) return FS2HIP_EINVAL;
  long long blocks = ((long long)B * Ho * Wo * 9 * (Cin >> 2) + 255) / 256;
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  im2col_s2_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(x, col, B, H, W, Cin, Ho, Wo);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_col2im_s2(const float* dcol, float* dx, int B, int H, int W, int Cin, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin & 3) || ((uintptr_t)dx % 16) || ((uintptr_t)dcol % 16)) return FS2HIP_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  long long blocks = ((long long)B * H * W * (Cin >> 2) + 255) / 256;
  if (blocks > 65535 * 16) blocks = 65535 * 16;
  col2im_s2_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(dcol, dx, B, H, W, Cin, Ho, Wo);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_conv2d_s2_wgrad_parts(int B, int H, int W) {
  const long long rows = (long long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
  return (int)((rows + CW_ROWS - 1) / CW_ROWS);
}

// partial: parts * 9*Cin*Cout floats; dw [3][3][Cin][Cout] is finished here
extern "C" int fs2hip_conv2d_s2_bwd_weight(const float* x, const float* dy, float* partial, float* dw, int B, int H, int W,
                                           int Cin, int Cout, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return FS2HIP_EINVAL;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int nw = 9 * Cin * Cout, parts = fs2hip_conv2d_s2_wgrad_parts(B, H, W);
  conv2d_s2_bwd_weight_kernel<<<dim3((nw + 255) / 256, parts), dim3(256), 0, S_>>>(x, dy, partial, B, H, W, Cin, Ho, Wo, Cout);
  FS2_LAUNCH_CHECK();
  return fs2hip_reduce_slabs(partial, dw, nw, parts, nw, stream);
}

extern "C" int fs2hip_gru_gate_fwd(const float* gi, long long gi_stride, const float* gh, const float* hprev, float* hnew,
                                   float* gates, int B, int U, void* stream) {
  if (B <= 0 || U <= 0) return FS2HIP_EINVAL;
  gru_gate_fwd_kernel<<<dim3((B * U + 255) / 256), dim3(256), 0, S_>>>(gi, gi_stride, gh, hprev, hnew, gates, B, U);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_gru_gate_bwd(const float* dh, const float* gates, const float* hprev, float* dgi, long long dgi_stride,
                                   float* dgh, float* dhprev, int B, int U, void* stream) {
  if (B <= 0 || U <= 0) return FS2HIP_EINVAL;
  gru_gate_bwd_kernel<<<dim3((B * U + 255) / 256), dim3(256), 0, S_>>>(dh, gates, hprev, dgi, dgi_stride, dgh, dhprev, B, U);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_gst_attn_fwd(const float* q, const float* k, const float* v, float* p, float* ctx, int B, int NT,
                                   int heads, void* stream) {
  if (B <= 0 || NT <= 0 || NT > GST_MAX_TOKENS || heads <= 0 || heads > 4) return FS2HIP_EINVAL;
  gst_attn_fwd_kernel<<<dim3(B), dim3(256), 0, S_>>>(q, k, v, p, ctx, B, NT, heads);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_gst_attn_bwd(const float* dctx, const float* q, const float* k, const float* v, const float* p,
                                   float* dq, float* dk_part, float* dv_part, int B, int NT, int heads, void* stream) {
  if (B <= 0 || NT <= 0 || NT > GST_MAX_TOKENS || heads <= 0 || heads > 4) return FS2HIP_EINVAL;
  gst_attn_bwd_kernel<<<dim3(B), dim3(256), 0, S_>>>(dctx, q, k, v, p, dq, dk_part, dv_part, B, NT, heads);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_act_apply(const float* x, float* out, long long n, int act, void* stream) {
  if (n <= 0) return FS2HIP_EINVAL;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  act_apply_kernel<<<dim3((unsigned)blocks), dim3(256), 0, S_>>>(x, out, n, act);
  FS2_LAUNCH_CHECK();
  return 0;
}
