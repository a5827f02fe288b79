// Learned-alignment path (reference fs2/attn/attention.py:195-251, fs2/attn/alignment.py:48-74,
// fs2/attn/attention_loss.py:22-73, fs2/variance_adaptor.py:160-222, :267-268).
//
//   attn_dist      : logits[b,t1,t2] = -0.0005 * sum_c (q[b,t1,c] - k[b,t2,c])^2   -- never materialises the
//                    reference's (B, C, T1, T2) difference tensor (850 MB at the benchmark shape)
//   attn_softmax   : attn_logprob = log_softmax(logits) + log(prior + 1e-8); attn_soft = softmax over valid keys
//   mas            : monotonic alignment search, one workgroup per utterance (row-sequential DP with the
//                    direction bits kept on chip, then a serial backtrack); adds and max only -> bit-exact
//   avg_variance   : phone-level mean of frame-level pitch/energy over each token's frames (non-zero count)
//   ctc            : CTC loss of attn_logprob (blank column -1, targets 1..L) value and gradient
//   bin            : binarisation loss value/gradient
//   softmax_bwd    : gradient of the logits; dist_bwd_q / dist_bwd_k : gradients of the projections
#include "common.h"

namespace {

constexpr float NEG_INF = -INFINITY;

// lane i <- lane i-1 (lane 0 gets `fill`) / lane i <- lane i+1 (lane 63 gets `fill`): one DPP move instead of a
// ds_bpermute round trip through the LDS crossbar -- these sit on the serial critical path of the recursions below
__device__ __forceinline__ float wave_shr1(float x, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, x),
                                                               0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float x, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, x),
                                                               0x130, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_dist_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         float* __restrict__ logits, int B, int T1, int T2, int C) {
  extern __shared__ float sm[];  // qs[16][C] | ks[64][C+1]
  float* qs = sm;
  float* ks = sm + 16 * C;
  const int b = blockIdx.y, t10 = blockIdx.x * 16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * C; i += 256) {
    int r = i / C, c = i % C;
    qs[i] = (t10 + r < T1) ? q[((long long)b * T1 + t10 + r) * C + c] : 0.f;
  }
  for (int t20 = 0; t20 < T2; t20 += 64) {
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * C; i += 256) {
      int r = i / C, c = i % C;
      ks[r * (C + 1) + c] = (t20 + r < T2) ? k[((long long)b * T2 + t20 + r) * C + c] : 0.f;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* kr = ks + lane * (C + 1);
    for (int c = 0; c < C; ++c) {
      const float kv = kr[c];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = qs[(wave + 4 * r) * C + c] - kv;
        acc[r] = fmaf(d, d, acc[r]);
      }
    }
    if (t20 + lane < T2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t1 = t10 + wave + 4 * r;
        if (t1 < T1) logits[((long long)b * T1 + t1) * T2 + t20 + lane] = -0.0005f * acc[r];
      }
    }
  }
}

// one wavefront per (b, t1) row
__global__ __launch_bounds__(256) void attn_softmax_kernel(const float* __restrict__ logits, const float* __restrict__ prior,
                                                            const int* __restrict__ key_lens, float* __restrict__ logprob,
                                                            float* __restrict__ soft, int B, int T1, int T2) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * T1) return;
  const int b = (int)(row / T1);
  const int L = key_lens[b];
  const float* x = logits + row * T2;
  float m = NEG_INF;
  for (int j = lane; j < T2; j += 64) m = fmaxf(m, x[j]);
  m = fs2_wave_max(m);
  float s = 0.f;
  for (int j = lane; j < T2; j += 64) s += expf(x[j] - m);
  const float lse = m + logf(fs2_wave_sum(s));
  float m2 = NEG_INF;
  for (int j = lane; j < T2; j += 64) {
    const float lp = (x[j] - lse) + logf(prior[row * T2 + j] + 1e-8f);
    logprob[row * T2 + j] = lp;
    if (j < L) m2 = fmaxf(m2, lp);
  }
  m2 = fs2_wave_max(m2);
  float s2 = 0.f;
  for (int j = lane; j < L && j < T2; j += 64) s2 += expf(logprob[row * T2 + j] - m2);
  s2 = fs2_wave_sum(s2);
  for (int j = lane; j < T2; j += 64) soft[row * T2 + j] = j < L ? expf(logprob[row * T2 + j] - m2) / s2 : 0.f;
}

// ---------------------------------------------------------------------------------------------------
// MAS: in = attn_soft (is_log = 0, log applied here) or log-probabilities (is_log = 1).
// dirs: uint32 bit matrix [T1][W] (bit j of row i = "move left when leaving (i, j)"), in LDS when it fits.
__global__ __launch_bounds__(256) void mas_kernel(const float* __restrict__ in, int is_log, const int* __restrict__ in_lens,
                                                   const int* __restrict__ out_lens, float* __restrict__ hard,
                                                   int* __restrict__ hard_idx, int* __restrict__ dur,
                                                   unsigned* __restrict__ dirs_global, int B, int Tm, int Ts, int W,
                                                   int dirs_in_lds) {
  extern __shared__ unsigned smem_u[];
  const int b = blockIdx.x, tid = threadIdx.x;
  int T1 = out_lens[b], T2 = in_lens[b];
  T1 = T1 < 0 ? 0 : (T1 > Tm ? Tm : T1);
  T2 = T2 < 1 ? 1 : (T2 > Ts ? Ts : T2);
  float* prev = reinterpret_cast<float*>(smem_u);           // [2][Ts]
  unsigned* dirs = dirs_in_lds ? smem_u + 2 * Ts : dirs_global + (long long)b * Tm * W;
  const float* src = in + (long long)b * Tm * Ts;
  float* hb = hard + (long long)b * Tm * Ts;
  for (long long i = tid; i < (long long)Tm * Ts; i += 256) hb[i] = 0.f;
  for (int j = tid; j < Ts; j += 256) dur[b * Ts + j] = 0;
  for (int i = tid; i < Tm; i += 256) hard_idx[b * Tm + i] = -1;
  if (T1 == 0) return;
  for (int j = tid; j < T2; j += 256) {
    float v = src[j];
    prev[j] = j == 0 ? (is_log ? v : logf(v)) : NEG_INF;
  }
  __syncthreads();
  int cur = 0;
  // The recursion over rows is serial, but its inputs are not: the log-probabilities of the next MAS_CHUNK rows are
  // fetched (and logged) into registers while the current chunk runs, so a row costs an LDS exchange + barrier and
  // no longer a dependent global load.  (Ts <= 256 tokens per pass of the column loop.)
  constexpr int MAS_CHUNK = 8;
  if (T2 <= 256) {
    const int j = tid;
    const bool act = j < T2;
    float nxt[MAS_CHUNK], curv[MAS_CHUNK];
    auto fetch = [&](int i0, float (&dst)[MAS_CHUNK]) {
#pragma unroll
      for (int r = 0; r < MAS_CHUNK; ++r) {
        const int i = i0 + r;
        float v = (act && i < T1) ? src[(long long)i * Ts + j] : 1.f;
        dst[r] = is_log ? v : logf(v);
      }
    };
    fetch(1, nxt);
    for (int i0 = 1; i0 < T1; i0 += MAS_CHUNK) {
#pragma unroll
      for (int r = 0; r < MAS_CHUNK; ++r) curv[r] = nxt[r];
      if (i0 + MAS_CHUNK < T1) fetch(i0 + MAS_CHUNK, nxt);
#pragma unroll
      for (int r = 0; r < MAS_CHUNK; ++r) {
        const int i = i0 + r;
        if (i >= T1) break;  // uniform
        const float* pr = prev + cur * Ts;
        float* nx = prev + (cur ^ 1) * Ts;
        bool left = false;
        if (act) {
          const float p1 = j > 0 ? pr[j - 1] : NEG_INF, p2 = pr[j];
          nx[j] = curv[r] + fmaxf(p1, p2);
          left = j > 0 && p1 >= p2;  // fs2/attn/alignment.py:68 (ties move left)
        }
        const unsigned long long bal = __ballot(left);
        if ((tid & 31) == 0 && j < T2 + 31) {
          const unsigned word = (tid & 32) ? (unsigned)(bal >> 32) : (unsigned)bal;
          const int w = j >> 5;
          if (w < W) dirs[(long long)i * W + w] = word;
        }
        __syncthreads();
        cur ^= 1;
      }
    }
  } else {
    for (int i = 1; i < T1; ++i) {
      const float* pr = prev + cur * Ts;
      float* nx = prev + (cur ^ 1) * Ts;
      for (int j0 = 0; j0 < T2; j0 += 256) {
        const int j = j0 + tid;
        bool left = false;
        if (j < T2) {
          const float p1 = j > 0 ? pr[j - 1] : NEG_INF, p2 = pr[j];
          float v = src[(long long)i * Ts + j];
          if (!is_log) v = logf(v);
          nx[j] = v + fmaxf(p1, p2);
          left = j > 0 && p1 >= p2;  // fs2/attn/alignment.py:68 (ties move left)
        }
        const unsigned long long bal = __ballot(left);
        if ((tid & 31) == 0 && j < T2 + 31) {
          const unsigned word = (tid & 32) ? (unsigned)(bal >> 32) : (unsigned)bal;
          const int w = j >> 5;
          if (w < W) dirs[(long long)i * W + w] = word;
        }
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  if (tid == 0) {
    int j = T2 - 1;
    int i = T1 - 1;
    for (; i >= 1; --i) {
      hb[(long long)i * Ts + j] = 1.f;
      hard_idx[b * Tm + i] = j;
      if (j > 0 && ((dirs[(long long)i * W + (j >> 5)] >> (j & 31)) & 1u)) {
        --j;
        if (j == 0) {
          for (int r = 1; r < i; ++r) {
            hb[(long long)r * Ts] = 1.f;
            hard_idx[b * Tm + r] = 0;
          }
          break;
        }
      }
    }
    hb[j] = 1.f;
    hard_idx[b * Tm] = j;
  }
  __syncthreads();
  // durations = column sums of the hard map (fs2/variance_adaptor.py:267-268)
  for (int i = tid; i < T1; i += 256) {
    const int j = hard_idx[b * Tm + i];
    if (j >= 0) atomicAdd(&dur[b * Ts + j], 1);
  }
}

// Same search for texts of at most 128 tokens (the usual case), ONE wavefront per utterance: lane l owns columns l
// and l + 64 in registers, the left neighbour comes by lane shift, so a row costs two shuffles and a ballot instead
// of an LDS exchange and a workgroup barrier (648 rows: 0.44 ms -> tens of microseconds).  Row inputs are fetched a
// chunk ahead.  Bit-identical to mas_kernel.
__global__ __launch_bounds__(64) void mas_wave_kernel(const float* __restrict__ in, int is_log,
                                                      const int* __restrict__ in_lens, const int* __restrict__ out_lens,
                                                      float* __restrict__ hard, int* __restrict__ hard_idx,
                                                      int* __restrict__ dur, int B, int Tm, int Ts) {
  extern __shared__ unsigned long long dirs64[];  // [Tm][2]: bit j of word j >> 6 = "move left when leaving (i, j)"
  const int b = blockIdx.x, lane = threadIdx.x;
  int T1 = out_lens[b], T2 = in_lens[b];
  T1 = T1 < 0 ? 0 : (T1 > Tm ? Tm : T1);
  T2 = T2 < 1 ? 1 : (T2 > Ts ? Ts : T2);
  const float* src = in + (long long)b * Tm * Ts;
  float* hb = hard + (long long)b * Tm * Ts;  // zeroed by the launcher (a single wavefront is slow at bulk stores)
  for (int j = lane; j < Ts; j += 64) dur[b * Ts + j] = 0;
  for (int i = lane; i < Tm; i += 64) hard_idx[b * Tm + i] = -1;
  if (T1 == 0) return;
  const int ja = lane, jb = lane + 64;
  const bool acta = ja < T2, actb = jb < T2;
  float pa = NEG_INF, pb = NEG_INF;
  if (lane == 0) {
    const float v = src[0];
    pa = is_log ? v : logf(v);
  }
  constexpr int CH = 8;
  float na[CH], nb[CH], va[CH], vb[CH];
  auto fetch = [&](int i0) {
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      const int i = i0 + r;
      const int ii = i < T1 ? i : T1 - 1;  // clamped rows / columns: loads without divergence, values unused when inactive
      const float xa = src[(long long)ii * Ts + (acta ? ja : 0)];
      const float xb = src[(long long)ii * Ts + (actb ? jb : 0)];
      na[r] = is_log ? xa : logf(xa);
      nb[r] = is_log ? xb : logf(xb);
    }
  };
  fetch(1);
  for (int i0 = 1; i0 < T1; i0 += CH) {
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      va[r] = na[r];
      vb[r] = nb[r];
    }
    if (i0 + CH < T1) fetch(i0 + CH);
#pragma unroll
    for (int r = 0; r < CH; ++r) {
      const int i = i0 + r;
      if (i >= T1) break;  // uniform
      const float a63 = __shfl(pa, 63, 64);                 // column 64's left neighbour is column 63
      const float la = wave_shr1(pa, NEG_INF);              // column 0 has no left neighbour
      const float lb = wave_shr1(pb, a63);
      const bool left_a = acta && ja > 0 && la >= pa;  // fs2/attn/alignment.py:68 (ties move left)
      const bool left_b = actb && lb >= pb;
      const float xa = va[r] + fmaxf(la, pa), xb = vb[r] + fmaxf(lb, pb);
      const unsigned long long ba = __ballot(left_a), bb = __ballot(left_b);
      if (lane == 0) {
        dirs64[2 * i] = ba;
        dirs64[2 * i + 1] = bb;
      }
      pa = acta ? xa : NEG_INF;
      pb = actb ? xb : NEG_INF;
    }
  }
  __syncthreads();
  if (lane == 0) {
    int j = T2 - 1;
    int i = T1 - 1;
    for (; i >= 1; --i) {
      hb[(long long)i * Ts + j] = 1.f;
      hard_idx[b * Tm + i] = j;
      if (j > 0 && ((dirs64[2 * i + (j >> 6)] >> (j & 63)) & 1ull)) {
        --j;
        if (j == 0) {
          for (int r = 1; r < i; ++r) {
            hb[(long long)r * Ts] = 1.f;
            hard_idx[b * Tm + r] = 0;
          }
          break;
        }
      }
    }
    hb[j] = 1.f;
    hard_idx[b * Tm] = j;
  }
  __syncthreads();
  for (int i = lane; i < T1; i += 64) {
    const int j = hard_idx[b * Tm + i];
    if (j >= 0) atomicAdd(&dur[b * Ts + j], 1);
  }
}

// fs2/variance_adaptor.py:207-222: mean over each token's frames counting non-zero frames only
__global__ void avg_variance_kernel(const float* __restrict__ var, const int* __restrict__ cum, float* __restrict__ out,
                                    int B, int Tm, int Ts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Ts) return;
  const int b = i / Ts, j = i % Ts;
  const int start = j > 0 ? cum[i - 1] : 0, end = min(cum[i], Tm);
  float s = 0.f;
  int n = 0;
  for (int t = start; t < end; ++t) {
    const float v = var[(long long)b * Tm + t];
    s += v;
    n += v != 0.f;
  }
  out[i] = n == 0 ? 0.f : s / (float)n;
}

// ---------------------------------------------------------------------------------------------------
// CTC (fs2/attn/attention_loss.py:22-62): classes 0 = blank (logit -1), k+1 = key k (attn_logprob, or -1e15
// beyond key_len); log_softmax over classes; targets 1..L; loss_b = nll_b / max(L, 1); mean over the batch.
// One workgroup per utterance; extended states s = 0..2L (even = blank, odd s = label (s+1)/2).
// (A one-wavefront-per-utterance version with five states per lane was measured at 1.6 ms against 1.4 ms for this
// kernel: the recursion is bound by the instruction count of a single wavefront, not by the barrier.  It was dropped;
// in a training step this kernel runs on the side stream under the decoder.)
__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  return m == NEG_INF ? NEG_INF : m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  return m == NEG_INF ? NEG_INF : m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

__global__ __launch_bounds__(256) void ctc_kernel(const float* __restrict__ logprob, const int* __restrict__ key_lens,
                                                   const int* __restrict__ query_lens, float* __restrict__ alpha_ws,
                                                   float* __restrict__ lse_ws, float* __restrict__ nll_out,
                                                   float* __restrict__ dlogprob, float weight, float blank_logit, int B,
                                                   int Tm, int Ts) {
  extern __shared__ float sf[];  // buf[2][S] | red[4]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int T = query_lens[b], L = key_lens[b];
  T = T > Tm ? Tm : T;
  L = L > Ts ? Ts : L;
  const int S = 2 * L + 1, Smax = 2 * Ts + 1;
  float* buf = sf;
  float* red = sf + 2 * Smax;
  const float* z = logprob + (long long)b * Tm * Ts;
  float* alpha = alpha_ws + (long long)b * Tm * Smax;
  float* lse = lse_ws + (long long)b * Tm;
  float* dz = dlogprob ? dlogprob + (long long)b * Tm * Ts : nullptr;
  if (dz)
    for (long long i = tid; i < (long long)Tm * Ts; i += 256) dz[i] = 0.f;
  if (T <= 0 || L <= 0) {  // empty input or target: handled as an infinite loss -> zeroed (zero_infinity)
    if (tid == 0) nll_out[b] = 0.f;
    return;
  }
  // (1) per-row log-sum-exp over the classes {blank} U {keys < L} (masked keys contribute exp(-1e15) = 0)
  for (int t = wave; t < T; t += 4) {
    float m = blank_logit;
    for (int k = lane; k < L; k += 64) m = fmaxf(m, z[(long long)t * Ts + k]);
    m = fs2_wave_max(m);
    float s = lane == 0 ? expf(blank_logit - m) : 0.f;
    for (int k = lane; k < L; k += 64) s += expf(z[(long long)t * Ts + k] - m);
    s = fs2_wave_sum(s);
    if (lane == 0) lse[t] = m + logf(s);
  }
  __syncthreads();
  auto lp = [&](int t, int s) -> float {  // log-probability of extended state s at time t
    return ((s & 1) ? z[(long long)t * Ts + (s >> 1)] : blank_logit) - lse[t];
  };
  // (2) alpha
  for (int s = tid; s < S; s += 256) {
    const float a = s == 0 ? lp(0, 0) : (s == 1 ? lp(0, 1) : NEG_INF);
    buf[s] = a;
    alpha[s] = a;
  }
  __syncthreads();
  int cur = 0;
  for (int t = 1; t < T; ++t) {
    const float* pr = buf + cur * Smax;
    float* nx = buf + (cur ^ 1) * Smax;
    for (int s = tid; s < S; s += 256) {
      const float a0 = pr[s], a1 = s >= 1 ? pr[s - 1] : NEG_INF, a2 = ((s & 1) && s >= 3) ? pr[s - 2] : NEG_INF;
      const float a = lse3(a0, a1, a2) + lp(t, s);
      nx[s] = a;
      alpha[(long long)t * Smax + s] = a;
    }
    __syncthreads();
    cur ^= 1;
  }
  const float* last = buf + cur * Smax;
  const float ll = lse2(last[S - 1], S >= 2 ? last[S - 2] : NEG_INF);
  const float nll = -ll;
  const bool inf = !(nll < INFINITY);  // zero_infinity=True
  __syncthreads();
  if (tid == 0) nll_out[b] = inf ? 0.f : nll / (float)(L > 1 ? L : 1);
  if (!dz || inf) return;
  const float scale = weight / ((float)(L > 1 ? L : 1) * (float)B);
  // (3) beta (kept in LDS only) + gradient: d/dz[t][k] = softmax - exp(alpha+beta + nll - lp) for the key classes
  for (int s = tid; s < S; s += 256) buf[s] = (s == S - 1 || s == S - 2) ? lp(T - 1, s) : NEG_INF;
  __syncthreads();
  cur = 0;
  for (int t = T - 1; t >= 0; --t) {
    const float* bt = buf + cur * Smax;
    for (int k = tid; k < L; k += 256) {
      const int s = 2 * k + 1;
      const float l = lp(t, s);
      const float ab = alpha[(long long)t * Smax + s] + bt[s];
      dz[(long long)t * Ts + k] = scale * (expf(l) - expf(ab + nll - l));
    }
    if (t == 0) break;
    float* nx = buf + (cur ^ 1) * Smax;
    for (int s = tid; s < S; s += 256) {
      const float b0 = bt[s], b1 = s + 1 < S ? bt[s + 1] : NEG_INF, b2 = ((s & 1) && s + 2 < S) ? bt[s + 2] : NEG_INF;
      nx[s] = lse3(b0, b1, b2) + lp(t - 1, s);
    }
    __syncthreads();
    cur ^= 1;
  }
  (void)red;
}

// loss = weight * mean_b(nll_b)  (nll_b already divided by the target length)
__global__ void ctc_finish_kernel(const float* __restrict__ nll, int B, float weight, float* __restrict__ out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < B; i += 64) s += (double)nll[i];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)B * (double)weight);
}

// fs2/attn/attention_loss.py:65-73: -sum(log(clamp(soft[hard == 1], 1e-12))) / sum(hard); one entry per frame
__global__ __launch_bounds__(256) void bin_loss_kernel(const float* __restrict__ soft, const int* __restrict__ hard_idx,
                                                        int B, int Tm, int Ts, float* __restrict__ partial) {
  __shared__ float red[4][2];
  float s = 0.f, n = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * Tm; i += gridDim.x * blockDim.x) {
    const int j = hard_idx[i];
    if (j >= 0) {
      s += logf(fmaxf(soft[(long long)i * Ts + j], 1e-12f));
      n += 1.f;
    }
  }
  s = fs2_wave_sum(s);
  n = fs2_wave_sum(n);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2 + 0] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    partial[blockIdx.x * 2 + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
}
// out[0] = loss * weight; out_coef[0] = -weight / count  (d loss / d soft[h] = coef / soft[h])
__global__ void bin_finish_kernel(const float* __restrict__ partial, int nparts, float weight, float* __restrict__ out,
                                  float* __restrict__ coef) {
  double s = 0.0, n = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 64) { s += (double)partial[2 * i]; n += (double)partial[2 * i + 1]; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); n += __shfl_xor(n, o, 64); }
  if (threadIdx.x == 0) {
    out[0] = n > 0 ? (float)(-s / n * (double)weight) : 0.f;
    coef[0] = n > 0 ? (float)(-(double)weight / n) : 0.f;
  }
}

// d logits of one row from (a) d attn_logprob (CTC, may be null) and (b) the binarisation loss acting on
// attn_soft[row][h]:  masked softmax backward, then log_softmax backward (the prior is a constant).
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ soft,
                                                                const float* __restrict__ dlogprob,
                                                                const int* __restrict__ hard_idx, const float* __restrict__ bin_coef,
                                                                float* __restrict__ dlogits, int B, int T1, int T2) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * T1) return;
  const float* x = logits + row * T2;
  const float* p = soft + row * T2;
  const int h = hard_idx ? hard_idx[row] : -1;
  float g = 0.f, ph = 0.f;
  if (h >= 0) {
    ph = p[h];
    g = ph > 1e-12f ? bin_coef[0] / ph : 0.f;  // clamp(min=1e-12) has zero slope below the clamp
  }
  float m = NEG_INF;
  for (int j = lane; j < T2; j += 64) m = fmaxf(m, x[j]);
  m = fs2_wave_max(m);
  float s = 0.f, dsum = 0.f;
  for (int j = lane; j < T2; j += 64) {
    s += expf(x[j] - m);
    float d = dlogprob ? dlogprob[row * T2 + j] : 0.f;
    d += p[j] * ((j == h ? g : 0.f) - g * ph);  // softmax backward of a one-hot upstream gradient
    dsum += d;
  }
  s = fs2_wave_sum(s);
  dsum = fs2_wave_sum(dsum);
  for (int j = lane; j < T2; j += 64) {
    float d = dlogprob ? dlogprob[row * T2 + j] : 0.f;
    d += p[j] * ((j == h ? g : 0.f) - g * ph);
    dlogits[row * T2 + j] = d - expf(x[j] - m) / s * dsum;
  }
}

// dq[b,t1,:] = -0.001 * sum_t2 dl[t1,t2] * (q[t1,:] - k[t2,:])      (one wavefront per row)
__global__ __launch_bounds__(256) void dist_bwd_q_kernel(const float* __restrict__ dl, const float* __restrict__ q,
                                                          const float* __restrict__ k, float* __restrict__ dq, int B,
                                                          int T1, int T2, int C) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * T1) return;
  const int b = (int)(row / T1);
  for (int c = lane; c < C; c += 64) {
    const float qv = q[row * C + c];
    float acc = 0.f;
    for (int j = 0; j < T2; ++j) acc = fmaf(dl[row * T2 + j], qv - k[((long long)b * T2 + j) * C + c], acc);
    dq[row * C + c] = -0.001f * acc;
  }
}
// dk[b,t2,:] = +0.001 * sum_t1 dl[t1,t2] * (q[t1,:] - k[t2,:])
__global__ __launch_bounds__(256) void dist_bwd_k_kernel(const float* __restrict__ dl, const float* __restrict__ q,
                                                          const float* __restrict__ k, float* __restrict__ dk, int B,
                                                          int T1, int T2, int C) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * T2) return;
  const int b = (int)(row / T2), j = (int)(row % T2);
  for (int c = lane; c < C; c += 64) {
    const float kv = k[row * C + c];
    float acc = 0.f;
    for (int t = 0; t < T1; ++t)
      acc = fmaf(dl[((long long)b * T1 + t) * T2 + j], q[((long long)b * T1 + t) * C + c] - kv, acc);
    dk[row * C + c] = 0.001f * acc;
  }
}

// ---- round 5: the two kernels above walk their reduction with one dependent global load per step (a wavefront per row,
// the same dl address in every lane): 0.18 + 0.37 ms of the learned-alignment step at the benchmark shape for 0.85 GFLOP.
// The tiled forms keep the other operand of an utterance in LDS and fetch dl coalesced, same arithmetic per term
// (dl * (q - k), summed in ascending order of the reduction index), so the results are the simple kernels' bit for bit.

// dq: a workgroup = 64 query rows of one utterance; k[b] ([T2][C], <= 64 KB) in LDS; a wavefront takes a row at a time,
// lanes along the channels (two channel slots per lane: C <= 128), dl[row][j0 .. j0 + 63] fetched as one coalesced load and
// handed round by readlane.
__global__ __launch_bounds__(256) void dist_bwd_q_tile_kernel(const float* __restrict__ dl, const float* __restrict__ q,
                                                               const float* __restrict__ k, float* __restrict__ dq, int T1,
                                                               int T2, int C) {
  extern __shared__ float ks[];  // [T2][C]
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* kb = k + (long long)b * T2 * C;
  for (int i = threadIdx.x; i < T2 * C; i += 256) ks[i] = kb[i];
  __syncthreads();
  const int c0 = lane, c1 = lane + 64;
  for (int r = wave; r < 64; r += 4) {
    const int t = blockIdx.x * 64 + r;
    if (t >= T1) break;
    const long long row = (long long)b * T1 + t;
    const float q0 = c0 < C ? q[row * C + c0] : 0.f, q1 = c1 < C ? q[row * C + c1] : 0.f;
    float a0 = 0.f, a1 = 0.f;
    for (int j0 = 0; j0 < T2; j0 += 64) {
      const float dv = j0 + lane < T2 ? dl[row * T2 + j0 + lane] : 0.f;
      const int n = min(64, T2 - j0);
#pragma unroll 8
      for (int jj = 0; jj < n; ++jj) {
        const float d = __shfl(dv, jj);
        const float* kr = ks + (j0 + jj) * C;
        if (c0 < C) a0 = fmaf(d, q0 - kr[c0], a0);
        if (c1 < C) a1 = fmaf(d, q1 - kr[c1], a1);
      }
    }
    if (c0 < C) dq[row * C + c0] = -0.001f * a0;
    if (c1 < C) dq[row * C + c1] = -0.001f * a1;
  }
}

// dk: a workgroup = 64 key positions of one utterance x all channels: lanes along the key positions (dl rows are read
// coalesced), the four wavefronts take a quarter of the channels each (CPT per thread, in registers); q[b] passes through
// LDS in tiles of 32 frames (broadcast reads).
template <int CPT>
__global__ __launch_bounds__(256) void dist_bwd_k_tile_kernel(const float* __restrict__ dl, const float* __restrict__ q,
                                                               const float* __restrict__ k, float* __restrict__ dk, int T1,
                                                               int T2, int C) {
  constexpr int TT = 32;
  __shared__ float qs[TT][4 * CPT];
  const int b = blockIdx.y, lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const bool jok = j < T2;
  float kv[CPT], acc[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = cg * CPT + i;
    kv[i] = (jok && c < C) ? k[((long long)b * T2 + j) * C + c] : 0.f;
    acc[i] = 0.f;
  }
  for (int t0 = 0; t0 < T1; t0 += TT) {
    __syncthreads();
    for (int i = threadIdx.x; i < TT * 4 * CPT; i += 256) {
      const int tt = i / (4 * CPT), c = i - tt * (4 * CPT);
      qs[tt][c] = (t0 + tt < T1 && c < C) ? q[((long long)b * T1 + t0 + tt) * C + c] : 0.f;
    }
    float dv[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) dv[tt] = (jok && t0 + tt < T1) ? dl[((long long)b * T1 + t0 + tt) * T2 + j] : 0.f;
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      if (t0 + tt >= T1) break;  // (uniform: keeps the summation exactly T1 terms long)
#pragma unroll
      for (int i = 0; i < CPT; ++i) acc[i] = fmaf(dv[tt], qs[tt][cg * CPT + i] - kv[i], acc[i]);
    }
  }
  if (jok) {
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = cg * CPT + i;
      if (c < C) dk[((long long)b * T2 + j) * C + c] = 0.001f * acc[i];
    }
  }
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int fs2hip_attn_dist(const float* q, const float* k, float* logits, int B, int T1, int T2, int C, void* stream) {
  if (B <= 0 || T1 <= 0 || T2 <= 0 || C <= 0) return FS2HIP_EINVAL;
  const size_t smem = (size_t)(16 * C + 64 * (C + 1)) * sizeof(float);
  if (smem > 64 * 1024) return FS2HIP_EINVAL;
  attn_dist_kernel<<<dim3((T1 + 15) / 16, B), dim3(256), smem, S_>>>(q, k, logits, B, T1, T2, C);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_attn_softmax(const float* logits, const float* prior, const int* key_lens, float* logprob,
                                   float* soft, int B, int T1, int T2, void* stream) {
  if (B <= 0 || T1 <= 0 || T2 <= 0) return FS2HIP_EINVAL;
  const long long rows = (long long)B * T1;
  attn_softmax_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_>>>(logits, prior, key_lens, logprob, soft, B, T1, T2);
  FS2_LAUNCH_CHECK();
  return 0;
}

// dirs_ws: B*Tm*ceil(Ts/32) uint32 scratch (used when the direction bits do not fit in LDS)
extern "C" int fs2hip_mas(const float* in, int is_log, const int* in_lens, const int* out_lens, float* hard,
                          int* hard_idx, int* dur, unsigned* dirs_ws, int B, int Tm, int Ts, void* stream) {
  if (B <= 0 || Tm <= 0 || Ts <= 0) return FS2HIP_EINVAL;
  if (Ts <= 128 && (size_t)Tm * 16 <= 60 * 1024) {  // one wavefront per utterance, everything on chip
    if (hipMemsetAsync(hard, 0, (size_t)B * Tm * Ts * sizeof(float), S_) != hipSuccess) return FS2HIP_EINVAL;
    mas_wave_kernel<<<dim3(B), dim3(64), (size_t)Tm * 16, S_>>>(in, is_log, in_lens, out_lens, hard, hard_idx, dur, B, Tm, Ts);
    FS2_LAUNCH_CHECK();
    return 0;
  }
  const int W = (Ts + 31) / 32;
  size_t smem = (size_t)2 * Ts * sizeof(float);
  const size_t dirs_bytes = (size_t)Tm * W * sizeof(unsigned);
  int in_lds = 0;
  if (smem + dirs_bytes <= 60 * 1024) {
    in_lds = 1;
    smem += dirs_bytes;
  } else if (!dirs_ws) {
    return FS2HIP_EINVAL;
  }
  mas_kernel<<<dim3(B), dim3(256), smem, S_>>>(in, is_log, in_lens, out_lens, hard, hard_idx, dur, dirs_ws, B, Tm, Ts, W, in_lds);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_avg_variance(const float* var, const int* cum, float* out, int B, int Tm, int Ts, void* stream) {
  if (B <= 0 || Tm <= 0 || Ts <= 0) return FS2HIP_EINVAL;
  avg_variance_kernel<<<dim3((B * Ts + 255) / 256), dim3(256), 0, S_>>>(var, cum, out, B, Tm, Ts);
  FS2_LAUNCH_CHECK();
  return 0;
}

// alpha_ws: B*Tm*(2*Ts+1) floats; lse_ws: B*Tm; nll_ws: B.  dlogprob may be NULL (value only).
extern "C" int fs2hip_attn_ctc_loss(const float* logprob, const int* key_lens, const int* query_lens, float* alpha_ws,
                                    float* lse_ws, float* nll_ws, float* dlogprob, float weight, float* loss_out, int B,
                                    int Tm, int Ts, void* stream) {
  if (B <= 0 || Tm <= 0 || Ts <= 0) return FS2HIP_EINVAL;
  const size_t smem = (size_t)(2 * (2 * Ts + 1) + 4) * sizeof(float);
  if (smem > 64 * 1024) return FS2HIP_EINVAL;
  ctc_kernel<<<dim3(B), dim3(256), smem, S_>>>(logprob, key_lens, query_lens, alpha_ws, lse_ws, nll_ws, dlogprob, weight,
                                              -1.0f, B, Tm, Ts);
  FS2_LAUNCH_CHECK();
  ctc_finish_kernel<<<dim3(1), dim3(64), 0, S_>>>(nll_ws, B, weight, loss_out);
  FS2_LAUNCH_CHECK();
  return 0;
}

// partial: 2*256 floats; bin_coef: 1 float (consumed by fs2hip_attn_softmax_bwd)
extern "C" int fs2hip_attn_bin_loss(const float* soft, const int* hard_idx, float* partial, float weight, float* loss_out,
                                    float* bin_coef, int B, int Tm, int Ts, void* stream) {
  if (B <= 0 || Tm <= 0 || Ts <= 0) return FS2HIP_EINVAL;
  int nb = (B * Tm + 255) / 256;
  if (nb > 256) nb = 256;
  bin_loss_kernel<<<dim3(nb), dim3(256), 0, S_>>>(soft, hard_idx, B, Tm, Ts, partial);
  FS2_LAUNCH_CHECK();
  bin_finish_kernel<<<dim3(1), dim3(64), 0, S_>>>(partial, nb, weight, loss_out, bin_coef);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_attn_softmax_bwd(const float* logits, const float* soft, const float* dlogprob, const int* hard_idx,
                                       const float* bin_coef, float* dlogits, int B, int T1, int T2, void* stream) {
  if (B <= 0 || T1 <= 0 || T2 <= 0 || (hard_idx && !bin_coef)) return FS2HIP_EINVAL;
  const long long rows = (long long)B * T1;
  attn_softmax_bwd_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_>>>(logits, soft, dlogprob, hard_idx, bin_coef,
                                                                                 dlogits, B, T1, T2);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_attn_dist_bwd(const float* dlogits, const float* q, const float* k, float* dq, float* dk, int B,
                                    int T1, int T2, int C, void* stream) {
  if (B <= 0 || T1 <= 0 || T2 <= 0 || C <= 0) return FS2HIP_EINVAL;
  const char* tile_env = getenv("FS2_DIST_BWD_TILE");  // "0": the simple one-wavefront-per-row kernels (measurement aid, tests)
  const bool tiled = !(tile_env && atoi(tile_env) == 0) && C <= 128 && B <= 65535;
  if (dq) {
    const size_t smem = (size_t)T2 * C * sizeof(float);
    if (tiled && smem <= 64 * 1024) {
      dist_bwd_q_tile_kernel<<<dim3((T1 + 63) / 64, B), dim3(256), smem, S_>>>(dlogits, q, k, dq, T1, T2, C);
    } else {
      const long long rows = (long long)B * T1;
      dist_bwd_q_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_>>>(dlogits, q, k, dq, B, T1, T2, C);
    }
    FS2_LAUNCH_CHECK();
  }
  if (dk) {
    if (tiled) {
      const dim3 grid((T2 + 63) / 64, B);
      if (C <= 32) dist_bwd_k_tile_kernel<8><<<grid, dim3(256), 0, S_>>>(dlogits, q, k, dk, T1, T2, C);
      else if (C <= 80) dist_bwd_k_tile_kernel<20><<<grid, dim3(256), 0, S_>>>(dlogits, q, k, dk, T1, T2, C);
      else dist_bwd_k_tile_kernel<32><<<grid, dim3(256), 0, S_>>>(dlogits, q, k, dk, T1, T2, C);
    } else {
      const long long rows = (long long)B * T2;
      dist_bwd_k_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_>>>(dlogits, q, k, dk, B, T1, T2, C);
    }
    FS2_LAUNCH_CHECK();
  }
  return 0;
}
