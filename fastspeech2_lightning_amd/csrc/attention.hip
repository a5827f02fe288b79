// Multi-head self-attention with key-padding mask, flash-style (scores never reach HBM), fp32
// on v_mfma_f32_16x16x4_f32 (or, on request, bf16-rounded operands on v_mfma_f32_16x16x32_bf16).  Replaces nn.MultiheadAttention's softmax(QK^T/sqrt(d) + mask)V
// inside torchaudio's ConformerLayer (call sites fs2/model.py:193, :241) forward and backward.
//
// Layout: qkv is the in_proj output [B*T][3*D] (q | k | v, head h = columns h*HD..), o is [B*T][D].
// A workgroup (4 wavefronts) owns 64 rows of one (batch, head); each wavefront owns 16 of them and
// walks the other sequence in tiles of 64 staged in LDS.  All products are computed TRANSPOSED
// (S^T = K Q^T, O^T = V^T P^T, ...) so that the owned row index sits on the MFMA column
// (lane & 15): softmax statistics and rescales are then lane-local, and an accumulator tile is
// directly the B operand of the next MFMA (any k-order is a valid reduction order for fp32).
#include "attention2.h"
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float xor_max16_32(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum16_32(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// MFMA operands of one lane.  Every product of these kernels pairs "this lane's own row/column" (registers) with
// rows of an LDS tile, four reduction values per lane group g = lane >> 4 at a time:
//   fp32        : four v_mfma_f32_16x16x4_f32, element e of A with element e of B;
//   "bf16-mixed": EIGHT values (two such groups of four) rounded to bf16 (RNE) for ONE v_mfma_f32_16x16x32_bf16,
//                 whose operand layout is: lane group g holds reduction slots 8g..8g+7 of A and of B -- A and B use
//                 the same (group, element) -> slot map, so which eight values share an instruction is free.
//                 Accumulation, softmax statistics and storage stay fp32.
// The conversions are left to the compiler (8-wide convert consumed whole by the MFMA = four v_cvt_pk_bf16_f32):
// a v_cvt_pk_bf16_f32 emitted through inline asm is invisible to the hazard recogniser, which then places the
// MFMA one wait state behind it -- one too few on gfx950 (measured: stale operand registers, tools/scratch/bf16_probe.hip).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 cvt8(float4 a, float4 b) {
  const f32x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return __builtin_convertvector(v, bf16x8);
}
__device__ __forceinline__ f32x4 mfma4(float4 a, float4 b, f32x4 c) {
  c = mfma16(a.x, b.x, c);
  c = mfma16(a.y, b.y, c);
  c = mfma16(a.z, b.z, c);
  return mfma16(a.w, b.w, c);
}
__device__ __forceinline__ f32x4 mfma8(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// this lane's own row: values d = 16 j + 4 g + e (j < HD/16), loaded through f(j) -> float4
template <int HD, bool BF>
struct Own;
template <int HD>
struct Own<HD, false> {
  float4 v[HD / 16];
  template <class F>
  __device__ __forceinline__ void load(F f) {
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) v[j] = f(j);
  }
};
template <int HD>
struct Own<HD, true> {
  static constexpr int NP = (HD / 16 + 1) / 2;
  bf16x8 v[NP];
  template <class F>
  __device__ __forceinline__ void load(F f) {
#pragma unroll
    for (int jj = 0; jj < NP; ++jj)
      v[jj] = cvt8(f(2 * jj), 2 * jj + 1 < HD / 16 ? f(2 * jj + 1) : make_float4(0, 0, 0, 0));
  }
};

// this lane's weights for TWO 16-row blocks of a tile: w0[r] / w1[r] belong to rows 4 g + r of block 2 kp / 2 kp + 1
template <bool BF>
struct W2;
template <>
struct W2<false> {
  float4 a, b;
  __device__ __forceinline__ void set(const f32x4& w0, const f32x4& w1) {
    a = make_float4(w0[0], w0[1], w0[2], w0[3]);
    b = make_float4(w1[0], w1[1], w1[2], w1[3]);
  }
};
template <>
struct W2<true> {
  bf16x8 v;
  __device__ __forceinline__ void set(const f32x4& w0, const f32x4& w1) {
    v = cvt8(make_float4(w0[0], w0[1], w0[2], w0[3]), make_float4(w1[0], w1[1], w1[2], w1[3]));
  }
};

struct AttnP {
  const float* qkv;
  const int* lens;
  int B, T, H;
  float scale;
  Fs2Drop drop;
};

// One 64-row x HD tile (columns from `col`) of the [*, ld] matrix: global -> registers (`fetch_rows`, issued one
// tile ahead so that the loads are in flight under the MFMAs of the current tile) and registers -> LDS
// [64][HD+4] (`commit_rows`, between the two barriers of an iteration).
template <int HD>
struct RowRegs {
  float4 v[(64 * (HD / 4) + 255) / 256];
};
template <int HD>
__device__ __forceinline__ void fetch_rows(RowRegs<HD>& regs, const float* __restrict__ src, int ld, int col, int row0,
                                           int nrows, int tid) {
  constexpr int F4 = HD / 4;
#pragma unroll
  for (int it = 0; it < (64 * F4 + 255) / 256; ++it) {
    int idx = tid + it * 256;
    int r = idx / F4, c4 = idx % F4;
    int row = row0 + r;
    regs.v[it] = (idx < 64 * F4 && row < nrows)
                     ? *reinterpret_cast<const float4*>(src + (long long)row * ld + col + c4 * 4)
                     : make_float4(0, 0, 0, 0);
  }
}
template <int HD>
__device__ __forceinline__ void commit_rows(float* __restrict__ dst, const RowRegs<HD>& regs, int tid) {
  constexpr int LDT = HD + 4, F4 = HD / 4;
#pragma unroll
  for (int it = 0; it < (64 * F4 + 255) / 256; ++it) {
    int idx = tid + it * 256;
    if (idx < 64 * F4) {
      int r = idx / F4, c4 = idx % F4;
      *reinterpret_cast<float4*>(dst + r * LDT + c4 * 4) = regs.v[it];
    }
  }
}

// X^T[rows of tile][own] = sum_d tile[row][d] * own[d]   (tile rows 16*kt + (lane&15))
template <int HD, bool BF>
__device__ __forceinline__ f32x4 dot_tile(const float* __restrict__ tile, const Own<HD, BF>& own, int kt, int c, int g) {
  constexpr int LDT = HD + 4, NJ = HD / 16;
  const float* row = tile + (16 * kt + c) * LDT + 4 * g;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (BF) {
#pragma unroll
    for (int jj = 0; jj < (NJ + 1) / 2; ++jj) {
      const float4 t0 = *reinterpret_cast<const float4*>(row + 32 * jj);
      const float4 t1 = 2 * jj + 1 < NJ ? *reinterpret_cast<const float4*>(row + 32 * jj + 16) : make_float4(0, 0, 0, 0);
      acc = mfma8(cvt8(t0, t1), own.v[jj], acc);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc = mfma4(*reinterpret_cast<const float4*>(row + 16 * j), own.v[j], acc);
  }
  return acc;
}

// The same product for NT consecutive 16-row blocks at once (blocks kt0 .. kt0+NT-1).  Source order = issue order
// the fp32 path wants: the operand rows of reduction slice j+1 are fetched from LDS before the MFMAs of slice j, and
// the NT accumulators are interleaved element by element, so that neither an LDS round trip (the compiler's own
// schedule read two rows, waited, issued eight MFMAs, read the next two ...) nor the back-to-back dependence of one
// accumulator chain is exposed.  Every accumulator still adds its products in the order (j, element): bit-identical
// to dot_tile.
template <int HD, bool BF, int NT>
__device__ __forceinline__ void dot_tiles(const float* __restrict__ tile, const Own<HD, BF>& own, int kt0, int c, int g,
                                          f32x4 (&out)[NT]) {
  constexpr int LDT = HD + 4, NJ = HD / 16;
  if constexpr (BF) {
#pragma unroll
    for (int u = 0; u < NT; ++u) out[u] = dot_tile<HD, true>(tile, own, kt0 + u, c, g);
  } else {
    const float* row = tile + (16 * kt0 + c) * LDT + 4 * g;
    float4 cur[NT], nxt[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      out[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      cur[u] = *reinterpret_cast<const float4*>(row + u * 16 * LDT);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) {
#pragma unroll
        for (int u = 0; u < NT; ++u) nxt[u] = *reinterpret_cast<const float4*>(row + u * 16 * LDT + 16 * (j + 1));
      }
#pragma unroll
      for (int u = 0; u < NT; ++u) out[u] = mfma16(cur[u].x, own.v[j].x, out[u]);
#pragma unroll
      for (int u = 0; u < NT; ++u) out[u] = mfma16(cur[u].y, own.v[j].y, out[u]);
#pragma unroll
      for (int u = 0; u < NT; ++u) out[u] = mfma16(cur[u].z, own.v[j].z, out[u]);
#pragma unroll
      for (int u = 0; u < NT; ++u) out[u] = mfma16(cur[u].w, own.v[j].w, out[u]);
#pragma unroll
      for (int u = 0; u < NT; ++u) cur[u] = nxt[u];
    }
  }
}

// acc^T[d = 16*dt + c][own] += sum over the 32 tile rows of block pair kp: tile[row][d] * w[row]
template <int HD, bool BF>
__device__ __forceinline__ f32x4 acc_pair(const float* __restrict__ tile, int kp, const W2<BF>& w, int dt, int c, int g, f32x4 acc) {
  constexpr int LDT = HD + 4;
  const float* q0 = tile + (32 * kp + 4 * g) * LDT + 16 * dt + c;
  const float* q1 = q0 + 16 * LDT;
  const float4 t0 = make_float4(q0[0], q0[LDT], q0[2 * LDT], q0[3 * LDT]);
  const float4 t1 = make_float4(q1[0], q1[LDT], q1[2 * LDT], q1[3 * LDT]);
  if constexpr (BF) {
    return mfma8(cvt8(t0, t1), w.v, acc);
  } else {
    return mfma4(t1, w.b, mfma4(t0, w.a, acc));
  }
}

template <int HD, bool BF>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnP p, float* __restrict__ o, float* __restrict__ lse) {
  constexpr int LDT = HD + 4, NJ = HD / 16;
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[64 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z, T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = blockIdx.x * 64 + wave * 16 + c;
  const int len = p.lens[b];
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const float* base = p.qkv + (long long)b * T * ld;
  Own<HD, BF> qr;
  qr.load([&](int j) {
    float4 v = q < T ? *reinterpret_cast<const float4*>(base + (long long)q * ld + h * HD + 16 * j + 4 * g)
                     : make_float4(0, 0, 0, 0);
    return make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
  });
  f32x4 oacc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) oacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const int kend = min(T, len);
  const unsigned long long rowidx = ((unsigned long long)(b * p.H + h) * T + q) * (unsigned long long)(T + (T & 1));
  RowRegs<HD> kreg, vreg;
  fetch_rows<HD>(kreg, base, ld, D + h * HD, 0, T, tid);
  fetch_rows<HD>(vreg, base, ld, 2 * D + h * HD, 0, T, tid);
  for (int key0 = 0; key0 < kend; key0 += 64) {
    __syncthreads();
    commit_rows<HD>(Ks, kreg, tid);
    commit_rows<HD>(Vs, vreg, tid);
    __syncthreads();
    if (key0 + 64 < kend) {  // next tile's loads fly under this tile's MFMAs
      fetch_rows<HD>(kreg, base, ld, D + h * HD, key0 + 64, T, tid);
      fetch_rows<HD>(vreg, base, ld, 2 * D + h * HD, key0 + 64, T, tid);
    }
    f32x4 s[4];
    float mx = -INFINITY;
    dot_tiles<HD, BF, 4>(Ks, qr, 0, c, g, s);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = key0 + 16 * kt + 4 * g + r;
        if (key >= len) s[kt][r] = -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    }
    mx = xor_max16_32(mx);
    const float mnew = fmaxf(m, mx);
    const float alpha = __expf(m - mnew);
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = __expf(s[kt][r] - mnew);  // v_exp_f32: ~1 ulp; precise expf costs 5x the instructions
        rs += pv;
        s[kt][r] = pv * fs2_drop_factor(drop, rowidx + (unsigned long long)(key0 + 16 * kt + 4 * g + r));
      }
    rs = xor_sum16_32(rs);
    l = l * alpha + rs;
    m = mnew;
    W2<BF> pw[2];
    pw[0].set(s[0], s[1]);
    pw[1].set(s[2], s[3]);
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt) {
      oacc[dt] *= alpha;
      oacc[dt] = acc_pair<HD, BF>(Vs, 1, pw[1], dt, c, g, acc_pair<HD, BF>(Vs, 0, pw[0], dt, c, g, oacc[dt]));
    }
  }
  if (q < T) {
    const float inv = 1.f / l;
    float* orow = o + ((long long)b * T + q) * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt)
      *reinterpret_cast<float4*>(orow + 16 * dt + 4 * g) =
          make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
    if (g == 0) lse[((long long)b * p.H + h) * T + q] = m + logf(l);
  }
}

// delta[b][h][t] = sum_d dO * O over the head's columns; one wavefront per row
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dout, const float* __restrict__ o,
                                                          float* __restrict__ delta, int B, int T, int H, int HD) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B * T) return;
  const int D = H * HD, f4 = D / 4, per_head = HD / 4;
  const int b = row / T, t = row % T;
  for (int i0 = 0; i0 < f4; i0 += 64) {
    int i = i0 + lane;
    float s = 0.f;
    if (i < f4) {
      float4 a = reinterpret_cast<const float4*>(dout + (long long)row * D)[i];
      float4 c = reinterpret_cast<const float4*>(o + (long long)row * D)[i];
      s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
    for (int w = 1; w < per_head && w < 64; w <<= 1) s += __shfl_xor(s, w, 64);
    if (i < f4 && (i % per_head) == 0) delta[((long long)b * H + i / per_head) * T + t] = s;
  }
}

// dQ: same walk as the forward (own = queries, tiles = keys)
template <int HD, bool BF>
__global__ __launch_bounds__(256, BF ? 2 : 1) void attn_bwd_dq_kernel(AttnP p, const float* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           float* __restrict__ dqkv) {
  constexpr int LDT = HD + 4, NJ = HD / 16;
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[64 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z, T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = blockIdx.x * 64 + wave * 16 + c;
  const int len = p.lens[b];
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const float* base = p.qkv + (long long)b * T * ld;
  Own<HD, BF> qr, dor;
  qr.load([&](int j) {
    float4 v = q < T ? *reinterpret_cast<const float4*>(base + (long long)q * ld + h * HD + 16 * j + 4 * g)
                     : make_float4(0, 0, 0, 0);
    return make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
  });
  dor.load([&](int j) {
    return q < T ? *reinterpret_cast<const float4*>(dout + ((long long)b * T + q) * D + h * HD + 16 * j + 4 * g)
                 : make_float4(0, 0, 0, 0);
  });
  const float lse_q = q < T ? lse[((long long)b * p.H + h) * T + q] : INFINITY;
  const float delta_q = q < T ? delta[((long long)b * p.H + h) * T + q] : 0.f;
  f32x4 dq[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) dq[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kend = min(T, len);
  const unsigned long long rowidx = ((unsigned long long)(b * p.H + h) * T + q) * (unsigned long long)(T + (T & 1));
  RowRegs<HD> kreg, vreg;
  fetch_rows<HD>(kreg, base, ld, D + h * HD, 0, T, tid);
  fetch_rows<HD>(vreg, base, ld, 2 * D + h * HD, 0, T, tid);
  for (int key0 = 0; key0 < kend; key0 += 64) {
    __syncthreads();
    commit_rows<HD>(Ks, kreg, tid);
    commit_rows<HD>(Vs, vreg, tid);
    __syncthreads();
    if (key0 + 64 < kend) {
      fetch_rows<HD>(kreg, base, ld, D + h * HD, key0 + 64, T, tid);
      fetch_rows<HD>(vreg, base, ld, 2 * D + h * HD, key0 + 64, T, tid);
    }
    f32x4 ds[4], s4[4], dp4[4];
    dot_tiles<HD, BF, 4>(Ks, qr, 0, c, g, s4);
    dot_tiles<HD, BF, 4>(Vs, dor, 0, c, g, dp4);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = key0 + 16 * kt + 4 * g + r;
        float pv = key < len ? __expf(s4[kt][r] - lse_q) : 0.f;
        float f = fs2_drop_factor(drop, rowidx + (unsigned long long)key);
        ds[kt][r] = pv * (dp4[kt][r] * f - delta_q);
      }
    }
    W2<BF> dsw[2];
    dsw[0].set(ds[0], ds[1]);
    dsw[1].set(ds[2], ds[3]);
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt)
      dq[dt] = acc_pair<HD, BF>(Ks, 1, dsw[1], dt, c, g, acc_pair<HD, BF>(Ks, 0, dsw[0], dt, c, g, dq[dt]));
  }
  if (q < T) {
    float* row = dqkv + ((long long)b * T + q) * ld + h * HD;
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt)
      *reinterpret_cast<float4*>(row + 16 * dt + 4 * g) =
          make_float4(dq[dt][0] * p.scale, dq[dt][1] * p.scale, dq[dt][2] * p.scale, dq[dt][3] * p.scale);
  }
}

// dK, dV: own = keys (registers), tiles = queries (Q and dO staged in LDS)
template <int HD, bool BF>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnP p, const float* __restrict__ dout,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            float* __restrict__ dqkv) {
  constexpr int LDT = HD + 4, NJ = HD / 16;
  __shared__ __attribute__((aligned(16))) float Qs[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Os[64 * LDT];
  __shared__ float lse_s[64], delta_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z, T = p.T, D = p.H * HD, ld = 3 * D;
  const int key = blockIdx.x * 64 + wave * 16 + c;
  const int len = p.lens[b];
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const float* base = p.qkv + (long long)b * T * ld;
  float* krow = dqkv + ((long long)b * T + key) * ld + D + h * HD;
  float* vrow = krow + D;
  if ((int)(blockIdx.x * 64) >= len) {  // every key of this workgroup is padding: gradients are zero
    if (key < T) {
#pragma unroll
      for (int dt = 0; dt < NJ; ++dt) {
        *reinterpret_cast<float4*>(krow + 16 * dt + 4 * g) = make_float4(0, 0, 0, 0);
        *reinterpret_cast<float4*>(vrow + 16 * dt + 4 * g) = make_float4(0, 0, 0, 0);
      }
    }
    return;
  }
  Own<HD, BF> kr, vr;
  kr.load([&](int j) {
    float4 v = key < T ? *reinterpret_cast<const float4*>(base + (long long)key * ld + D + h * HD + 16 * j + 4 * g)
                       : make_float4(0, 0, 0, 0);
    return make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
  });
  vr.load([&](int j) {
    return key < T ? *reinterpret_cast<const float4*>(base + (long long)key * ld + 2 * D + h * HD + 16 * j + 4 * g)
                   : make_float4(0, 0, 0, 0);
  });
  f32x4 dk[NJ], dv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    dk[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool key_ok = key < len;
  const unsigned long long headidx = (unsigned long long)(b * p.H + h) * T;
  RowRegs<HD> qreg, oreg;
  fetch_rows<HD>(qreg, base, ld, h * HD, 0, T, tid);
  fetch_rows<HD>(oreg, dout + (long long)b * T * D, D, h * HD, 0, T, tid);
  for (int q0 = 0; q0 < T; q0 += 64) {
    __syncthreads();
    commit_rows<HD>(Qs, qreg, tid);
    commit_rows<HD>(Os, oreg, tid);
    if (tid < 64) {
      int qq = q0 + tid;
      lse_s[tid] = qq < T ? lse[((long long)b * p.H + h) * T + qq] : INFINITY;
      delta_s[tid] = qq < T ? delta[((long long)b * p.H + h) * T + qq] : 0.f;
    }
    __syncthreads();
    if (q0 + 64 < T) {
      fetch_rows<HD>(qreg, base, ld, h * HD, q0 + 64, T, tid);
      fetch_rows<HD>(oreg, dout + (long long)b * T * D, D, h * HD, q0 + 64, T, tid);
    }
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {  // two 16-query blocks at a time (one bf16 MFMA covers both)
      f32x4 pd[2], ds[2], s2[2], dp2[2];
      dot_tiles<HD, BF, 2>(Qs, kr, 2 * qp, c, g, s2);
      dot_tiles<HD, BF, 2>(Os, vr, 2 * qp, c, g, dp2);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * qp + u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int ql = 16 * qt + 4 * g + r;
          float pv = key_ok ? __expf(s2[u][r] - lse_s[ql]) : 0.f;
          float f = fs2_drop_factor(drop, (headidx + (unsigned long long)(q0 + ql)) * (unsigned long long)(T + (T & 1)) + (unsigned long long)key);
          pd[u][r] = pv * f;
          ds[u][r] = pv * (dp2[u][r] * f - delta_s[ql]);
        }
      }
      W2<BF> pdw, dsw;
      pdw.set(pd[0], pd[1]);
      dsw.set(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < NJ; ++dt) {
        dv[dt] = acc_pair<HD, BF>(Os, qp, pdw, dt, c, g, dv[dt]);
        dk[dt] = acc_pair<HD, BF>(Qs, qp, dsw, dt, c, g, dk[dt]);
      }
    }
  }
  if (key < T) {
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt) {
      *reinterpret_cast<float4*>(krow + 16 * dt + 4 * g) =
          make_float4(dk[dt][0] * p.scale, dk[dt][1] * p.scale, dk[dt][2] * p.scale, dk[dt][3] * p.scale);
      *reinterpret_cast<float4*>(vrow + 16 * dt + 4 * g) = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
    }
  }
}

bool attn_args_ok(const void* qkv, int B, int T, int H, int HD) {
  if (B <= 0 || T <= 0 || H <= 0) return false;
  if (HD != 16 && HD != 32 && HD != 64 && HD != 128) return false;
  if ((uintptr_t)qkv % 16) return false;
  return true;
}

}  // namespace

#define ATTN_DISPATCH_HD(HD_, CALL)                     \
  switch (HD_) {                                        \
    case 16: { constexpr int HDc = 16; CALL; } break;   \
    case 32: { constexpr int HDc = 32; CALL; } break;   \
    case 64: { constexpr int HDc = 64; CALL; } break;   \
    default: { constexpr int HDc = 128; CALL; } break;  \
  }
#define ATTN_DISPATCH(HD_, BF_, CALL)                              \
  if (BF_) { constexpr bool BFc = true; ATTN_DISPATCH_HD(HD_, CALL) } \
  else { constexpr bool BFc = false; ATTN_DISPATCH_HD(HD_, CALL) }

extern "C" int fs2hip_attention_fwd(const float* qkv, const int* lens, float* o, float* lse, int B, int T, int H,
                                    int HD, float drop_p, unsigned long long drop_seed,
                                    const unsigned long long* drop_step, int operand_bf16, void* stream) {
  if (!attn_args_ok(qkv, B, T, H, HD) || ((uintptr_t)o % 16)) return FS2HIP_EINVAL;
  static const bool old_only = getenv("FS2_ATTN_GEN1") != nullptr;  // measurement aid: first-generation kernels everywhere
  if (!old_only && fs2_attn2_supported(HD, operand_bf16)) {
    Attn2Args a2{qkv, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step),
                 operand_bf16 == 2 ? 3 : operand_bf16, nullptr};
    return fs2_attn2_fwd(a2, o, lse, (hipStream_t)stream);
  }
  AttnP p{qkv, lens, B, T, H, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step)};
  dim3 grid((T + 63) / 64, H, B);
  ATTN_DISPATCH(HD, operand_bf16 == 1, (attn_fwd_kernel<HDc, BFc><<<grid, dim3(256), 0, (hipStream_t)stream>>>(p, o, lse)));
  FS2_LAUNCH_CHECK();
  return 0;
}

// The forward pass that also writes the masked scores out (operand_bf16 0: fp32 MFMA path, 2: "32-split"; head dims 64 / 128) for
// fs2hip_attention_bwd_spill_s: `scores` holds at least B * H * T * (T rounded up to 32) floats.
extern "C" int fs2hip_attention_fwd_s(const float* qkv, const int* lens, float* o, float* lse, float* scores,
                                      long long score_floats, int B, int T, int H, int HD, float drop_p,
                                      unsigned long long drop_seed, const unsigned long long* drop_step, int operand_bf16,
                                      void* stream) {
  if (operand_bf16 != 0 && operand_bf16 != 2) return FS2HIP_EINVAL;  // exact fp32 or three exact bf16 planes
  if (!attn_args_ok(qkv, B, T, H, HD) || ((uintptr_t)o % 16) || ((uintptr_t)scores % 16) || !scores) return FS2HIP_EINVAL;
  static const bool old_only = getenv("FS2_ATTN_GEN1") != nullptr;
  if (old_only || (HD != 64 && HD != 128)) return FS2HIP_EINVAL;
  if (score_floats < (long long)B * H * T * ((T + 31) & ~31)) return FS2HIP_EINVAL;
  Attn2Args a2{qkv, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step),
               operand_bf16 == 2 ? 3 : 0, nullptr};
  return fs2_attn2_fwd(a2, o, lse, (hipStream_t)stream, scores);
}

extern "C" int fs2hip_attention_bwd_spill_s(const float* qkv, const int* lens, const float* o, const float* dout,
                                            const float* lse, const float* scores, float* aux, float* ds, long long ds_floats,
                                            float* dqkv, int B, int T, int H, int HD, float drop_p,
                                            unsigned long long drop_seed, const unsigned long long* drop_step, void* stream);

extern "C" int fs2hip_attention_bwd_spill_supported(int HD) {
  static const bool old_only = getenv("FS2_ATTN_GEN1") != nullptr;
  return (!old_only && (HD == 64 || HD == 128)) ? 1 : 0;
}

extern "C" int fs2hip_attention_bwd_spill(const float* qkv, const int* lens, const float* o, const float* dout,
                                          const float* lse, float* aux, float* ds, long long ds_floats, float* dqkv, int B,
                                          int T, int H, int HD, float drop_p, unsigned long long drop_seed,
                                          const unsigned long long* drop_step, void* stream) {
  if (!attn_args_ok(qkv, B, T, H, HD) || ((uintptr_t)o % 16) || ((uintptr_t)dout % 16) || ((uintptr_t)dqkv % 16) ||
      ((uintptr_t)ds % 16) || !fs2hip_attention_bwd_spill_supported(HD))
    return FS2HIP_EINVAL;
  Attn2Args a2{qkv, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step), 0, nullptr};
  const long long need = fs2_attn2_bwd_spill_elems(a2);
  if (need == 0 || ds_floats < need) return FS2HIP_EINVAL;
  return fs2_attn2_bwd_spill(a2, o, dout, lse, aux, ds, dqkv, (hipStream_t)stream);
}

extern "C" int fs2hip_attention_bwd(const float* qkv, const int* lens, const float* o, const float* dout,
                                    const float* lse, float* delta, float* dqkv, int B, int T, int H, int HD,
                                    float drop_p, unsigned long long drop_seed,
                                    const unsigned long long* drop_step, int operand_bf16, void* stream) {
  if (!attn_args_ok(qkv, B, T, H, HD) || ((uintptr_t)o % 16) || ((uintptr_t)dout % 16) || ((uintptr_t)dqkv % 16))
    return FS2HIP_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static const bool old_only = getenv("FS2_ATTN_GEN1") != nullptr;
  if (!old_only && fs2_attn2_supported(HD, operand_bf16)) {
    Attn2Args a2{qkv, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step),
                 operand_bf16 == 2 ? 3 : operand_bf16, nullptr};
    return fs2_attn2_bwd(a2, o, dout, lse, delta, dqkv, s);
  }
  attn_delta_kernel<<<dim3((B * T + 3) / 4), dim3(256), 0, s>>>(dout, o, delta, B, T, H, HD);
  FS2_LAUNCH_CHECK();
  AttnP p{qkv, lens, B, T, H, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step)};
  dim3 grid((T + 63) / 64, H, B);
  ATTN_DISPATCH(HD, operand_bf16 == 1, (attn_bwd_dq_kernel<HDc, BFc><<<grid, dim3(256), 0, s>>>(p, dout, lse, delta, dqkv)));
  FS2_LAUNCH_CHECK();
  ATTN_DISPATCH(HD, operand_bf16 == 1, (attn_bwd_dkv_kernel<HDc, BFc><<<grid, dim3(256), 0, s>>>(p, dout, lse, delta, dqkv)));
  FS2_LAUNCH_CHECK();
  return 0;
}

// fs2hip_attention_bwd_spill with the scores of fs2hip_attention_fwd_s: the dK/dV kernel reads them instead of recomputing
// K.Q^T (3 products instead of 4, the forward pass's probabilities to the bit)
extern "C" int fs2hip_attention_bwd_spill_s(const float* qkv, const int* lens, const float* o, const float* dout,
                                            const float* lse, const float* scores, float* aux, float* ds, long long ds_floats,
                                            float* dqkv, int B, int T, int H, int HD, float drop_p,
                                            unsigned long long drop_seed, const unsigned long long* drop_step, void* stream) {
  if (!attn_args_ok(qkv, B, T, H, HD) || ((uintptr_t)o % 16) || ((uintptr_t)dout % 16) || ((uintptr_t)dqkv % 16) ||
      ((uintptr_t)ds % 16) || ((uintptr_t)scores % 16) || !scores || !fs2hip_attention_bwd_spill_supported(HD))
    return FS2HIP_EINVAL;
  Attn2Args a2{qkv, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step), 0, nullptr};
  const long long need = fs2_attn2_bwd_spill_elems(a2);
  if (need == 0 || ds_floats < need) return FS2HIP_EINVAL;
  return fs2_attn2_bwd_spill(a2, o, dout, lse, aux, ds, dqkv, (hipStream_t)stream, scores);
}
