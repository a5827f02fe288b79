// Multi-head self-attention with key-padding mask on operands that ARE bf16 in memory (precision "bf16-mixed" with
// bf16 activation storage): qkv, o, dout and dqkv are bf16 tensors, softmax statistics and accumulation are fp32.
// Head dim 128 (the Conformer's 2 x 128).  Replaces nn.MultiheadAttention's softmax(QK^T/sqrt(d) + mask) V inside
// torchaudio's ConformerLayer under Lightning's bf16-mixed autocast (call sites fs2/model.py:193, :241), forward and
// backward.
//
// Why a family of its own (attention2.hip's PL = 1 mode rounds fp32 tiles in registers): at the bf16 MFMA rate a 32 x 32
// block of scores costs 16 MFMAs = 512 cycles forward, and the fp32 design spent 2 500 on it -- fp32 LDS images (twice the
// LDS bytes, conversion instructions on every operand), single-buffered tiles with two barriers per 32 keys.  Here:
//   * tiles are [64 rows][128] bf16 images (16 KB), double-buffered, one barrier per 64 rows, landed by LDS-DMA;
//   * ONE image serves both ways an operand is read: by rows (k = head dim contiguous: K in K.Q^T, V in V.dO^T, Q / dO in
//     the dK/dV kernel) with ds_read_b128, and TRANSPOSED (k = the tile's rows: V in P.V, K in dS.K, Q / dO in the
//     gradient products) with ds_read_b64_tr_b16.  The 16-byte chunk index of a row is XOR-ed with
//     ((row & 3) << 2) | ((row >> 2) & 3): sixteen consecutive rows of one chunk column cover all bank groups (row reads),
//     and the four rows a transposing read touches per 32-lane half land in four different 64-byte segments;
//   * products are computed transposed as in attention2.hip (the owned row on the MFMA column = the lane): statistics are
//     lane-local, an accumulator block IS the B fragment of the next product after packing -- the reduction order
//     of that product is the accumulator's register order (rows 8g + 4h + i), and the transposing read fetches the other
//     operand's rows in exactly that order;
//   * two workgroups per CU for the forward and the dQ kernel (<= 256 registers): one wavefront's softmax arithmetic runs
//     under the other's MFMAs without any hand interleaving.
// The keep mask of the probabilities' dropout is attention2.hip's, bit for bit (attention_util.h).
#include "attention2.h"
#include "attention_util.h"
#include "gemm_bf16_core.h"

namespace {

constexpr int AB_HD = 128;
constexpr int AB_ROWB = AB_HD * 2;       // bytes per tile row
constexpr int AB_KT = 64;                // rows per tile
constexpr int AB_TILE = AB_KT * AB_ROWB;  // 16 KB

struct AttnBArgs {
  const u16* qkv;   // [B*T][3*D] bf16: q | k | v
  const int* lens;  // [B]
  int B, T, H, HD;
  float scale;
  Fs2Drop drop;
};

__device__ __forceinline__ int ab_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// One [64][128] bf16 tile of a row-major matrix (row stride ld elements) -> LDS by DMA, 256 threads, 4 pieces each.
// Piece (it, tid) is LDS chunk tid & 15 of row 16 it + (tid >> 4) (the DMA's destination is linear in the lane) and
// holds global chunk (tid & 15) ^ swz(row); rows at or beyond `nrows` are zero-filled (out-of-range offset).
struct TileDmaB {
  int voff0, step, r0;
  __device__ __forceinline__ void setup(int ld, int tid) {
    r0 = tid >> 4;
    voff0 = (r0 * ld + (((tid & 15) ^ ab_swz(r0)) << 3)) * 2;
    step = 16 * ld * 2;
  }
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t r, char* tile, int row0, int nrows, int ld, int wave) const {
    const int soff = row0 * ld * 2;
    if (row0 + AB_KT <= nrows) {
#pragma unroll
      for (int it = 0; it < 4; ++it) b_dma16(r, voff0 + it * step, soff, tile + (it * 256 + wave * 64) * 16);
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it)
        b_dma16(r, row0 + 16 * it + r0 < nrows ? voff0 + it * step : B_OOB, soff, tile + (it * 256 + wave * 64) * 16);
    }
  }
};

// per-lane LDS byte addresses (relative to a 32-row block of a tile; the block's offset is an instruction immediate)
struct RowRdB {  // fragment kk of the lane's row: head-dim steps 16 kk + 8 hi .. + 7
  unsigned a[8];
  __device__ __forceinline__ void setup(int l32, int hi, unsigned lds0) {
    const unsigned x = hi ^ ab_swz(l32);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a[kk] = lds0 + l32 * AB_ROWB + ((x ^ (2 * kk)) << 4);
  }
};
struct TrRdB {  // transposing reads: lane 4 q4 + pp of 16-lane group (h, ch) addresses row 4 h + q4 (+ 8 for `hi`) of a
                // 16-row reduction step, head-dim columns 32 db + 16 ch + 4 pp .. + 3
  unsigned lo[4], hi[4];  // (rows + 8: the chunk index's bit 1 flips under the swizzle; the base is 64-byte aligned)
  __device__ __forceinline__ void setup(int lane, unsigned lds0) {
    const int h = lane >> 5, ch = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int cg = 4 * db + 2 * ch + (pp >> 1);
      lo[db] = lds0 + (4 * h + q4) * AB_ROWB + ((cg ^ ((q4 << 2) | h)) << 4) + 8 * (pp & 1);
      hi[db] = lo[db] ^ 32u;
    }
  }
};

__device__ __forceinline__ void ab_pin(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void ab_pin(u32x2& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ f32x16 ab_mfma(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc[key row of the block, own column] = sum over the head dim of tile row . own row: 8 row-fragment reads (four in
// flight) and 8 MFMAs.  OFF = LDS byte offset of the 32-row block.
template <int OFF>
__device__ __forceinline__ f32x16 ab_dot_rows(const RowRdB& rd, const u32x4 (&own)[8]) {
  u32x4 f[8];
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  sfor<4>([&](auto kc) { b_rd128<OFF>(f[decltype(kc)::value], rd.a[decltype(kc)::value]); });
  sfor<8>([&](auto kc) {
    constexpr int kk = decltype(kc)::value;
    if constexpr (kk + 4 < 8) {
      b_rd128<OFF>(f[kk + 4], rd.a[kk + 4]);
      b_lds_wait<4>();
    } else {
      b_lds_wait<7 - kk>();
    }
    ab_pin(f[kk]);
    acc = ab_mfma(f[kk], own[kk], acc);
  });
  return acc;
}

// acc[db][head-dim row, own column] += sum over the block's 32 rows of tile[row][32 db + .] * w[row, own column], where
// w[m] packs the accumulator registers 8 m .. 8 m + 7 of the weights' block (rows 16 m + 4 h + i and 16 m + 8 + 4 h + i):
// 16 transposing reads, 8 MFMAs.  OFF = LDS byte offset of the 32-row block.
template <int OFF>
__device__ __forceinline__ void ab_acc_cols(const TrRdB& tr, const u32x4 (&w)[2], f32x16 (&acc)[4]) {
  u32x2 lo[8], hi[8];  // index 4 m + db
  auto rd = [&](auto gc) {  // group g = 2 m + (db >> 1): the two fragments db = 2 (g & 1), + 1
    constexpr int g = decltype(gc)::value, m = g >> 1;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int db = 2 * (g & 1) + e;
      b_rdtr<OFF + (16 * m) * AB_ROWB>(lo[4 * m + db], tr.lo[db]);
      b_rdtr<OFF + (16 * m + 8) * AB_ROWB>(hi[4 * m + db], tr.hi[db]);
    }
  };
  rd(std::integral_constant<int, 0>{});
  rd(std::integral_constant<int, 1>{});
  sfor<4>([&](auto gc) {
    constexpr int g = decltype(gc)::value, m = g >> 1;
    if constexpr (g + 2 < 4) {
      rd(std::integral_constant<int, g + 2>{});
      b_lds_wait<8>();
    } else {
      b_lds_wait<(3 - g) * 4>();
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int db = 2 * (g & 1) + e;
      ab_pin(lo[4 * m + db]);
      ab_pin(hi[4 * m + db]);
      const u32x4 a = {lo[4 * m + db][0], lo[4 * m + db][1], hi[4 * m + db][0], hi[4 * m + db][1]};
      acc[db] = ab_mfma(a, w[m], acc[db]);
    }
  });
}

// the own row's eight fragments (head-dim steps 16 kk + 8 hi .. + 7) from global memory; zeros outside the matrix
__device__ __forceinline__ void ab_load_own(u32x4 (&f)[8], const u16* row, bool ok, int hi) {
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *reinterpret_cast<const u32x4*>(row + 16 * kk + 8 * hi);
    f[kk] = v;
  }
}

// The transposed accumulators acc[db][head-dim 32 db + 8 g + 4 h + i] of the lane's row, times `mul`, as bf16 -- through
// LDS: a lane holds 4 consecutive head-dim values of ITS row per (db, g), i.e. 8-byte pieces at a
// row stride -- 32 cache lines touched per store instruction, a quarter of each filled (6 us of the forward kernel's
// 61 in the phase probe).  The wavefront's 32 rows pass through a private LDS region ([32][128] bf16, rows padded to
// 272 bytes: 16-byte aligned rows, a two-way conflict on the 8-byte writes) and leave as 16 bytes per lane, 256
// contiguous bytes per row.  `stage` must no longer be read by anybody (the caller's barrier after the last tile).
constexpr int AB_STG_ROW = 272, AB_STG_BYTES = 32 * AB_STG_ROW;
__device__ __forceinline__ void ab_store_rows_staged(char* stage, u16* rows0, long long ld, int nvalid,
                                                     const f32x16 (&acc)[4], float mul, int lane) {
  const int l32 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 v;
      v[0] = pack_bf16x2(acc[db][4 * g] * mul, acc[db][4 * g + 1] * mul);
      v[1] = pack_bf16x2(acc[db][4 * g + 2] * mul, acc[db][4 * g + 3] * mul);
      *reinterpret_cast<u32x2*>(stage + l32 * AB_STG_ROW + (32 * db + 8 * g + 4 * hi) * 2) = v;
    }
  const int rsub = lane >> 4, chunk = lane & 15;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = 4 * it + rsub;
    const u32x4 v = *reinterpret_cast<const u32x4*>(stage + r * AB_STG_ROW + 16 * chunk);
    if (r < nvalid) *reinterpret_cast<u32x4*>(rows0 + (long long)r * ld + 8 * chunk) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------
// forward.  Workgroup = 4 wavefronts = 4 x 32 queries sharing the K / V tiles of the utterance's valid keys.
// ------------------------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attnb_fwd_kernel(AttnBArgs p, u16* __restrict__ o, float* __restrict__ lse) {
  constexpr int HD = AB_HD;
  __shared__ __attribute__((aligned(1024))) char smem[4 * AB_TILE];  // stage s: K at 2 s, V at 2 s + 1 (tiles)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  int qb, b, h, len;
  work_unit(p, (p.T + 127) / 128, qb, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = qb * 128 + wave * 32 + l32;
  const bool active = qb * 128 + wave * 32 < T;  // (wave-uniform) a wavefront without rows still stages tiles and syncs
  const int kend = min(T, len);
  const int nt = (kend + AB_KT - 1) / AB_KT;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  PairHash ph;
  ph.setup(drop);
  const float lg_dscale = DROP ? __builtin_amdgcn_logf(drop.scale) : 0.f;  // log2 of the dropout scale 1/(1-p)
  const u16* base = p.qkv + (long long)b * T * ld;
  const __amdgpu_buffer_rsrc_t rk = b_rsrc(base + D + h * HD), rv = b_rsrc(base + 2 * D + h * HD);
  TileDmaB dma;
  dma.setup(ld, tid);
  if (nt > 0) {
    dma.issue(rk, smem, 0, T, ld, wave);
    dma.issue(rv, smem + AB_TILE, 0, T, ld, wave);
  }
  const float qscale = p.scale * 1.44269504088896f;  // scores in log2 units
  u32x4 qf[8];
  ab_load_own(qf, base + (long long)q * ld + h * HD, q < T, hi);
  f32x16 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
  float m = -INFINITY, l = 0.f;
  const uint32_t rowidx = (uint32_t)(((unsigned long long)(b * p.H + h) * T + q) * (unsigned long long)(T + (T & 1)));
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  RowRdB rd;
  rd.setup(l32, hi, lds0);
  TrRdB tr;
  tr.setup(lane, lds0);

  auto tile = [&](auto stc, int j) {
    constexpr int ST = decltype(stc)::value;
    constexpr int KOFF = ST * 2 * AB_TILE, VOFF = KOFF + AB_TILE;
    b_wait_vmcnt_barrier<0>();  // tile j landed for everybody; everybody has left tile j - 1, whose stage is free
    if (j + 1 < nt) {
      dma.issue(rk, smem + (ST ^ 1) * 2 * AB_TILE, (j + 1) * AB_KT, T, ld, wave);
      dma.issue(rv, smem + (ST ^ 1) * 2 * AB_TILE + AB_TILE, (j + 1) * AB_KT, T, ld, wave);
    }
    if (!active) return;
    const int key0 = j * AB_KT;
    const bool two = key0 + 32 < kend;  // (wave-uniform) the second 32-key block holds valid keys
    f32x16 s[2];
    s[0] = ab_dot_rows<KOFF>(rd, qf);
    if (two) s[1] = ab_dot_rows<KOFF + 32 * AB_ROWB>(rd, qf);
    if (key0 + AB_KT > kend) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (key0 + 32 * kb + 8 * (i >> 2) + 4 * hi + (i & 3) >= kend) s[kb][i] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
    if (two) {
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
    }
    mx = pair_max(mx) * qscale;
    // lazy reference maximum (attention2.hip): rescale only when some row's maximum grew by more than 2^8
    float mnew = m, alpha = 1.f;
    if (__builtin_amdgcn_ballot_w64(mx > m + 8.f)) {
      mnew = fmaxf(m, mx);
      alpha = __builtin_amdgcn_exp2f(m - mnew);
#pragma unroll
      for (int d = 0; d < 4; ++d) oacc[d] *= alpha;
    }
    const float mref = mnew - lg_dscale;  // the weights carry the dropout scale
    float rs = 0.f;
    auto weights = [&](const f32x16& sb, int kb, u32x4 (&w)[2]) {
      const uint32_t pair0 = (rowidx + (uint32_t)(key0 + 32 * kb + 4 * hi)) >> 1;
      float e[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        e[i] = __builtin_amdgcn_exp2f(fmaf(sb[i], qscale, -mref));
        rs += e[i];
      }
      if constexpr (DROP) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {  // pair t: registers 2 t, 2 t + 1 (keys 8 (t >> 1) + 4 hi + 2 (t & 1), + 1)
          const uint32_t hsh = ph.hash(pair0 + (uint32_t)(4 * (t >> 1) + (t & 1)));
          e[2 * t] = ph.template keep<0>(hsh) ? e[2 * t] : 0.f;
          e[2 * t + 1] = ph.template keep<1>(hsh) ? e[2 * t + 1] : 0.f;
        }
      }
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int c = 0; c < 4; ++c) w[mm][c] = pack_bf16x2(e[8 * mm + 2 * c], e[8 * mm + 2 * c + 1]);
    };
    u32x4 w0[2], w1[2];
    weights(s[0], 0, w0);
    if (two) weights(s[1], 1, w1);
    ab_acc_cols<VOFF>(tr, w0, oacc);
    if (two) ab_acc_cols<VOFF + 32 * AB_ROWB>(tr, w1, oacc);
    l = l * alpha + pair_sum(rs);
    m = mnew;
  };
  for (int j = 0; j < nt; j += 2) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) tile(std::integral_constant<int, 1>{}, j + 1);
  }
  __syncthreads();  // nobody reads a tile any more: the stages become the wavefronts' store regions
  if (active) {
    const float dscale = DROP ? drop.scale : 1.f;
    const float lt = l / dscale;  // the row sums carry the dropout scale, the accumulators do too
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    const int row0 = qb * 128 + wave * 32;
    ab_store_rows_staged(smem + wave * AB_STG_BYTES, o + ((long long)b * T + row0) * D + h * HD, D, T - row0, oacc, inv, lane);
    if (hi == 0 && q < T) lse[((long long)b * p.H + h) * T + q] = (m + log2f(lt)) * 0.693147180559945f;
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward.  prep (per row and head: lse' = lse log2e - log2(dropout scale), delta' = sum_d dO O / dropout scale),
// dQ (own = queries, tiles = keys / values), dK/dV (own = keys, tiles = queries / dO).
// p' = 2^(s - lse') is the kept element's weight, dS = keep(p') dP - p' delta'.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attnb_prep_kernel(const u16* __restrict__ dout, const u16* __restrict__ o,
                                                         const float* __restrict__ lse, float2* __restrict__ aux, int B,
                                                         int T, int H, float lg_dscale, float inv_dscale) {
  // one 16-lane group per (row, head): 16 lanes x 8 bf16 = 128 head-dim columns
  const int g = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
  const int row = g / H, hh = g - row * H;
  float sm = 0.f;
  if (row < B * T) {
    const long long off = ((long long)row * H + hh) * AB_HD + 8 * sub;
    const u32x4 a = *reinterpret_cast<const u32x4*>(dout + off);
    const u32x4 c = *reinterpret_cast<const u32x4*>(o + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) sm += bf16_lo(a[i]) * bf16_lo(c[i]) + bf16_hi(a[i]) * bf16_hi(c[i]);
  }
  sm += __shfl_xor(sm, 1, 64);
  sm += __shfl_xor(sm, 2, 64);
  sm += __shfl_xor(sm, 4, 64);
  sm += __shfl_xor(sm, 8, 64);
  if (row < B * T && sub == 0) {
    const int b = row / T, t = row - b * T;
    const long long k = ((long long)b * H + hh) * T + t;
    aux[k] = make_float2(lse[k] * 1.44269504088896f - lg_dscale, sm * inv_dscale);
  }
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attnb_bwd_dq_kernel(AttnBArgs p, const u16* __restrict__ dout,
                                                              const float2* __restrict__ aux, u16* __restrict__ dqkv) {
  constexpr int HD = AB_HD;
  __shared__ __attribute__((aligned(1024))) char smem[4 * AB_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  int qb, b, h, len;
  work_unit(p, (p.T + 127) / 128, qb, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = qb * 128 + wave * 32 + l32;
  const bool active = qb * 128 + wave * 32 < T;
  const int kend = min(T, len);
  const int nt = (kend + AB_KT - 1) / AB_KT;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  PairHash ph;
  ph.setup(drop);
  const u16* base = p.qkv + (long long)b * T * ld;
  const __amdgpu_buffer_rsrc_t rk = b_rsrc(base + D + h * HD), rv = b_rsrc(base + 2 * D + h * HD);
  TileDmaB dma;
  dma.setup(ld, tid);
  if (nt > 0) {
    dma.issue(rk, smem, 0, T, ld, wave);
    dma.issue(rv, smem + AB_TILE, 0, T, ld, wave);
  }
  const float qscale = p.scale * 1.44269504088896f;
  u32x4 qf[8], gf[8];
  ab_load_own(qf, base + (long long)q * ld + h * HD, q < T, hi);
  ab_load_own(gf, dout + ((long long)b * T + q) * D + h * HD, q < T, hi);
  float2 ax = make_float2(0.f, 0.f);
  if (q < T) ax = aux[((long long)b * p.H + h) * T + q];
  f32x16 dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
  const uint32_t rowidx = (uint32_t)(((unsigned long long)(b * p.H + h) * T + q) * (unsigned long long)(T + (T & 1)));
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  RowRdB rd;
  rd.setup(l32, hi, lds0);
  TrRdB tr;
  tr.setup(lane, lds0);

  auto block = [&](auto koffc, auto voffc, int key0) {  // one 32-key block
    constexpr int KOFF = decltype(koffc)::value, VOFF = decltype(voffc)::value;
    f32x16 s = ab_dot_rows<KOFF>(rd, qf);
    f32x16 dp = ab_dot_rows<VOFF>(rd, gf);
    const uint32_t pair0 = (rowidx + (uint32_t)(key0 + 4 * hi)) >> 1;
    float ds[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      uint32_t hsh = 0;
      if constexpr (DROP) hsh = ph.hash(pair0 + (uint32_t)(4 * (t >> 1) + (t & 1)));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = 2 * t + e;
        float pp = __builtin_amdgcn_exp2f(fmaf(s[i], qscale, -ax.x));
        if (key0 + 32 > kend) pp = key0 + 8 * (i >> 2) + 4 * hi + (i & 3) < kend ? pp : 0.f;
        float pd = pp;
        if constexpr (DROP) pd = (e ? ph.template keep<1>(hsh) : ph.template keep<0>(hsh)) ? pp : 0.f;
        ds[i] = fmaf(pd, dp[i], -pp * ax.y);
      }
    }
    u32x4 w[2];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int c = 0; c < 4; ++c) w[mm][c] = pack_bf16x2(ds[8 * mm + 2 * c], ds[8 * mm + 2 * c + 1]);
    ab_acc_cols<KOFF>(tr, w, dq);
  };
  auto tile = [&](auto stc, int j) {
    constexpr int ST = decltype(stc)::value;
    constexpr int KOFF = ST * 2 * AB_TILE, VOFF = KOFF + AB_TILE;
    b_wait_vmcnt_barrier<0>();
    if (j + 1 < nt) {
      dma.issue(rk, smem + (ST ^ 1) * 2 * AB_TILE, (j + 1) * AB_KT, T, ld, wave);
      dma.issue(rv, smem + (ST ^ 1) * 2 * AB_TILE + AB_TILE, (j + 1) * AB_KT, T, ld, wave);
    }
    if (!active) return;
    const int key0 = j * AB_KT;
    block(std::integral_constant<int, KOFF>{}, std::integral_constant<int, VOFF>{}, key0);
    if (key0 + 32 < kend)
      block(std::integral_constant<int, KOFF + 32 * AB_ROWB>{}, std::integral_constant<int, VOFF + 32 * AB_ROWB>{}, key0 + 32);
  };
  for (int j = 0; j < nt; j += 2) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) tile(std::integral_constant<int, 1>{}, j + 1);
  }
  __syncthreads();
  if (active) {
    const int row0 = qb * 128 + wave * 32;
    ab_store_rows_staged(smem + wave * AB_STG_BYTES, dqkv + ((long long)b * T + row0) * ld + h * HD, ld, T - row0, dq, p.scale,
                         lane);
  }
}

// dQ from the spilled dS (attnb_bwd_dkv_kernel<.., SPILL>): dq^T[d][q] = scale * sum_key K[key][d] * dS[q][key].  A wavefront
// owns 32 queries; K tiles (64 keys, 16 KB) go through LDS, two stages; a lane's weights of a 32-key block are the 32
// contiguous bytes the dK/dV kernel laid out for it, requested one block ahead.  No exponentials, no hashes, two products
// fewer than the recomputing kernel.
__global__ __launch_bounds__(256, 2) void attnb_bwd_dq_ds_kernel(AttnBArgs p, const u16* __restrict__ ds, u16* __restrict__ dqkv) {
  constexpr int HD = AB_HD;
  constexpr int SMEM = 2 * AB_TILE > 4 * AB_STG_BYTES ? 2 * AB_TILE : 4 * AB_STG_BYTES;  // K stages; the rows' way out at the end
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  int qb, b, h, len;
  work_unit(p, (p.T + 127) / 128, qb, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D, Tp32 = (T + 31) & ~31;
  const int q = qb * 128 + wave * 32 + l32;
  const bool active = qb * 128 + wave * 32 < T;
  const int kend = min(T, len);
  const int nt = (kend + AB_KT - 1) / AB_KT;
  const u16* base = p.qkv + (long long)b * T * ld;
  const __amdgpu_buffer_rsrc_t rk = b_rsrc(base + D + h * HD);
  TileDmaB dma;
  dma.setup(ld, tid);
  if (nt > 0) dma.issue(rk, smem, 0, T, ld, wave);
  f32x16 dq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  TrRdB tr;
  tr.setup(lane, lds0);
  // this lane's dS row (rows past T: never stored, the loads are skipped)
  const u16* dsrow = ds + (((long long)(b * p.H + h) * T + (q < T ? q : 0)) * Tp32) + 16 * hi;
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  auto load_w = [&](u32x4 (&w)[2], int key0) {  // the sixteen weights of keys key0 .. key0 + 31 (key0 < kend)
    if (q < T && key0 < kend) {
      w[0] = *reinterpret_cast<const u32x4*>(dsrow + key0);
      w[1] = *reinterpret_cast<const u32x4*>(dsrow + key0 + 8);
    } else {
      w[0] = z4;
      w[1] = z4;
    }
  };
  u32x4 wa[2], wb[2];
  load_w(wa, 0);
  auto tile = [&](auto stc, int j) {
    constexpr int ST = decltype(stc)::value;
    constexpr int KOFF = ST * AB_TILE;
    b_wait_vmcnt_barrier<0>();
    if (j + 1 < nt) dma.issue(rk, smem + (ST ^ 1) * AB_TILE, (j + 1) * AB_KT, T, ld, wave);
    if (!active) return;
    const int key0 = j * AB_KT;
    load_w(wb, key0 + 32);
    ab_acc_cols<KOFF>(tr, wa, dq);
    load_w(wa, key0 + 64);
    if (key0 + 32 < kend) ab_acc_cols<KOFF + 32 * AB_ROWB>(tr, wb, dq);
  };
  for (int j = 0; j < nt; j += 2) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) tile(std::integral_constant<int, 1>{}, j + 1);
  }
  __syncthreads();
  if (active) {
    const int row0 = qb * 128 + wave * 32;
    ab_store_rows_staged(smem + wave * AB_STG_BYTES, dqkv + ((long long)b * T + row0) * ld + h * HD, ld, T - row0, dq, p.scale,
                         lane);
  }
}

// dK / dV.  Workgroup = 4 wavefronts = 4 x 32 keys sharing the Q / dO tiles (64 queries) and their {lse', delta'} pairs.
// S and dP come out with the LANE on the key and the registers on the queries, so the per-query statistics are read
// per register (broadcast reads of the staged pairs) and neighbouring keys -- which share a dropout hash -- sit in
// neighbouring lanes: each lane hashes its own elements.  One wavefront per SIMD (the two accumulator sets and the two
// own-row fragment sets are 256 registers by themselves).
// SPILL: dS -- masked to the utterance's keys and rounded to bf16, i.e. exactly the operand the dQ product consumes -- is
// also written out: ds[b][h][q][Tp32] (Tp32 = T rounded up to 32 keys), the 32 keys of a block in the order of the dQ
// kernel's operand registers (position 16 hi' + 4 a + r for key 8 a + 4 hi' + r), so that a lane of
// attnb_bwd_dq_ds_kernel fetches its sixteen weights of a key block as 32 contiguous bytes.  That kernel then needs no S,
// no dP and none of the softmax / dropout vector work the recomputing dQ kernel is bound by.
template <bool DROP, bool SPILL = false>
__global__ __launch_bounds__(256, 1) void attnb_bwd_dkv_kernel(AttnBArgs p, const u16* __restrict__ dout,
                                                               const float2* __restrict__ aux, u16* __restrict__ dqkv,
                                                               u16* __restrict__ ds = nullptr) {
  constexpr int HD = AB_HD;
  // the {lse', delta'} pairs of stage s at 1024 s, then the tiles: stage s has its Q tile at 2 s, its dO tile at 2 s + 1
  constexpr int TILES = 2 * 1024;
  __shared__ __attribute__((aligned(1024))) char smem[TILES + 4 * AB_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  int kb, b, h, len;
  work_unit(p, (p.T + 127) / 128, kb, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D;
  const int key = kb * 128 + wave * 32 + l32;
  const int kend = min(T, len);
  const bool active = kb * 128 + wave * 32 < kend;  // (wave-uniform) some of the wavefront's keys are valid
  const bool any = kb * 128 < kend;                 // (workgroup-uniform)
  const int nt = any ? (T + AB_KT - 1) / AB_KT : 0;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  PairHash ph;
  ph.setup(drop);
  const u16* base = p.qkv + (long long)b * T * ld;
  const __amdgpu_buffer_rsrc_t rq = b_rsrc(base + h * HD), rg = b_rsrc(dout + (long long)b * T * D + h * HD);
  const __amdgpu_buffer_rsrc_t ra = b_rsrc(aux + ((long long)b * p.H + h) * T);
  TileDmaB dmaq, dmag;
  dmaq.setup(ld, tid);
  dmag.setup(D, tid);
  auto stage_in = [&](int st, int q0) {
    dmaq.issue(rq, smem + TILES + st * 2 * AB_TILE, q0, T, ld, wave);
    dmag.issue(rg, smem + TILES + st * 2 * AB_TILE + AB_TILE, q0, T, D, wave);
    if (wave == 0) {  // 64 {lse', delta'} pairs = 128 floats, one per lane and instruction; zeros beyond the last row
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = it * 64 + lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(smem + st * 1024 + it * 256), 4,
                                                 q0 + (idx >> 1) < T ? idx * 4 : B_OOB, q0 * 8, 0, 0);
      }
    }
  };
  if (nt > 0) stage_in(0, 0);
  const float qscale = p.scale * 1.44269504088896f;
  u32x4 kf[8], vf[8];
  ab_load_own(kf, base + (long long)key * ld + D + h * HD, key < T, hi);
  ab_load_own(vf, base + (long long)key * ld + 2 * D + h * HD, key < T, hi);
  f32x16 dk[4], dv[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dk[d][i] = dv[d][i] = 0.f;
  // element index of (query row, key) = row * Tp + key with Tp even: pair index = row * (Tp / 2) + (key >> 1)
  const uint32_t half_tp = (uint32_t)(T + (T & 1)) >> 1;
  const uint32_t unit_row0 = (uint32_t)((unsigned long long)(b * p.H + h) * T);
  const uint32_t pair_lane = (uint32_t)(4 * hi) * half_tp + (uint32_t)(key >> 1);
  const int odd = key & 1;
  // dS slab of this (utterance, head): rows past T fall to the range check (the SGPR offset is part of it on gfx950)
  const int Tp32 = (T + 31) & ~31;
  const __amdgpu_buffer_rsrc_t rds = __builtin_amdgcn_make_buffer_rsrc(
      SPILL ? (void*)(ds + ((long long)(b * p.H + h) * T) * Tp32) : (void*)dqkv, 0, SPILL ? T * Tp32 * 2 : 0, 0x00020000);
  const int ds_pos = 16 * ((l32 >> 2) & 1) + 4 * (l32 >> 3) + (l32 & 3);
  const int ds_voff = (4 * hi * Tp32 + (key & ~31) + ds_pos) * 2;  // (key < Tp32 for every key of an active wavefront)
  const bool key_valid = key < len;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
  RowRdB rd;
  rd.setup(l32, hi, lds0 + TILES);
  TrRdB tr;
  tr.setup(lane, lds0 + TILES);
  const unsigned aux_lane = lds0 + 32 * hi;

  auto block = [&](auto qoffc, auto goffc, auto aoffc, int q0) {  // one 32-query block
    constexpr int QOFF = decltype(qoffc)::value, GOFF = decltype(goffc)::value, AOFF = decltype(aoffc)::value;
    f32x16 s = ab_dot_rows<QOFF>(rd, kf);
    f32x16 dp = ab_dot_rows<GOFF>(rd, vf);
    // (read here, under the draining MFMAs, and consumed right after the wait: a register that an asm read is still
    // filling must not live across code in which the allocator may move it)
    u32x4 axr[8];  // pairs of queries q0 + 8 g + 4 hi + {0,1} (axr[2 g]) and + {2,3} (axr[2 g + 1])
    sfor<8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      b_rd128<AOFF + (i >> 1) * 64 + (i & 1) * 16>(axr[i], aux_lane);
    });
    b_lds_wait<0>();
    float lsev[16], delv[16];  // of the query of accumulator register i = 4 g + r: pair r & 1 of read 2 g + (r >> 1)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ab_pin(axr[j]);
      // (the whole vector, then the element: __builtin_bit_cast of a vector ELEMENT expression reads the vector's first
      // bytes whatever the index -- clang 19 / ROCm 7.2, tools/probes/bitcast_vector_element.hip)
      const f32x4 pr = __builtin_bit_cast(f32x4, axr[j]);
      lsev[2 * j] = pr[0];
      delv[2 * j] = pr[1];
      lsev[2 * j + 1] = pr[2];
      delv[2 * j + 1] = pr[3];
    }
    float pdv[16], dsv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int g = i >> 2, r = i & 3;
      const float lsep = lsev[i], delp = delv[i];
      const float pp = __builtin_amdgcn_exp2f(fmaf(s[i], qscale, -lsep));
      float pd = pp;
      if constexpr (DROP) {
        const uint32_t hsh = ph.hash((unit_row0 + (uint32_t)(q0 + 8 * g + r)) * half_tp + pair_lane);  // (uniform product)
        const uint32_t field = odd ? (hsh >> 16) : (hsh & 0xffffu);
        pd = field >= ph.thresh ? pp : 0.f;
      }
      pdv[i] = pd;
      dsv[i] = fmaf(pd, dp[i], -pp * delp);
      if constexpr (SPILL) {  // dS[q0 + 8 g + 4 hi + r][key], zero at the keys behind the utterance's end
        const unsigned w = pack_bf16x2(key_valid ? dsv[i] : 0.f, 0.f);
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(w & 0xffffu), rds, ds_voff, (q0 + 8 * g + r) * Tp32 * 2, 0);
      }
    }
    u32x4 wp[2], wd[2];
#pragma unroll
    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        wp[mm][c] = pack_bf16x2(pdv[8 * mm + 2 * c], pdv[8 * mm + 2 * c + 1]);
        wd[mm][c] = pack_bf16x2(dsv[8 * mm + 2 * c], dsv[8 * mm + 2 * c + 1]);
      }
    ab_acc_cols<GOFF>(tr, wp, dv);
    ab_acc_cols<QOFF>(tr, wd, dk);
  };
  auto tile = [&](auto stc, int j) {
    constexpr int ST = decltype(stc)::value;
    constexpr int QOFF = ST * 2 * AB_TILE, GOFF = QOFF + AB_TILE, AOFF = ST * 1024;
    // tile j landed for everybody.  SPILL: the 32 dS stores of the previous tile (two blocks: every tile but the last has
    // both) are younger than this tile's DMA and may stay in flight -- for the wavefronts that made any
    if (SPILL && active && j > 0) b_wait_vmcnt_barrier<32>(); else b_wait_vmcnt_barrier<0>();
    if (j + 1 < nt) stage_in(ST ^ 1, (j + 1) * AB_KT);
    if (!active) return;
    const int q0 = j * AB_KT;
    block(std::integral_constant<int, QOFF>{}, std::integral_constant<int, GOFF>{}, std::integral_constant<int, AOFF>{}, q0);
    if (q0 + 32 < T)
      block(std::integral_constant<int, QOFF + 32 * AB_ROWB>{}, std::integral_constant<int, GOFF + 32 * AB_ROWB>{},
            std::integral_constant<int, AOFF + 256>{}, q0 + 32);
  };
  for (int j = 0; j < nt; j += 2) {
    tile(std::integral_constant<int, 0>{}, j);
    if (j + 1 < nt) tile(std::integral_constant<int, 1>{}, j + 1);
  }
  __syncthreads();
  const int row0 = kb * 128 + wave * 32;
  if (row0 < T) {
    u16* krows = dqkv + ((long long)b * T + row0) * ld + D + h * HD;
    const bool valid = key < len;  // a padded key has no weight in any row: its gradients are zero
    char* stage = smem + TILES + wave * AB_STG_BYTES;
    ab_store_rows_staged(stage, krows, ld, T - row0, dk, valid ? p.scale : 0.f, lane);
    ab_store_rows_staged(stage, krows + D, ld, T - row0, dv, valid ? 1.f : 0.f, lane);
  }
}

}  // namespace

bool fs2_attnb_supported(int HD) { return HD == AB_HD; }

int fs2_attnb_fwd(const Attn2Args& a, const void* qkv, void* o, float* lse, hipStream_t s) {
  if (a.HD != AB_HD || a.B <= 0 || a.T <= 0 || a.H <= 0) return FS2HIP_EINVAL;
  if ((double)a.B * a.H * a.T * (a.T + (a.T & 1)) >= 4294967296.0) return FS2HIP_EINVAL;
  if ((long long)a.T * 3 * a.H * a.HD * 2 >= 0x7fffffffLL) return FS2HIP_EINVAL;  // an utterance's rows within one buffer range
  AttnBArgs p{(const u16*)qkv, a.lens, a.B, a.T, a.H, a.HD, a.scale, a.drop};
  dim3 grid(((a.T + 127) / 128) * a.H * a.B);
  if (a.drop.on) attnb_fwd_kernel<true><<<grid, dim3(256), 0, s>>>(p, (u16*)o, lse);
  else attnb_fwd_kernel<false><<<grid, dim3(256), 0, s>>>(p, (u16*)o, lse);
  FS2_LAUNCH_CHECK();
  return 0;
}

int fs2_attnb_bwd(const Attn2Args& a, const void* qkv, const void* o, const void* dout, const float* lse, float* aux,
                  void* dqkv, hipStream_t s, void* ds) {
  if (a.HD != AB_HD || a.B <= 0 || a.T <= 0 || a.H <= 0) return FS2HIP_EINVAL;
  if ((double)a.B * a.H * a.T * (a.T + (a.T & 1)) >= 4294967296.0) return FS2HIP_EINVAL;
  if ((long long)a.T * 3 * a.H * a.HD * 2 >= 0x7fffffffLL) return FS2HIP_EINVAL;
  AttnBArgs p{(const u16*)qkv, a.lens, a.B, a.T, a.H, a.HD, a.scale, a.drop};
  const float dscale = a.drop.on ? a.drop.scale : 1.f;
  const long long groups = (long long)a.B * a.T * a.H;
  attnb_prep_kernel<<<dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, s>>>((const u16*)dout, (const u16*)o, lse,
                                                                               reinterpret_cast<float2*>(aux), a.B, a.T, a.H,
                                                                               log2f(dscale), 1.f / dscale);
  FS2_LAUNCH_CHECK();
  dim3 grid(((a.T + 127) / 128) * a.H * a.B);
  const float2* ax = reinterpret_cast<const float2*>(aux);
  if (ds) {  // dS spilled by the dK/dV kernel, dQ as a product of its own (ds: B * H * T * (T rounded up to 32) bf16)
    if ((long long)a.T * ((a.T + 31) & ~31) * 2 >= 0x7fffffffLL || ((uintptr_t)ds % 16)) return FS2HIP_EINVAL;
    if (a.drop.on) attnb_bwd_dkv_kernel<true, true><<<grid, dim3(256), 0, s>>>(p, (const u16*)dout, ax, (u16*)dqkv, (u16*)ds);
    else attnb_bwd_dkv_kernel<false, true><<<grid, dim3(256), 0, s>>>(p, (const u16*)dout, ax, (u16*)dqkv, (u16*)ds);
    FS2_LAUNCH_CHECK();
    attnb_bwd_dq_ds_kernel<<<grid, dim3(256), 0, s>>>(p, (const u16*)ds, (u16*)dqkv);
    FS2_LAUNCH_CHECK();
    return 0;
  }
  if (a.drop.on) {
    attnb_bwd_dq_kernel<true><<<grid, dim3(256), 0, s>>>(p, (const u16*)dout, ax, (u16*)dqkv);
    FS2_LAUNCH_CHECK();
    attnb_bwd_dkv_kernel<true><<<grid, dim3(256), 0, s>>>(p, (const u16*)dout, ax, (u16*)dqkv);
  } else {
    attnb_bwd_dq_kernel<false><<<grid, dim3(256), 0, s>>>(p, (const u16*)dout, ax, (u16*)dqkv);
    FS2_LAUNCH_CHECK();
    attnb_bwd_dkv_kernel<false><<<grid, dim3(256), 0, s>>>(p, (const u16*)dout, ax, (u16*)dqkv);
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_attention_fwd_b(const void* qkv, const int* lens, void* o, float* lse, int B, int T, int H, int HD,
                                      float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                      void* stream) {
  if (!qkv || !lens || !o || !lse || ((uintptr_t)qkv % 16) || ((uintptr_t)o % 16)) return FS2HIP_EINVAL;
  Attn2Args a{nullptr, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step), 1, nullptr};
  return fs2_attnb_fwd(a, qkv, o, lse, (hipStream_t)stream);
}

extern "C" int fs2hip_attention_bwd_b(const void* qkv, const int* lens, const void* o, const void* dout, const float* lse,
                                      float* aux, void* dqkv, int B, int T, int H, int HD, float drop_p,
                                      unsigned long long drop_seed, const unsigned long long* drop_step, void* stream) {
  if (!qkv || !lens || !o || !dout || !lse || !aux || !dqkv || ((uintptr_t)qkv % 16) || ((uintptr_t)o % 16) ||
      ((uintptr_t)dout % 16) || ((uintptr_t)dqkv % 16) || ((uintptr_t)aux % 16))
    return FS2HIP_EINVAL;
  Attn2Args a{nullptr, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step), 1, nullptr};
  return fs2_attnb_bwd(a, qkv, o, dout, lse, aux, dqkv, (hipStream_t)stream);
}

extern "C" int fs2hip_attention_b_supported(int HD) { return fs2_attnb_supported(HD) ? 1 : 0; }

// fs2hip_attention_bwd_b with dS written out by the dK/dV kernel (bf16, B * H * T * (T rounded up to 32) elements of scratch
// in `ds`) and dQ = scale * dS . K as a product of its own: the recomputing dQ kernel's S, dP and softmax / dropout arithmetic
// are not run a second time.
extern "C" int fs2hip_attention_bwd_b_spill(const void* qkv, const int* lens, const void* o, const void* dout, const float* lse,
                                            float* aux, void* ds, long long ds_elems, void* dqkv, int B, int T, int H, int HD,
                                            float drop_p, unsigned long long drop_seed, const unsigned long long* drop_step,
                                            void* stream) {
  if (!qkv || !lens || !o || !dout || !lse || !aux || !dqkv || !ds || ((uintptr_t)qkv % 16) || ((uintptr_t)o % 16) ||
      ((uintptr_t)dout % 16) || ((uintptr_t)dqkv % 16) || ((uintptr_t)aux % 16) || ((uintptr_t)ds % 16))
    return FS2HIP_EINVAL;
  if (ds_elems < (long long)B * H * T * ((T + 31) & ~31)) return FS2HIP_EINVAL;
  Attn2Args a{nullptr, lens, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(drop_p, drop_seed, drop_step), 1, nullptr};
  return fs2_attnb_bwd(a, qkv, o, dout, lse, aux, dqkv, (hipStream_t)stream, ds);
}
