// Multi-head self-attention with key-padding mask for head dims 64 / 128, fp32 on v_mfma_f32_32x32x2_f32 -- the
// second generation of attention.hip's kernels (which stay for head dims 16 / 32 and for bf16 operands).
// Replaces nn.MultiheadAttention's softmax(QK^T/sqrt(d) + mask) V inside torchaudio's ConformerLayer (call sites
// fs2/model.py:193, :241), forward and backward.
//
// What changed against the first generation, and why (profiles/r01_*: MFMA pipe busy 0.43 there):
//   * a wavefront owns 32 rows on the 32x32x2 MFMA (half the LDS operand bytes and half the instructions per FLOP of
//     16 rows on 16x16x4); a workgroup is TWO wavefronts = 64 rows, four workgroups per CU: all 704 row blocks of the
//     benchmark shape are resident at once (2.75 per CU: one round, against two rounds with the second 37 % full);
//   * key / value tiles (32 keys) go HBM/L2 -> LDS by DMA (`buffer_load ... lds`): no staging registers, no
//     register -> LDS commit phase, and the next K tile is in flight under the softmax and P.V of the current one,
//     the next V tile under the next K.Q^T and softmax;
//   * LDS images are unpadded [32][HD] with the 16-byte chunk index XOR-swizzled by (row & 15) (a DMA cannot pad
//     rows): the K.Q^T operand is one conflict-free ds_read_b128 per four MFMAs, the P.V operand a conflict-free
//     ds_read_b32 whose swizzle folds into four per-lane base addresses + compile-time offsets (no address VALU);
//   * operand reads are inline assembly with counted lgkmcnt waits, as in the GEMM cores (a compiler-visible LDS
//     read behind an LDS-DMA costs a vmcnt(0) that drains the tile in flight).
// As before all products are computed TRANSPOSED (S^T = K Q^T, O^T = V^T P^T): the owned row sits on the MFMA
// column (lane & 31), so softmax statistics are lane-local (one cross-half shuffle) and an accumulator is directly
// the B operand of the next product -- any k-order is a valid fp32 reduction order as long as A and B agree on it.
#include "attention2.h"

#include <utility>

#include "gemm2_core.h"

namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: the body gets the index as an integral_constant (ds_read immediates need constants)
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int KT = 32;  // keys (or queries) per LDS tile

// In-kernel phase timing for the diagnostic build (no stamp executes in the product: the macro is empty there)
#ifdef FS2_ATTN_STAMPS
#define STAMP_DECL long long st_prev = __builtin_amdgcn_s_memtime(), st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i)                                         \
  {                                                      \
    const long long st_now = __builtin_amdgcn_s_memtime(); \
    st_sum[i] += st_now - st_prev;                       \
    st_prev = st_now;                                    \
  }
#define STAMP_FLUSH(ptr, slot)                                                     \
  if ((ptr) && (threadIdx.x & 63) == 0) {                                          \
    for (int i_ = 0; i_ < 8; ++i_) (ptr)[((long long)(slot) * 4 + (threadIdx.x >> 6)) * 8 + i_] += st_sum[i_]; \
  }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_FLUSH(ptr, slot)
#endif

// One [KT][HD] tile of a row-major matrix -> LDS by DMA, NT threads.  Piece p = it * NT + tid lands at LDS byte
// 16 p (the DMA's destination is linear in the lane); it holds row p / CPR, global chunk (p % CPR) ^ (row & 15).
template <int HD, int NT>
struct TileDma {
  static constexpr int CPR = HD / 4, NP = KT * CPR / NT, RPI = NT / CPR;  // pieces per thread, rows per `it`
  int voff[NP];
  int r0;
  __device__ __forceinline__ void setup(int ld, int tid) {
    r0 = tid / CPR;
    const int cp = tid % CPR;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int r = it * RPI + r0;
      voff[it] = (r * ld + 4 * (cp ^ (r & 15))) * 4;
    }
  }
  // piece IT of the tile at rows row0 .. (interleaved one at a time into an MFMA stream: an LDS-DMA instruction
  // costs 60-180 issue cycles, which hide under MFMAs in flight but not between phases)
  template <int IT>
  __device__ __forceinline__ void piece(__amdgpu_buffer_rsrc_t rs, float* tile, int row0, int nrows, int ld, int wave) const {
    if (row0 + KT <= nrows) {  // (wave-uniform) the whole tile is inside the matrix: no vector instruction at all
      blds16(rs, voff[IT], row0 * ld * 4, tile + (IT * NT + wave * 64) * 4);
    } else {
      const int rem = nrows - row0 - r0;
      blds16(rs, IT * RPI < rem ? voff[IT] : FS2_OOB, row0 * ld * 4, tile + (IT * NT + wave * 64) * 4);
    }
  }
  // rows row0 .. row0 + 31 of the matrix behind `rs` (rows >= nrows are written as zeros, nothing is read)
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rs, float* tile, int row0, int nrows, int ld, int wave) const {
    const int soff = row0 * ld * 4, rem = nrows - row0 - r0;
    if (__builtin_amdgcn_ballot_w64(rem < KT) == 0) {  // the whole tile is inside the matrix: no per-piece select
      sfor<NP>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
        blds16(rs, voff[it], soff, tile + (it * NT + wave * 64) * 4);
      });
    } else {
      sfor<NP>([&](auto ic) {
        constexpr int it = decltype(ic)::value;
        blds16(rs, it * RPI < rem ? voff[it] : FS2_OOB, soff, tile + (it * NT + wave * 64) * 4);
      });
    }
  }
};

// ds_read_b128 address of chunk 2j + hi of row `row` (this lane's MFMA row): base + ((2j ^ u) << 4), u = hi ^ (row & 15)
struct RowRd {
  unsigned base, u16;
  template <int HD>
  __device__ __forceinline__ void setup(int row, int hi) {
    base = row * HD * 4;
    u16 = (unsigned)(hi ^ (row & 15)) << 4;
  }
  __device__ __forceinline__ unsigned addr(int j) const { return base + (((unsigned)(2 * j) << 4) ^ u16); }
};

// acc^T[32 tile rows][32 own] = sum_k tile[row][k] * own[k]   (own[j][e] = own row's value k = 8 j + 4 hi + e)
struct NoHook {
  template <class C>
  __device__ __forceinline__ void operator()(C) const {}
};

// `hook(j)` runs after the four MFMAs of reduction group j (work that should hide under them: DMA issue)
template <int HD, class Hook = NoHook>
__device__ __forceinline__ f32x16 dot_rows(const RowRd& rd, unsigned tile_base, const f32x4 (&own)[HD / 8], Hook&& hook = Hook()) {
  constexpr int NJ = HD / 8;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  f32x4 a[3];
  lds_rd128<0>(a[0], tile_base + rd.addr(0));
  lds_rd128<0>(a[1], tile_base + rd.addr(1));
  sfor<NJ>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j + 2 < NJ) {
      lds_rd128<0>(a[(j + 2) % 3], tile_base + rd.addr(j + 2));
      lds_wait<2>();
    } else if constexpr (j + 1 < NJ) {
      lds_wait<1>();
    } else {
      lds_wait<0>();
    }
    pin(a[j % 3]);
    acc = mfma32(a[j % 3][0], own[j][0], acc);
    acc = mfma32(a[j % 3][1], own[j][1], acc);
    acc = mfma32(a[j % 3][2], own[j][2], acc);
    acc = mfma32(a[j % 3][3], own[j][3], acc);
    hook(jc);
  });
  return acc;
}

// Per-lane base addresses for the transposed read tile[key][d] with key = 8a + 4hi + r, d = 32 db + (lane & 31):
// byte = [(8a + r) HD 4 + ((db ^ (a & 1)) << 7)]  (compile time)  +  [4hi HD 4 + ((l4 ^ (4hi + r)) << 4) + 4 (lane & 3)]
struct ColRd {
  unsigned base[4];
  template <int HD>
  __device__ __forceinline__ void setup(int l32, int hi) {
    const int l4 = l32 >> 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) base[r] = 4 * hi * HD * 4 + ((l4 ^ (4 * hi + r)) << 4) + 4 * (l32 & 3);
  }
};
template <int HD, int T_, int DB>
constexpr int col_off() {  // step t = 4a + r of d-block DB
  return ((8 * (T_ >> 2) + (T_ & 3)) * HD * 4) + ((DB ^ ((T_ >> 2) & 1)) << 7);
}

// acc^T[d = 32 db + (lane & 31)][own] += sum over the tile's 32 rows: tile[row][d] * w[row][own].
// Step-major: the NDB MFMAs of row 8a + 4hi + r (step t = 4a + r) go back to back on independent accumulators, and the
// vector work that makes the NEXT step's weight is cut into NDB slices, one behind each MFMA.  The matrix pipe takes
// one MFMA at a time from a wavefront (issue blocks until the previous one is nearly through), so only the vector
// instructions that sit between two MFMAs hide under the first of them -- 64 cycles' worth per gap; a step's whole
// softmax / dS arithmetic in front of its MFMAs leaves the pipe idle (measured: 375 cycles per step instead of 256).
//   first(t)        -> weight of step t in one piece (t = 0 only: the one exposed element of a tile)
//   slice<k>(t)     -> slice k of NDB of the weight of step t; the last slice returns the weight
template <int HD, class Wt, class Hook = NoHook>
__device__ __forceinline__ void acc_cols_pipelined(const ColRd& cr, unsigned tile_base, f32x16 (&acc)[HD / 32], Wt& wt,
                                                   Hook&& hook = Hook()) {
  constexpr int NDB = HD / 32;
  float v[3][NDB];  // operand rows of steps t, t+1, t+2: the LDS round trip is longer than one MFMA
  sfor<NDB>([&](auto dc) {
    constexpr int db = decltype(dc)::value;
    lds_rd32<col_off<HD, 0, db>()>(v[0][db], tile_base + cr.base[0]);
  });
  sfor<NDB>([&](auto dc) {
    constexpr int db = decltype(dc)::value;
    lds_rd32<col_off<HD, 1, db>()>(v[1][db], tile_base + cr.base[1]);
  });
  float w = wt.first();
  sfor<16>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if constexpr (t + 1 < 16) lds_wait<NDB>(); else lds_wait<0>();
#pragma unroll
    for (int db = 0; db < NDB; ++db) pin(v[t % 3][db]);
    float wn = 0.f;
    sfor<NDB>([&](auto dc) {
      constexpr int db = decltype(dc)::value;
      acc[db] = mfma32(v[t % 3][db], w, acc[db]);
#ifndef FS2_PV_NOREAD
      if constexpr (t + 2 < 16) lds_rd32<col_off<HD, t + 2, db>()>(v[(t + 2) % 3][db], tile_base + cr.base[(t + 2) & 3]);
#endif
#ifndef FS2_PV_NOVALU
      if constexpr (t + 1 < 16) wn = wt.template slice<db, NDB, t + 1>(wn);
#else
      wn = w;
#endif
      if constexpr (db == NDB - 1) hook(tc);
      __builtin_amdgcn_sched_barrier(0);
    });
    w = wn;
  });
}

// max / sum over the lane pair (l, l ^ 32) -- the two halves of a row's 32 keys -- with v_permlane32_swap (vector
// pipe; ds_bpermute would be an LDS round trip in the one stretch of the tile that is not under MFMAs)
__device__ __forceinline__ float pair_max(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)p;
}

// fs2_drop_factor (common.h) for the products' weights, bit-identical to it: one hash per PAIR of neighbouring keys
// (steps t = 4a + r with r = 0,1 and r = 2,3 are neighbours; the mask rows are padded to an even length, so a pair
// never straddles a hash).  The element index stays below 2^32 (checked by the launcher).
struct PairHash {
  uint32_t thresh, s_lo, s_hi;
  __device__ __forceinline__ void setup(const Fs2Drop& d) {
    thresh = d.thresh; s_lo = (uint32_t)d.seed; s_hi = (uint32_t)(d.seed >> 32);
  }
  __device__ __forceinline__ uint32_t hash(uint32_t pair_idx) const {
    uint32_t x = pair_idx ^ s_lo;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= s_hi;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
  }
  template <int ODD>
  __device__ __forceinline__ bool keep(uint32_t h) const { return (ODD ? (h >> 16) : (h & 0xffffu)) >= thresh; }
};

// Weights of the forward's P.V product in the log2 domain: w(t) = 2^(s[t] - mref) [* dropout], where the dropout
// scale 1/(1-p) is already inside mref (mref = reference maximum - log2(scale)), so a kept element costs nothing
// extra and the row sum `rs` comes out scaled by it as well.  Vector instructions per element: subtract, v_exp, add
// and, with dropout, half a hash (7 + 4), compare, select.
template <bool DROP>
struct SoftmaxWeights {
  const f32x16& s;
  float mref;
  float rs = 0.f;
  PairHash ph;
  uint32_t pair0;  // (element index of key offset 0 of this tile, for this lane's row and half) >> 1
  uint32_t h = 0;
  __device__ __forceinline__ SoftmaxWeights(const f32x16& s_, float mref_, const PairHash& ph_, uint32_t pair0_)
      : s(s_), mref(mref_), ph(ph_), pair0(pair0_) {}
  template <int T_>
  __device__ __forceinline__ float weight() {
    const float e = __builtin_amdgcn_exp2f(s[T_] - mref);
    rs += e;
    if constexpr (DROP) {
      if constexpr ((T_ & 1) == 0) h = ph.hash(pair0 + (uint32_t)(4 * (T_ >> 2) + ((T_ & 3) >> 1)));
      return ph.template keep<T_ & 1>(h) ? e : 0.f;
    }
    return e;
  }
  __device__ __forceinline__ float first() { return weight<0>(); }
  // slice K of N of the weight of step T_: everything in the last slice but the hash of an even step
  template <int K, int N, int T_>
  __device__ __forceinline__ float slice(float carry) {
    if constexpr (K == N - 1) return weight<T_>();
    return carry;
  }
};

// ------------------------------------------------------------------------------------------------------------
// forward.  Workgroup = 4 wavefronts = 2 row blocks (32 queries each) x 2 key groups: the key tiles of the sequence
// are cut in two halves, each half has its own K / V tiles in LDS (staged by its two wavefronts) and its own online
// softmax state; the two partial (O, m, l) of a row block are merged through LDS at the end.  That makes 2816
// wavefront-sized work units of the benchmark shape instead of 1408 -- 2.75 per SIMD in three rounds of one
// workgroup per CU (92 % full) against 1.375 in two (69 %).
// Per tile and wavefront: 64 MFMAs K.Q^T back to back, the row maximum (the one exposed piece of vector work), then
// P.V step by step -- exp / dropout of element t is issued under the four MFMAs of element t-1.
// Tiles are single-buffered with two barriers per tile: barrier X (start of K.Q^T; everybody is done with the V tile
// -> the next V tile's DMA starts and lands under K.Q^T + softmax) and barrier Y (start of P.V; everybody is done
// with the K tile -> the next K tile's DMA starts and lands under P.V).
// ------------------------------------------------------------------------------------------------------------
template <int HD, bool DROP>
__global__ __launch_bounds__(256, 2) void attn2_fwd_kernel(Attn2Args p, float* __restrict__ o, float* __restrict__ lse) {
  constexpr int NJ = HD / 8, NDB = HD / 32;
  // K tile, V tile of key group 0; then of group 1: 64 KB, two workgroups per CU.  (fp32 MFMAs and fp32 vector
  // instructions share the arithmetic, so the second wavefront per SIMD overlaps no arithmetic -- but it does cover
  // the barrier / DMA / LDS waits: one workgroup per CU measured 15-20 % slower.)
  __shared__ __attribute__((aligned(1024))) float smem[4 * KT * HD];
  // (the wavefront index through readfirstlane: everything derived from it -- key group, tile numbers, DMA scalar
  // offsets -- is then provably wave-uniform; as a plain tid >> 6 the DMA's scalar offset compiled to a waterfall loop)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  const int rb = wave & 1, kg = wave >> 1, gtid = tid & 127;
  // XCD-aware order: the row blocks of one (batch, head) are consecutive work ids on ONE XCD, so that the K / V rows
  // they all stream stay in that XCD's L2 (dealt round-robin they would be fetched by all eight)
  const int nqb = (p.T + 63) / 64;
  const int wid = fs2_xcd_remap(blockIdx.x, gridDim.x);
  const int qb = wid % nqb, bh = wid / nqb;
  const int h = bh % p.H, b = bh / p.H, T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = qb * 64 + rb * 32 + l32;
  const int len = p.lens[b];
  const int kend = min(T, len);
  const int nt = (kend + KT - 1) / KT, n0 = (nt + 1) / 2;  // key tiles: group 0 takes [0, n0), group 1 [n0, nt)
  const int tile0 = kg ? n0 : 0, mine = kg ? nt - n0 : n0;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  PairHash ph;
  ph.setup(drop);
  const float lg_dscale = DROP ? __builtin_amdgcn_logf(drop.scale) : 0.f;  // log2 of the dropout scale 1/(1-p)
  const float* base = p.qkv + (long long)b * T * ld;
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(base + D + h * HD), rv = make_rsrc(base + 2 * D + h * HD);
  TileDma<HD, 128> dma;
  dma.setup(ld, gtid);
  float* Kt = smem + kg * 2 * KT * HD;
  float* Vt = Kt + KT * HD;
  dma.issue(rk, Kt, tile0 * KT, T, ld, rb);
  f32x4 qv[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < T) v = *reinterpret_cast<const f32x4*>(base + (long long)q * ld + h * HD + 8 * j + 4 * hi);
    qv[j] = v * (p.scale * 1.44269504088896f);  // scores in log2 units: the exponentials are bare v_exp_f32
  }
  f32x16 oacc[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[d][i] = 0.f;
  float m = -INFINITY, l = 0.f;
  const uint32_t rowidx = (uint32_t)(((unsigned long long)(b * p.H + h) * T + q) * (unsigned long long)(T + (T & 1)));
  RowRd rd;
  rd.setup<HD>(l32, hi);
  ColRd cr;
  cr.setup<HD>(l32, hi);
  const unsigned ks = lds_addr(Kt), vs = lds_addr(Vt);
  STAMP_DECL;
  for (int j = 0; j < n0; ++j) {
    const int key0 = (tile0 + j) * KT;
    const bool act = j < mine;  // (group 1 may have one tile less: it still keeps the barriers)
    const bool actn = j + 1 < mine;
    STAMP(0)  // prologue / loop overhead
    wait_vmcnt_barrier<0>();    // X: my K pieces landed; after the barrier everybody's did, and the V tile is free
    STAMP(1)  // wait at X
    f32x16 s;
    if (act) {
      s = dot_rows<HD>(rd, ks, qv, [&](auto jc) {  // the V tile's DMA, piece by piece under the MFMAs
        constexpr int it = decltype(jc)::value;
        if constexpr (it < TileDma<HD, 128>::NP) dma.template piece<it>(rv, Vt, key0, T, ld, rb);
      });
    }
    STAMP(2)  // K.Q^T
    float mnew = m, alpha = 1.f;
    if (act) {
      float mx = -INFINITY;
      if (key0 + KT > len) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (key0 + 8 * (i >> 2) + 4 * hi + (i & 3) >= len) s[i] = -INFINITY;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[i]);
      mx = pair_max(mx);
      // Lazy reference maximum: the accumulators are only rescaled when some row's maximum has grown by more than
      // LAZY (then to the exact new maximum); otherwise p = exp(s - m_ref) with s - m_ref <= LAZY, far inside fp32
      // range -- the final O / l does not depend on the reference.  After the first tiles this is (almost) never taken.
      constexpr float LAZY = 8.f;  // (log2 units)
      if (__builtin_amdgcn_ballot_w64(mx > m + LAZY)) {
        mnew = fmaxf(m, mx);
        alpha = __builtin_amdgcn_exp2f(m - mnew);
#pragma unroll
        for (int d = 0; d < NDB; ++d) oacc[d] *= alpha;
      }
    }
    STAMP(3)  // row maximum
    wait_vmcnt_barrier<0>();    // Y: the V tile landed everywhere, and the K tile is free
    STAMP(4)  // wait at Y
    if (act) {
      if (actn) dma.issue(rk, Kt, key0 + KT, T, ld, rb);
      STAMP(6)
      SoftmaxWeights<DROP> wt(s, mnew - lg_dscale, ph, (rowidx + (uint32_t)(key0 + 4 * hi)) >> 1);
      acc_cols_pipelined<HD>(cr, vs, oacc, wt);
      l = l * alpha + pair_sum(wt.rs);  // (sum of the UNdropped probabilities, times the dropout scale)
      m = mnew;
    }
    STAMP(5)  // P.V
  }
  STAMP_FLUSH(p.stamps, blockIdx.x)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // every DMA landed, nobody reads a tile any more
  // merge the two key groups of each row block: group 1 parks (O^T, m, l) in LDS in register order
  static_assert(2 * (NDB * 16 + 2) * 64 <= 4 * KT * HD, "parking area");
  float* park = smem + rb * (NDB * 16 + 2) * 64;
  if (kg == 1) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) park[(d * 16 + i) * 64 + lane] = oacc[d][i];
    park[(NDB * 16) * 64 + lane] = m;
    park[(NDB * 16 + 1) * 64 + lane] = l;
  }
  __syncthreads();
  if (kg == 0 && q < T) {
    const float m1 = park[(NDB * 16) * 64 + lane], l1 = park[(NDB * 16 + 1) * 64 + lane];
    const float mm = fmaxf(m, m1);
    const float a0 = __builtin_amdgcn_exp2f(m - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
    const float dscale = DROP ? drop.scale : 1.f;
    const float lt = (l * a0 + l1 * a1) / dscale;  // the row sums carry the dropout scale, the accumulators do too
    const float inv = 1.f / lt;
    float* orow = o + ((long long)b * T + q) * D + h * HD;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (oacc[d][4 * a + r] * a0 + park[(d * 16 + 4 * a + r) * 64 + lane] * a1) * inv;
        *reinterpret_cast<f32x4*>(orow + 32 * d + 8 * a + 4 * hi) = v;
      }
    if (hi == 0) lse[((long long)b * p.H + h) * T + q] = (mm + log2f(lt)) * 0.693147180559945f;
  }
}

}  // namespace

bool fs2_attn2_supported(int HD, int operand_bf16) { return !operand_bf16 && (HD == 64 || HD == 128); }

int fs2_attn2_fwd(const Attn2Args& a, float* o, float* lse, hipStream_t s) {
  if ((double)a.B * a.H * a.T * a.T >= 4294967296.0) return FS2HIP_EINVAL;  // 32-bit dropout element index
  dim3 grid(((a.T + 63) / 64) * a.H * a.B);
  if (a.HD == 128) {
    if (a.drop.on) attn2_fwd_kernel<128, true><<<grid, dim3(256), 0, s>>>(a, o, lse);
    else attn2_fwd_kernel<128, false><<<grid, dim3(256), 0, s>>>(a, o, lse);
  } else {
    if (a.drop.on) attn2_fwd_kernel<64, true><<<grid, dim3(256), 0, s>>>(a, o, lse);
    else attn2_fwd_kernel<64, false><<<grid, dim3(256), 0, s>>>(a, o, lse);
  }
  FS2_LAUNCH_CHECK();
  return 0;
}
