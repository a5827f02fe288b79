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
#include "attention_util.h"

namespace {

template <int OFF>
__device__ __forceinline__ void lds_rd64(float2& v, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pin(float2& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// The same MFMA with its accumulator pinned to AGPRs.  The backward kernels hold 64-128 accumulator registers that
// no vector instruction ever touches next to ~200 registers of operands: left to itself the allocator parks and
// unparks accumulators around every phase (v_accvgpr_read/write are vector instructions -- 200-350 of them per
// tile, each one ADDED to the MFMA time).  The "a" constraint keeps them where they belong.  The two wait states
// in front cover a vector write of an operand register in the instruction before (the hazard recogniser does not
// see through inline asm).
__device__ __forceinline__ void mfma32_agpr(f32x16& acc, float a, float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// ... and with the accumulator pinned to VGPRs (results that vector instructions consume: S, dP); `first` starts
// the chain from zero (inline constant: no 16-register clear)
__device__ __forceinline__ void mfma32_vgpr_first(f32x16& acc, float a, float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_vgpr(f32x16& acc, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// ... and with the B operand (the wavefront's own row, which nothing but MFMAs ever reads) in an AGPR: the backward
// kernels keep 128 such values; in VGPRs they pushed the allocator past 256 and into parking them around every use
__device__ __forceinline__ void mfma32_vgpr_first_ob(f32x16& acc, float a, float b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma32_vgpr_ob(f32x16& acc, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
// gives a value its home in an AGPR (defined by asm into the "a" class: later "a" uses need no copy)
__device__ __forceinline__ float to_agpr(float v) {
  float r;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ f32x4 to_agpr(f32x4 v) {
  f32x4 r;
  r[0] = to_agpr(v[0]); r[1] = to_agpr(v[1]); r[2] = to_agpr(v[2]); r[3] = to_agpr(v[3]);
  return r;
}
// before the first compiler-visible read of such an accumulator: the last MFMA's 16 passes have to retire
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

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
template <int HD, bool OWN_AGPR = false, class Hook = NoHook>
__device__ __forceinline__ f32x16 dot_rows(const RowRd& rd, unsigned tile_base, const f32x4 (&own)[HD / 8], Hook&& hook = Hook()) {
  constexpr int NJ = HD / 8;
  f32x16 acc;
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
    if constexpr (OWN_AGPR) {
      if constexpr (j == 0) mfma32_vgpr_first_ob(acc, a[0][0], own[0][0]);
      else mfma32_vgpr_ob(acc, a[j % 3][0], own[j][0]);
      mfma32_vgpr_ob(acc, a[j % 3][1], own[j][1]);
      mfma32_vgpr_ob(acc, a[j % 3][2], own[j][2]);
      mfma32_vgpr_ob(acc, a[j % 3][3], own[j][3]);
    } else {
      if constexpr (j == 0) mfma32_vgpr_first(acc, a[0][0], own[0][0]);
      else mfma32_vgpr(acc, a[j % 3][0], own[j][0]);
      mfma32_vgpr(acc, a[j % 3][1], own[j][1]);
      mfma32_vgpr(acc, a[j % 3][2], own[j][2]);
      mfma32_vgpr(acc, a[j % 3][3], own[j][3]);
    }
    hook(jc);
  });
  mfma_drain();  // (the result is consumed by vector instructions right away)
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

// ------------------------------------------------------------------------------------------------------------
// Operand modes on the bf16 matrix pipe (template parameter PL = number of bf16 planes per operand value):
//   PL = 1  "bf16-mixed": values rounded to bf16 (RNE), one v_mfma_f32_32x32x16_bf16 per 16 reduction steps
//   PL = 3  "32-split"  : values cut exactly into three bf16 planes, six partial products per product (gemm2_core.h)
// Both take operands from the same fp32 LDS images as the fp32 kernels; the MFMA's C/D layout is the 32x32 one, so
// everything downstream of an accumulator (softmax, masks, stores) is shared.  Unlike fp32 MFMAs these run BESIDE
// the vector ALUs, so the softmax / dropout / cut arithmetic overlaps them instead of adding to them.
// ------------------------------------------------------------------------------------------------------------
template <int PL>
struct Pl {
  bf16x8 p[PL];
};
template <int PL>
__device__ __forceinline__ Pl<PL> make_planes(const float (&x)[8]) {
  Pl<PL> r;
  if constexpr (PL == 1) {
    const f32x8 v = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
    r.p[0] = __builtin_convertvector(v, bf16x8);
  } else {
    u32x4 q0, q1, q2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned a0, a1, a2;
      split_pair(x[2 * e], x[2 * e + 1], a0, a1, a2);
      q0[e] = a0; q1[e] = a1; q2[e] = a2;
    }
    r.p[0] = __builtin_bit_cast(bf16x8, q0);
    r.p[1] = __builtin_bit_cast(bf16x8, q1);
    r.p[2] = __builtin_bit_cast(bf16x8, q2);
  }
  return r;
}
template <int PL>
__device__ __forceinline__ Pl<PL> make_planes(const f32x4& lo, const f32x4& hi) {
  const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return make_planes<PL>(x);
}
template <int PL>
__device__ __forceinline__ f32x16 mfma_planes(const Pl<PL>& a, const Pl<PL>& b, f32x16 c) {
  if constexpr (PL == 1) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], c, 0, 0, 0);
  } else {  // smallest partial products first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], c, 0, 0, 0);
  }
}

// this lane's own row for the bf16 products: step s covers k = 16 s + 8 hi + 0..7; load(j4) returns the four values
// k = 4 j4 .. 4 j4 + 3 of the row
template <int HD, int PL, class Ld>
__device__ __forceinline__ void own_planes(Pl<PL> (&own)[HD / 16], int hi, Ld&& load) {
#pragma unroll
  for (int s_ = 0; s_ < HD / 16; ++s_) own[s_] = make_planes<PL>(load(4 * s_ + 2 * hi), load(4 * s_ + 2 * hi + 1));
}

// acc^T[32 tile rows][32 own] = sum_k tile[row][k] * own[k] on the bf16 pipe: per 16 reduction steps two
// ds_read_b128 (chunks 4s + 2hi, + 1 of the lane's tile row), the cut / rounding, and PL (PL + 1) / 2 ... MFMAs
template <int HD, int PL, class Hook = NoHook>
__device__ __forceinline__ f32x16 dot_rows_pl(int row, int hi, unsigned tile_base, const Pl<PL> (&own)[HD / 16], Hook&& hook = Hook()) {
  constexpr int NS = HD / 16;
  const unsigned base = tile_base + row * HD * 4, sw = (unsigned)(row & 15) << 4;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  f32x4 a[3][2];
  auto rd = [&](int s_, f32x4 (&dst)[2]) {
    lds_rd128<0>(dst[0], base + ((((unsigned)(4 * s_ + 2 * hi)) << 4) ^ sw));
    lds_rd128<0>(dst[1], base + ((((unsigned)(4 * s_ + 2 * hi + 1)) << 4) ^ sw));
  };
  rd(0, a[0]);
  if constexpr (NS > 1) rd(1, a[1]);
  sfor<NS>([&](auto sc) {
    constexpr int s_ = decltype(sc)::value;
    if constexpr (s_ + 2 < NS) {
      rd(s_ + 2, a[(s_ + 2) % 3]);
      lds_wait<4>();
    } else if constexpr (s_ + 1 < NS) {
      lds_wait<2>();
    } else {
      lds_wait<0>();
    }
    pin(a[s_ % 3][0]);
    pin(a[s_ % 3][1]);
    acc = mfma_planes<PL>(make_planes<PL>(a[s_ % 3][0], a[s_ % 3][1]), own[s_], acc);
    hook(sc);
  });
  return acc;
}

// acc[k]^T[d][own] += sum over the tile's 32 rows tile_k[row][d] * w_k[row][own] on the bf16 pipe, for NW (tile,
// weight) sets at once (forward: P.V; dQ: K.dS; dK/dV: dO.P and Q.dS).  Two 16-row steps; the eight weights of a step
// are S-layout registers 8 s .. 8 s + 7 (rows 16 s + 8 (e >> 2) + 4 hi + (e & 3)), made by weight(t, w[NW]) right
// before; a tile operand is eight ds_read_b32 per d-block and step, read one MFMA group ahead.
template <int HD, int PL, int NW, class Wf>
__device__ __forceinline__ void acc_cols_pl(const ColRd& cr, const unsigned (&tile)[NW], f32x16 (&acc)[NW][HD / 32], Wf&& weight) {
  constexpr int NDB = HD / 32, PER = NW * NDB, N = 2 * PER;
  float v[2][8];
  auto rd = [&](auto nc, float (&dst)[8]) {
    constexpr int n = decltype(nc)::value, s_ = n / PER, k = (n % PER) / NDB, db = n % NDB;
    sfor<8>([&](auto ec) {
      constexpr int t = 8 * s_ + decltype(ec)::value;
      lds_rd32<col_off<HD, t, db>()>(dst[decltype(ec)::value], tile[k] + cr.base[t & 3]);
    });
  };
  rd(std::integral_constant<int, 0>{}, v[0]);
  sfor<2>([&](auto sc) {
    constexpr int s_ = decltype(sc)::value;
    float w[NW][8];
    sfor<8>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      float one[NW];
      weight(std::integral_constant<int, 8 * s_ + e>{}, one);
#pragma unroll
      for (int k = 0; k < NW; ++k) w[k][e] = one[k];
    });
    Pl<PL> wp[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) wp[k] = make_planes<PL>(w[k]);
    sfor<PER>([&](auto uc) {
      constexpr int u = decltype(uc)::value, n = s_ * PER + u, k = u / NDB, db = u % NDB;
      if constexpr (n + 1 < N) {
        rd(std::integral_constant<int, n + 1>{}, v[(n + 1) & 1]);
        lds_wait<8>();
      } else {
        lds_wait<0>();
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) pin(v[n & 1][e]);
      acc[k][db] = mfma_planes<PL>(make_planes<PL>(v[n & 1]), wp[k], acc[k][db]);
    });
  });
}

__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) float*)p;
}

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
// SS (fp32 MFMA path and the three-plane "32-split" path, training): the masked scores -- log2 units, scale folded in -- are also written out as
// s_out[b][h][q][key] (rows of Tp32 = T rounded up to 32 floats) for the dK/dV kernel, which then needs neither the S product
// nor its own K rows (attn2_bwd_dkv_kernel<.., SIN>).  Four 16-byte stores per lane and key tile, younger than the V tile's
// DMA: the barrier behind the row maximum waits with vmcnt(4).
template <int HD, bool DROP, int PL, bool SS = false>
__global__ __launch_bounds__(256, 2) void attn2_fwd_kernel(Attn2Args p, float* __restrict__ o, float* __restrict__ lse,
                                                           float* __restrict__ s_out = nullptr) {
  static_assert(!SS || PL == 0 || PL == 3, "scores are written out by the fp32-accurate paths only");
  constexpr int NJ = HD / 8, NDB = HD / 32;
  // K tile, V tile of key group 0; then of group 1: 64 KB, two workgroups per CU.  (fp32 MFMAs and fp32 vector
  // instructions share the arithmetic, so the second wavefront per SIMD overlaps no arithmetic -- but it does cover
  // the barrier / DMA / LDS waits: one workgroup per CU measured 15-20 % slower.)
  __shared__ __attribute__((aligned(1024))) float smem[4 * KT * HD];
  // (the wavefront index through readfirstlane: everything derived from it -- key group, tile numbers, DMA scalar
  // offsets -- is then provably wave-uniform; as a plain tid >> 6 the DMA's scalar offset compiled to a waterfall loop)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  const int rb = wave & 1, kg = wave >> 1, gtid = tid & 127;
  int qb, b, h, len;
  work_unit(p, (p.T + 63) / 64, qb, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = qb * 64 + rb * 32 + l32;
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
  const float qscale = p.scale * 1.44269504088896f;  // scores in log2 units: the exponentials are bare v_exp_f32
  f32x4 qv[PL == 0 ? NJ : 1];
  Pl<(PL == 0 ? 1 : PL)> qp[HD / 16];
  if constexpr (PL == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < T) v = *reinterpret_cast<const f32x4*>(base + (long long)q * ld + h * HD + 8 * j + 4 * hi);
      qv[j] = v * qscale;
    }
  } else {
    own_planes<HD, PL>(qp, hi, [&](int j4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < T) v = *reinterpret_cast<const f32x4*>(base + (long long)q * ld + h * HD + 4 * j4);
      return v * qscale;
    });
  }
  f32x16 oacc_[1][NDB];
  auto& oacc = oacc_[0];
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
  const int Tp32 = (T + 31) & ~31;
  const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(
      SS ? (void*)(s_out + ((long long)(b * p.H + h) * T) * Tp32) : (void*)o, 0, SS ? T * Tp32 * 4 : 0, 0x00020000);
  const int ss_voff = q < T ? (q * Tp32 + 4 * hi) * 4 : FS2_OOB;
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
      auto vdma = [&](auto jc) {  // the V tile's DMA, piece by piece under the MFMAs
        constexpr int it = decltype(jc)::value;
        if constexpr (it < TileDma<HD, 128>::NP) dma.template piece<it>(rv, Vt, key0, T, ld, rb);
      };
      if constexpr (PL == 0) s = dot_rows<HD>(rd, ks, qv, vdma);
      else s = dot_rows_pl<HD, PL>(l32, hi, ks, qp, vdma);
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
      if constexpr (SS) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const f32x4 v = {s[4 * a], s[4 * a + 1], s[4 * a + 2], s[4 * a + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rss, ss_voff, (key0 + 8 * a) * 4, 0);
        }
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
    // Y: the V tile landed everywhere, and the K tile is free (SS: the four score stores are younger than its DMA)
    if (SS && act) wait_vmcnt_barrier<4>(); else wait_vmcnt_barrier<0>();
    STAMP(4)  // wait at Y
    if (act) {
      if (actn) dma.issue(rk, Kt, key0 + KT, T, ld, rb);
      STAMP(6)
      SoftmaxWeights<DROP> wt(s, mnew - lg_dscale, ph, (rowidx + (uint32_t)(key0 + 4 * hi)) >> 1);
      if constexpr (PL == 0) acc_cols_pipelined<HD>(cr, vs, oacc, wt);
      else acc_cols_pl<HD, PL, 1>(cr, {vs}, oacc_, [&](auto tc, float (&w1)[1]) { w1[0] = wt.template weight<decltype(tc)::value>(); });
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

// ------------------------------------------------------------------------------------------------------------
// backward.  Three launches: prep (per row and head: lse and delta = sum_d dO*O, pre-digested), dQ (own = queries,
// tiles = keys), dK/dV (own = keys, tiles = queries).  Both gradient kernels use the forward's decomposition --
// 4 wavefronts = 2 row blocks x 2 groups over the other sequence, partial results merged through LDS -- but with
// double-buffered tiles and ONE barrier per tile: a tile is read from the start (S, dP) to the end (the step-major
// gradient products) of its iteration, so the next tile's DMA goes into the other buffer, piece by piece under the
// MFMAs of S and dP.  128 KB of LDS: one workgroup per CU; the accumulators live in AGPRs (512 registers per lane).
// Vector work per element (it is ADDED to the MFMA time on this chip): subtract, v_exp, [hash], compare, select,
// multiply, fma -- lse arrives as lse*log2(e) - log2(dropout scale) and delta as delta / (dropout scale), so that
// p' = 2^(s - lse') is already the kept-element weight and dS = keep(p')*dP - p'*delta'.
// ------------------------------------------------------------------------------------------------------------

// aux[b][h][t] = {lse * log2e - log2(dscale), delta / dscale}, delta = sum over the head's columns of dO * O
__global__ __launch_bounds__(256) void attn2_prep_kernel(const float* __restrict__ dout, const float* __restrict__ o,
                                                         const float* __restrict__ lse, float2* __restrict__ aux, int B,
                                                         int T, int H, int HD, float lg_dscale, float inv_dscale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  // The two pairs of slack behind the last row: the dK/dV kernel fetches pairs two at a time, and with T odd the last
  // piece of the last (utterance, head) reads one pair past the end.  Its row is padding (dO = 0 there), but 0 * NaN is
  // NaN: the slack must hold numbers, not whatever the allocator left (seen with poisoned memory: NaN gradients).
  if (blockIdx.x == 0 && threadIdx.x < 2) aux[(long long)B * H * T + threadIdx.x] = make_float2(0.f, 0.f);
  if (row >= B * T) return;
  const int D = H * HD, f4 = D / 4, per_head = HD / 4;
  const int b = row / T, t = row % T;
  for (int i0 = 0; i0 < f4; i0 += 64) {
    const int i = i0 + lane;
    float sm = 0.f;
    if (i < f4) {
      const float4 a = reinterpret_cast<const float4*>(dout + (long long)row * D)[i];
      const float4 c = reinterpret_cast<const float4*>(o + (long long)row * D)[i];
      sm = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    }
    for (int w = 1; w < per_head && w < 64; w <<= 1) sm += __shfl_xor(sm, w, 64);
    if (i < f4 && (i % per_head) == 0) {
      const long long k = ((long long)b * H + i / per_head) * T + t;
      aux[k] = make_float2(lse[k] * 1.44269504088896f - lg_dscale, sm * inv_dscale);
    }
  }
}

// Stream of (one LDS operand read, one MFMA) pairs with the reads running LOOK pairs ahead: before MFMA i the
// wavefront waits until at most LOOK - 1 reads are outstanding (read i has landed), after it issues read i + LOOK.
// rd(ic, dst) issues read number i into dst; mf(ic, v) performs MFMA i with operand v.
template <int N, int LOOK, class Rd, class Mf>
__device__ __forceinline__ void read_mfma_stream(Rd&& rd, Mf&& mf) {
  float v[2 * LOOK];
  sfor<LOOK>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < N) rd(ic, v[i % (2 * LOOK)]);
  });
  sfor<N>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    constexpr int left = N - 1 - i;  // reads issued after read i so far
    lds_wait<(left < LOOK - 1 ? left : LOOK - 1)>();
    pin(v[i % (2 * LOOK)]);
    mf(ic, v[i % (2 * LOOK)]);
    if constexpr (i + LOOK < N) rd(std::integral_constant<int, i + LOOK>{}, v[(i + LOOK) % (2 * LOOK)]);
  });
}

// dQ: own = 32 queries per wavefront (Q' = Q * scale * log2e and dO in registers), tiles = keys (K and V)
template <int HD, bool DROP, int PL>
__global__ __launch_bounds__(256, 1) void attn2_bwd_dq_kernel(Attn2Args p, const float* __restrict__ dout,
                                                              const float2* __restrict__ aux, float* __restrict__ dqkv) {
  constexpr int NJ = HD / 8, NDB = HD / 32, TILE = KT * HD, NP = TileDma<HD, 128>::NP;
  __shared__ __attribute__((aligned(1024))) float smem[8 * TILE];  // [key group][stage][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  const int rb = wave & 1, kg = wave >> 1, gtid = tid & 127;
  int qb, b, h, len;
  work_unit(p, (p.T + 63) / 64, qb, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = qb * 64 + rb * 32 + l32;
  const int kend = min(T, len);
  const int nt = (kend + KT - 1) / KT, n0 = (nt + 1) / 2;
  const int tile0 = kg ? n0 : 0, mine = kg ? nt - n0 : n0;
  PairHash ph;
  ph.setup(fs2_resolve_drop(p.drop));
  const float* base = p.qkv + (long long)b * T * ld;
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(base + D + h * HD), rv = make_rsrc(base + 2 * D + h * HD);
  TileDma<HD, 128> dma;
  dma.setup(ld, gtid);
  float* gbuf = smem + kg * 4 * TILE;
  if (mine > 0) {
    dma.issue(rk, gbuf, tile0 * KT, T, ld, rb);
    dma.issue(rv, gbuf + TILE, tile0 * KT, T, ld, rb);
  }
  const float qscale = p.scale * 1.44269504088896f;
  f32x4 qv[PL == 0 ? NJ : 1], dov[PL == 0 ? NJ : 1];
  Pl<(PL == 0 ? 1 : PL)> qp[HD / 16], dop[HD / 16];
  if constexpr (PL == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
      if (q < T) {
        v = *reinterpret_cast<const f32x4*>(base + (long long)q * ld + h * HD + 8 * j + 4 * hi);
        w = *reinterpret_cast<const f32x4*>(dout + ((long long)b * T + q) * D + h * HD + 8 * j + 4 * hi);
      }
      qv[j] = to_agpr(v * qscale);
      dov[j] = to_agpr(w);
    }
  } else {
    own_planes<HD, PL>(qp, hi, [&](int j4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < T) v = *reinterpret_cast<const f32x4*>(base + (long long)q * ld + h * HD + 4 * j4);
      return v * qscale;
    });
    own_planes<HD, PL>(dop, hi, [&](int j4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < T) v = *reinterpret_cast<const f32x4*>(dout + ((long long)b * T + q) * D + h * HD + 4 * j4);
      return v;
    });
  }
  float2 ax = make_float2(INFINITY, 0.f);  // (rows past T: p = 2^(s - inf) = 0)
  if (q < T) ax = aux[((long long)b * p.H + h) * T + q];
  const float lse2 = ax.x, deltap = ax.y;
  f32x16 dq_[1][NDB];
  auto& dq = dq_[0];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
  const uint32_t rowidx = (uint32_t)(((unsigned long long)(b * p.H + h) * T + q) * (unsigned long long)(T + (T & 1)));
  RowRd rd;
  rd.setup<HD>(l32, hi);
  ColRd cr;
  cr.setup<HD>(l32, hi);
  const unsigned g0 = lds_addr(gbuf);
  STAMP_DECL;
  for (int j = 0; j < n0; ++j) {
    const int key0 = (tile0 + j) * KT;
    const bool act = j < mine, actn = j + 1 < mine;
    const int st = j & 1;
    STAMP(0)
    wait_vmcnt_barrier<0>();  // tile j landed for everybody, and everybody is done with tile j - 1 (the other stage)
    STAMP(1)
    if (!act) continue;
    const unsigned kb = g0 + st * 2 * TILE * 4, vb = kb + TILE * 4;
    float* nxt = gbuf + (st ^ 1) * 2 * TILE;
    auto kdma = [&](auto jc) {
      constexpr int it = decltype(jc)::value;
      if constexpr (it < NP) {
        if (actn) dma.template piece<it>(rk, nxt, key0 + KT, T, ld, rb);
      }
    };
    auto vdma = [&](auto jc) {
      constexpr int it = decltype(jc)::value;
      if constexpr (it < NP) {
        if (actn) dma.template piece<it>(rv, nxt + TILE, key0 + KT, T, ld, rb);
      }
    };
    f32x16 s, dp;
    if constexpr (PL == 0) s = dot_rows<HD, true>(rd, kb, qv, kdma);
    else s = dot_rows_pl<HD, PL>(l32, hi, kb, qp, kdma);
    STAMP(2)
    if constexpr (PL == 0) dp = dot_rows<HD, true>(rd, vb, dov, vdma);
    else dp = dot_rows_pl<HD, PL>(l32, hi, vb, dop, vdma);
    STAMP(3)
    if (key0 + KT > len) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (key0 + 8 * (i >> 2) + 4 * hi + (i & 3) >= len) s[i] = -INFINITY;
    }
    // dq^T[d][q] += sum_key K[key][d] * dS[key][q], step-major; dS of step t is made right before its MFMAs
    const uint32_t pair0 = (rowidx + (uint32_t)(key0 + 4 * hi)) >> 1;
    uint32_t hsh = 0;
    float w = 0.f;
    auto weight = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const float pp = __builtin_amdgcn_exp2f(s[t] - lse2);
      float pd = pp;
      if constexpr (DROP) {
        if constexpr ((t & 1) == 0) hsh = ph.hash(pair0 + (uint32_t)(4 * (t >> 2) + ((t & 3) >> 1)));
        pd = ph.template keep<t & 1>(hsh) ? pp : 0.f;
      }
      return fmaf(-pp, deltap, pd * dp[t]);
    };
    if constexpr (PL == 0) {
      w = weight(std::integral_constant<int, 0>{});
      STAMP(4)
      read_mfma_stream<16 * NDB, 8>(
          [&](auto ic, float& dst) {
            constexpr int i = decltype(ic)::value, t = i / NDB, db = i % NDB;
            lds_rd32<col_off<HD, t, db>()>(dst, kb + cr.base[t & 3]);
          },
          [&](auto ic, float v) {
            constexpr int i = decltype(ic)::value, t = i / NDB, db = i % NDB;
            mfma32_agpr(dq[db], v, w);
            if constexpr (db == NDB - 1 && t + 1 < 16) w = weight(std::integral_constant<int, t + 1>{});
          });
    } else {
      STAMP(4)
      acc_cols_pl<HD, PL, 1>(cr, {kb}, dq_, [&](auto tc, float (&w1)[1]) { w1[0] = weight(tc); });
    }
    STAMP(5)
  }
  STAMP_FLUSH(p.stamps, blockIdx.x)
  mfma_drain();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  // merge the two key groups: group 1 parks its partial dq, group 0 adds, scales and stores
  float* park = smem + rb * NDB * 16 * 64;
  if (kg == 1) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) park[(d * 16 + i) * 64 + lane] = dq[d][i];
  }
  __syncthreads();
  if (kg == 0 && q < T) {
    float* row = dqkv + ((long long)b * T + q) * ld + h * HD;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (dq[d][4 * a + r] + park[(d * 16 + 4 * a + r) * 64 + lane]) * p.scale;
        *reinterpret_cast<f32x4*>(row + 32 * d + 8 * a + 4 * hi) = v;
      }
  }
}

// dQ from the spilled dS (see attn2_bwd_dkv_kernel<.., SPILL>): dq^T[d][q] = scale * sum_key K[key][d] * dS[q][key].
// own = 32 queries per wavefront; tiles = keys (K only: 16 KB per tile, two stages per key group = 64 KB: two workgroups per
// CU).  A lane's sixteen weights of a key tile -- dS[q][key0 + 8 a + 4 hi + r], the accumulator layout of the other
// kernels' S -- are four 16-byte loads from its own dS row, requested one tile ahead.
template <int HD>
__global__ __launch_bounds__(256, 2) void attn2_bwd_dq_ds_kernel(Attn2Args p, const float* __restrict__ ds,
                                                                 float* __restrict__ dqkv) {
  constexpr int NDB = HD / 32, TILE = KT * HD;
  __shared__ __attribute__((aligned(1024))) float smem[4 * TILE];  // [key group][stage] K
  static_assert(2 * NDB * 16 * 64 <= 4 * TILE, "parking area");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  const int rb = wave & 1, kg = wave >> 1, gtid = tid & 127;
  int qb, b, h, len;
  work_unit(p, (p.T + 63) / 64, qb, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D, Tp32 = (T + 31) & ~31;
  const int q = qb * 64 + rb * 32 + l32;
  const int kend = min(T, len);
  const int nt = (kend + KT - 1) / KT, n0 = (nt + 1) / 2;
  const int tile0 = kg ? n0 : 0, mine = kg ? nt - n0 : n0;
  const float* base = p.qkv + (long long)b * T * ld;
  const __amdgpu_buffer_rsrc_t rk = make_rsrc(base + D + h * HD);
  TileDma<HD, 128> dma;
  dma.setup(ld, gtid);
  float* gbuf = smem + kg * 2 * TILE;
  if (mine > 0) dma.issue(rk, gbuf, tile0 * KT, T, ld, rb);
  const float* dsrow = ds + (((long long)(b * p.H + h) * T + (q < T ? q : 0)) * Tp32) + 4 * hi;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 wc[4] = {zero4, zero4, zero4, zero4}, wn[4] = {zero4, zero4, zero4, zero4};
  if (mine > 0 && q < T) {
#pragma unroll
    for (int a = 0; a < 4; ++a) wc[a] = *reinterpret_cast<const f32x4*>(dsrow + tile0 * KT + 8 * a);
  }
  f32x16 dq[NDB];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[d][i] = 0.f;
  ColRd cr;
  cr.setup<HD>(l32, hi);
  const unsigned g0 = lds_addr(gbuf);
  for (int j = 0; j < n0; ++j) {
    const int key0 = (tile0 + j) * KT;
    const bool act = j < mine, actn = j + 1 < mine;
    const int st = j & 1;
    wait_vmcnt_barrier<0>();  // tile j (and this lane's weights for it) landed; everybody is done with tile j - 1
    if (!act) continue;
    const unsigned kb = g0 + st * TILE * 4;
    if (actn) {
      dma.issue(rk, gbuf + (st ^ 1) * TILE, key0 + KT, T, ld, rb);
      if (q < T) {
#pragma unroll
        for (int a = 0; a < 4; ++a) wn[a] = *reinterpret_cast<const f32x4*>(dsrow + key0 + KT + 8 * a);
      }
    }
    read_mfma_stream<16 * NDB, 8>(
        [&](auto ic, float& dst) {
          constexpr int i = decltype(ic)::value, t = i / NDB, db = i % NDB;
          lds_rd32<col_off<HD, t, db>()>(dst, kb + cr.base[t & 3]);
        },
        [&](auto ic, float v) {
          constexpr int i = decltype(ic)::value, t = i / NDB, db = i % NDB;
          mfma32_agpr(dq[db], v, wc[t >> 2][t & 3]);
        });
#pragma unroll
    for (int a = 0; a < 4; ++a) wc[a] = wn[a];
  }
  mfma_drain();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  // merge the two key groups: group 1 parks its partial dq, group 0 adds, scales and stores
  float* park = smem + rb * NDB * 16 * 64;
  if (kg == 1) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) park[(d * 16 + i) * 64 + lane] = dq[d][i];
  }
  __syncthreads();
  if (kg == 0 && q < T) {
    float* row = dqkv + ((long long)b * T + q) * ld + h * HD;
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (dq[d][4 * a + r] + park[(d * 16 + 4 * a + r) * 64 + lane]) * p.scale;
        *reinterpret_cast<f32x4*>(row + 32 * d + 8 * a + 4 * hi) = v;
      }
  }
}

// dK, dV: own = 32 keys per wavefront (K' = K * scale * log2e and V in registers), tiles = queries (Q, dO rows and
// their {lse', delta'} pairs)
// SPILL (fp32 MFMA path): dS -- the kernel makes it anyway, element by element, for dK -- is also written out as
// ds[b][h][q][key] (rows of Tp32 = T rounded up to 32 floats), and dQ = scale * dS . K becomes a product of its own
// (attn2_bwd_dq_ds_kernel) instead of a second kernel that recomputes S, dP and the softmax arithmetic: 5 products per
// (32 x 32) block instead of 9 in the two gradient kernels together, and the vector work of the recomputation -- which
// on this chip is added to the fp32 MFMA time -- is gone.  One 4-byte store per lane and step: a step's 32 keys of one
// query row are 128 contiguous bytes.
// SIN (with SPILL): the scores come from memory -- s_in[b][h][q][key], written by attn2_fwd_kernel<.., SS> -- instead of from
// a K.Q^T product of this kernel's own: 3 products per block instead of 4, no own K rows in registers, and the
// probabilities are the forward pass's to the bit.  A tile's sixteen scores per lane are requested right behind the tile
// barrier and used behind the dP product.
template <int HD, bool DROP, int PL, bool SPILL = false, bool SIN = false>
__global__ __launch_bounds__(256, 1) void attn2_bwd_dkv_kernel(Attn2Args p, const float* __restrict__ dout,
                                                               const float2* __restrict__ aux, float* __restrict__ dqkv,
                                                               float* __restrict__ ds = nullptr,
                                                               const float* __restrict__ s_in = nullptr) {
  static_assert(!SPILL || PL == 0, "dS is spilled by the fp32 MFMA path only");
  static_assert(!SIN || SPILL, "scores from memory: only in the spilled-dS form");
  constexpr int NJ = HD / 8, NDB = HD / 32, TILE = KT * HD, NP = TileDma<HD, 128>::NP, STAGE = 2 * TILE + 256;
  __shared__ __attribute__((aligned(1024))) float smem[4 * STAGE];  // [query group][stage][Q | dO | aux (1 KB)]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, l32 = lane & 31;
  const int rb = wave & 1, qg = wave >> 1, gtid = tid & 127;
  int kblk, b, h, len;
  work_unit(p, (p.T + 63) / 64, kblk, b, h, len);
  const int T = p.T, D = p.H * HD, ld = 3 * D;
  const int key = kblk * 64 + rb * 32 + l32;
  const float* base = p.qkv + (long long)b * T * ld;
  float* krow = dqkv + ((long long)b * T + key) * ld + D + h * HD;
  float* vrow = krow + D;
  if (kblk * 64 >= len) {  // every key of this workgroup is padding: its gradients are zero
    if (qg == 0 && key < T) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          *reinterpret_cast<f32x4*>(krow + 32 * d + 8 * a + 4 * hi) = z;
          *reinterpret_cast<f32x4*>(vrow + 32 * d + 8 * a + 4 * hi) = z;
        }
    }
    return;
  }
  const int nt = (T + KT - 1) / KT, n0 = (nt + 1) / 2;  // query tiles: every row of the padded sequence has a gradient
  const int tile0 = qg ? n0 : 0, mine = qg ? nt - n0 : n0;
  PairHash ph;
  ph.setup(fs2_resolve_drop(p.drop));
  const __amdgpu_buffer_rsrc_t rq = make_rsrc(base + h * HD), rdo = make_rsrc(dout + (long long)b * T * D + h * HD);
  const __amdgpu_buffer_rsrc_t rax = make_rsrc(reinterpret_cast<const float*>(aux + ((long long)b * p.H + h) * T));
  TileDma<HD, 128> dmaq, dmao;
  dmaq.setup(ld, gtid);
  dmao.setup(D, gtid);
  float* gbuf = smem + qg * 2 * STAGE;
  // the 32 {lse', delta'} pairs of a query tile: 16 pieces of 16 bytes, issued by the first 16 lanes of the row
  // block-0 wavefront (the other lanes carry the out-of-range offset: zeros into the slack of the 1 KB slot)
  auto issue_aux = [&](float* stage, int q0) {
    if (rb == 0) {
      const int piece = lane;  // 2 queries per piece
      const bool ok = lane < 16 && q0 + 2 * piece < T;  // (T even or not: a piece past T-1 reads one pair too many -> aux is padded by the launcher)
      blds16(rax, ok ? piece * 16 : FS2_OOB, q0 * 8, stage + 2 * TILE);
    }
  };
  if (mine > 0) {
    dmaq.issue(rq, gbuf, tile0 * KT, T, ld, rb);
    dmao.issue(rdo, gbuf + TILE, tile0 * KT, T, D, rb);
    issue_aux(gbuf, tile0 * KT);
  }
  const float kscale = p.scale * 1.44269504088896f;
  f32x4 kv[PL == 0 ? NJ : 1], vv[PL == 0 ? NJ : 1];
  Pl<(PL == 0 ? 1 : PL)> kp[HD / 16], vp[HD / 16];
  if constexpr (PL == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, w = {0.f, 0.f, 0.f, 0.f};
      if (key < T) {
        if constexpr (!SIN) v = *reinterpret_cast<const f32x4*>(base + (long long)key * ld + D + h * HD + 8 * j + 4 * hi);
        w = *reinterpret_cast<const f32x4*>(base + (long long)key * ld + 2 * D + h * HD + 8 * j + 4 * hi);
      }
      if constexpr (!SIN) kv[j] = to_agpr(v * kscale);
      vv[j] = to_agpr(w);
    }
  } else {
    own_planes<HD, PL>(kp, hi, [&](int j4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (key < T) v = *reinterpret_cast<const f32x4*>(base + (long long)key * ld + D + h * HD + 4 * j4);
      return v * kscale;
    });
    own_planes<HD, PL>(vp, hi, [&](int j4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (key < T) v = *reinterpret_cast<const f32x4*>(base + (long long)key * ld + 2 * D + h * HD + 4 * j4);
      return v;
    });
  }
  f32x16 g_[2][NDB];  // [0]: dV, [1]: dK
  auto& dv = g_[0];
  auto& dk = g_[1];
#pragma unroll
  for (int d = 0; d < NDB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dk[d][i] = 0.f;
      dv[d][i] = 0.f;
    }
  const bool key_ok = key < len;
  const uint32_t Tp = (uint32_t)(T + (T & 1));
  const uint32_t head0 = (uint32_t)((unsigned long long)(b * p.H + h) * T) * Tp;  // (wraps like the other kernels' 32-bit index)
  const int sh16 = 16 * (key & 1);
  // dS slab of this (utterance, head): [T][Tp32]; rows past T fall to the range check, keys past Tp32 carry the sentinel
  const int Tp32 = (T + 31) & ~31;
  const __amdgpu_buffer_rsrc_t rds = __builtin_amdgcn_make_buffer_rsrc(
      SPILL ? (void*)(ds + ((long long)(b * p.H + h) * T) * Tp32) : (void*)dqkv, 0, SPILL ? T * Tp32 * 4 : 0, 0x00020000);
  const int ds_voff = key < Tp32 ? (4 * hi * Tp32 + key) * 4 : FS2_OOB;
  const __amdgpu_buffer_rsrc_t rsi = __builtin_amdgcn_make_buffer_rsrc(
      SIN ? (void*)(s_in + ((long long)(b * p.H + h) * T) * Tp32) : (void*)dqkv, 0, SIN ? T * Tp32 * 4 : 0, 0x00020000);
  RowRd rd;
  rd.setup<HD>(l32, hi);
  ColRd cr;
  cr.setup<HD>(l32, hi);
  const unsigned g0 = lds_addr(gbuf);
  STAMP_DECL;
  for (int j = 0; j < n0; ++j) {
    const int q0 = (tile0 + j) * KT;
    const bool act = j < mine, actn = j + 1 < mine;
    const int st = j & 1;
    STAMP(0)
    // tile j landed for everybody; the sixteen dS stores of the previous tile are younger than its DMA and may stay in flight
    if (SPILL && j > 0) wait_vmcnt_barrier<16>(); else wait_vmcnt_barrier<0>();
    STAMP(1)
    if (!act) continue;
    const unsigned qb_ = g0 + st * STAGE * 4, ob = qb_ + TILE * 4, ab = ob + TILE * 4;
    float* nxt = gbuf + (st ^ 1) * STAGE;
    const unsigned ab_hi = ab + 4 * hi * 8;  // {lse', delta'} of tile row 8a + 4hi + r: broadcast ds_read_b64 at (8a + r) * 8
    if (actn) issue_aux(nxt, q0 + KT);
    auto qdma = [&](auto jc) {
      constexpr int it = decltype(jc)::value;
      if constexpr (it < NP) {
        if (actn) dmaq.template piece<it>(rq, nxt, q0 + KT, T, ld, rb);
      }
    };
    auto odma = [&](auto jc) {
      constexpr int it = decltype(jc)::value;
      if constexpr (it < NP) {
        if (actn) dmao.template piece<it>(rdo, nxt + TILE, q0 + KT, T, D, rb);
      }
    };
    f32x16 s, dp;
    if constexpr (SIN) {
      // S[q0 + 8 (t >> 2) + 4 hi + (t & 3)][key]: the layout of the dS stores below; rows past T and keys past Tp32 read zeros
      // (such keys are masked by key_ok, such rows' gradients are never stored)
      sfor<16>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        s[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, ds_voff, (q0 + 8 * (t >> 2) + (t & 3)) * Tp32 * 4, 0));
      });
      // both operand tiles' DMA pieces go out under the one product that is left in front of the gradient stream
      auto qodma = [&](auto jc) {
        qdma(jc);
        odma(jc);
      };
      dp = dot_rows<HD, true>(rd, ob, vv, qodma);
    } else {
      if constexpr (PL == 0) s = dot_rows<HD, true>(rd, qb_, kv, qdma);
      else s = dot_rows_pl<HD, PL>(l32, hi, qb_, kp, qdma);
      STAMP(2)
      if constexpr (PL == 0) dp = dot_rows<HD, true>(rd, ob, vv, odma);
      else dp = dot_rows_pl<HD, PL>(l32, hi, ob, vp, odma);
    }
    STAMP(3)
    const uint32_t rowkey = head0 + (uint32_t)(q0 + 4 * hi) * Tp + (uint32_t)key;  // element index of (row q0 + 4hi, own key)
    float wv = 0.f, wk = 0.f;
    float2 axs[2];  // the pairs of steps t and t + 1 (two in flight: a pair is read a whole step before its use)
    auto aux_read = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      lds_rd64<(8 * (t >> 2) + (t & 3)) * 8>(axs[t & 1], ab_hi);
    };
    auto weights = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      pin(axs[t & 1]);
      const float pp = key_ok ? __builtin_amdgcn_exp2f(s[t] - axs[t & 1].x) : 0.f;
      float pd = pp;
      if constexpr (DROP) {
        const uint32_t idx = rowkey + (uint32_t)(8 * (t >> 2) + (t & 3)) * Tp;
        const uint32_t hh = ph.hash(idx >> 1);
        pd = ((hh >> sh16) & 0xffffu) >= ph.thresh ? pp : 0.f;
      }
      wv = pd;
      wk = fmaf(-pp, axs[t & 1].y, pd * dp[t]);
      if constexpr (SPILL)  // dS[q0 + 8 (t >> 2) + 4 hi + (t & 3)][key]
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, wk), rds, ds_voff,
                                              (q0 + 8 * (t >> 2) + (t & 3)) * Tp32 * 4, 0);
    };
    if constexpr (PL == 0) {
      aux_read(std::integral_constant<int, 0>{});
      aux_read(std::integral_constant<int, 1>{});
      lds_wait<1>();
      weights(std::integral_constant<int, 0>{});
      float wv_c = wv, wk_c = wk;
      STAMP(4)
      // per step 2 NDB MFMAs: dv^T[d][key] += dO[q][d] * pd[q][key], dk^T[d][key] += Q[q][d] * dS[q][key].  The aux pair
      // of step t + 2 is read behind the last MFMA of step t: by the time weights(t + 2) runs, the stream's own waits
      // (at most 7 younger reads outstanding before every MFMA) have long covered it.
      read_mfma_stream<32 * NDB, 8>(
          [&](auto ic, float& dst) {
            constexpr int i = decltype(ic)::value, t = i / (2 * NDB), u = i % (2 * NDB), db = u % NDB;
            lds_rd32<col_off<HD, t, db>()>(dst, (u < NDB ? ob : qb_) + cr.base[t & 3]);
          },
          [&](auto ic, float v) {
            constexpr int i = decltype(ic)::value, t = i / (2 * NDB), u = i % (2 * NDB), db = u % NDB;
            if constexpr (u < NDB) mfma32_agpr(dv[db], v, wv_c);
            else mfma32_agpr(dk[db], v, wk_c);
            if constexpr (u == 2 * NDB - 1 && t + 1 < 16) {
              weights(std::integral_constant<int, t + 1>{});
              wv_c = wv;
              wk_c = wk;
              if constexpr (t + 2 < 16) aux_read(std::integral_constant<int, t + 2>{});
            }
          });
    } else {
      // the eight {lse', delta'} pairs of a 16-row step are read (one wait) right before the step's weights
      float2 axa[8];
      STAMP(4)
      acc_cols_pl<HD, PL, 2>(cr, {ob, qb_}, g_, [&](auto tc, float (&w2)[2]) {
        constexpr int t = decltype(tc)::value;
        if constexpr ((t & 7) == 0) {
          sfor<8>([&](auto ec) {
            constexpr int u = t + decltype(ec)::value;
            lds_rd64<(8 * (u >> 2) + (u & 3)) * 8>(axa[u & 7], ab_hi);
          });
          lds_wait<0>();
        }
        pin(axa[t & 7]);
        const float pp = key_ok ? __builtin_amdgcn_exp2f(s[t] - axa[t & 7].x) : 0.f;
        float pd = pp;
        if constexpr (DROP) {
          const uint32_t idx = rowkey + (uint32_t)(8 * (t >> 2) + (t & 3)) * Tp;
          const uint32_t hh = ph.hash(idx >> 1);
          pd = ((hh >> sh16) & 0xffffu) >= ph.thresh ? pp : 0.f;
        }
        w2[0] = pd;
        w2[1] = fmaf(-pp, axa[t & 7].y, pd * dp[t]);
      });
    }
    STAMP(5)
  }
  STAMP_FLUSH(p.stamps, blockIdx.x)
  mfma_drain();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  float* park = smem + rb * 2 * NDB * 16 * 64;
  static_assert(2 * 2 * NDB * 16 * 64 <= 4 * STAGE, "parking area");
  if (qg == 1) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        park[(d * 16 + i) * 64 + lane] = dk[d][i];
        park[((NDB + d) * 16 + i) * 64 + lane] = dv[d][i];
      }
  }
  __syncthreads();
  if (qg == 0 && key < T) {
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4 gk, gv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gk[r] = (dk[d][4 * a + r] + park[(d * 16 + 4 * a + r) * 64 + lane]) * p.scale;
          gv[r] = dv[d][4 * a + r] + park[((NDB + d) * 16 + 4 * a + r) * 64 + lane];
        }
        *reinterpret_cast<f32x4*>(krow + 32 * d + 8 * a + 4 * hi) = gk;
        *reinterpret_cast<f32x4*>(vrow + 32 * d + 8 * a + 4 * hi) = gv;
      }
  }
}

}  // namespace

bool fs2_attn2_supported(int HD, int operand_bf16) { return (operand_bf16 >= 0 && operand_bf16 <= 2) && (HD == 64 || HD == 128); }

int fs2_attn2_fwd(const Attn2Args& a, float* o, float* lse, hipStream_t s, float* s_out) {
  // 32-bit dropout element index; the mask's rows are padded to an even length (one hash serves two neighbouring keys),
  // so the index stride is T + (T & 1) -- the same bound in the forward and the backward launcher
  if ((double)a.B * a.H * a.T * (a.T + (a.T & 1)) >= 4294967296.0) return FS2HIP_EINVAL;
  dim3 grid(((a.T + 63) / 64) * a.H * a.B);
  if (s_out && (a.planes == 1 || (long long)a.T * ((a.T + 31) & ~31) * 4 >= 0x7fffffffLL)) return FS2HIP_EINVAL;
#define FS2_ATTN2_FWD(HD_, DROP_)                                                                 \
  if (a.planes == 3 && s_out) attn2_fwd_kernel<HD_, DROP_, 3, true><<<grid, dim3(256), 0, s>>>(a, o, lse, s_out); \
  else if (a.planes == 3) attn2_fwd_kernel<HD_, DROP_, 3><<<grid, dim3(256), 0, s>>>(a, o, lse);    \
  else if (a.planes == 1) attn2_fwd_kernel<HD_, DROP_, 1><<<grid, dim3(256), 0, s>>>(a, o, lse);    \
  else if (s_out) attn2_fwd_kernel<HD_, DROP_, 0, true><<<grid, dim3(256), 0, s>>>(a, o, lse, s_out); \
  else attn2_fwd_kernel<HD_, DROP_, 0><<<grid, dim3(256), 0, s>>>(a, o, lse);
  if (a.HD == 128) {
    if (a.drop.on) { FS2_ATTN2_FWD(128, true) } else { FS2_ATTN2_FWD(128, false) }
  } else {
    if (a.drop.on) { FS2_ATTN2_FWD(64, true) } else { FS2_ATTN2_FWD(64, false) }
  }
#undef FS2_ATTN2_FWD
  FS2_LAUNCH_CHECK();
  return 0;
}

// `delta` is the caller's [B][H][T] scratch of the first-generation interface; the second generation needs a
// {lse', delta'} pair per row and head plus one pair of slack, so it is only used when the caller passes `aux`.
int fs2_attn2_bwd(const Attn2Args& a, const float* o, const float* dout, const float* lse, float* aux, float* dqkv,
                  hipStream_t s) {
  if ((double)a.B * a.H * a.T * (a.T + (a.T & 1)) >= 4294967296.0) return FS2HIP_EINVAL;
  const float dscale = a.drop.on ? a.drop.scale : 1.f;
  attn2_prep_kernel<<<dim3((a.B * a.T + 3) / 4), dim3(256), 0, s>>>(dout, o, lse, reinterpret_cast<float2*>(aux), a.B, a.T,
                                                                     a.H, a.HD, log2f(dscale), 1.f / dscale);
  FS2_LAUNCH_CHECK();
  dim3 grid(((a.T + 63) / 64) * a.H * a.B);
  const float2* ax = reinterpret_cast<const float2*>(aux);
// "32-split": dQ on the bf16 pipe (237 vs 277 us at the decoder's shape), dK/dV stays on the fp32 MFMAs -- with one
// wavefront per SIMD its cut arithmetic (two gradient tiles per step) is not hidden and the split version is slower
// (430 vs 340 us).  "bf16-mixed": everything on the bf16 pipe.
#define FS2_ATTN2_BWD_PL(HD_, DROP_, PQ_, PKV_)                                                \
  attn2_bwd_dq_kernel<HD_, DROP_, PQ_><<<grid, dim3(256), 0, s>>>(a, dout, ax, dqkv);          \
  FS2_LAUNCH_CHECK();                                                                          \
  attn2_bwd_dkv_kernel<HD_, DROP_, PKV_><<<grid, dim3(256), 0, s>>>(a, dout, ax, dqkv);
#define FS2_ATTN2_BWD(HD_, DROP_)                                     \
  if (a.planes == 3) { FS2_ATTN2_BWD_PL(HD_, DROP_, 3, 0) }           \
  else if (a.planes == 1) { FS2_ATTN2_BWD_PL(HD_, DROP_, 1, 1) }      \
  else { FS2_ATTN2_BWD_PL(HD_, DROP_, 0, 0) }
  if (a.HD == 128) {
    if (a.drop.on) { FS2_ATTN2_BWD(128, true) } else { FS2_ATTN2_BWD(128, false) }
  } else {
    if (a.drop.on) { FS2_ATTN2_BWD(64, true) } else { FS2_ATTN2_BWD(64, false) }
  }
#undef FS2_ATTN2_BWD_PL
#undef FS2_ATTN2_BWD
  FS2_LAUNCH_CHECK();
  return 0;
}

// The backward pass with dS spilled by the dK/dV kernel and dQ as a product of its own (fp32 MFMA path only: the caller
// checks fs2_attn2_bwd_spill_elems first).  `ds`: B * H * T * (T rounded up to 32) floats of scratch.
long long fs2_attn2_bwd_spill_elems(const Attn2Args& a) {
  if (a.planes != 0 || (a.HD != 64 && a.HD != 128)) return 0;
  const long long n = (long long)a.B * a.H * a.T * ((a.T + 31) & ~31);
  return n * 4 < 0x7fffffffLL * 16 ? n : 0;
}
int fs2_attn2_bwd_spill(const Attn2Args& a, const float* o, const float* dout, const float* lse, float* aux, float* ds,
                        float* dqkv, hipStream_t s, const float* s_in) {
  if ((double)a.B * a.H * a.T * (a.T + (a.T & 1)) >= 4294967296.0 || fs2_attn2_bwd_spill_elems(a) == 0) return FS2HIP_EINVAL;
  if ((long long)a.T * ((a.T + 31) & ~31) * 4 >= 0x7fffffffLL) return FS2HIP_EINVAL;  // one (utterance, head) slab per resource
  const float dscale = a.drop.on ? a.drop.scale : 1.f;
  attn2_prep_kernel<<<dim3((a.B * a.T + 3) / 4), dim3(256), 0, s>>>(dout, o, lse, reinterpret_cast<float2*>(aux), a.B, a.T,
                                                                     a.H, a.HD, log2f(dscale), 1.f / dscale);
  FS2_LAUNCH_CHECK();
  dim3 grid(((a.T + 63) / 64) * a.H * a.B);
  const float2* ax = reinterpret_cast<const float2*>(aux);
#define FS2_ATTN2_SPILL(HD_, DROP_)                                                                              \
  if (s_in) attn2_bwd_dkv_kernel<HD_, DROP_, 0, true, true><<<grid, dim3(256), 0, s>>>(a, dout, ax, dqkv, ds, s_in); \
  else attn2_bwd_dkv_kernel<HD_, DROP_, 0, true><<<grid, dim3(256), 0, s>>>(a, dout, ax, dqkv, ds);                  \
  FS2_LAUNCH_CHECK();                                                                                            \
  attn2_bwd_dq_ds_kernel<HD_><<<grid, dim3(256), 0, s>>>(a, ds, dqkv);
  if (a.HD == 128) {
    if (a.drop.on) { FS2_ATTN2_SPILL(128, true) } else { FS2_ATTN2_SPILL(128, false) }
  } else {
    if (a.drop.on) { FS2_ATTN2_SPILL(64, true) } else { FS2_ATTN2_SPILL(64, false) }
  }
#undef FS2_ATTN2_SPILL
  FS2_LAUNCH_CHECK();
  return 0;
}

#ifdef FS2_ATTN_STAMPS
// diagnostic build: one of the two gradient kernels alone (which = 0: dQ, 1: dK/dV), HD = 128, after a prep launch
int fs2_attn2_bwd_one(const Attn2Args& a, const float* dout, const float* aux, float* dqkv, int which, hipStream_t s) {
  dim3 grid(((a.T + 63) / 64) * a.H * a.B);
  const float2* ax = reinterpret_cast<const float2*>(aux);
#define FS2_ONE(K_, DROP_, P3_)                                                               \
  if (a.planes == 3) K_<128, DROP_, P3_><<<grid, dim3(256), 0, s>>>(a, dout, ax, dqkv);         \
  else if (a.planes == 1) K_<128, DROP_, 1><<<grid, dim3(256), 0, s>>>(a, dout, ax, dqkv);      \
  else K_<128, DROP_, 0><<<grid, dim3(256), 0, s>>>(a, dout, ax, dqkv);
  if (which == 0) {
    if (a.drop.on) { FS2_ONE(attn2_bwd_dq_kernel, true, 3) } else { FS2_ONE(attn2_bwd_dq_kernel, false, 3) }
  } else {
    if (a.drop.on) { FS2_ONE(attn2_bwd_dkv_kernel, true, 0) } else { FS2_ONE(attn2_bwd_dkv_kernel, false, 0) }
  }
#undef FS2_ONE
  return 0;
}
#endif
