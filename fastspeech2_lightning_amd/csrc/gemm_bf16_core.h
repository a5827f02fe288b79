// GEMM core for operands that ARE bf16 in memory (Fs2GemmArgs.operand_bf16 == 4; precision "bf16-mixed" with bf16
// activation storage).  v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 or bf16 results.
//
// What differs from the fp32 cores (gemm2.hip), beyond the element type:
//  * every operand orientation is read straight from its natural layout -- a k-contiguous operand ([row][k]) through
//    the chunk-swizzled [rows][64] image and ds_read_b128, a REDUCTION-MAJOR operand ([k][row]: the weight of the
//    data-gradient GEMM, both operands of the weight-gradient GEMM) through a [64][rows] image and the transposing
//    read ds_read_b64_tr_b16, which hands each lane four consecutive reduction steps of its own row.  Activations
//    and weights therefore exist once, in bf16, and no transposed copy of anything is made;
//  * K-tiles are 64 reduction steps deep (rows of a k-contiguous image stay 128 bytes);
//  * the MFMA takes the B fragment as its first operand and the A fragment as its second, so an accumulator holds
//    C TRANSPOSED: the lane is the output row, registers 4t..4t+3 are four consecutive output columns.  The
//    epilogue then moves 16 bytes (fp32) / 8 bytes (bf16) per lane and store instead of one element, the bias is the
//    same for all lanes of a half wave, and the two elements that share one dropout hash sit in one lane;
//  * the weight-gradient form can also sum the columns of its A operand (the bias gradient: A^T . 1) with one more
//    MFMA against a fragment of ones, in the workgroups of the first tile column.
#pragma once
#include "gemm_common.h"

namespace {

constexpr int BKE = 64;  // reduction elements per K-tile
typedef __attribute__((address_space(3))) void lds_void;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

constexpr int B_OOB = (int)0x80000000;  // with num_records = 2^31: offset + anything >= num_records -> zeros in LDS

__device__ __forceinline__ __amdgpu_buffer_rsrc_t b_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, B_OOB, 0x00020000);
}
__device__ __forceinline__ void b_dma16(__amdgpu_buffer_rsrc_t r, int voff, int soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// conv-tap modes (as in gemm2_core.h)
constexpr int BT_NONE = 0;  // plain GEMM
constexpr int BT_RED = 1;   // shift_operand == 0, Rper % 64 == 0: the tap is a per-K-tile scalar (row shift of A, slice of B)
constexpr int BT_ROWS = 2;  // shift_operand == 1 (weight gradient): reduction rows of B shifted by the slice's tap, T >= 64

// chunk swizzle of a reduction-major image [64][ROWS]: the transposing read of a 32-lane half touches four k-rows x
// 64 contiguous bytes; XOR-ing the 64-byte segment index with the k-row puts them on all 64 banks
template <int ROWS>
__device__ __forceinline__ int red_swz(int k) {
  return ROWS >= 128 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2);
}

template <int ROWS>
struct PiecesB {
  int voff[ROWS / 32];  // byte offset of the piece in K-tile 0 (B_OOB: outside the matrix)
  int t[ROWS / 32];     // conv taps: time index of the piece's row (A rows, BT_RED) / of its reduction row (B rows, BT_ROWS)
};

// reduction element (KC) / reduction row (!KC) of piece `it` inside a K-tile
template <int ROWS, bool KC>
__device__ __forceinline__ int piece_k(int it, int tid) {
  const int pidx = it * 256 + tid;
  if (KC) return ((pidx & 7) ^ (((pidx >> 3) >> 1) & 7)) << 3;
  return pidx / (ROWS / 8);
}

template <int ROWS, bool KC, bool IS_A, int TAPS>
__device__ __forceinline__ void setup_pieces_b(PiecesB<ROWS>& pc, const GemmP& p, int row0, int r_begin, int tid) {
  const Fs2GemmArgs& a = p.a;
  const int ld = IS_A ? a.lda : a.ldb;
  const int nrows = IS_A ? a.Mc : a.Nc;
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const int pidx = it * 256 + tid;
    pc.t[it] = 0;
    if (KC) {
      const int row = pidx >> 3, gr = row0 + row;
      pc.voff[it] = gr < nrows ? (gr * ld + piece_k<ROWS, true>(it, tid)) * 2 : B_OOB;
      if (TAPS == BT_RED && IS_A) pc.t[it] = gr % a.T;
    } else {
      constexpr int CH = ROWS / 8;
      const int k = pidx / CH, c = pidx % CH;
      const int col = row0 + ((c ^ red_swz<ROWS>(k)) << 3);
      pc.voff[it] = col < nrows ? (k * ld + col) * 2 : B_OOB;
      if (TAPS == BT_ROWS && !IS_A) pc.t[it] = (r_begin + k) % a.T;
    }
  }
}

template <int ROWS, bool KC>
__device__ __forceinline__ void issue_plain(char* tile, __amdgpu_buffer_rsrc_t r, const PiecesB<ROWS>& pc, int soff,
                                            int rem, int wave, int tid) {
  if (rem >= BKE) {
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it) b_dma16(r, pc.voff[it], soff, tile + (it * 256 + wave * 64) * 16);
  } else {  // the reduction ends inside this K-tile (R is a multiple of 8: whole pieces)
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it)
      b_dma16(r, piece_k<ROWS, KC>(it, tid) < rem ? pc.voff[it] : B_OOB, soff, tile + (it * 256 + wave * 64) * 16);
  }
}
// BT_RED, A operand (k-contiguous): rows shifted by the K-tile's tap; a row whose shifted time index leaves [0, T) is
// the convolution's zero padding
template <int ROWS>
__device__ __forceinline__ void issue_shift_rows(char* tile, __amdgpu_buffer_rsrc_t r, const PiecesB<ROWS>& pc, int soff,
                                                 int shift, int T, int wave) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const bool ok = (unsigned)(pc.t[it] + shift) < (unsigned)T;
    b_dma16(r, ok ? pc.voff[it] : B_OOB, soff, tile + (it * 256 + wave * 64) * 16);
  }
}
// BT_ROWS, B operand (reduction-major): the reduction index is the (b, t) row itself; advances the pieces' time index
// by one K-tile (T >= 64)
template <int ROWS>
__device__ __forceinline__ void issue_shift_red(char* tile, __amdgpu_buffer_rsrc_t r, PiecesB<ROWS>& pc, int soff, int rem,
                                                int shift, int T, int wave, int tid) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const bool ok = piece_k<ROWS, false>(it, tid) < rem && (unsigned)(pc.t[it] + shift) < (unsigned)T;
    b_dma16(r, ok ? pc.voff[it] : B_OOB, soff, tile + (it * 256 + wave * 64) * 16);
    const int t = pc.t[it] + BKE;
    pc.t[it] = t >= T ? t - T : t;
  }
}

template <bool AKC, bool BKC, int TAPS>
struct StreamB {
  __amdgpu_buffer_rsrc_t ra, rb;
  int r_begin, r_end, kt;
  int tap, kin, shift_min, shift_z;

  __device__ __forceinline__ void begin(const GemmP& p, int r_begin_, int r_end_, int shift_z_) {
    const Fs2GemmArgs& a = p.a;
    r_begin = r_begin_;
    r_end = r_end_;
    shift_z = shift_z_;
    kt = tap = kin = shift_min = 0;
    const u16* A = (const u16*)a.A;
    const u16* B = (const u16*)a.B;
    if (TAPS == BT_RED) {
      tap = r_begin / p.Rper;
      kin = r_begin - tap * p.Rper;
      shift_min = a.tap_add + (a.tap_mul < 0 ? a.tap_mul * (a.taps - 1) : 0);  // folded into A's base: soff stays >= 0
      A += (long long)shift_min * a.lda;
    } else if (TAPS == BT_ROWS) {
      B += (long long)shift_z * a.ldb;
    }
    ra = b_rsrc(A);
    rb = b_rsrc(B);
  }

  template <int BM, int BN>
  __device__ __forceinline__ void issue(const GemmP& p, char* At, char* Bt, PiecesB<BM>& pa, PiecesB<BN>& pb, int wave,
                                        int tid) {
    const Fs2GemmArgs& a = p.a;
    const int r0 = r_begin + kt * BKE, rem = r_end - r0;
    if constexpr (TAPS == BT_RED) {
      const int shift = tap * a.tap_mul + a.tap_add;
      issue_shift_rows<BM>(At, ra, pa, ((shift - shift_min) * a.lda + kin) * 2, shift, a.T, wave);
      issue_plain<BN, BKC>(Bt, rb, pb, (int)(((long long)tap * a.b_tap_stride + (BKC ? kin : kin * a.ldb)) * 2), BKE, wave, tid);
      kin += BKE;
      if (kin == p.Rper) {
        kin = 0;
        ++tap;
      }
    } else if constexpr (TAPS == BT_ROWS) {
      issue_plain<BM, AKC>(At, ra, pa, (AKC ? r0 : r0 * a.lda) * 2, rem, wave, tid);
      issue_shift_red<BN>(Bt, rb, pb, r0 * a.ldb * 2, rem, shift_z, a.T, wave, tid);
    } else {
      issue_plain<BM, AKC>(At, ra, pa, (AKC ? r0 : r0 * a.lda) * 2, rem, wave, tid);
      issue_plain<BN, BKC>(Bt, rb, pb, (BKC ? r0 : r0 * a.ldb) * 2, rem, wave, tid);
    }
    ++kt;
  }
};

template <int N>
__device__ __forceinline__ void b_wait_vmcnt_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void b_lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}
template <int OFF>
__device__ __forceinline__ void b_rd128(u32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void b_rdtr(u32x2& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}

// MFMA fragments of one 16-deep reduction step for T row blocks of 32
template <int T, bool KC>
struct FragB;
template <int T>
struct FragB<T, true> {
  u32x4 q[T];
  static constexpr int READS = T;
  __device__ __forceinline__ bf16x8 get(int i) const { return __builtin_bit_cast(bf16x8, q[i]); }
  __device__ __forceinline__ void pin_all() {
#pragma unroll
    for (int i = 0; i < T; ++i) asm volatile("" : "+v"(q[i]));
  }
};
template <int T>
struct FragB<T, false> {
  u32x2 lo[T], hi[T];
  static constexpr int READS = 2 * T;
  __device__ __forceinline__ bf16x8 get(int i) const {
    const u32x4 v = {lo[i][0], lo[i][1], hi[i][0], hi[i][1]};
    return __builtin_bit_cast(bf16x8, v);
  }
  __device__ __forceinline__ void pin_all() {
#pragma unroll
    for (int i = 0; i < T; ++i) {
      asm volatile("" : "+v"(lo[i]));
      asm volatile("" : "+v"(hi[i]));
    }
  }
};

// per-lane LDS byte address of the operand reads (relative to the operand's image in a stage)
template <int ROWS, bool KC>
struct RdB {
  unsigned a[KC ? 4 : 1];
  __device__ __forceinline__ void setup(int wrow0, int lane) {
    if (KC) {  // [ROWS][64 bf16] = 128-byte rows, chunk index XOR ((row >> 1) & 7); lane half h takes chunk 2g + h
      const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
      for (int g = 0; g < 4; ++g) a[g] = (wrow0 + l31) * 128 + (((2 * g + h) ^ ((l31 >> 1) & 7)) << 4);
    } else {   // [64][ROWS bf16]: lane 4q+p of 16-lane group (h, ch) addresses k-row 8h + q, columns 16 ch + 4p .. +3
      const int grp = lane >> 4, h = grp >> 1, ch = grp & 1, q = (lane & 15) >> 2, pp = lane & 3;
      const int krow = 8 * h + q, colb = wrow0 + 16 * ch + 4 * pp;
      a[0] = krow * (ROWS * 2) + (((colb >> 3) ^ red_swz<ROWS>(krow)) << 4) + 8 * (pp & 1);
    }
  }
};

template <int G, int ROWS, int T, bool KC>
__device__ __forceinline__ void frag_read_b(FragB<T, KC>& f, const RdB<ROWS, KC>& rd, unsigned base) {
  if constexpr (KC) {
    b_rd128<0>(f.q[0], rd.a[G] + base);
    if constexpr (T > 1) b_rd128<32 * 128>(f.q[1], rd.a[G] + base);
  } else {
    constexpr int RB = ROWS * 2;
    const unsigned ad = rd.a[0] + base;
    b_rdtr<(16 * G) * RB>(f.lo[0], ad);
    b_rdtr<(16 * G + 4) * RB>(f.hi[0], ad);
    if constexpr (T > 1) {  // the next row block: chunk index + 4, i.e. byte bit 6 flipped under the XOR swizzle
      const unsigned ad1 = ad ^ 64u;
      b_rdtr<(16 * G) * RB>(f.lo[1], ad1);
      b_rdtr<(16 * G + 4) * RB>(f.hi[1], ad1);
    }
  }
}

// MFMAs of one K-tile.  acc[i][j] holds the TRANSPOSED 32x32 block: lane & 31 = row of A's block i, registers =
// columns of B's block j.  cs[i] (COLSUM): column sums of A's block i (= the bias gradient of a weight-gradient GEMM).
template <int BM, int BN, bool AKC, bool BKC, bool COLSUM>
__device__ __forceinline__ void compute_ktile_b(f32x16 (&acc)[BM / 64][BN / 64], f32x16 (&cs)[BM / 64],
                                                const RdB<BM, AKC>& rda, const RdB<BN, BKC>& rdb, unsigned sa,
                                                unsigned sb, bool do_cs) {
  constexpr int TM = BM / 64, TN = BN / 64;
  FragB<TM, AKC> fa[2];
  FragB<TN, BKC> fb[2];
  constexpr int RD = FragB<TM, AKC>::READS + FragB<TN, BKC>::READS;
  frag_read_b<0, BM>(fa[0], rda, sa);
  frag_read_b<0, BN>(fb[0], rdb, sb);
#define FS2_BSTEP(G)                                                                                            \
  {                                                                                                             \
    if (G < 3) {                                                                                                \
      frag_read_b<(G + 1) & 3, BM>(fa[(G + 1) & 1], rda, sa);                                                   \
      frag_read_b<(G + 1) & 3, BN>(fb[(G + 1) & 1], rdb, sb);                                                   \
      b_lds_wait<RD>();                                                                                         \
    } else {                                                                                                    \
      b_lds_wait<0>();                                                                                          \
    }                                                                                                           \
    fa[G & 1].pin_all();                                                                                        \
    fb[G & 1].pin_all();                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                              \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                              \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[G & 1].get(j), fa[G & 1].get(i), acc[i][j], 0, 0, 0); \
    if (COLSUM && do_cs) {                                                                                      \
      const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};                                  \
      _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                            \
          cs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), fa[G & 1].get(i), cs[i], 0, 0, 0); \
    }                                                                                                           \
  }
  FS2_BSTEP(0)
  FS2_BSTEP(1)
  FS2_BSTEP(2)
  FS2_BSTEP(3)
#undef FS2_BSTEP
}

// ---- epilogue -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// EPI: -1 raw fp32 partial sums (split-K slab), else Fs2GemmArgs.epi.  OBF: C (and out_pre) are bf16.
// ACT: the activation as a compile-time constant (the run-time switch inside the unrolled quads made every element
// pay for a three-way branch with tanhf in one arm: 25 us on a 60 us GEMM), or -1 = take it from the arguments
template <int ACT>
__device__ __forceinline__ float act_b(int act, float x) { return fs2_act(ACT >= 0 ? ACT : act, x); }
template <int ACT>
__device__ __forceinline__ float dact_b(int act, float x) { return fs2_dact(ACT >= 0 ? ACT : act, x); }

template <int BM, int BN, int EPI, bool OBF, int ACT = -1>
__device__ __forceinline__ void epilogue_b(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], void* Cv, int ldc, int m0,
                                           int n0, int wm, int wn, int lane) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const Fs2GemmArgs& a = p.a;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const int l31 = lane & 31, h = lane >> 5;
  constexpr int ES = OBF ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, a.Mc * ldc * ES, 0x00020000);
  const bool aux_bf = (a.io_bf16 & 2) != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + 32 * i + l31;
    const bool rowok = m < a.Mc;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = n0 + wn * (BN / 2) + 32 * j + 8 * t + 4 * h;
        const bool ok = rowok && n < a.Nc;  // (Nc is a multiple of 4: the quad is inside or outside as a whole)
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * t + e];
        if (EPI >= 0) {
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (a.bias && n < a.Nc) b = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = a.alpha * v[e] + b[e];
        }
        if (EPI == FS2_EPI_ACT) {
          if (a.out_pre) {
            if (OBF) {
              const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_pre, 0, a.Mc * a.ldpre * 2, 0x00020000);
              const u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
              __builtin_amdgcn_raw_buffer_store_b64(w, rp, ok ? (m * a.ldpre + n) * 2 : B_OOB, 0, 0);
              // the activation sees what the backward pass will read back: the rounded pre-activation
              v[0] = bf16_lo(w[0]); v[1] = bf16_hi(w[0]); v[2] = bf16_lo(w[1]); v[3] = bf16_hi(w[1]);
            } else {
              const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_pre, 0, a.Mc * a.ldpre * 4, 0x00020000);
              const f32x4 w = {v[0], v[1], v[2], v[3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), rp, ok ? (m * a.ldpre + n) * 4 : B_OOB, 0, 0);
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_b<ACT>(a.act, v[e]);
        } else if (EPI == FS2_EPI_DACT) {
          float x[4];
          if (aux_bf) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.Mc * a.ldaux * 2, 0x00020000);
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rx, ok ? (m * a.ldaux + n) * 2 : B_OOB, 0, 0);
            x[0] = bf16_lo(w[0]); x[1] = bf16_hi(w[0]); x[2] = bf16_lo(w[1]); x[3] = bf16_hi(w[1]);
          } else {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.Mc * a.ldaux * 4, 0x00020000);
            const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (m * a.ldaux + n) * 4 : B_OOB, 0, 0));
            x[0] = w[0]; x[1] = w[1]; x[2] = w[2]; x[3] = w[3];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= dact_b<ACT>(a.act, x[e]);
        }
        if (EPI > 0 && drop.on) {  // element index m * ldc + n, as everywhere else this mask is used
          const unsigned idx = (unsigned)(m * ldc + n);
          if ((ldc & 1) == 0) {  // n is a multiple of 4: the quad is two whole hash pairs
            float f[4];
            fs2_drop_quad(drop, idx, f);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= f[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= fs2_drop_factor(drop, (unsigned long long)idx + e);
          }
        }
        if (EPI == FS2_EPI_RESID) {
          const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0, a.Mc * a.ldr * 4, 0x00020000);
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (m * a.ldr + n) * 4 : B_OOB, 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = x[e] + a.res_scale * v[e];
        }
        if (OBF) {
          const u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          __builtin_amdgcn_raw_buffer_store_b64(w, rc, ok ? (m * ldc + n) * 2 : B_OOB, 0, 0);
        } else {
          const f32x4 w = {v[0], v[1], v[2], v[3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, w), rc, ok ? (m * ldc + n) * 4 : B_OOB, 0, 0);
        }
      }
    }
  }
}

// ---- the same epilogue with whole-row stores ------------------------------------------------------------------------
// The direct form above stores 8 (bf16) / 16 (fp32) bytes per lane at a row stride: every store instruction touches 32
// cache lines and fills an eighth / a quarter of each (measured: a 41 472 x 1024 x 256 GEMM takes the same 60 us
// whether it writes 85 MB of bf16 or 170 MB of fp32 -- it is bound by write requests, not bytes).  Here a wavefront
// passes each 32-row block of its sub-tile through LDS (its own region: no workgroup barrier) and writes it back out
// with the lanes along the rows: 16 bytes per lane, 64 .. 256 contiguous bytes per row and instruction.  `stg` is this
// wavefront's staging region (32 rows of the sub-tile + 16 bytes of padding per row).
template <int ROWB>
struct Stager {
  static constexpr int RS = ROWB + 16;      // padded row stride: rows of one quad column spread over the banks
  static constexpr int LPR = ROWB / 16;     // lanes per row on the way out
  static constexpr int RPP = 64 / LPR;      // rows per store instruction
  static constexpr int BYTES = 32 * RS;
};

template <int BM, int BN, int EPI, bool OBF, int ACT = -1>
__device__ __forceinline__ void epilogue_staged_b(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], void* Cv, int ldc,
                                                  int m0, int n0, int wm, int wn, int lane, char* stg) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int ES = OBF ? 2 : 4;
  typedef Stager<(BN / 2) * ES> SG;
  const Fs2GemmArgs& a = p.a;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const int l31 = lane & 31, h = lane >> 5;
  const bool aux_bf = (a.io_bf16 & 2) != 0;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, a.Mc * ldc * ES, 0x00020000);
  const bool two = EPI == FS2_EPI_ACT && a.out_pre != nullptr;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_pre, 0, two ? a.Mc * a.ldpre * ES : 0, 0x00020000);
  const int rr = lane / SG::LPR, cc = lane % SG::LPR;  // this lane's (row in a pass, 16-byte chunk) on the way out
  const int ncol = n0 + wn * (BN / 2) + cc * (16 / ES);
  char* const wr_base = stg + l31 * SG::RS + 4 * h * ES;
  const char* const rd_base = stg + rr * SG::RS + cc * 16;

  // everything staged so far goes out: rows 32 i .. + 31 of the wavefront's sub-tile into tensor `r` (leading dim ld)
  auto flush = [&](__amdgpu_buffer_rsrc_t r, int ld, int i) {
#pragma unroll
    for (int ps = 0; ps < 32 / SG::RPP; ++ps) {
      const int row = ps * SG::RPP + rr;
      const int m = m0 + wm * (BM / 2) + 32 * i + row;
      const u32x4 w = *reinterpret_cast<const u32x4*>(rd_base + ps * SG::RPP * SG::RS);
      const bool ok = m < a.Mc && ncol + 16 / ES <= a.Nc;
      __builtin_amdgcn_raw_buffer_store_b128(w, r, ok ? (m * ld + ncol) * ES : B_OOB, 0, 0);
      if (OBF && !ok && m < a.Mc && ncol + 4 <= a.Nc) {  // a bf16 row that ends on half a chunk (Nc is a multiple of 4)
        const u32x2 w2 = {w[0], w[1]};
        __builtin_amdgcn_raw_buffer_store_b64(w2, r, (m * ld + ncol) * ES, 0, 0);
      }
    }
  };
  auto put = [&](int j, int t, const float (&v)[4]) {
    char* const ad = wr_base + (32 * j + 8 * t) * ES;
    if (OBF) {
      const u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      *reinterpret_cast<u32x2*>(ad) = w;
    } else {
      const f32x4 w = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(ad) = w;
    }
  };

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + 32 * i + l31;
    const bool rowok = m < a.Mc;
    float v[TN][4][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = n0 + wn * (BN / 2) + 32 * j + 8 * t + 4 * h;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][t][e] = acc[i][j][4 * t + e];
        if (EPI >= 0) {
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (a.bias && n < a.Nc) b = *reinterpret_cast<const f32x4*>(a.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][t][e] = a.alpha * v[j][t][e] + b[e];
        }
        if (two) put(j, t, v[j][t]);
      }
    if (two) flush(rp, a.ldpre, i);  // (LDS serves a wavefront's accesses in order: the next puts land behind these reads)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = n0 + wn * (BN / 2) + 32 * j + 8 * t + 4 * h;
        const bool ok = rowok && n < a.Nc;
        float(&q)[4] = v[j][t];
        if (EPI == FS2_EPI_ACT) {
          if (OBF && two) {  // the activation sees what the backward pass will read back: the rounded pre-activation
            const unsigned w0 = pack_bf16x2(q[0], q[1]), w1 = pack_bf16x2(q[2], q[3]);
            q[0] = bf16_lo(w0); q[1] = bf16_hi(w0); q[2] = bf16_lo(w1); q[3] = bf16_hi(w1);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = act_b<ACT>(a.act, q[e]);
        } else if (EPI == FS2_EPI_DACT) {
          float x[4];
          if (aux_bf) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.Mc * a.ldaux * 2, 0x00020000);
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rx, ok ? (m * a.ldaux + n) * 2 : B_OOB, 0, 0);
            x[0] = bf16_lo(w[0]); x[1] = bf16_hi(w[0]); x[2] = bf16_lo(w[1]); x[3] = bf16_hi(w[1]);
          } else {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.Mc * a.ldaux * 4, 0x00020000);
            const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (m * a.ldaux + n) * 4 : B_OOB, 0, 0));
            x[0] = w[0]; x[1] = w[1]; x[2] = w[2]; x[3] = w[3];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] *= dact_b<ACT>(a.act, x[e]);
        }
        if (EPI > 0 && drop.on) {
          const unsigned idx = (unsigned)(m * ldc + n);
          if ((ldc & 1) == 0) {  // n is a multiple of 4: the quad is two whole hash pairs
            float f[4];
            fs2_drop_quad(drop, idx, f);
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] *= f[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] *= fs2_drop_factor(drop, (unsigned long long)idx + e);
          }
        }
        if (EPI == FS2_EPI_RESID) {
          const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0, a.Mc * a.ldr * 4, 0x00020000);
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (m * a.ldr + n) * 4 : B_OOB, 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = x[e] + a.res_scale * q[e];
        }
        put(j, t, q);
      }
    flush(rc, ldc, i);
  }
}

// stg: this wavefront's LDS staging region for whole-row stores (nullptr: the direct form), used when the launcher
// found every output row 16-byte aligned (GemmP.staged)
// CAN_STAGE: the caller has a staging region at all (the persistent kernels do not: their ring is in use)
template <int BM, int BN, bool CAN_STAGE>
__device__ __forceinline__ void epilogue_dispatch_b(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0,
                                                    int wm, int wn, int lane, int split, int tapz, char* stg) {
  const Fs2GemmArgs& a = p.a;
  const bool staged = CAN_STAGE && p.staged;
  if (a.splitk > 1) {
    float* slab = a.workspace + ((long long)split * a.taps + tapz) * ((long long)a.Mc * a.Nc);
    if constexpr (CAN_STAGE) {
      if (staged) {
        epilogue_staged_b<BM, BN, -1, false>(p, acc, slab, a.Nc, m0, n0, wm, wn, lane, stg);
        return;
      }
    }
    epilogue_b<BM, BN, -1, false>(p, acc, slab, a.Nc, m0, n0, wm, wn, lane);
    return;
  }
  const bool obf = (a.io_bf16 & 1) != 0;
  char* C = (char*)a.C;
  if (a.shift_operand == 1) C += (long long)tapz * a.c_tap_stride * (obf ? 2 : 4);
#define FS2_BEPI(E, A)                                                                                     \
  {                                                                                                        \
    bool done = false;                                                                                     \
    if constexpr (CAN_STAGE) {                                                                             \
      if (staged) {                                                                                        \
        if (obf) epilogue_staged_b<BM, BN, E, true, A>(p, acc, C, a.ldc, m0, n0, wm, wn, lane, stg);       \
        else epilogue_staged_b<BM, BN, E, false, A>(p, acc, C, a.ldc, m0, n0, wm, wn, lane, stg);          \
        done = true;                                                                                       \
      }                                                                                                    \
    }                                                                                                      \
    if (!done) {                                                                                           \
      if (obf) epilogue_b<BM, BN, E, true, A>(p, acc, C, a.ldc, m0, n0, wm, wn, lane);                     \
      else epilogue_b<BM, BN, E, false, A>(p, acc, C, a.ldc, m0, n0, wm, wn, lane);                        \
    }                                                                                                      \
  }
  // SiLU (the Conformer feed-forward modules) has its own instances; the other activations read the code at run time
  switch (a.epi) {
    case FS2_EPI_ACT:
      if (a.act == FS2_ACT_SILU) FS2_BEPI(FS2_EPI_ACT, FS2_ACT_SILU)
      else FS2_BEPI(FS2_EPI_ACT, -1)
      break;
    case FS2_EPI_RESID: FS2_BEPI(FS2_EPI_RESID, -1) break;
    case FS2_EPI_DACT:
      if (a.act == FS2_ACT_SILU) FS2_BEPI(FS2_EPI_DACT, FS2_ACT_SILU)
      else FS2_BEPI(FS2_EPI_DACT, -1)
      break;
    default: FS2_BEPI(0, -1) break;
  }
#undef FS2_BEPI
}

// ---- work units of the persistent kernels (gemm_bf16p.hip, gemm_bf16q.hip): one output tile of one split-K / tap slice ----
struct UnitB {
  int m0, n0, tapz, split, r_begin, r_end, nkt, shift_z, tile_n;
};
__device__ __forceinline__ UnitB decode_unit_b(const GemmP& p, int u, int nunits, int tiles, int BM, int BN) {
  const Fs2GemmArgs& a = p.a;
  UnitB q;
  const int uu = fs2_xcd_remap(u, nunits);
  const int z = uu / tiles, t = uu - z * tiles;
  const int ntap = a.shift_operand == 1 ? a.taps : 1;  // slice order (split, tap)
  q.split = z / ntap;
  q.tapz = z - q.split * ntap;
  q.r_begin = q.split * p.r_chunk;
  q.r_end = min(a.R, q.r_begin + p.r_chunk);
  const int tile_m = t / p.tiles_n;
  q.tile_n = t - tile_m * p.tiles_n;
  q.m0 = tile_m * BM;
  q.n0 = q.tile_n * BN;
  q.nkt = q.r_end > q.r_begin ? (q.r_end - q.r_begin + BKE - 1) / BKE : 0;
  q.shift_z = q.tapz * a.tap_mul + a.tap_add;
  return q;
}

}  // namespace
