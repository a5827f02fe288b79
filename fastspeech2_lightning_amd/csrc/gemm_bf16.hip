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
template <int BM, int BN, int EPI, bool OBF>
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
          for (int e = 0; e < 4; ++e) v[e] = fs2_act(a.act, v[e]);
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
          for (int e = 0; e < 4; ++e) v[e] *= fs2_dact(a.act, x[e]);
        }
        if (EPI > 0 && drop.on) {  // element index m * ldc + n, as everywhere else this mask is used
          const unsigned long long idx = (unsigned long long)(unsigned)(m * ldc + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= fs2_drop_factor(drop, idx + e);
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

template <int BM, int BN>
__device__ __forceinline__ void epilogue_dispatch_b(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0,
                                                    int wm, int wn, int lane, int split, int tapz) {
  const Fs2GemmArgs& a = p.a;
  if (a.splitk > 1) {
    float* slab = a.workspace + ((long long)split * a.taps + tapz) * ((long long)a.Mc * a.Nc);
    epilogue_b<BM, BN, -1, false>(p, acc, slab, a.Nc, m0, n0, wm, wn, lane);
    return;
  }
  const bool obf = (a.io_bf16 & 1) != 0;
  char* C = (char*)a.C;
  if (a.shift_operand == 1) C += (long long)tapz * a.c_tap_stride * (obf ? 2 : 4);
#define FS2_BEPI(E)                                                                   \
  if (obf) epilogue_b<BM, BN, E, true>(p, acc, C, a.ldc, m0, n0, wm, wn, lane);      \
  else epilogue_b<BM, BN, E, false>(p, acc, C, a.ldc, m0, n0, wm, wn, lane);
  switch (a.epi) {
    case FS2_EPI_ACT: FS2_BEPI(FS2_EPI_ACT) break;
    case FS2_EPI_RESID: FS2_BEPI(FS2_EPI_RESID) break;
    case FS2_EPI_DACT: FS2_BEPI(FS2_EPI_DACT) break;
    default: FS2_BEPI(0) break;
  }
#undef FS2_BEPI
}

template <int BM, int BN, bool AKC, bool BKC, int TAPS, int NST, bool COLSUM>
__global__ __launch_bounds__(256) void gemmb_kernel(GemmP p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int L = BM / 32 + BN / 32;  // LDS-DMA instructions per thread per K-tile
  static_assert(NST >= 2 && NST <= 4 && (NST - 2) * L < 64, "ring depth");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntile = p.tiles_m * p.tiles_n;
  const int unit = fs2_xcd_remap(blockIdx.x, gridDim.x);
  const int z = unit / ntile;
  const int wg = unit - z * ntile;
  const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int tapz = 0, split = 0;
  if (a.shift_operand == 1 || a.splitk > 1) {  // slice order (split, tap)
    const int ntap = a.shift_operand == 1 ? a.taps : 1;
    split = z / ntap;
    tapz = z - split * ntap;
  }
  const int r_begin = split * p.r_chunk;
  const int r_end = min(a.R, r_begin + p.r_chunk);
  // (the chunk is rounded up to whole K-tiles, so the last slices of a split can be empty: they write zero slabs)
  const int nkt = r_end > r_begin ? (r_end - r_begin + BKE - 1) / BKE : 0;
  const int shift_z = tapz * a.tap_mul + a.tap_add;

  f32x16 acc[TM][TN];
  f32x16 cs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // the bias gradient comes from the workgroups of the first tile column (and, for a convolution, of the centre-most
  // tap only: every tap slice sees the same A rows)
  const bool do_cs = COLSUM && a.colsum != nullptr && tile_n == 0 && tapz == 0;

  PiecesB<BM> pa;
  PiecesB<BN> pb;
  StreamB<AKC, BKC, TAPS> st;
  setup_pieces_b<BM, AKC, true, TAPS>(pa, p, m0, r_begin, tid);
  setup_pieces_b<BN, BKC, false, TAPS>(pb, p, n0, r_begin, tid);
  st.begin(p, r_begin, r_end, shift_z);
  auto issue = [&](int stage) {
    char* At = lds + stage * STAGE;
    st.template issue<BM, BN>(p, At, At + A_BYTES, pa, pb, wave, tid);
  };

  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  RdB<BM, AKC> rda;
  RdB<BN, BKC> rdb;
  rda.setup(wm * (BM / 2), lane);
  rdb.setup(wn * (BN / 2), lane);

  for (int kt = 0; kt < NST - 1 && kt < nkt; ++kt) issue(kt);
  int stage = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int newer = min(NST - 2, nkt - 1 - kt);
    if (NST >= 4 && newer == 2) b_wait_vmcnt_barrier<2 * L>();
    else if (NST >= 3 && newer == 1) b_wait_vmcnt_barrier<L>();
    else b_wait_vmcnt_barrier<0>();
    if (kt + NST - 1 < nkt) issue(stage == 0 ? NST - 1 : stage - 1);  // the stage read in iteration kt - 1
    const unsigned sa = lds0 + stage * STAGE, sb = sa + A_BYTES;
    compute_ktile_b<BM, BN, AKC, BKC, COLSUM>(acc, cs, rda, rdb, sa, sb, do_cs);
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  epilogue_dispatch_b<BM, BN>(p, acc, m0, n0, wm, wn, lane, split, tapz);
  if (COLSUM && do_cs && wn == 0 && lane < 32) {  // every row of cs[i] is the same sum: take the lane's first register
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * (BM / 2) + 32 * i + lane;
      if (m < a.Mc) a.colsum[(long long)split * a.Mc + m] = cs[i][0];
    }
  }
}

// ---- persistent variant -----------------------------------------------------------------------------------------------
// The GEMMs of this model are short (K = 256 .. 1024: 4 .. 16 K-tiles) and, at the bf16 MFMA rate, bound by the
// memory side: with one tile per workgroup every tile pays the DMA latency of its first K-tiles and drains before the
// next workgroup starts.  Here a launch fills every workgroup slot once and a workgroup walks the units u = blockIdx,
// blockIdx + grid, ...; the K-tiles of all its units form ONE stream through the two LDS stages -- the first K-tiles of
// the next unit are in flight under the last MFMAs and the whole epilogue of the current one (gemm2p.hip, same scheme).
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

template <int BM, int BN, bool AKC, bool BKC, int TAPS, bool COLSUM>
__global__ __launch_bounds__(256) void gemmbp_kernel(GemmP p, int nunits, int tiles) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;

  f32x16 acc[TM][TN];
  f32x16 cs[TM];

  // ---- producer: the next K-tile of this workgroup's stream (units with an empty reduction slice are skipped) --------
  int u_p = blockIdx.x, nkt_p = 0;
  PiecesB<BM> pa;
  PiecesB<BN> pb;
  StreamB<AKC, BKC, TAPS> st;
  auto enter_unit = [&]() {
    while (u_p < nunits) {
      const UnitB up = decode_unit_b(p, u_p, nunits, tiles, BM, BN);
      if (up.nkt > 0) {
        setup_pieces_b<BM, AKC, true, TAPS>(pa, p, up.m0, up.r_begin, tid);
        setup_pieces_b<BN, BKC, false, TAPS>(pb, p, up.n0, up.r_begin, tid);
        st.begin(p, up.r_begin, up.r_end, up.shift_z);
        nkt_p = up.nkt;
        return;
      }
      u_p += G;
    }
  };
  auto produce = [&](int stage) {
    if (u_p >= nunits) return;
    char* At = lds + stage * STAGE;
    st.template issue<BM, BN>(p, At, At + A_BYTES, pa, pb, wave, tid);
    if (st.kt == nkt_p) {
      u_p += G;
      enter_unit();
    }
  };

  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  RdB<BM, AKC> rda;
  RdB<BN, BKC> rdb;
  rda.setup(wm * (BM / 2), lane);
  rdb.setup(wn * (BN / 2), lane);

  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[i][r] = 0.f;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
  };

  enter_unit();
  produce(0);
  int stage = 0;
  b_wait_vmcnt_barrier<0>();  // K-tile 0 of the first unit has landed for everybody
  produce(1);
  for (int u_c = blockIdx.x; u_c < nunits; u_c += G) {
    const UnitB uc = decode_unit_b(p, u_c, nunits, tiles, BM, BN);
    clear();
    const bool do_cs = COLSUM && a.colsum != nullptr && uc.tile_n == 0 && uc.tapz == 0;
    if (uc.nkt > 0) {
      compute_ktile_b<BM, BN, AKC, BKC, COLSUM>(acc, cs, rda, rdb, lds0 + stage * STAGE, lds0 + stage * STAGE + A_BYTES, do_cs);
      stage ^= 1;
      for (int kt = 1; kt < uc.nkt; ++kt) {
        b_wait_vmcnt_barrier<0>();  // this K-tile has landed for everybody; the other stage is no longer read
        produce(stage ^ 1);
        compute_ktile_b<BM, BN, AKC, BKC, COLSUM>(acc, cs, rda, rdb, lds0 + stage * STAGE, lds0 + stage * STAGE + A_BYTES, do_cs);
        stage ^= 1;
      }
      // first K-tile of the next unit: its sync and the following DMA go ahead of the epilogue, whose stores then
      // drain under that K-tile's MFMAs (uniform: every wave of the workgroup sees the same stream state)
      b_wait_vmcnt_barrier<0>();
      produce(stage ^ 1);
    }
    epilogue_dispatch_b<BM, BN>(p, acc, uc.m0, uc.n0, wm, wn, lane, uc.split, uc.tapz);
    if (COLSUM && do_cs && wn == 0 && lane < 32) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = uc.m0 + wm * (BM / 2) + 32 * i + lane;
        if (m < a.Mc) a.colsum[(long long)uc.split * a.Mc + m] = cs[i][0];
      }
    }
  }
}

template <int BM, int BN, int WG_PER_CU>
int launch_bp(GemmP& p, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return FS2HIP_EINVAL;
    n_cu = prop.multiProcessorCount;
  }
  p.tiles_m = (a.Mc + BM - 1) / BM;
  p.tiles_n = (a.Nc + BN - 1) / BN;
  const int tiles = p.tiles_m * p.tiles_n;
  const long long nunits_ll = (long long)tiles * nz;
  if (nunits_ll > 0x7fffffffLL) return FS2HIP_EINVAL;
  const int nunits = (int)nunits_ll;
  const int slots = n_cu * WG_PER_CU;
  dim3 grid(nunits < slots ? nunits : slots), block(256);
  int mode = BT_NONE;
  if (a.taps > 1) {
    if (a.shift_operand == 0) {
      if (p.Rper % BKE) return FS2HIP_EINVAL;
      mode = BT_RED;
    } else {
      if (a.T < BKE) return FS2HIP_EINVAL;
      mode = BT_ROWS;
    }
  }
  if (a.a_kcontig && a.b_kcontig) {
    if (mode == BT_RED) gemmbp_kernel<BM, BN, true, true, BT_RED, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else if (mode == BT_NONE) gemmbp_kernel<BM, BN, true, true, BT_NONE, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else return FS2HIP_EINVAL;
  } else if (a.a_kcontig && !a.b_kcontig) {
    if (mode == BT_RED) gemmbp_kernel<BM, BN, true, false, BT_RED, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else if (mode == BT_NONE) gemmbp_kernel<BM, BN, true, false, BT_NONE, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else return FS2HIP_EINVAL;
  } else if (!a.a_kcontig && !a.b_kcontig) {
    if (mode == BT_ROWS) gemmbp_kernel<BM, BN, false, false, BT_ROWS, true><<<grid, block, 0, s>>>(p, nunits, tiles);
    else if (mode == BT_NONE) gemmbp_kernel<BM, BN, false, false, BT_NONE, true><<<grid, block, 0, s>>>(p, nunits, tiles);
    else return FS2HIP_EINVAL;
  } else {
    return FS2HIP_EINVAL;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

template <int BM, int BN, int NST>
int launch_b(GemmP& p, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  p.tiles_m = (a.Mc + BM - 1) / BM;
  p.tiles_n = (a.Nc + BN - 1) / BN;
  if ((long long)p.tiles_m * p.tiles_n * nz > 0x7fffffffLL) return FS2HIP_EINVAL;
  dim3 grid(p.tiles_m * p.tiles_n * nz), block(256);
  int mode = BT_NONE;
  if (a.taps > 1) {
    if (a.shift_operand == 0) {
      if (p.Rper % BKE) return FS2HIP_EINVAL;
      mode = BT_RED;
    } else {
      if (a.T < BKE) return FS2HIP_EINVAL;
      mode = BT_ROWS;
    }
  }
  if (a.a_kcontig && a.b_kcontig) {  // forward
    if (mode == BT_RED) gemmb_kernel<BM, BN, true, true, BT_RED, NST, false><<<grid, block, 0, s>>>(p);
    else if (mode == BT_NONE) gemmb_kernel<BM, BN, true, true, BT_NONE, NST, false><<<grid, block, 0, s>>>(p);
    else return FS2HIP_EINVAL;
  } else if (a.a_kcontig && !a.b_kcontig) {  // data gradient: the weight as it is stored, read by rows of the reduction
    if (mode == BT_RED) gemmb_kernel<BM, BN, true, false, BT_RED, NST, false><<<grid, block, 0, s>>>(p);
    else if (mode == BT_NONE) gemmb_kernel<BM, BN, true, false, BT_NONE, NST, false><<<grid, block, 0, s>>>(p);
    else return FS2HIP_EINVAL;
  } else if (!a.a_kcontig && !a.b_kcontig) {  // weight gradient (+ bias gradient)
    if (mode == BT_ROWS) gemmb_kernel<BM, BN, false, false, BT_ROWS, NST, true><<<grid, block, 0, s>>>(p);
    else if (mode == BT_NONE) gemmb_kernel<BM, BN, false, false, BT_NONE, NST, true><<<grid, block, 0, s>>>(p);
    else return FS2HIP_EINVAL;
  } else {
    return FS2HIP_EINVAL;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// tile ids of the bf16-storage core: 20 = 128x128 (2 stages, 2 workgroups / CU), 21 = 128x128 (3 stages),
// 22 = 128x64 (2 stages, 3 workgroups / CU), 23 = 64x64 (3 stages, 3 workgroups / CU); persistent (one K-tile stream
// per workgroup across its tiles): 24 = 128x128 (2 / CU), 25 = 128x64 (3 / CU), 26 = 64x64 (4 / CU)
int fs2_gemmb_launch(GemmP& p, int tile, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  const int chunk = (a.R + a.splitk - 1) / a.splitk;
  p.r_chunk = ((chunk + BKE - 1) / BKE) * BKE;
  switch (tile) {
    case 20: return launch_b<128, 128, 2>(p, nz, s);
    case 21: return launch_b<128, 128, 3>(p, nz, s);
    case 22: return launch_b<128, 64, 2>(p, nz, s);
    case 23: return launch_b<64, 64, 3>(p, nz, s);
    case 24: return launch_bp<128, 128, 2>(p, nz, s);
    case 25: return launch_bp<128, 64, 3>(p, nz, s);
    case 26: return launch_bp<64, 64, 4>(p, nz, s);
    default: return FS2HIP_EINVAL;
  }
}
