// Helpers shared by the direct-to-LDS GEMM kernels (gemm2.hip: one tile per workgroup; gemm2p.hip: persistent).
#pragma once
#include "gemm_common.h"

namespace {


constexpr int BK2 = 32;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __attribute__((aligned(16))) float fs2_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

// One K-tile of one operand -> LDS.  ROWS = tile extent along the operand's non-reduction dimension.
//   KC  : operand stored [row][k]  (image [ROWS][32], chunk-swizzled)   else [k][row] (image [32][ROWS])
//   IS_A: A operand (conv-tap row shift in NT/NN mode)                   else B (tap -> weight slice)
template <int ROWS, bool KC, bool IS_A>
__device__ __forceinline__ void issue_tile(float* __restrict__ tile, const GemmP& p, int row0, int r0, int r_end,
                                           int shift_z, int tid, int wave) {
  const Fs2GemmArgs& a = p.a;
  const float* src = IS_A ? a.A : a.B;
  const int ld = IS_A ? a.lda : a.ldb;
  const int nrows = IS_A ? a.Mc : a.Nc;
  const bool taps0 = a.taps > 1 && a.shift_operand == 0;  // reduction runs over (tap, k)
  const bool taps1 = a.taps > 1 && a.shift_operand == 1;  // weight gradient: reduction rows of B are shifted
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const int pidx = it * 256 + tid;
    const float* ptr;
    bool ok;
    if (KC) {
      const int row = pidx >> 3, pc = pidx & 7;
      const int kg = r0 + ((pc ^ ((row >> 1) & 7)) << 2);
      const int gr = row0 + row;
      ok = gr < nrows && kg < r_end;
      if (taps0) {
        const int tap = kg / p.Rper, kin = kg - tap * p.Rper;
        if (IS_A) {
          const int shift = tap * a.tap_mul + a.tap_add;
          const int t = gr % a.T + shift;
          ok = ok && t >= 0 && t < a.T;
          ptr = src + (long long)(gr + shift) * ld + kin;
        } else {
          ptr = src + (long long)tap * a.b_tap_stride + (long long)gr * ld + kin;
        }
      } else {
        ptr = src + (long long)gr * ld + kg;
      }
    } else {
      const int k = pidx / (ROWS / 4), r4 = pidx % (ROWS / 4);
      const int kg = r0 + k, col = row0 + r4 * 4;
      ok = kg < r_end && col < nrows;
      if (taps0 && !IS_A) {  // NN conv backward-data: B = W[tap] stored [Rper][Nc]
        const int tap = kg / p.Rper, kin = kg - tap * p.Rper;
        ptr = src + (long long)tap * a.b_tap_stride + (long long)kin * ld + col;
      } else if (taps1 && !IS_A) {  // TN conv weight gradient: x rows shifted by the tap of this launch slice
        const int t = kg % a.T + shift_z;
        ok = ok && t >= 0 && t < a.T;
        ptr = src + (long long)(kg + shift_z) * ld + col;
      } else {
        ptr = src + (long long)kg * ld + col;
      }
    }
    glds16(ok ? ptr : fs2_zero_page, tile + (it * 256 + wave * 64) * 4);
  }
}

// ---- hoisted addressing: buffer loads to LDS ------------------------------------------------------------------
// All modes but TAPS_GENERIC address an operand through a raw buffer resource (base pointer in 4 SGPRs) plus
//   * a loop-invariant per-piece byte offset in ONE VGPR (the piece's row / column inside the operand; a piece
//     whose row or column lies outside the matrix carries the out-of-range sentinel instead), and
//   * a per-K-tile scalar byte offset (the reduction advance, the conv tap's weight slice and row shift).
// `buffer_load_dwordx4 ... lds` writes ZEROS to LDS for a lane whose offset is out of range (checked on this
// chip: tools/scratch/buflds_test.hip; voffset + soffset is what is compared with num_records), so matrix
// edges, conv 'same' padding and reduction tails need no zero page and no 64-bit address arithmetic: a
// K-tile's DMA issue is VALU-free in the plain case and one compare + select per piece at an edge.
// Operand extents must stay below 2 GiB (checked by the launcher).
constexpr int FS2_OOB = (int)0x80000000;  // with num_records = 2^31: offset + anything >= num_records

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, FS2_OOB, 0x00020000);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, int voff, int soff, float* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int ROWS>
struct Pieces {
  int voff[ROWS / 32];  // byte offset of the piece in K-tile 0 (FS2_OOB: row / column outside the matrix)
  int t[ROWS / 32];     // conv taps: time index of the piece's row (A rows, TAPS_RED) or of its reduction row in
                        // the current K-tile (B rows, TAPS_ROWS)
};

// kernel variants by conv-tap mode
constexpr int TAPS_NONE = 0;     // plain GEMM
constexpr int TAPS_RED = 1;      // shift_operand == 0, Rper % 32 == 0: every K-tile lies inside one tap -> the tap
                                 // (row shift of A, weight slice of B) is a per-tile scalar
constexpr int TAPS_ROWS = 2;     // shift_operand == 1 (weight gradient): reduction rows of B shifted by the
                                 // tap of this launch slice; T >= 32
constexpr int TAPS_GENERIC = 3;  // any Rper / T: per-piece address decode in every K-tile (slow; small convs)

// reduction offset of piece `it` inside a K-tile (k-contiguous operands: the swizzled 16-byte chunk)
template <int ROWS, bool KC>
__device__ __forceinline__ int piece_koff(int it, int tid) {
  const int pidx = it * 256 + tid;
  if (KC) return (((pidx & 7) ^ (((pidx >> 3) >> 1) & 7)) << 2);
  return pidx / (ROWS / 4);
}

template <int ROWS, bool KC, bool IS_A, int TAPS>
__device__ __forceinline__ void setup_pieces(Pieces<ROWS>& pc_, const GemmP& p, int row0, int r_begin, int tid) {
  const Fs2GemmArgs& a = p.a;
  const int ld = IS_A ? a.lda : a.ldb;
  const int nrows = IS_A ? a.Mc : a.Nc;
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const int pidx = it * 256 + tid;
    pc_.t[it] = 0;
    if (KC) {
      const int row = pidx >> 3, gr = row0 + row;
      pc_.voff[it] = gr < nrows ? (gr * ld + piece_koff<ROWS, KC>(it, tid)) * 4 : FS2_OOB;
      if (TAPS == TAPS_RED && IS_A) pc_.t[it] = gr % a.T;
    } else {
      const int k = pidx / (ROWS / 4), col = row0 + (pidx % (ROWS / 4)) * 4;
      pc_.voff[it] = col < nrows ? (k * ld + col) * 4 : FS2_OOB;
      if (TAPS == TAPS_ROWS && !IS_A) pc_.t[it] = (r_begin + k) % a.T;
    }
  }
}

// soff = byte offset of this K-tile (scalar); rem = reduction elements left from the start of this K-tile
template <int ROWS, bool KC>
__device__ __forceinline__ void issue_fast(float* __restrict__ tile, __amdgpu_buffer_rsrc_t r, const Pieces<ROWS>& pc_,
                                           int soff, int rem, int wave, int tid) {
  if (rem >= BK2) {
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it) blds16(r, pc_.voff[it], soff, tile + (it * 256 + wave * 64) * 4);
  } else {  // the reduction ends inside this K-tile
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it)
      blds16(r, piece_koff<ROWS, KC>(it, tid) < rem ? pc_.voff[it] : FS2_OOB, soff, tile + (it * 256 + wave * 64) * 4);
  }
}

// TAPS_RED, A operand: rows shifted by the K-tile's tap (the shift is part of soff); a row whose shifted
// time index leaves [0, T) is the convolution's zero padding.
template <int ROWS>
__device__ __forceinline__ void issue_shifted_rows(float* __restrict__ tile, __amdgpu_buffer_rsrc_t r,
                                                   const Pieces<ROWS>& pc_, int soff, int shift, int T, int wave) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const bool ok = (unsigned)(pc_.t[it] + shift) < (unsigned)T;
    blds16(r, ok ? pc_.voff[it] : FS2_OOB, soff, tile + (it * 256 + wave * 64) * 4);
  }
}

// TAPS_ROWS, B operand: the reduction index is the (b, t) row itself (the slice's row shift is in the
// resource's base); advances the pieces' time index by one K-tile (T >= 32, so one conditional
// subtraction keeps it in [0, T)).
template <int ROWS>
__device__ __forceinline__ void issue_shifted_red(float* __restrict__ tile, __amdgpu_buffer_rsrc_t r, Pieces<ROWS>& pc_,
                                                  int soff, int rem, int shift, int T, int wave, int tid) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const bool ok = piece_koff<ROWS, false>(it, tid) < rem && (unsigned)(pc_.t[it] + shift) < (unsigned)T;
    blds16(r, ok ? pc_.voff[it] : FS2_OOB, soff, tile + (it * 256 + wave * 64) * 4);
    const int t = pc_.t[it] + BK2;
    pc_.t[it] = t >= T ? t - T : t;
  }
}

// Scalar side of the hoisted addressing: resources and per-K-tile byte offsets of both operands for one work
// unit (a tile of one split-K / conv-tap slice).  K-tiles are issued in order.
template <bool AKC, bool BKC, int TAPS>
struct Stream {
  __amdgpu_buffer_rsrc_t ra, rb;
  int r_begin, r_end, kt;  // reduction range of the unit, next K-tile to issue
  int tap, kin;            // TAPS_RED: tap and offset inside the tap of the next K-tile
  int shift_min, shift_z;

  __device__ __forceinline__ void begin(const GemmP& p, int r_begin_, int r_end_, int shift_z_) {
    const Fs2GemmArgs& a = p.a;
    r_begin = r_begin_;
    r_end = r_end_;
    shift_z = shift_z_;
    kt = 0;
    tap = kin = shift_min = 0;
    const float* A = a.A;
    const float* B = a.B;
    if (TAPS == TAPS_RED) {
      tap = r_begin / p.Rper;
      kin = r_begin - tap * p.Rper;
      // most negative row shift over the taps: folded into A's base so that the scalar offset stays >= 0
      shift_min = a.tap_add + (a.tap_mul < 0 ? a.tap_mul * (a.taps - 1) : 0);
      A += (long long)shift_min * a.lda;
    } else if (TAPS == TAPS_ROWS) {
      B += (long long)shift_z * a.ldb;
    }
    ra = make_rsrc(A);
    rb = make_rsrc(B);
  }

  template <int BM, int BN>
  __device__ __forceinline__ void issue(const GemmP& p, float* At, float* Bt, Pieces<BM>& pa, Pieces<BN>& pb, int wave,
                                        int tid) {
    const Fs2GemmArgs& a = p.a;
    const int r0 = r_begin + kt * BK2, rem = r_end - r0;
    if constexpr (TAPS == TAPS_RED) {
      const int shift = tap * a.tap_mul + a.tap_add;
      issue_shifted_rows<BM>(At, ra, pa, ((shift - shift_min) * a.lda + kin) * 4, shift, a.T, wave);
      issue_fast<BN, BKC>(Bt, rb, pb, (int)(((long long)tap * a.b_tap_stride + (BKC ? kin : kin * a.ldb)) * 4), BK2, wave, tid);
      kin += BK2;
      if (kin == p.Rper) {
        kin = 0;
        ++tap;
      }
    } else if constexpr (TAPS == TAPS_ROWS) {
      issue_fast<BM, AKC>(At, ra, pa, (AKC ? r0 : r0 * a.lda) * 4, rem, wave, tid);
      issue_shifted_red<BN>(Bt, rb, pb, r0 * a.ldb * 4, rem, shift_z, a.T, wave, tid);
    } else {
      issue_fast<BM, AKC>(At, ra, pa, (AKC ? r0 : r0 * a.lda) * 4, rem, wave, tid);
      issue_fast<BN, BKC>(Bt, rb, pb, (BKC ? r0 : r0 * a.ldb) * 4, rem, wave, tid);
    }
    ++kt;
  }
};

// every byte offset the hoisted addressing can form must stay below 2^31
inline bool fs2_gemm2_offsets_fit(const Fs2GemmArgs& a) {
  const long long lim = 0x7fffffffLL - 4096;
  const long long a_rows = a.a_kcontig ? a.Mc : a.R, b_rows = a.b_kcontig ? a.Nc : a.R;
  long long ea = 4LL * (a_rows + 2LL * a.taps) * a.lda + 4LL * a.R;
  long long eb = 4LL * (b_rows + 2LL * a.taps) * a.ldb + 4LL * a.R + 4LL * a.taps * (a.b_tap_stride > 0 ? a.b_tap_stride : 0);
  return ea < lim && eb < lim;
}

// wait until at most N of this wave's vector-memory operations (here: LDS-DMA pieces) are outstanding, then
// the workgroup barrier.  One asm statement with a memory clobber: the compiler tracks neither the DMA's
// LDS writes nor the counter, so no LDS access may move across it.
template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// MFMA operand fetch from LDS in inline assembly.  A compiler-visible LDS load after an LDS-DMA makes the
// waitcnt pass insert `s_waitcnt vmcnt(0)` in front of it (it cannot tell which DMA the read depends on),
// which drains the tile that was just put in flight and serialises DMA and MFMA inside a wave.  With the
// reads in asm the only vmcnt waits are the counted ones above; the price is that the lgkmcnt bookkeeping
// is ours as well: `lds_wait<N>()` + `pin()` on every register the following MFMAs consume.
template <int OFF>
__device__ __forceinline__ void lds_rd128(f32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_rd32(float& v, unsigned addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

// MFMA operands of one 8-deep reduction group (4 MFMA k-steps) for T row blocks of 32.
//   KC : LDS image [ROWS][32], chunk-swizzled: one ds_read_b128 per row block (lane half h: k = 8g+4h..+3)
//   !KC: LDS image [32][ROWS]: four conflict-free ds_read_b32 per row block
template <int T, bool KC>
struct Frag;
template <int T>
struct Frag<T, true> {
  f32x4 q[T];
  static constexpr int READS = T;
  __device__ __forceinline__ float get(int i, int j) const { return q[i][j]; }
  __device__ __forceinline__ void pin_all() {
#pragma unroll
    for (int i = 0; i < T; ++i) pin(q[i]);
  }
};
template <int T>
struct Frag<T, false> {
  float q[T][4];
  static constexpr int READS = 4 * T;
  __device__ __forceinline__ float get(int i, int j) const { return q[i][j]; }
  __device__ __forceinline__ void pin_all() {
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) pin(q[i][j]);
  }
};

// per-lane LDS byte addresses of the operand reads, relative to the operand tile of stage 0
template <int ROWS, bool KC>
struct RdAddr {
  unsigned g[KC ? 4 : 1];
  __device__ __forceinline__ void setup(int wrow0, int l31, int h) {
    if (KC) {
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) g[gg] = ((wrow0 + l31) * BK2 + (((2 * gg + h) ^ ((l31 >> 1) & 7)) << 2)) * 4;
    } else {
      g[0] = (4 * h * ROWS + wrow0 + l31) * 4;
    }
  }
};

template <int G, int ROWS, int T, bool KC>
__device__ __forceinline__ void frag_read(Frag<T, KC>& f, const RdAddr<ROWS, KC>& ra, unsigned stage_base) {
  if constexpr (KC) {
    lds_rd128<0>(f.q[0], ra.g[G] + stage_base);
    if constexpr (T > 1) lds_rd128<32 * BK2 * 4>(f.q[1], ra.g[G] + stage_base);
  } else {
    const unsigned ad = ra.g[0] + stage_base;
    lds_rd32<(8 * G + 0) * ROWS * 4>(f.q[0][0], ad);
    lds_rd32<(8 * G + 1) * ROWS * 4>(f.q[0][1], ad);
    lds_rd32<(8 * G + 2) * ROWS * 4>(f.q[0][2], ad);
    lds_rd32<(8 * G + 3) * ROWS * 4>(f.q[0][3], ad);
    if constexpr (T > 1) {
      lds_rd32<(8 * G + 0) * ROWS * 4 + 128>(f.q[1][0], ad);
      lds_rd32<(8 * G + 1) * ROWS * 4 + 128>(f.q[1][1], ad);
      lds_rd32<(8 * G + 2) * ROWS * 4 + 128>(f.q[1][2], ad);
      lds_rd32<(8 * G + 3) * ROWS * 4 + 128>(f.q[1][3], ad);
    }
  }
}


// Column sums of a reduction-major A operand (the bias gradient dY^T . 1 of the layer whose weight gradient the launch
// computes, Fs2GemmArgs.colsum): the lane already holds four reduction steps of its row per group -- four adds per row
// block and group beside sixteen MFMAs, in the wavefronts of the first tile column only.
template <int TM, bool KC>
__device__ __forceinline__ void colsum_group(float (&cs)[TM], const Frag<TM, KC>& f) {
#pragma unroll
  for (int i = 0; i < TM; ++i) cs[i] += (f.get(i, 0) + f.get(i, 1)) + (f.get(i, 2) + f.get(i, 3));
}

// MFMAs of one K-tile (BK2 = 32 deep) from the LDS stage at byte addresses sa (A image) / sb (B image).
// LDS reads run one reduction group (8 deep) ahead of the MFMAs that consume them (two register sets).
template <int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void compute_ktile(f32x16 (&acc)[BM / 64][BN / 64], const RdAddr<BM, AKC>& rda,
                                              const RdAddr<BN, BKC>& rdb, unsigned sa, unsigned sb,
                                              float (&cs)[BM / 64], bool do_cs) {
  constexpr int TM = BM / 64, TN = BN / 64;
    Frag<TM, AKC> fa[2];
    Frag<TN, BKC> fb[2];
    constexpr int RD = Frag<TM, AKC>::READS + Frag<TN, BKC>::READS;
    frag_read<0, BM>(fa[0], rda, sa);
    frag_read<0, BN>(fb[0], rdb, sb);
#define FS2_GROUP(G)                                                                                       \
  {                                                                                                        \
    if (G < 3) {                                                                                           \
      frag_read<(G + 1) & 3, BM>(fa[(G + 1) & 1], rda, sa);                                                \
      frag_read<(G + 1) & 3, BN>(fb[(G + 1) & 1], rdb, sb);                                                \
      lds_wait<RD>();                                                                                      \
    } else {                                                                                               \
      lds_wait<0>();                                                                                       \
    }                                                                                                      \
    fa[G & 1].pin_all();                                                                                   \
    fb[G & 1].pin_all();                                                                                   \
    if (!AKC && do_cs) colsum_group<TM, AKC>(cs, fa[G & 1]);                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                         \
    _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                      \
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[G & 1].get(i, j), fb[G & 1].get(jn, j), acc[i][jn], 0, 0, 0); \
  }
    FS2_GROUP(0)
    FS2_GROUP(1)
    FS2_GROUP(2)
    FS2_GROUP(3)
#undef FS2_GROUP
}

// "bf16-mixed" operands (Fs2GemmArgs.operand_bf16): the same fp32 LDS images and the same reads, but two
// reduction groups (lane half h: k = 8g+4h..+3 for g = 2P, 2P+1) are rounded to eight bf16 and go through ONE
// v_mfma_f32_32x32x16_bf16 (8 passes) instead of eight v_mfma_f32_32x32x2_f32 (16 passes each).  A and B use the
// same (g, h, j) -> k-slot map, so the k order inside the instruction is again irrelevant.  At this MFMA rate
// the K-tile is bound by its LDS traffic, so all reads of the tile are issued at once and waited for in halves.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

template <int T, bool KC>
__device__ __forceinline__ bf16x8 frag_bf16(const Frag<T, KC>& lo, const Frag<T, KC>& hi, int i) {
  const f32x8 v = {lo.get(i, 0), lo.get(i, 1), lo.get(i, 2), lo.get(i, 3), hi.get(i, 0), hi.get(i, 1), hi.get(i, 2), hi.get(i, 3)};
  return __builtin_convertvector(v, bf16x8);  // v_cvt_pk_bf16_f32, round to nearest even
}

template <int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void compute_ktile_bf16(f32x16 (&acc)[BM / 64][BN / 64], const RdAddr<BM, AKC>& rda,
                                                   const RdAddr<BN, BKC>& rdb, unsigned sa, unsigned sb,
                                                   float (&cs)[BM / 64], bool do_cs) {
  constexpr int TM = BM / 64, TN = BN / 64;
  Frag<TM, AKC> fa[4];
  Frag<TN, BKC> fb[4];
  constexpr int RD = Frag<TM, AKC>::READS + Frag<TN, BKC>::READS;
  frag_read<0, BM>(fa[0], rda, sa);
  frag_read<0, BN>(fb[0], rdb, sb);
  frag_read<1, BM>(fa[1], rda, sa);
  frag_read<1, BN>(fb[1], rdb, sb);
  frag_read<2, BM>(fa[2], rda, sa);
  frag_read<2, BN>(fb[2], rdb, sb);
  frag_read<3, BM>(fa[3], rda, sa);
  frag_read<3, BN>(fb[3], rdb, sb);
#define FS2_PAIR(P, WAIT)                                                                                   \
  {                                                                                                          \
    lds_wait<WAIT>();                                                                                        \
    fa[2 * P].pin_all();                                                                                     \
    fa[2 * P + 1].pin_all();                                                                                 \
    fb[2 * P].pin_all();                                                                                     \
    fb[2 * P + 1].pin_all();                                                                                 \
    if (!AKC && do_cs) {                                                                                     \
      colsum_group<TM, AKC>(cs, fa[2 * P]);                                                                  \
      colsum_group<TM, AKC>(cs, fa[2 * P + 1]);                                                              \
    }                                                                                                        \
    bf16x8 ua[TM], ub[TN];                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) ua[i] = frag_bf16(fa[2 * P], fa[2 * P + 1], i);           \
    _Pragma("unroll") for (int jn = 0; jn < TN; ++jn) ub[jn] = frag_bf16(fb[2 * P], fb[2 * P + 1], jn);      \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                           \
    _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                        \
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[i], ub[jn], acc[i][jn], 0, 0, 0);            \
  }
  FS2_PAIR(0, 2 * RD)
  FS2_PAIR(1, 0)
#undef FS2_PAIR
}

// "32-split" operands (Fs2GemmArgs.operand_bf16 == 2): fp32 accuracy on the bf16 matrix pipe.  Every operand value is
// cut -- exactly, by truncation -- into three bf16 planes x = x0 + x1 + x2 (8 + 8 + 8 = the 24 significant bits of an
// fp32), and a product a*b is taken as the six partial products a_i*b_j with i + j <= 2, each one EXACT in the
// MFMA's fp32 accumulation (8 x 8 significant bits), smallest first.  What is dropped (a1*b2, a2*b1, a2*b2) is below
// 2^-24 |a||b|, i.e. below the rounding of the fp32 product itself: the result meets the fp32 kernels' error bound
// (tests/test_gemm_split_gpu.py holds it to the same tolerance, against float64).  Cost: six v_mfma_f32_32x32x16_bf16
// (8 passes each) per sixteen reduction steps instead of eight v_mfma_f32_32x32x2_f32 (16 passes each) -- 48 pipe
// cycles instead of 128 per 32x32x16 block -- plus 5.5 vector instructions per operand value for the cut, which run
// beside the bf16 MFMAs (unlike fp32 MFMAs, they do not share the vector ALUs).  Same fp32 LDS images, same reads.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
  p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);  // {hi16(b), hi16(a)}: two truncated bf16
  const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u), rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
  const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
  p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
  const float sa = ra - __builtin_bit_cast(float, va & 0xffff0000u), sb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
  p2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
}

struct Planes {
  bf16x8 p[3];
};
template <int T, bool KC>
__device__ __forceinline__ Planes frag_split3(const Frag<T, KC>& lo, const Frag<T, KC>& hi, int i) {
  u32x4 q0, q1, q2;
  unsigned a0, a1, a2;
  split_pair(lo.get(i, 0), lo.get(i, 1), a0, a1, a2); q0[0] = a0; q1[0] = a1; q2[0] = a2;
  split_pair(lo.get(i, 2), lo.get(i, 3), a0, a1, a2); q0[1] = a0; q1[1] = a1; q2[1] = a2;
  split_pair(hi.get(i, 0), hi.get(i, 1), a0, a1, a2); q0[2] = a0; q1[2] = a1; q2[2] = a2;
  split_pair(hi.get(i, 2), hi.get(i, 3), a0, a1, a2); q0[3] = a0; q1[3] = a1; q2[3] = a2;
  Planes r;
  r.p[0] = __builtin_bit_cast(bf16x8, q0);
  r.p[1] = __builtin_bit_cast(bf16x8, q1);
  r.p[2] = __builtin_bit_cast(bf16x8, q2);
  return r;
}

template <int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void compute_ktile_split(f32x16 (&acc)[BM / 64][BN / 64], const RdAddr<BM, AKC>& rda,
                                                    const RdAddr<BN, BKC>& rdb, unsigned sa, unsigned sb,
                                                    float (&cs)[BM / 64], bool do_cs) {
  constexpr int TM = BM / 64, TN = BN / 64;
  Frag<TM, AKC> fa[4];
  Frag<TN, BKC> fb[4];
  constexpr int RD = Frag<TM, AKC>::READS + Frag<TN, BKC>::READS;
  frag_read<0, BM>(fa[0], rda, sa);
  frag_read<0, BN>(fb[0], rdb, sb);
  frag_read<1, BM>(fa[1], rda, sa);
  frag_read<1, BN>(fb[1], rdb, sb);
  frag_read<2, BM>(fa[2], rda, sa);
  frag_read<2, BN>(fb[2], rdb, sb);
  frag_read<3, BM>(fa[3], rda, sa);
  frag_read<3, BN>(fb[3], rdb, sb);
  lds_wait<0>();
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    fa[g].pin_all();
    fb[g].pin_all();
    if (!AKC && do_cs) colsum_group<TM, AKC>(cs, fa[g]);
  }
  // The cut of the second half of the K-tile is issued between the MFMAs of the first half (the bf16 matrix pipe
  // and the vector ALUs run side by side, but a wavefront issues in order: a block of 170 vector instructions in
  // front of 24 MFMAs leaves the pipe idle for its whole length).  7 vector instructions per 32-cycle MFMA.
  Planes ua[2][TM], ub[2][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) ua[0][i] = frag_split3(fa[0], fa[1], i);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) ub[0][jn] = frag_split3(fb[0], fb[1], jn);
#pragma unroll
  for (int i = 0; i < TM; ++i) ua[1][i] = frag_split3(fa[2], fa[3], i);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) ub[1][jn] = frag_split3(fb[2], fb[3], jn);
#pragma unroll
  for (int P = 0; P < 2; ++P)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        f32x16 c = acc[i][jn];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[P][i].p[2], ub[P][jn].p[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[P][i].p[0], ub[P][jn].p[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[P][i].p[1], ub[P][jn].p[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[P][i].p[1], ub[P][jn].p[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[P][i].p[0], ub[P][jn].p[1], c, 0, 0, 0);
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[P][i].p[0], ub[P][jn].p[0], c, 0, 0, 0);
      }
  // schedule: [cut of half 0] then 6 TM TN x {1 MFMA, 7 vector} (half 0's MFMAs over half 1's cut), then the rest
  constexpr int CUT = 44 * (TM + TN);  // vector instructions of one half's cut
  __builtin_amdgcn_sched_group_barrier(0x002, CUT, 0);
#pragma unroll
  for (int k = 0; k < 6 * TM * TN; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, (CUT + 6 * TM * TN - 1) / (6 * TM * TN), 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x008, 6 * TM * TN, 0);
}

// Operands that ARE bf16 in memory (Fs2GemmArgs.operand_bf16 == 3; both k-contiguous).  The loader is the fp32 one
// with the reduction counted in 4-byte slots: a row of a K-tile is still one 128-byte line, now 64 reduction steps
// deep, and the 16-byte chunk (2g + h) a lane half reads for "group g" is exactly the eight bf16 (k = 16g + 8h .. +7)
// v_mfma_f32_32x32x16_bf16 wants from it: four MFMAs per accumulator and K-tile, no conversion, half the HBM, L2 and
// LDS bytes per flop of the register-rounding mode.
template <int BM, int BN>
__device__ __forceinline__ void compute_ktile_bf16s(f32x16 (&acc)[BM / 64][BN / 64], const RdAddr<BM, true>& rda,
                                                    const RdAddr<BN, true>& rdb, unsigned sa, unsigned sb) {
  constexpr int TM = BM / 64, TN = BN / 64;
  Frag<TM, true> fa[4];
  Frag<TN, true> fb[4];
  constexpr int RD = TM + TN;
  frag_read<0, BM>(fa[0], rda, sa);
  frag_read<0, BN>(fb[0], rdb, sb);
  frag_read<1, BM>(fa[1], rda, sa);
  frag_read<1, BN>(fb[1], rdb, sb);
  frag_read<2, BM>(fa[2], rda, sa);
  frag_read<2, BN>(fb[2], rdb, sb);
  frag_read<3, BM>(fa[3], rda, sa);
  frag_read<3, BN>(fb[3], rdb, sb);
#define FS2_STEP(G)                                                                                              \
  {                                                                                                              \
    lds_wait<(3 - G) * RD>();                                                                                    \
    fa[G].pin_all();                                                                                             \
    fb[G].pin_all();                                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
    _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                                            \
        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[G].q[i]),             \
                                                             __builtin_bit_cast(bf16x8, fb[G].q[jn]), acc[i][jn], 0, 0, 0); \
  }
  FS2_STEP(0)
  FS2_STEP(1)
  FS2_STEP(2)
  FS2_STEP(3)
#undef FS2_STEP
}

// BF: 0 = fp32 MFMA, 1 = operands rounded to bf16 ("bf16-mixed"), 2 = three-plane split ("32-split"), 3 = bf16 in memory
template <int BF, int BM, int BN, bool AKC, bool BKC>
__device__ __forceinline__ void compute_ktile_any(f32x16 (&acc)[BM / 64][BN / 64], const RdAddr<BM, AKC>& rda,
                                                  const RdAddr<BN, BKC>& rdb, unsigned sa, unsigned sb,
                                                  float (&cs)[BM / 64], bool do_cs) {
  if constexpr (BF == 3) {
    static_assert(AKC && BKC, "bf16 storage: k-contiguous operands only");
    compute_ktile_bf16s<BM, BN>(acc, rda, rdb, sa, sb);
  } else if constexpr (BF == 2) compute_ktile_split<BM, BN, AKC, BKC>(acc, rda, rdb, sa, sb, cs, do_cs);
  else if constexpr (BF == 1) compute_ktile_bf16<BM, BN, AKC, BKC>(acc, rda, rdb, sa, sb, cs, do_cs);
  else compute_ktile<BM, BN, AKC, BKC>(acc, rda, rdb, sa, sb, cs, do_cs);
}

// after the epilogue of a unit: the two halves of a wavefront hold the sums over different reduction steps of the same
// rows; lanes 0..31 of the wavefronts of the first tile column write colsum[split][Mc]
template <int BM>
__device__ __forceinline__ void colsum_store(const Fs2GemmArgs& a, float (&cs)[BM / 64], int m0, int wm, int lane, int split) {
#pragma unroll
  for (int i = 0; i < BM / 64; ++i) {
    const float t = cs[i] + __shfl_xor(cs[i], 32, 64);
    const int m = m0 + wm * (BM / 2) + 32 * i + lane;
    if (lane < 32 && m < a.Mc) a.colsum[(long long)split * a.Mc + m] = t;
  }
}

}  // namespace
