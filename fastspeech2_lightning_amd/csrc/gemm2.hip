// GEMM core v2 for gfx950: fp32 v_mfma_f32_32x32x2_f32 fed by direct-to-LDS loads.
//
//  * BK = 32: a k-contiguous operand row contributes one whole 128-byte line per K-tile;
//  * operands go HBM/L2 -> LDS with `global_load_lds` (16 B per lane, no VGPR staging, no ds_write, no
//    VALU transposition).  The LDS destination of that instruction is lane-linear, so
//      - k-contiguous operands ([row][k]) keep their row-major image [rows][32]; the 16-byte chunk index is
//        XOR-swizzled with (row & 7) on the SOURCE address and again on the read, and the MFMA operands are
//        fetched with ds_read_b128 (lane half h takes k = 8g+4h .. +3, element j feeds MFMA j: any k order is
//        a valid fp32 reduction order as long as A and B agree);
//      - reduction-major operands ([k][rows]) land as [32][rows] and are read with conflict-free ds_read_b32;
//    rows/taps outside the matrix read a 16-byte zero page instead (conv 'same' padding, M/N/K edges);
//  * an LDS ring of NST stages with NST-1 K-tiles of DMA in flight: per K-tile ONE raw s_barrier behind a
//    COUNTED `s_waitcnt vmcnt(N)` (N = loads of the newer tiles, never 0 in steady state) -- the wait
//    retires this wave's oldest tile, the barrier publishes everybody's and also retires all reads of the
//    stage that the next DMA (issued right after it) overwrites;
//  * XCD-aware workgroup order (tiles sharing an M-tile's A rows run on one XCD's L2).
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
      const int kg = r0 + ((pc ^ (row & 7)) << 2);
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

// Loop-invariant part of the per-piece source addresses (no conv taps): computed once per workgroup so
// that a K-tile's DMA issue is one 64-bit add + one compare per piece instead of multiplies, swizzle and
// bounds logic competing with the wave's own MFMA issue.
template <int ROWS>
struct Pieces {
  const float* base[ROWS / 32];  // address of the piece in K-tile 0
  int koff[ROWS / 32];           // reduction offset of the piece inside a K-tile (k-contiguous: swizzled chunk)
  int t[ROWS / 32];              // conv taps: time index of the piece's row (A rows, TAPS == 1) or of its
                                 // reduction row in the current K-tile (B rows, TAPS == 2)
  bool ok[ROWS / 32];            // row / column inside the matrix
};

// kernel variants by conv-tap mode
constexpr int TAPS_NONE = 0;     // plain GEMM
constexpr int TAPS_RED = 1;      // shift_operand == 0, Rper % 32 == 0: every K-tile lies inside one tap -> the tap
                                 // (row shift of A, weight slice of B) is a per-tile scalar
constexpr int TAPS_ROWS = 2;     // shift_operand == 1 (weight gradient): reduction rows of B shifted by the
                                 // tap of this launch slice; T >= 32
constexpr int TAPS_GENERIC = 3;  // any Rper / T: per-piece address decode in every K-tile (slow; small convs)

// TAPS_RED keeps the reduction offset out of the base (the per-tile scalar (tap, k-in-tap) supplies it);
// TAPS_ROWS bakes the launch slice's row shift into B's base and tracks the row's time index.
template <int ROWS, bool KC, bool IS_A, int TAPS>
__device__ __forceinline__ void setup_pieces(Pieces<ROWS>& pc_, const GemmP& p, int row0, int r_begin, int shift_z,
                                             int tid) {
  const Fs2GemmArgs& a = p.a;
  const float* src = IS_A ? a.A : a.B;
  const int ld = IS_A ? a.lda : a.ldb;
  const int nrows = IS_A ? a.Mc : a.Nc;
  const int rb = TAPS == TAPS_RED ? 0 : r_begin;
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const int pidx = it * 256 + tid;
    pc_.t[it] = 0;
    if (KC) {
      const int row = pidx >> 3, pc = pidx & 7, gr = row0 + row;
      pc_.koff[it] = (pc ^ (row & 7)) << 2;
      pc_.ok[it] = gr < nrows;
      pc_.base[it] = src + (long long)gr * ld + rb + pc_.koff[it];
      if (TAPS == TAPS_RED && IS_A) pc_.t[it] = gr % a.T;
    } else {
      const int k = pidx / (ROWS / 4), col = row0 + (pidx % (ROWS / 4)) * 4;
      pc_.koff[it] = k;
      pc_.ok[it] = col < nrows;
      if (TAPS == TAPS_ROWS && !IS_A) {
        pc_.t[it] = (r_begin + k) % a.T;
        pc_.base[it] = src + (long long)(r_begin + k + shift_z) * ld + col;
      } else {
        pc_.base[it] = src + (long long)(rb + k) * ld + col;
      }
    }
  }
}

// kstep = floats between consecutive K-tiles of a piece (32 for k-contiguous, 32*ld otherwise);
// rem = reduction elements left from the start of this K-tile
template <int ROWS>
__device__ __forceinline__ void issue_fast(float* __restrict__ tile, const Pieces<ROWS>& pc_, long long koffset, int rem,
                                           int wave) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const bool ok = pc_.ok[it] && pc_.koff[it] < rem;
    glds16(ok ? pc_.base[it] + koffset : fs2_zero_page, tile + (it * 256 + wave * 64) * 4);
  }
}

// TAPS_RED, A operand: rows shifted by the K-tile's tap; a row whose shifted time index leaves [0, T) is
// the convolution's zero padding.  offset = shift * lda + k-in-tap (floats).
template <int ROWS>
__device__ __forceinline__ void issue_shifted_rows(float* __restrict__ tile, const Pieces<ROWS>& pc_, long long offset,
                                                   int shift, int T, int wave) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const bool ok = pc_.ok[it] && (unsigned)(pc_.t[it] + shift) < (unsigned)T;
    glds16(ok ? pc_.base[it] + offset : fs2_zero_page, tile + (it * 256 + wave * 64) * 4);
  }
}

// TAPS_ROWS, B operand: the reduction index is the (b, t) row itself; advances the pieces' time index by one
// K-tile (T >= 32, so one conditional subtraction keeps it in [0, T)).
template <int ROWS>
__device__ __forceinline__ void issue_shifted_red(float* __restrict__ tile, Pieces<ROWS>& pc_, long long koffset, int rem,
                                                  int shift, int T, int wave) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const bool ok = pc_.ok[it] && pc_.koff[it] < rem && (unsigned)(pc_.t[it] + shift) < (unsigned)T;
    glds16(ok ? pc_.base[it] + koffset : fs2_zero_page, tile + (it * 256 + wave * 64) * 4);
    const int t = pc_.t[it] + BK2;
    pc_.t[it] = t >= T ? t - T : t;
  }
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
      for (int gg = 0; gg < 4; ++gg) g[gg] = ((wrow0 + l31) * BK2 + (((2 * gg + h) ^ (l31 & 7)) << 2)) * 4;
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

template <int BM, int BN, bool AKC, bool BKC, int NST, int TAPS>
__global__ __launch_bounds__(256) void gemm2_kernel(GemmP p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_TILE = BM * BK2, B_TILE = BN * BK2, STAGE = A_TILE + B_TILE;
  constexpr int L = BM / 32 + BN / 32;  // LDS-DMA instructions per thread per K-tile
  static_assert(NST >= 2 && NST <= 4 && (NST - 2) * L < 64, "ring depth");
  __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = fs2_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int tapz = 0, split = 0;
  if (a.shift_operand == 1 || a.splitk > 1) {
    tapz = blockIdx.z / a.splitk;
    split = blockIdx.z % a.splitk;
  }
  const int r_begin = split * p.r_chunk;
  const int r_end = min(a.R, r_begin + p.r_chunk);
  const int nkt = (r_end - r_begin + BK2 - 1) / BK2;
  const int shift_z = tapz * a.tap_mul + a.tap_add;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // loop-invariant per-piece addresses (all modes but TAPS_GENERIC, which decodes every piece in every K-tile)
  Pieces<BM> pa;
  Pieces<BN> pb;
  if constexpr (TAPS != TAPS_GENERIC) {
    setup_pieces<BM, AKC, true, TAPS>(pa, p, m0, r_begin, shift_z, tid);
    setup_pieces<BN, BKC, false, TAPS>(pb, p, n0, r_begin, shift_z, tid);
  }
  const long long stepA = AKC ? BK2 : (long long)BK2 * a.lda, stepB = BKC ? BK2 : (long long)BK2 * a.ldb;
  // TAPS_RED: (tap, offset inside the tap) of the next K-tile to issue; K-tiles are issued in order
  int tap_i = TAPS == TAPS_RED ? r_begin / p.Rper : 0;
  int kin_i = TAPS == TAPS_RED ? r_begin - tap_i * p.Rper : 0;
  auto issue = [&](int kt, int stage) {
    float* At = lds + stage * STAGE;
    float* Bt = At + A_TILE;
    const int r0 = r_begin + kt * BK2;
    if constexpr (TAPS == TAPS_GENERIC) {
      issue_tile<BM, AKC, true>(At, p, m0, r0, r_end, shift_z, tid, wave);
      issue_tile<BN, BKC, false>(Bt, p, n0, r0, r_end, shift_z, tid, wave);
    } else if constexpr (TAPS == TAPS_RED) {
      const int shift = tap_i * a.tap_mul + a.tap_add;
      issue_shifted_rows<BM>(At, pa, (long long)shift * a.lda + kin_i, shift, a.T, wave);
      issue_fast<BN>(Bt, pb, (long long)tap_i * a.b_tap_stride + (BKC ? (long long)kin_i : (long long)kin_i * a.ldb),
                     BK2, wave);
      kin_i += BK2;
      if (kin_i == p.Rper) {
        kin_i = 0;
        ++tap_i;
      }
    } else if constexpr (TAPS == TAPS_ROWS) {
      issue_fast<BM>(At, pa, kt * stepA, r_end - r0, wave);
      issue_shifted_red<BN>(Bt, pb, kt * stepB, r_end - r0, shift_z, a.T, wave);
    } else {
      issue_fast<BM>(At, pa, kt * stepA, r_end - r0, wave);
      issue_fast<BN>(Bt, pb, kt * stepB, r_end - r0, wave);
    }
  };

  const int l31 = lane & 31, h = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
  RdAddr<BM, AKC> rda;
  RdAddr<BN, BKC> rdb;
  rda.setup(wm * (BM / 2), l31, h);
  rdb.setup(wn * (BN / 2), l31, h);

  for (int kt = 0; kt < NST - 1 && kt < nkt; ++kt) issue(kt, kt);
  int stage = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt is this wave's oldest outstanding DMA; min(NST-2, tiles left) newer ones may stay in flight
    const int newer = min(NST - 2, nkt - 1 - kt);
    if (NST >= 4 && newer == 2) wait_vmcnt_barrier<2 * L>();
    else if (NST >= 3 && newer == 1) wait_vmcnt_barrier<L>();
    else wait_vmcnt_barrier<0>();
    if (kt + NST - 1 < nkt) issue(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);  // the stage read in iteration kt-1
    const unsigned sa = lds0 + stage * (STAGE * 4), sb = sa + A_TILE * 4;
    // LDS reads run one reduction group ahead of the MFMAs that consume them (two register sets)
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
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  gemm_epilogue<BM, BN>(p, acc, m0, n0, wm, wn, lane, split, tapz);
}

template <int BM, int BN, int NST, bool GENERIC_TOO>
int launch_tile(GemmP& p, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  p.tiles_m = (a.Mc + BM - 1) / BM;
  p.tiles_n = (a.Nc + BN - 1) / BN;
  dim3 grid(p.tiles_m * p.tiles_n, 1, nz), block(256);
  int mode = TAPS_NONE;
  if (a.taps > 1) {
    if (a.shift_operand == 0) mode = (p.Rper % BK2 == 0) ? TAPS_RED : TAPS_GENERIC;
    else mode = a.T >= BK2 ? TAPS_ROWS : TAPS_GENERIC;
  }
  if (mode == TAPS_GENERIC) {
    if constexpr (GENERIC_TOO) {
      if (a.a_kcontig && a.b_kcontig) gemm2_kernel<BM, BN, true, true, NST, TAPS_GENERIC><<<grid, block, 0, s>>>(p);
      else if (a.a_kcontig) gemm2_kernel<BM, BN, true, false, NST, TAPS_GENERIC><<<grid, block, 0, s>>>(p);
      else if (!a.b_kcontig) gemm2_kernel<BM, BN, false, false, NST, TAPS_GENERIC><<<grid, block, 0, s>>>(p);
      else return FS2HIP_EINVAL;
    } else {
      return FS2HIP_EINVAL;  // odd tap widths: only the 64x64 2-stage tile carries the generic decode
    }
  } else if (a.a_kcontig && a.b_kcontig) {  // forward: taps only as TAPS_RED
    if (mode == TAPS_RED) gemm2_kernel<BM, BN, true, true, NST, TAPS_RED><<<grid, block, 0, s>>>(p);
    else if (mode == TAPS_NONE) gemm2_kernel<BM, BN, true, true, NST, TAPS_NONE><<<grid, block, 0, s>>>(p);
    else return FS2HIP_EINVAL;
  } else if (a.a_kcontig && !a.b_kcontig) {  // backward data
    if (mode == TAPS_RED) gemm2_kernel<BM, BN, true, false, NST, TAPS_RED><<<grid, block, 0, s>>>(p);
    else if (mode == TAPS_NONE) gemm2_kernel<BM, BN, true, false, NST, TAPS_NONE><<<grid, block, 0, s>>>(p);
    else return FS2HIP_EINVAL;
  } else if (!a.a_kcontig && !a.b_kcontig) {  // weight gradient
    if (mode == TAPS_ROWS) gemm2_kernel<BM, BN, false, false, NST, TAPS_ROWS><<<grid, block, 0, s>>>(p);
    else if (mode == TAPS_NONE) gemm2_kernel<BM, BN, false, false, NST, TAPS_NONE><<<grid, block, 0, s>>>(p);
    else return FS2HIP_EINVAL;
  } else {
    return FS2HIP_EINVAL;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int fs2_gemm2_launch(GemmP& p, int tile, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  // 16-byte pieces: k-contiguous operands need Rper % 4 == 0 (checked by the caller); the reduction
  // chunk of a split must be a multiple of this core's BK
  const int chunk = (a.R + a.splitk - 1) / a.splitk;
  p.r_chunk = ((chunk + BK2 - 1) / BK2) * BK2;
  switch (tile) {
    case 4: return launch_tile<128, 128, 3, false>(p, nz, s);  // 96 KiB ring, 1 workgroup / CU
    case 5: return launch_tile<128, 64, 3, false>(p, nz, s);   // 72 KiB ring, 2 workgroups / CU
    case 6: return launch_tile<64, 64, 4, false>(p, nz, s);    // 64 KiB ring, 2 workgroups / CU
    case 7: return launch_tile<64, 64, 2, true>(p, nz, s);     // 32 KiB, 5 workgroups / CU (occupancy instead of depth)
    case 8: return launch_tile<128, 64, 2, false>(p, nz, s);   // 48 KiB, 3 workgroups / CU
    case 9: return launch_tile<128, 128, 2, false>(p, nz, s);  // 64 KiB, 2 workgroups / CU
    default: return FS2HIP_EINVAL;
  }
}
