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
  bool ok[ROWS / 32];            // row / column inside the matrix
};

template <int ROWS, bool KC, bool IS_A>
__device__ __forceinline__ void setup_pieces(Pieces<ROWS>& pc_, const GemmP& p, int row0, int r_begin, int tid) {
  const Fs2GemmArgs& a = p.a;
  const float* src = IS_A ? a.A : a.B;
  const int ld = IS_A ? a.lda : a.ldb;
  const int nrows = IS_A ? a.Mc : a.Nc;
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const int pidx = it * 256 + tid;
    if (KC) {
      const int row = pidx >> 3, pc = pidx & 7, gr = row0 + row;
      pc_.koff[it] = (pc ^ (row & 7)) << 2;
      pc_.ok[it] = gr < nrows;
      pc_.base[it] = src + (long long)gr * ld + r_begin + pc_.koff[it];
    } else {
      const int k = pidx / (ROWS / 4), col = row0 + (pidx % (ROWS / 4)) * 4;
      pc_.koff[it] = k;
      pc_.ok[it] = col < nrows;
      pc_.base[it] = src + (long long)(r_begin + k) * ld + col;
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

// wait until at most N of this wave's vector-memory operations (here: LDS-DMA pieces) are outstanding, then
// the workgroup barrier.  One asm statement with a memory clobber: the compiler tracks neither the DMA's
// LDS writes nor the counter, so no LDS access may move across it.
template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int BM, int BN, bool AKC, bool BKC, int NST>
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

  const bool tapmode = a.taps > 1;  // conv taps: per-tile address decode; otherwise hoisted pointers
  Pieces<BM> pa;
  Pieces<BN> pb;
  if (!tapmode) {
    setup_pieces<BM, AKC, true>(pa, p, m0, r_begin, tid);
    setup_pieces<BN, BKC, false>(pb, p, n0, r_begin, tid);
  }
  const long long stepA = AKC ? BK2 : (long long)BK2 * a.lda, stepB = BKC ? BK2 : (long long)BK2 * a.ldb;
  auto issue = [&](int kt, int stage) {
    float* At = lds + stage * STAGE;
    float* Bt = At + A_TILE;
    const int r0 = r_begin + kt * BK2;
    if (tapmode) {
      issue_tile<BM, AKC, true>(At, p, m0, r0, r_end, shift_z, tid, wave);
      issue_tile<BN, BKC, false>(Bt, p, n0, r0, r_end, shift_z, tid, wave);
    } else {
      issue_fast<BM>(At, pa, kt * stepA, r_end - r0, wave);
      issue_fast<BN>(Bt, pb, kt * stepB, r_end - r0, wave);
    }
  };

  const int l31 = lane & 31, h = lane >> 5;
  for (int kt = 0; kt < NST - 1 && kt < nkt; ++kt) issue(kt, kt);
  int stage = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt is this wave's oldest outstanding DMA; min(NST-2, tiles left) newer ones may stay in flight
    const int newer = min(NST - 2, nkt - 1 - kt);
    if (NST >= 4 && newer == 2) wait_vmcnt_barrier<2 * L>();
    else if (NST >= 3 && newer == 1) wait_vmcnt_barrier<L>();
    else wait_vmcnt_barrier<0>();
    if (kt + NST - 1 < nkt) issue(kt + NST - 1, stage == 0 ? NST - 1 : stage - 1);  // the stage read in iteration kt-1
    const float* At = lds + stage * STAGE;
    const float* Bt = At + A_TILE;
#pragma unroll
    for (int g = 0; g < BK2 / 8; ++g) {
      float av[TM][4], bv[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / 2) + i * 32 + l31;
        if (AKC) {
          const float4 v = *reinterpret_cast<const float4*>(At + row * BK2 + (((2 * g + h) ^ (row & 7)) << 2));
          av[i][0] = v.x; av[i][1] = v.y; av[i][2] = v.z; av[i][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) av[i][j] = At[(8 * g + 4 * h + j) * BM + row];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int row = wn * (BN / 2) + i * 32 + l31;
        if (BKC) {
          const float4 v = *reinterpret_cast<const float4*>(Bt + row * BK2 + (((2 * g + h) ^ (row & 7)) << 2));
          bv[i][0] = v.x; bv[i][1] = v.y; bv[i][2] = v.z; bv[i][3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[i][j] = Bt[(8 * g + 4 * h + j) * BN + row];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j], bv[jn][j], acc[i][jn], 0, 0, 0);
    }
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  gemm_epilogue<BM, BN>(p, acc, m0, n0, wm, wn, lane, split, tapz);
}

template <int BM, int BN, int NST>
int launch_tile(GemmP& p, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  p.tiles_m = (a.Mc + BM - 1) / BM;
  p.tiles_n = (a.Nc + BN - 1) / BN;
  dim3 grid(p.tiles_m * p.tiles_n, 1, nz), block(256);
  if (a.a_kcontig && a.b_kcontig) gemm2_kernel<BM, BN, true, true, NST><<<grid, block, 0, s>>>(p);
  else if (a.a_kcontig && !a.b_kcontig) gemm2_kernel<BM, BN, true, false, NST><<<grid, block, 0, s>>>(p);
  else if (!a.a_kcontig && !a.b_kcontig) gemm2_kernel<BM, BN, false, false, NST><<<grid, block, 0, s>>>(p);
  else return FS2HIP_EINVAL;
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
    case 4: return launch_tile<128, 128, 3>(p, nz, s);  // 96 KiB ring, 1 workgroup / CU
    case 5: return launch_tile<128, 64, 3>(p, nz, s);   // 72 KiB ring, 2 workgroups / CU
    case 6: return launch_tile<64, 64, 4>(p, nz, s);    // 64 KiB ring, 2 workgroups / CU
    case 7: return launch_tile<64, 64, 2>(p, nz, s);    // 32 KiB, 5 workgroups / CU (occupancy instead of depth)
    case 8: return launch_tile<128, 64, 2>(p, nz, s);   // 48 KiB, 3 workgroups / CU
    case 9: return launch_tile<128, 128, 2>(p, nz, s);  // 64 KiB, 2 workgroups / CU
    default: return FS2HIP_EINVAL;
  }
}
