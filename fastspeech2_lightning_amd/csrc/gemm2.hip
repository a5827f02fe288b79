// GEMM core v2 for gfx950: fp32 v_mfma_f32_32x32x2_f32 fed by direct-to-LDS loads.
//
//  * BK = 32: a k-contiguous operand row contributes one whole 128-byte line per K-tile;
//  * operands go HBM/L2 -> LDS with `global_load_lds` (16 B per lane, no VGPR staging, no ds_write, no
//    VALU transposition).  The LDS destination of that instruction is lane-linear, so
//      - k-contiguous operands ([row][k]) keep their row-major image [rows][32]; the 16-byte chunk index is
//        XOR-swizzled with ((row >> 1) & 7) on the SOURCE address and again on the read (ds_read_b128 is served
//        in the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} per half wave: within a group the rows of
//        one parity then hit eight different chunks, i.e. all 64 banks once), and the MFMA operands are
//        fetched with ds_read_b128 (lane half h takes k = 8g+4h .. +3, element j feeds MFMA j: any k order is
//        a valid fp32 reduction order as long as A and B agree);
//      - reduction-major operands ([k][rows]) land as [32][rows] and are read with conflict-free ds_read_b32;
//    rows/taps outside the matrix read a 16-byte zero page instead (conv 'same' padding, M/N/K edges);
//  * an LDS ring of NST stages with NST-1 K-tiles of DMA in flight: per K-tile ONE raw s_barrier behind a
//    COUNTED `s_waitcnt vmcnt(N)` (N = loads of the newer tiles, never 0 in steady state) -- the wait
//    retires this wave's oldest tile, the barrier publishes everybody's and also retires all reads of the
//    stage that the next DMA (issued right after it) overwrites;
//  * XCD-aware workgroup order (tiles sharing an M-tile's A rows, and the tiles of one split-K / tap slice, run on
//    one XCD's L2).
#include "gemm2_core.h"
namespace {

// ``bid`` of ``nblk``: the workgroup's index inside its launch -- or, in a grouped launch, inside its member's share of it
template <int BM, int BN, bool AKC, bool BKC, int NST, int TAPS, int BF>
__device__ __forceinline__ void gemm2_body(const GemmP& p, const int bid, const int nblk) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_TILE = BM * BK2, B_TILE = BN * BK2, STAGE = A_TILE + B_TILE;
  constexpr int L = BM / 32 + BN / 32;  // LDS-DMA instructions per thread per K-tile
  static_assert(NST >= 2 && NST <= 4 && (NST - 2) * L < 64, "ring depth");
  __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // one grid dimension over (reduction slice, tile): the XCD remap then keeps the tiles of one split-K / tap slice
  // -- the workgroups that read the same reduction chunk of both operands -- on one XCD's L2.  [With the slices in
  // gridDim.z a slice's tiles were dealt over all eight XCDs and every XCD fetched the chunk: the 1024x256
  // weight gradient read X eight times, 255 MB for 106 MB of operands.]
  const int ntile = p.tiles_m * p.tiles_n;
  const int unit = fs2_xcd_remap(bid, nblk);
  const int z = unit / ntile;
  const int wg = unit - z * ntile;
  const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int tapz = 0, split = 0;
  if (a.shift_operand == 1 || a.splitk > 1) {  // slice order (split, tap): the taps of one reduction chunk are neighbours
    const int ntap = a.shift_operand == 1 ? a.taps : 1;
    split = z / ntap;
    tapz = z - split * ntap;
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

  // loop-invariant per-piece offsets + scalar stream state (all modes but TAPS_GENERIC, which decodes every
  // piece in every K-tile)
  Pieces<BM> pa;
  Pieces<BN> pb;
  Stream<AKC, BKC, TAPS> st;
  if constexpr (TAPS != TAPS_GENERIC) {
    setup_pieces<BM, AKC, true, TAPS>(pa, p, m0, r_begin, tid);
    setup_pieces<BN, BKC, false, TAPS>(pb, p, n0, r_begin, tid);
    st.begin(p, r_begin, r_end, shift_z);
  }
  auto issue = [&](int kt, int stage) {  // K-tiles are issued in order: kt == st.kt
    float* At = lds + stage * STAGE;
    float* Bt = At + A_TILE;
    if constexpr (TAPS == TAPS_GENERIC) {
      const int r0 = r_begin + kt * BK2;
      issue_tile<BM, AKC, true>(At, p, m0, r0, r_end, shift_z, tid, wave);
      issue_tile<BN, BKC, false>(Bt, p, n0, r0, r_end, shift_z, tid, wave);
    } else {
      st.template issue<BM, BN>(p, At, Bt, pa, pb, wave, tid);
    }
  };

  const int l31 = lane & 31, h = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
  RdAddr<BM, AKC> rda;
  RdAddr<BN, BKC> rdb;
  rda.setup(wm * (BM / 2), l31, h);
  rdb.setup(wn * (BN / 2), l31, h);

  // bias gradient beside a weight gradient (Fs2GemmArgs.colsum): wavefronts of the first tile column, first tap slice
  float cs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) cs[i] = 0.f;
  const bool do_cs = !AKC && !BKC && a.colsum != nullptr && tile_n == 0 && tapz == 0 && wn == 0;

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
    compute_ktile_any<BF, BM, BN, AKC, BKC>(acc, rda, rdb, sa, sb, cs, do_cs);
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  gemm_epilogue<BM, BN>(p, acc, m0, n0, wm, wn, lane, split, tapz);
  if (do_cs) colsum_store<BM>(a, cs, m0, wm, lane, split);
}

template <int BM, int BN, bool AKC, bool BKC, int NST, int TAPS, int BF>
__global__ __launch_bounds__(256) void gemm2_kernel(GemmP p) {
  gemm2_body<BM, BN, AKC, BKC, NST, TAPS, BF>(p, blockIdx.x, gridDim.x);
}

// grouped launch (GemmPG, gemm_common.h): the member that owns this workgroup, then the same code on ITS arguments
template <int BM, int BN, bool AKC, bool BKC, int NST>
__global__ __launch_bounds__(256) void gemm2g_kernel(GemmPG g) {
  const int i = fs2_group_member(g, blockIdx.x);
  gemm2_body<BM, BN, AKC, BKC, NST, TAPS_NONE, 0>(g.m[i], blockIdx.x - g.start[i], g.start[i + 1] - g.start[i]);
}

// fp32 or bf16-operand instance of one kernel shape
#define FS2_GO(AKC_, BKC_, TAPS_)                                                          \
  do {                                                                                     \
    if (a.operand_bf16 == 2) gemm2_kernel<BM, BN, AKC_, BKC_, NST, TAPS_, 2><<<grid, block, 0, s>>>(p);  \
    else if (a.operand_bf16) gemm2_kernel<BM, BN, AKC_, BKC_, NST, TAPS_, 1><<<grid, block, 0, s>>>(p);  \
    else gemm2_kernel<BM, BN, AKC_, BKC_, NST, TAPS_, 0><<<grid, block, 0, s>>>(p);        \
  } while (0)

template <int BM, int BN, int NST, bool GENERIC_TOO>
int launch_tile(GemmP& p, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  p.tiles_m = (a.Mc + BM - 1) / BM;
  p.tiles_n = (a.Nc + BN - 1) / BN;
  if ((long long)p.tiles_m * p.tiles_n * nz > 0x7fffffffLL) return FS2HIP_EINVAL;
  dim3 grid(p.tiles_m * p.tiles_n * nz), block(256);
  int mode = TAPS_NONE;
  if (a.operand_bf16 == 3 && !(a.a_kcontig && a.b_kcontig)) return FS2HIP_EINVAL;
  if (a.taps > 1) {
    if (a.shift_operand == 0) mode = (p.Rper % BK2 == 0) ? TAPS_RED : TAPS_GENERIC;
    else mode = a.T >= BK2 ? TAPS_ROWS : TAPS_GENERIC;
  }
  if (mode == TAPS_GENERIC && a.operand_bf16 == 3) return FS2HIP_EINVAL;
  if (mode == TAPS_GENERIC) {
    if constexpr (GENERIC_TOO) {
      if (a.a_kcontig && a.b_kcontig) FS2_GO(true, true, TAPS_GENERIC);
      else if (a.a_kcontig) FS2_GO(true, false, TAPS_GENERIC);
      else if (!a.b_kcontig) FS2_GO(false, false, TAPS_GENERIC);
      else return FS2HIP_EINVAL;
    } else {
      return FS2HIP_EINVAL;  // odd tap widths: only the 64x64 2-stage tile carries the generic decode
    }
  } else if (a.a_kcontig && a.b_kcontig) {  // forward: taps only as TAPS_RED
    if (a.operand_bf16 == 3) {  // bf16 in memory: these two instances only
      if (mode == TAPS_RED) gemm2_kernel<BM, BN, true, true, NST, TAPS_RED, 3><<<grid, block, 0, s>>>(p);
      else if (mode == TAPS_NONE) gemm2_kernel<BM, BN, true, true, NST, TAPS_NONE, 3><<<grid, block, 0, s>>>(p);
      else return FS2HIP_EINVAL;
    } else if (mode == TAPS_RED) FS2_GO(true, true, TAPS_RED);
    else if (mode == TAPS_NONE) FS2_GO(true, true, TAPS_NONE);
    else return FS2HIP_EINVAL;
  } else if (a.a_kcontig && !a.b_kcontig) {  // backward data
    if (mode == TAPS_RED) FS2_GO(true, false, TAPS_RED);
    else if (mode == TAPS_NONE) FS2_GO(true, false, TAPS_NONE);
    else return FS2HIP_EINVAL;
  } else if (!a.a_kcontig && !a.b_kcontig) {  // weight gradient
    if (mode == TAPS_ROWS) FS2_GO(false, false, TAPS_ROWS);
    else if (mode == TAPS_NONE) FS2_GO(false, false, TAPS_NONE);
    else return FS2HIP_EINVAL;
  } else {
    return FS2HIP_EINVAL;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

#undef FS2_GO

template <int BM, int BN, int NST>
int launch_grouped(GemmPG& g, hipStream_t s) {
  long long total = 0;
  for (int i = 0; i < g.n; ++i) {
    GemmP& p = g.m[i];
    const Fs2GemmArgs& a = p.a;
    const int chunk = (a.R + a.splitk - 1) / a.splitk;
    p.r_chunk = ((chunk + BK2 - 1) / BK2) * BK2;
    p.tiles_m = (a.Mc + BM - 1) / BM;
    p.tiles_n = (a.Nc + BN - 1) / BN;
    if (!fs2_gemm2_offsets_fit(a)) return FS2HIP_EINVAL;
    g.start[i] = (int)total;
    total += (long long)p.tiles_m * p.tiles_n * a.splitk;
    if (total > 0x7fffffffLL) return FS2HIP_EINVAL;
  }
  for (int i = g.n; i <= FS2_GEMM_GROUP_MAX; ++i) g.start[i] = (int)total;
  const Fs2GemmArgs& a = g.m[0].a;
  dim3 grid((unsigned)total), block(256);
  if (a.a_kcontig && a.b_kcontig) gemm2g_kernel<BM, BN, true, true, NST><<<grid, block, 0, s>>>(g);
  else if (a.a_kcontig && !a.b_kcontig) gemm2g_kernel<BM, BN, true, false, NST><<<grid, block, 0, s>>>(g);
  else if (!a.a_kcontig && !a.b_kcontig) gemm2g_kernel<BM, BN, false, false, NST><<<grid, block, 0, s>>>(g);
  else return FS2HIP_EINVAL;
  FS2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int fs2_gemm2_launch_grouped(GemmPG& g, int tile, hipStream_t s) {
  switch (tile) {
    case 7: return launch_grouped<64, 64, 2>(g, s);
    case 8: return launch_grouped<128, 64, 2>(g, s);
    default: return FS2HIP_EINVAL;
  }
}

int fs2_gemm2_launch(GemmP& p, int tile, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  // 16-byte pieces: k-contiguous operands need Rper % 4 == 0 (checked by the caller); the reduction
  // chunk of a split must be a multiple of this core's BK
  const int chunk = (a.R + a.splitk - 1) / a.splitk;
  p.r_chunk = ((chunk + BK2 - 1) / BK2) * BK2;
  if (!fs2_gemm2_offsets_fit(a)) return FS2HIP_EINVAL;  // > 2 GiB operands: core v1
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
