// One output tile per workgroup on the bf16-storage GEMM core (gemm_bf16_core.h): the forward and data-gradient GEMMs
// (short reductions, whole-row stores through LDS).  The persistent form is gemm_bf16p.hip.
#include "gemm_bf16_core.h"

namespace {

// ``bid`` of ``nblk``: the workgroup's index inside its launch -- or, in a grouped launch, inside its member's share of it
template <int BM, int BN, bool AKC, bool BKC, int TAPS, int NST, bool COLSUM>
__device__ __forceinline__ void gemmb_body(const GemmP& p, const int bid, const int nblk) {
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
  const int unit = fs2_xcd_remap(bid, nblk);
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
#ifdef FS2_PROBES
    if (p.probe & 2) return;
#endif
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
#ifdef FS2_PROBES
    if (!(p.probe & 1))
#endif
    compute_ktile_b<BM, BN, AKC, BKC, COLSUM>(acc, cs, rda, rdb, sa, sb, do_cs);
    stage = stage + 1 == NST ? 0 : stage + 1;
  }
  // the stages are free once every wavefront has left the last K-tile: each takes a region of them for its stores
  static_assert(4 * Stager<(BN / 2) * 4>::BYTES <= NST * STAGE, "staging regions fit in the ring");
  __builtin_amdgcn_s_barrier();
#ifdef FS2_PROBES
  if (p.probe & 4) return;
#endif
  epilogue_dispatch_b<BM, BN, true>(p, acc, m0, n0, wm, wn, lane, split, tapz, lds + wave * Stager<(BN / 2) * 4>::BYTES);
  if (COLSUM && do_cs && wn == 0 && lane < 32) {  // every row of cs[i] is the same sum: take the lane's first register
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * (BM / 2) + 32 * i + lane;
      if (m < a.Mc) a.colsum[(long long)split * a.Mc + m] = cs[i][0];
    }
  }
}

template <int BM, int BN, bool AKC, bool BKC, int TAPS, int NST, bool COLSUM>
__global__ __launch_bounds__(256) void gemmb_kernel(GemmP p) {
  gemmb_body<BM, BN, AKC, BKC, TAPS, NST, COLSUM>(p, blockIdx.x, gridDim.x);
}

// grouped launch (GemmPG, gemm_common.h): the member that owns this workgroup, then the same code on ITS arguments
template <int BM, int BN, bool AKC, bool BKC, int NST, bool COLSUM>
__global__ __launch_bounds__(256) void gemmbg_kernel(GemmPG g) {
  const int i = fs2_group_member(g, blockIdx.x);
  gemmb_body<BM, BN, AKC, BKC, BT_NONE, NST, COLSUM>(g.m[i], blockIdx.x - g.start[i], g.start[i + 1] - g.start[i]);
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

template <int BM, int BN, int NST>
int launch_b_grouped(GemmPG& g, hipStream_t s) {
  long long total = 0;
  for (int i = 0; i < g.n; ++i) {
    GemmP& p = g.m[i];
    const Fs2GemmArgs& a = p.a;
    const int chunk = (a.R + a.splitk - 1) / a.splitk;
    p.r_chunk = ((chunk + BKE - 1) / BKE) * BKE;
    p.tiles_m = (a.Mc + BM - 1) / BM;
    p.tiles_n = (a.Nc + BN - 1) / BN;
    g.start[i] = (int)total;
    total += (long long)p.tiles_m * p.tiles_n * a.splitk;
    if (total > 0x7fffffffLL) return FS2HIP_EINVAL;
  }
  for (int i = g.n; i <= FS2_GEMM_GROUP_MAX; ++i) g.start[i] = (int)total;
  const Fs2GemmArgs& a = g.m[0].a;
  dim3 grid((unsigned)total), block(256);
  if (a.a_kcontig && a.b_kcontig) gemmbg_kernel<BM, BN, true, true, NST, false><<<grid, block, 0, s>>>(g);
  else if (a.a_kcontig && !a.b_kcontig) gemmbg_kernel<BM, BN, true, false, NST, false><<<grid, block, 0, s>>>(g);
  else if (!a.a_kcontig && !a.b_kcontig) gemmbg_kernel<BM, BN, false, false, NST, true><<<grid, block, 0, s>>>(g);
  else return FS2HIP_EINVAL;
  FS2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

int fs2_gemmb_launch_grouped(GemmPG& g, int tile, hipStream_t s) {
  switch (tile) {
    case 22: return launch_b_grouped<128, 64, 2>(g, s);
    case 23: return launch_b_grouped<64, 64, 3>(g, s);
    case 26: return launch_b_grouped<64, 64, 2>(g, s);
    default: return FS2HIP_EINVAL;
  }
}

int fs2_gemmbp_launch(GemmP& p, int tile, int nz, hipStream_t s);  // gemm_bf16p.hip

// tile ids of the bf16-storage core: 20 = 128x128 (2 stages, 2 workgroups / CU), 22 = 128x64 (2 stages, 3 / CU),
// 23 = 64x64 (3 stages, 3 / CU); persistent (one K-tile stream per workgroup across its tiles): 24 = 128x128 (2 / CU),
// 25 = 128x64 (3 / CU).  [A deep-ring form -- one workgroup per CU, a ring of 4-5 stages with counted waits across tile
// boundaries and epilogues -- was built, is correct, and is slower on every shape (one wavefront per SIMD cannot overlap
// DMA issue, LDS reads, MFMAs and the epilogue): kept as tools/probes/gemm_bf16_deep_ring.hip.txt, see DESIGN.md.]
int fs2_gemmb_launch(GemmP& p, int tile, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  const int chunk = (a.R + a.splitk - 1) / a.splitk;
  p.r_chunk = ((chunk + BKE - 1) / BKE) * BKE;
  switch (tile) {
    case 20: return launch_b<128, 128, 2>(p, nz, s);
    case 21: return launch_b<128, 64, 3>(p, nz, s);  // 72 KiB: 2 workgroups / CU, two K-tiles in flight each
    case 26: return launch_b<64, 64, 2>(p, nz, s);   // 32 KiB: 5 workgroups / CU
    case 22: return launch_b<128, 64, 2>(p, nz, s);
    case 23: return launch_b<64, 64, 3>(p, nz, s);
    case 24:
    case 25: return fs2_gemmbp_launch(p, tile, nz, s);
    default: return FS2HIP_EINVAL;
  }
}
