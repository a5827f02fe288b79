// Weights-stationary streaming GEMM, exact fp32 (v_mfma_f32_32x32x2_f32): C[M][N] = epi(A[M][256] . W[N][256]^T), both
// operands k-contiguous fp32 -- the Conformer's K = 256 projections in "32-true" (first feed-forward GEMM, QKV / out /
// pointwise projections and, through the transposed weight mirror, their data gradients): 4 of the 14 ms of GEMM
// time of the benchmark step.
//
// The tiled fp32 kernels run these shapes at 72-115 TFLOP/s of 157: a 64 x 64 x 256 tile is 8 K-tiles of MFMAs behind a
// prologue (operand DMA latency) and in front of an epilogue that shares the vector pipe with the MFMAs, the tile
// counts of 20 736 rows leave a partly filled last round (5 184 tiles on 1 280 workgroup slots = 4.05 rounds), and W is
// re-fetched into LDS for every row tile.  Here, as in gemm_ws.hip:
//   * a workgroup (8 wavefronts, one per CU) owns 256 output columns: wavefront w keeps W[32 w .. 32 w + 31][0 .. 255]
//     in 128 registers as MFMA fragments for the whole launch;
//   * A streams through LDS in row tiles of 32 rows x 256 (32 KB, LDS-DMA, two stages): all eight wavefronts read it,
//     32 ds_read_b128 + 128 MFMAs per wavefront and tile (8 192 matrix-pipe cycles), one barrier per tile;
//   * workgroups walk row streams, so nothing is launched per tile, nothing is re-fetched, and the imbalance is one row
//     tile in ~10 (N = 1024: 648 row tiles over 64 streams) instead of a partly filled round.
// An interval is ~7 us of MFMAs per SIMD, far longer than a DMA round trip: the tile for interval i + 1 is requested at
// the top of interval i and has long landed when it is needed, so the epilogue is the tiled kernels' own
// (gemm_epilogue_impl: same arithmetic, same dropout element index, compiler-managed loads) and the only hand-counted
// wait is the one that lets the previous tile's stores stay in flight across the barrier.
//
// Round 5: the two wavefronts of a SIMD run HALF AN INTERVAL APART.  Rounds 4's kernel ran all eight in lockstep: MFMAs,
// then epilogue, then the barrier -- so both wavefronts of a SIMD left the matrix pipe idle together for the length of
// an epilogue (address arithmetic, 16-32 stores, their issue stalls): 103.6 us where the MFMA count says 76
// (profiles/r05_clock_probe.txt: the clock is NOT the difference).  One wavefront's single dependent accumulation chain
// fills the fp32 matrix pipe by itself (tools/probes/chain_probe.hip: 0.988 of peak with one chain, one wavefront per
// SIMD), so nothing is lost while a SIMD's other wavefront is elsewhere: wavefronts 0-3 run tile i's MFMAs and then its
// epilogue as before; wavefronts 4-7 carry tile i - 1's accumulator across the barrier, run ITS epilogue first -- beside
// the MFMAs of their SIMD's early wavefront -- and then tile i's MFMAs, beside the early wavefront's epilogue.  Same
// barrier per tile, same stage discipline (at barrier i every wavefront has finished READING tile i - 1), same
// arithmetic in the same order: still bit-identical to the tiled kernels.
#include "gemm2_core.h"

namespace {

constexpr int W32_THREADS = 512, W32_KT = 8;         // K = 256: eight 32-deep K-tile images per row tile
constexpr int W32_TILE_FLOATS = W32_KT * 32 * 32;    // [KT][32 rows][32 floats]
// (measured and rejected: two accumulation chains per wavefront -- 105.0 us against 103.6 us for the 20 736 x 1024 store
// form: the second wavefront of the SIMD already fills the gaps behind a dependent MFMA.  -DW32_TWO_CHAINS=1 rebuilds it;
// the sums then differ from the tiled kernels' in the last bit.)
#ifndef W32_TWO_CHAINS
#define W32_TWO_CHAINS 0
#endif

// STORES: vector-memory stores of one epilogue per wavefront (16 accumulator registers, twice with a pre-activation output)
template <int EPI, bool TWO>
__global__ __launch_bounds__(W32_THREADS) void gemmws32_kernel(GemmP p, int n_slices, int n_streams, int n_row_tiles) {
  __shared__ __attribute__((aligned(16))) float lds[2 * W32_TILE_FLOATS];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;  // (XCD, index inside it): the slices of a row stream share an XCD
  const int slice = bj % n_slices;
  const int stream = (bj / n_slices) * 8 + bx;
  const int ns0 = slice * 256, nw0 = ns0 + wave * 32;
  const int cnt = stream < n_row_tiles ? (n_row_tiles - 1 - stream) / n_streams + 1 : 0;

  // ---- W fragments: group q = 4 kt + g holds W[nw0 + l31][8 q + 4 h .. + 3]; element j feeds MFMA j of the group -------
  f32x4 wf[32];
  {
    const float* wrow = a.B + (long long)(nw0 + l31) * a.ldb + 4 * h;
    const bool ok = nw0 + l31 < a.Nc;
#pragma unroll
    for (int q = 0; q < 32; ++q) wf[q] = ok ? *reinterpret_cast<const f32x4*>(wrow + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- A stream ----------------------------------------------------------------------------------------------------------
  // piece q = it * 512 + tid: K-tile q >> 8, row (q >> 3) & 31, chunk q & 7 (swizzled on the source side); the K-tile
  // index grows by 2 per `it` = 256 bytes along the row, carried in the scalar offset.  Rows past Mc read zeros.
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.Mc * a.lda * 4, 0x00020000);
  const int avoff = (((tid >> 3) & 31) * a.lda + (tid >> 8) * 32 + (((tid & 7) ^ ((tid >> 4) & 7)) << 2)) * 4;
  const int tile_stride = 32 * a.lda * 4;
  auto dma = [&](int k) {
    if (k >= cnt) return;
    const int soff = (stream + k * n_streams) * tile_stride;
    float* dst = lds + (k & 1) * W32_TILE_FLOATS;
#pragma unroll
    for (int it = 0; it < 4; ++it) blds16(ra, avoff, soff + 256 * it, dst + (it * W32_THREADS + wave * 64) * 4);
  };
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
  unsigned ard[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) ard[g] = lds0 + (l31 * 32 + (((2 * g + h) ^ ((l31 >> 1) & 7)) << 2)) * 4;

  f32x16 acc[1][1];
  constexpr int STORES = TWO ? 32 : 16;
  float* C = a.C;
  const bool late = wave >= 4;  // (wave-uniform) the SIMD's second wavefront: epilogues run one interval later
  dma(0);
  for (int i = 0; i < cnt; ++i) {
    // tile i has landed for everybody (the stores of the previous epilogue, younger than its DMA, may stay in flight --
    // a late wavefront has no epilogue behind DMA(1) yet); everybody has left the MFMAs of tile i - 1, whose stage the
    // next DMA overwrites
    if (i == 0 || (late && i == 1)) wait_vmcnt_barrier<0>(); else wait_vmcnt_barrier<STORES>();
    dma(i + 1);
    if (late && i > 0)
      gemm_epilogue_impl<64, 64, EPI>(p, acc, C, a.ldc, (stream + (i - 1) * n_streams) * 32, ns0, 0, wave, lane);
    const unsigned sb = (i & 1) * (W32_TILE_FLOATS * 4);
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = acc2[r] = 0.f;
    // 32 groups of four K-steps; the fragment of group q + 2 is requested while group q is in the MFMAs
    f32x4 af[3];
#define W32_RD(Q_) lds_rd128<((Q_) >> 2) * 4096>(af[(Q_) % 3], ard[(Q_) & 3] + sb);
#define W32_MM(Q_, PENDING)                                                                                        \
  lds_wait<PENDING>();                                                                                             \
  pin(af[(Q_) % 3]);                                                                                               \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                  \
    if (W32_TWO_CHAINS && (j & 1))                                                                                 \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(Q_) % 3][j], wf[Q_][j], acc2, 0, 0, 0);                       \
    else                                                                                                           \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[(Q_) % 3][j], wf[Q_][j], acc[0][0], 0, 0, 0);            \
  }
#define W32_STEP(Q_) W32_RD((Q_) + 2) W32_MM(Q_, 2)
    W32_RD(0) W32_RD(1)
    W32_STEP(0) W32_STEP(1) W32_STEP(2) W32_STEP(3) W32_STEP(4) W32_STEP(5) W32_STEP(6) W32_STEP(7)
    W32_STEP(8) W32_STEP(9) W32_STEP(10) W32_STEP(11) W32_STEP(12) W32_STEP(13) W32_STEP(14) W32_STEP(15)
    W32_STEP(16) W32_STEP(17) W32_STEP(18) W32_STEP(19) W32_STEP(20) W32_STEP(21) W32_STEP(22) W32_STEP(23)
    W32_STEP(24) W32_STEP(25) W32_STEP(26) W32_STEP(27) W32_STEP(28) W32_STEP(29)
    W32_MM(30, 1) W32_MM(31, 0)
#undef W32_STEP
#undef W32_RD
#undef W32_MM
    if (W32_TWO_CHAINS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += acc2[r];
    }
    if (!late) gemm_epilogue_impl<64, 64, EPI>(p, acc, C, a.ldc, (stream + i * n_streams) * 32, ns0, 0, wave, lane);
  }
  if (late && cnt > 0)
    gemm_epilogue_impl<64, 64, EPI>(p, acc, C, a.ldc, (stream + (cnt - 1) * n_streams) * 32, ns0, 0, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI>
int go32(GemmP& p, dim3 grid, int n_slices, int n_streams, int n_row_tiles, hipStream_t s) {
  if (p.a.out_pre) gemmws32_kernel<EPI, true><<<grid, dim3(W32_THREADS), 0, s>>>(p, n_slices, n_streams, n_row_tiles);
  else gemmws32_kernel<EPI, false><<<grid, dim3(W32_THREADS), 0, s>>>(p, n_slices, n_streams, n_row_tiles);
  FS2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// tile id 32: exact fp32, forward orientation (both operands k-contiguous), K = 256, no conv taps, no split
int fs2_gemmws32_launch(GemmP& p, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  if (a.operand_bf16 != 0 || !a.a_kcontig || !a.b_kcontig || a.taps != 1 || a.splitk != 1 || a.R != 256 || a.colsum)
    return FS2HIP_EINVAL;
  if ((a.ldb % 4) || ((uintptr_t)a.B % 16) || a.epi < FS2_EPI_STORE || a.epi > FS2_EPI_DACT) return FS2HIP_EINVAL;
  if (a.out_pre && a.epi != FS2_EPI_ACT) return FS2HIP_EINVAL;
  if ((long long)(a.Mc + 64) * a.lda * 4 >= 0x7fffffffLL) return FS2HIP_EINVAL;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return FS2HIP_EINVAL;
    n_cu = prop.multiProcessorCount;
  }
  const int n_slices = (a.Nc + 255) / 256;
  const int per_xcd = n_cu / 8;
  if (per_xcd < 1 || n_slices > per_xcd) return FS2HIP_EINVAL;
  const int n_row_tiles = (a.Mc + 31) / 32;
  int streams_per_xcd = per_xcd / n_slices;
  while (streams_per_xcd > 1 && (streams_per_xcd - 1) * 8 >= n_row_tiles) --streams_per_xcd;
  const int n_streams = streams_per_xcd * 8;
  dim3 grid(8 * streams_per_xcd * n_slices);
  switch (a.epi) {
    case FS2_EPI_ACT: return go32<FS2_EPI_ACT>(p, grid, n_slices, n_streams, n_row_tiles, s);
    case FS2_EPI_RESID: return go32<FS2_EPI_RESID>(p, grid, n_slices, n_streams, n_row_tiles, s);
    case FS2_EPI_DACT: return go32<FS2_EPI_DACT>(p, grid, n_slices, n_streams, n_row_tiles, s);
    default: return go32<0>(p, grid, n_slices, n_streams, n_row_tiles, s);
  }
}
