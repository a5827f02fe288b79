// Persistent form of the bf16-storage GEMM core (gemm_bf16_core.h): the weight-gradient GEMMs (long reductions cut into
// split-K slices, fp32 slabs) and the PostNet convolutions.
#include "gemm_bf16_core.h"

namespace {

// ---- persistent variant -----------------------------------------------------------------------------------------------
// The GEMMs of this model are short (K = 256 .. 1024: 4 .. 16 K-tiles) and, at the bf16 MFMA rate, bound by the
// memory side: with one tile per workgroup every tile pays the DMA latency of its first K-tiles and drains before the
// next workgroup starts.  Here a launch fills every workgroup slot once and a workgroup walks the units u = blockIdx,
// blockIdx + grid, ...; the K-tiles of all its units form ONE stream through the two LDS stages -- the first K-tiles of
// the next unit are in flight under the last MFMAs and the whole epilogue of the current one (gemm2p.hip, same scheme).
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
    epilogue_dispatch_b<BM, BN, false>(p, acc, uc.m0, uc.n0, wm, wn, lane, uc.split, uc.tapz, nullptr);  // (the ring is in use)
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

}  // namespace

int fs2_gemmbp_launch(GemmP& p, int tile, int nz, hipStream_t s) {
  switch (tile) {
    case 24: return launch_bp<128, 128, 2>(p, nz, s);
    case 25: return launch_bp<128, 64, 3>(p, nz, s);
    default: return FS2HIP_EINVAL;
  }
}
