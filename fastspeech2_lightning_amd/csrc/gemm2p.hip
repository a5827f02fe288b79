// Persistent variant of the direct-to-LDS GEMM core (gemm2.hip) for gfx950.
//
// One launch fills every workgroup slot of the chip once (grid = CUs x workgroups per CU, or the number of
// work units if smaller); a workgroup walks the units u = blockIdx, blockIdx + grid, ... (a unit = one output
// tile of one split-K / conv-tap slice).  The K-tiles of all its units form ONE stream through the two LDS
// stages, so
//   * the DMA of the next unit's first K-tile is in flight while the last K-tile of the current unit is in
//     the MFMAs: no per-tile prologue latency;
//   * the epilogue of a unit is executed after the barrier and DMA issue of the next unit's first K-tile: its
//     global stores are not waited for until the following K-tile (every wait of
//     this two-stage ring is vmcnt(0), which also covers them), i.e. they drain under one K-tile of MFMAs
//     instead of at the end of a workgroup, where every co-resident workgroup would write at once.
// Work-unit ids go through the XCD remap, so the units that the workgroups of one XCD process in the same
// round are consecutive tiles (shared A rows stay in that XCD's L2).
#include "gemm2_core.h"

namespace {

struct Unit {
  int m0, n0, tapz, split, r_begin, r_end, nkt, shift_z, tile_n;
  int slice;  // >= 0: a reduction slice of a tail tile (raw partial sums into the workspace slab `slice`)
};

// Split tail (tiles 13/14): with T tiles on G workgroup slots the last T mod G tiles would occupy a fraction of
// the chip for a whole tile time.  The tiles of the last M-tile rows are therefore cut into S reduction slices
// each (S * tail tiles <= G): units [0, u_full) are whole tiles, the rest are slices, which write raw partial sums
// that fs2_tail_fixup adds up (+ bias).  S == 0: no split.
struct Hybrid {
  int u_full, S, chunk, m_tail0;
  float* ws;
  long long slab;
};

__device__ __forceinline__ Unit decode_unit(const GemmP& p, const Hybrid& hy, int u, int nunits, int tiles, int BM,
                                            int BN) {
  const Fs2GemmArgs& a = p.a;
  Unit q;
  int t;
  if (hy.S == 0) {
    const int uu = fs2_xcd_remap(u, nunits);
    const int z = uu / tiles;
    t = uu - z * tiles;
    const int ntap = a.shift_operand == 1 ? a.taps : 1;  // slice order (split, tap): the taps of one reduction chunk
    q.split = z / ntap;                                    // (same dY rows, X rows shifted by one) share an XCD's L2
    q.tapz = z - q.split * ntap;
    q.r_begin = q.split * p.r_chunk;
    q.r_end = min(a.R, q.r_begin + p.r_chunk);
    q.slice = -1;
  } else if (u < hy.u_full) {  // whole tiles and slices are remapped separately: every XCD gets the same mix
    t = fs2_xcd_remap(u, hy.u_full);
    q.tapz = q.split = 0;
    q.r_begin = 0;
    q.r_end = a.R;
    q.slice = -1;
  } else {
    const int v = fs2_xcd_remap(u - hy.u_full, nunits - hy.u_full);
    t = hy.u_full + v / hy.S;
    q.slice = v - (v / hy.S) * hy.S;
    q.tapz = q.split = 0;
    q.r_begin = q.slice * hy.chunk;
    q.r_end = min(a.R, q.r_begin + hy.chunk);
  }
  const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
  q.m0 = tile_m * BM;
  q.n0 = tile_n * BN;
  q.tile_n = tile_n;
  q.nkt = q.r_end > q.r_begin ? (q.r_end - q.r_begin + BK2 - 1) / BK2 : 0;
  q.shift_z = q.tapz * a.tap_mul + a.tap_add;
  return q;
}

template <int BM, int BN, bool AKC, bool BKC, int TAPS, int BF>
__global__ __launch_bounds__(256) void gemm2p_kernel(GemmP p, Hybrid hy, int nunits, int tiles) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_TILE = BM * BK2, B_TILE = BN * BK2, STAGE = A_TILE + B_TILE;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;

  f32x16 acc[TM][TN];

  // ---- producer: the next K-tile of this workgroup's stream -------------------------------------------------
  int u_p = blockIdx.x, nkt_p = 0;
  Pieces<BM> pa;
  Pieces<BN> pb;
  Stream<AKC, BKC, TAPS> st;
  auto enter_unit = [&]() {
    if (u_p >= nunits) return;
    const Unit up = decode_unit(p, hy, u_p, nunits, tiles, BM, BN);
    setup_pieces<BM, AKC, true, TAPS>(pa, p, up.m0, up.r_begin, tid);
    setup_pieces<BN, BKC, false, TAPS>(pb, p, up.n0, up.r_begin, tid);
    st.begin(p, up.r_begin, up.r_end, up.shift_z);
    nkt_p = up.nkt;
  };
  auto produce = [&](int stage) {
    if (u_p >= nunits) return;
    float* At = lds + stage * STAGE;
    st.template issue<BM, BN>(p, At, At + A_TILE, pa, pb, wave, tid);
    if (st.kt == nkt_p) {
      u_p += G;
      enter_unit();
    }
  };

  const int l31 = lane & 31, h = lane >> 5;
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
  RdAddr<BM, AKC> rda;
  RdAddr<BN, BKC> rdb;
  rda.setup(wm * (BM / 2), l31, h);
  rdb.setup(wn * (BN / 2), l31, h);

  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };

  // The accumulators are touched by non-MFMA code in exactly one place, after the inner K loop (as in the
  // one-tile-per-workgroup kernel).  [A flat loop over the stream with the epilogue under `if (kt == 0)` made
  // the compiler carry VGPR copies of all accumulators around every iteration, and its copy of the last
  // registers raced with the MFMA that writes them.]  Every unit has nkt >= 1 (the launcher rejects
  // split-K chunkings with empty slices).
  enter_unit();
  produce(0);
  int stage = 0;
  wait_vmcnt_barrier<0>();  // K-tile 0 of the first unit has landed for everybody
  produce(1);
  for (int u_c = blockIdx.x; u_c < nunits; u_c += G) {
    const Unit uc = decode_unit(p, hy, u_c, nunits, tiles, BM, BN);
    clear();
    float cs[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) cs[i] = 0.f;
    const bool do_cs = !AKC && !BKC && a.colsum != nullptr && uc.tile_n == 0 && uc.tapz == 0 && wn == 0;
    compute_ktile_any<BF, BM, BN, AKC, BKC>(acc, rda, rdb, lds0 + stage * (STAGE * 4), lds0 + stage * (STAGE * 4) + A_TILE * 4, cs, do_cs);
    stage ^= 1;
    for (int kt = 1; kt < uc.nkt; ++kt) {
      wait_vmcnt_barrier<0>();  // this K-tile has landed for everybody; the other stage is no longer read
      produce(stage ^ 1);
      compute_ktile_any<BF, BM, BN, AKC, BKC>(acc, rda, rdb, lds0 + stage * (STAGE * 4), lds0 + stage * (STAGE * 4) + A_TILE * 4, cs, do_cs);
      stage ^= 1;
    }
    if (u_c + G < nunits) {  // first K-tile of the next unit: its sync and the following DMA go ahead of the
      wait_vmcnt_barrier<0>();  // epilogue, whose stores then drain under that K-tile's MFMAs
      produce(stage ^ 1);
    }
    if (uc.slice >= 0)  // raw partial sums; the slab is addressed with the tile's absolute row numbers
      gemm_epilogue_impl<BM, BN, -1>(p, acc, hy.ws + uc.slice * hy.slab - (long long)hy.m_tail0 * a.Nc, a.Nc, uc.m0, uc.n0,
                                     wm, wn, lane);
    else
      gemm_epilogue<BM, BN>(p, acc, uc.m0, uc.n0, wm, wn, lane, uc.split, uc.tapz);
    if (do_cs) colsum_store<BM>(a, cs, uc.m0, wm, lane, uc.split);
  }
}

// Finishes the tail tiles of a split-tail launch whose epilogue is more than "+ bias": sums the S slices and applies
// gemm_epilogue_impl's arithmetic element by element (same order of operations, same element index m * ldc + n for the
// dropout mask, so forward and backward masks of a site agree whichever path wrote an element).
__global__ __launch_bounds__(256) void tail_fixup_epi_kernel(GemmP p, Hybrid hy) {
  const Fs2GemmArgs& a = p.a;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const int n4 = a.Nc >> 2;
  const long long total = (long long)(a.Mc - hy.m_tail0) * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / n4), n = (int)(i - (long long)r * n4) * 4, m = hy.m_tail0 + r;
    const float* src = hy.ws + (long long)r * a.Nc + n;
    float4 sum = *reinterpret_cast<const float4*>(src);
    for (int k = 1; k < hy.S; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(src + k * hy.slab);
      sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
    }
    const float4 b = a.bias ? *reinterpret_cast<const float4*>(a.bias + n) : make_float4(0, 0, 0, 0);
    float v[4] = {a.alpha * sum.x + b.x, a.alpha * sum.y + b.y, a.alpha * sum.z + b.z, a.alpha * sum.w + b.w};
    if (a.epi == FS2_EPI_ACT) {
      if (a.out_pre) *reinterpret_cast<float4*>(a.out_pre + (long long)m * a.ldpre + n) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fs2_act(a.act, v[e]);
    } else if (a.epi == FS2_EPI_DACT) {
      const float4 x = *reinterpret_cast<const float4*>(a.aux + (long long)m * a.ldaux + n);
      v[0] *= fs2_dact(a.act, x.x); v[1] *= fs2_dact(a.act, x.y); v[2] *= fs2_dact(a.act, x.z); v[3] *= fs2_dact(a.act, x.w);
    }
    if (a.epi > 0 && drop.on) {
      if ((a.ldc & 1) == 0) {  // n is a multiple of 4: two whole hash pairs
        float f[4];
        fs2_drop_quad(drop, (unsigned)(m * a.ldc + n), f);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= f[e];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= fs2_drop_factor(drop, (unsigned long long)(unsigned)(m * a.ldc + n + e));
      }
    }
    if (a.epi == FS2_EPI_RESID) {
      const float4 x = *reinterpret_cast<const float4*>(a.resid + (long long)m * a.ldr + n);
      v[0] = x.x + a.res_scale * v[0]; v[1] = x.y + a.res_scale * v[1]; v[2] = x.z + a.res_scale * v[2]; v[3] = x.w + a.res_scale * v[3];
    }
    *reinterpret_cast<float4*>(a.C + (long long)m * a.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

#define FS2_GO(AKC_, BKC_, TAPS_)                                                                             \
  do {                                                                                                        \
    if (a.operand_bf16 == 2) gemm2p_kernel<BM, BN, AKC_, BKC_, TAPS_, 2><<<grid, block, 0, s>>>(p, hy, nunits, tiles);  \
    else if (a.operand_bf16) gemm2p_kernel<BM, BN, AKC_, BKC_, TAPS_, 1><<<grid, block, 0, s>>>(p, hy, nunits, tiles);  \
    else gemm2p_kernel<BM, BN, AKC_, BKC_, TAPS_, 0><<<grid, block, 0, s>>>(p, hy, nunits, tiles);            \
  } while (0)

template <int BM, int BN, int WG_PER_CU, bool SPLIT_TAIL>
int launch_persistent(GemmP& p, int nz, hipStream_t s) {
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
  int nunits = (int)nunits_ll;
  const int slots = n_cu * WG_PER_CU;
  Hybrid hy{0, 0, 0, 0, nullptr, 0};
  if (SPLIT_TAIL) {
    // only where a separate pass can finish the tiles: one reduction range, rows of every tensor the epilogue touches
    // addressable as float4
    auto vec_ok = [](const float* ptr, int ld) { return ((uintptr_t)ptr % 16) == 0 && (ld % 4) == 0; };
    if (nz != 1 || a.splitk != 1 || a.colsum || (a.Nc % 4) || !a.workspace || !vec_ok(a.C, a.ldc) ||
        (a.bias && ((uintptr_t)a.bias % 16)) || (a.out_pre && (a.epi != FS2_EPI_ACT || !vec_ok(a.out_pre, a.ldpre))) ||
        (a.epi == FS2_EPI_RESID && !vec_ok(a.resid, a.ldr)) || (a.epi == FS2_EPI_DACT && !vec_ok(a.aux, a.ldaux)) ||
        a.epi < FS2_EPI_STORE || a.epi > FS2_EPI_DACT)
      return FS2HIP_EINVAL;
    const int rem = tiles % slots;
    if (rem == 0 || rem * 5 >= slots * 4) return FS2HIP_EINVAL;  // no tail worth cutting: tiles 11/12 do this shape
    const int tail_mt = (rem + p.tiles_n - 1) / p.tiles_n;         // whole M-tile rows
    const int tail_tiles = tail_mt * p.tiles_n;
    if (tail_tiles > tiles) return FS2HIP_EINVAL;
    const int nkt = (a.R + BK2 - 1) / BK2;
    int S = slots / tail_tiles;
    // at least 4 K-tiles per slice; 8 when the second pass also has an epilogue to apply (K = 256 slices of 4 K-tiles
    // plus that pass measured slower inside the step than one-tile-per-workgroup kernels, although faster in isolation)
    const int min_kt = a.epi == FS2_EPI_STORE ? 4 : 8;
    if (S > nkt / min_kt) S = nkt / min_kt;
    if (S < 2) return FS2HIP_EINVAL;
    hy.S = S;
    hy.chunk = ((nkt + S - 1) / S) * BK2;
    if ((long long)(S - 1) * hy.chunk >= a.R) return FS2HIP_EINVAL;
    hy.u_full = tiles - tail_tiles;
    hy.m_tail0 = (p.tiles_m - tail_mt) * BM;
    hy.slab = (long long)(a.Mc - hy.m_tail0) * a.Nc;
    hy.ws = a.workspace;
    if (hy.slab * S > a.workspace_floats || (hy.slab % 4) || ((uintptr_t)a.workspace % 16)) return FS2HIP_EINVAL;
    nunits = hy.u_full + tail_tiles * S;
  }
  dim3 grid(nunits < slots ? nunits : slots), block(256);
  int mode = TAPS_NONE;
  if (a.taps > 1) {
    if (a.shift_operand == 0) mode = (p.Rper % BK2 == 0) ? TAPS_RED : TAPS_GENERIC;
    else mode = a.T >= BK2 ? TAPS_ROWS : TAPS_GENERIC;
  }
  if (mode == TAPS_GENERIC) return FS2HIP_EINVAL;  // odd tap widths: gemm2.hip tile 7
  if (a.a_kcontig && a.b_kcontig) {
    if (mode == TAPS_RED) FS2_GO(true, true, TAPS_RED);
    else if (mode == TAPS_NONE) FS2_GO(true, true, TAPS_NONE);
    else return FS2HIP_EINVAL;
  } else if (a.a_kcontig && !a.b_kcontig) {
    if (mode == TAPS_RED) FS2_GO(true, false, TAPS_RED);
    else if (mode == TAPS_NONE) FS2_GO(true, false, TAPS_NONE);
    else return FS2HIP_EINVAL;
  } else if (!a.a_kcontig && !a.b_kcontig) {
    if (mode == TAPS_ROWS) FS2_GO(false, false, TAPS_ROWS);
    else if (mode == TAPS_NONE) FS2_GO(false, false, TAPS_NONE);
    else return FS2HIP_EINVAL;
  } else {
    return FS2HIP_EINVAL;
  }
  FS2_LAUNCH_CHECK();
  if (hy.S) {
    if (a.epi == FS2_EPI_STORE)
      return fs2_tail_fixup(hy.ws, hy.S, hy.slab, a.C, a.ldc, a.bias, a.alpha, hy.m_tail0, a.Mc, a.Nc, s);
    const long long total = (long long)(a.Mc - hy.m_tail0) * (a.Nc >> 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    tail_fixup_epi_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p, hy);
    FS2_LAUNCH_CHECK();
  }
  return 0;
}

#undef FS2_GO

}  // namespace

// tile ids 10-12: persistent 64x64 (4 workgroups / CU), 128x64 (3), 128x128 (2); 13/14/15: 128x128 / 128x64 / 64x64 + split tail
int fs2_gemm2p_launch(GemmP& p, int tile, int nz, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  const int chunk = (a.R + a.splitk - 1) / a.splitk;
  p.r_chunk = ((chunk + BK2 - 1) / BK2) * BK2;
  if ((long long)(a.splitk - 1) * p.r_chunk >= a.R) return FS2HIP_EINVAL;  // an empty split-K slice
  if (!fs2_gemm2_offsets_fit(a)) return FS2HIP_EINVAL;
  switch (tile) {
    case 10: return launch_persistent<64, 64, 4, false>(p, nz, s);  // 128 VGPRs: 4 waves per SIMD
    case 11: return launch_persistent<128, 64, 3, false>(p, nz, s);
    case 12: return launch_persistent<128, 128, 2, false>(p, nz, s);
    case 13: return launch_persistent<128, 128, 2, true>(p, nz, s);
    case 14: return launch_persistent<128, 64, 3, true>(p, nz, s);
    case 15: return launch_persistent<64, 64, 4, true>(p, nz, s);
    default: return FS2HIP_EINVAL;
  }
}
