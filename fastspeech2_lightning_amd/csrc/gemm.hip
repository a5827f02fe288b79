// fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32), with the conv-tap
// addressing modes and fused epilogues described in include/fs2hip.h.
//
// Tiling (wave64): a 256-thread workgroup = 4 wavefronts in a 2x2 arrangement computes a
// BM x BN tile; every wavefront owns (BM/2)x(BN/2) as TM x TN accumulators of 32x32 (16 VGPRs
// each).  K advances in steps of BK = 16 through two LDS buffers; both operand tiles are kept
// k-major in LDS ([k][m] / [k][n]) so that the MFMA operand fetch (lane l: row l&31, k = l>>5)
// is a conflict-free ds_read_b32 of 32 consecutive floats per half-wave.  Global->LDS goes
// through registers (the loader applies the conv-tap row shift / zero fill and transposes
// k-contiguous sources on the way), issued one K-tile ahead of the MFMAs.
#include <cstdlib>

#include <string.h>

#include "gemm_common.h"

namespace {

constexpr int BK = 16;
constexpr int LDS_PAD = 4;

template <int BM, int BN>
struct Tile {
  static constexpr int TM = BM / 64;
  static constexpr int TN = BN / 64;
  static constexpr int LDA = BM + LDS_PAD;
  static constexpr int LDB = BN + LDS_PAD;
  static constexpr int A_F4 = BM * BK / 4 / 256;  // float4 per thread per tile
  static constexpr int B_F4 = BN * BK / 4 / 256;
};

// Loads one [BK x ROWS] operand tile into registers.
//  kcontig: source is [row][k] (ld), thread reads float4 along k at (row = i*64 + tid/4, k4 = tid%4)
//  else   : source is [k][row] (ld), thread reads float4 along rows at (k = i*8 + tid/32, row4 = tid%32)
template <int ROWS, int NF4>
__device__ __forceinline__ void load_tile(float4 (&reg)[NF4], const float* __restrict__ src, int ld,
                                          bool kcontig, int row0, int nrows, int k0, int klimit,
                                          int shift, int T, bool shift_rows, bool shift_k, int tid) {
  if (kcontig) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      int r = row0 + i * 64 + (tid >> 2);
      int k = k0 + (tid & 3) * 4;
      bool ok = r < nrows && k < klimit;
      int rs = r;
      if (shift_rows) {
        int t = r % T + shift;
        ok = ok && t >= 0 && t < T;
        rs = r + shift;
      }
      reg[i] = ok ? *reinterpret_cast<const float4*>(src + (long long)rs * ld + k) : make_float4(0, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      int k = k0 + i * (1024 / ROWS) + (tid / (ROWS / 4));
      int r = row0 + (tid % (ROWS / 4)) * 4;
      bool ok = k < klimit && r < nrows;
      int ks = k;
      if (shift_k) {
        int t = k % T + shift;
        ok = ok && t >= 0 && t < T;
        ks = k + shift;
      }
      reg[i] = ok ? *reinterpret_cast<const float4*>(src + (long long)ks * ld + r) : make_float4(0, 0, 0, 0);
    }
  }
}

template <int ROWS, int NF4, int LD>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const float4 (&reg)[NF4], bool kcontig, int tid) {
  if (kcontig) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      int r = i * 64 + (tid >> 2);
      int k = (tid & 3) * 4;
      lds[(k + 0) * LD + r] = reg[i].x;
      lds[(k + 1) * LD + r] = reg[i].y;
      lds[(k + 2) * LD + r] = reg[i].z;
      lds[(k + 3) * LD + r] = reg[i].w;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      int k = i * (1024 / ROWS) + (tid / (ROWS / 4));
      int r = (tid % (ROWS / 4)) * 4;
      *reinterpret_cast<float4*>(lds + k * LD + r) = reg[i];
    }
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  using TL = Tile<BM, BN>;
  constexpr int TM = TL::TM, TN = TL::TN, LDA = TL::LDA, LDB = TL::LDB;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (LDA + LDB)];
  float* const As0 = lds;                 // [2][BK][LDA]
  float* const Bs0 = lds + 2 * BK * LDA;  // [2][BK][LDB]

  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = fs2_xcd_remap(blockIdx.x, gridDim.x);  // tiles sharing an M-tile's A rows on one XCD's L2
  const int tile_m = wg / p.tiles_n, tile_n = wg % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // reduction range of this workgroup
  int tapz = 0, split = 0;
  if (a.shift_operand == 1 || a.splitk > 1) {
    tapz = blockIdx.z / a.splitk;
    split = blockIdx.z % a.splitk;
  }
  const int r_begin = split * p.r_chunk;
  const int r_end = min(a.R, r_begin + p.r_chunk);
  const int nkt = (r_end - r_begin + BK - 1) / BK;

  const bool a_shift_rows = a.shift_operand == 0 && a.taps > 1;
  const bool b_shift_k = a.shift_operand == 1 && a.taps > 1;
  const int shift_z = tapz * a.tap_mul + a.tap_add;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[TL::A_F4], rb[TL::B_F4];

  auto issue = [&](int kt) {
    int r0 = r_begin + kt * BK;
    int tap = 0, kin = r0;
    const float* Bsrc = a.B;
    int shift = shift_z;
    if (a_shift_rows) {
      tap = r0 / p.Rper;
      kin = r0 - tap * p.Rper;
      shift = tap * a.tap_mul + a.tap_add;
      Bsrc += (long long)tap * a.b_tap_stride;
    }
    // A operand: kcontig -> [Mc][Rper]; else [R][Mc]
    load_tile<BM, TL::A_F4>(ra, a.A, a.lda, a.a_kcontig != 0, m0, a.Mc, a_shift_rows ? kin : r0,
                            a_shift_rows ? p.Rper : r_end, shift, a.T, a_shift_rows, false, tid);
    load_tile<BN, TL::B_F4>(rb, Bsrc, a.ldb, a.b_kcontig != 0, n0, a.Nc, a_shift_rows ? kin : r0,
                            a_shift_rows ? p.Rper : r_end, shift, a.T, false, b_shift_k, tid);
  };

  if (nkt > 0) {
    issue(0);
    store_tile<BM, TL::A_F4, LDA>(As0, ra, a.a_kcontig != 0, tid);
    store_tile<BN, TL::B_F4, LDB>(Bs0, rb, a.b_kcontig != 0, tid);
  }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) issue(kt + 1);
    const float* Ab = As0 + cur * (BK * LDA) + wm * (BM / 2) + (lane & 31);
    const float* Bb = Bs0 + cur * (BK * LDB) + wn * (BN / 2) + (lane & 31);
    const int kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = Ab[(kk * 2 + kh) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bb[(kk * 2 + kh) * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      store_tile<BM, TL::A_F4, LDA>(As0 + (cur ^ 1) * (BK * LDA), ra, a.a_kcontig != 0, tid);
      store_tile<BN, TL::B_F4, LDB>(Bs0 + (cur ^ 1) * (BK * LDB), rb, a.b_kcontig != 0, tid);
    }
    __syncthreads();
    cur ^= 1;
  }

  gemm_epilogue<BM, BN>(p, acc, m0, n0, wm, wn, lane, split, tapz);
}

}  // namespace

extern "C" int fs2hip_version(void) { return 1; }

namespace {
// which launcher takes a prepared GemmP: the bf16-storage core, the direct-to-LDS fp32 cores (tile >= 4), the
// register-staged core (tiles 1-3)
enum { ROUTE_B = 1, ROUTE_V2 = 2, ROUTE_V1 = 3 };
struct GemmRoute {
  int core, tile, nz;
};

// fs2hip_gemm's argument checks and derived fields (p.a normalised, Rper, staged, drop, r_chunk of the BK=16 core) and
// the choice of core and tile; nothing is enqueued
int gemm_prepare(const Fs2GemmArgs* args, GemmP& p, GemmRoute& route) {
  p.staged = 0;
#ifdef FS2_PROBES  // phase-ablation builds only (FS2_BUILD_PROBES=1 python -m fastspeech2_lightning_amd.build --force;
  // tools/probe_phases_bf16.sh): wrong results, timing only -- never in the shipped library
  static const int env_probe = getenv("FS2_GEMM_PROBE") ? atoi(getenv("FS2_GEMM_PROBE")) : 0;
  p.probe = env_probe;
#else
  p.probe = 0;
#endif
  p.a = *args;
  Fs2GemmArgs& a = p.a;
  if (a.Mc <= 0 || a.Nc <= 0 || a.R <= 0) return FS2HIP_EINVAL;
  if (a.taps < 1) a.taps = 1;
  if (a.splitk < 1) a.splitk = 1;
  if (a.operand_bf16 == 4) {  // bf16 in memory, any orientation: its own core and its own preconditions
    if (((uintptr_t)a.A % 16) || ((uintptr_t)a.B % 16) || (a.lda % 8) || (a.ldb % 8)) return FS2HIP_EINVAL;
    if ((a.a_kcontig || a.b_kcontig) && (a.R / (a.shift_operand == 0 && a.taps > 1 ? a.taps : 1)) % 8) return FS2HIP_EINVAL;
    if (!a.a_kcontig && (a.b_kcontig || a.lda < ((a.Mc + 7) / 8) * 8)) return FS2HIP_EINVAL;
    if (!a.b_kcontig && a.ldb < ((a.Nc + 7) / 8) * 8) return FS2HIP_EINVAL;
    if ((a.Nc % 4) || (a.ldc % 4) || ((uintptr_t)a.C % 16) || (a.bias && ((uintptr_t)a.bias % 16))) return FS2HIP_EINVAL;
    const bool wgrad = !a.a_kcontig && !a.b_kcontig;
    if (a.splitk > 1 && (!wgrad || !a.workspace || ((uintptr_t)a.workspace % 16) || (a.io_bf16 & 1))) return FS2HIP_EINVAL;
    if (a.colsum && !wgrad) return FS2HIP_EINVAL;
    p.Rper = a.R;
    if (a.taps > 1) {
      if (a.T <= 0) return FS2HIP_EINVAL;
      if (a.shift_operand == 0) {
        if (!a.a_kcontig || a.R % a.taps || a.Mc % a.T) return FS2HIP_EINVAL;
        p.Rper = a.R / a.taps;
      } else if (!wgrad || a.R % a.T) {
        return FS2HIP_EINVAL;
      }
    }
    if (a.epi == FS2_EPI_RESID && (!a.resid || (a.ldr % 4) || ((uintptr_t)a.resid % 16))) return FS2HIP_EINVAL;
    if (a.epi == FS2_EPI_DACT && (!a.aux || (a.ldaux % 4) || ((uintptr_t)a.aux % 16))) return FS2HIP_EINVAL;
    if (a.out_pre && ((a.ldpre % 4) || ((uintptr_t)a.out_pre % 16))) return FS2HIP_EINVAL;
    {  // 32-bit byte offsets everywhere: every tensor below 2 GiB
      const long long lim = 0x7fffffffLL - 65536;
      const long long a_rows = a.a_kcontig ? a.Mc : a.R, b_rows = a.b_kcontig ? a.Nc : a.R;
      if (2LL * (a_rows + 2LL * a.taps) * a.lda + 2LL * a.R >= lim) return FS2HIP_EINVAL;
      if (2LL * (b_rows + 2LL * a.taps) * a.ldb + 2LL * a.R + 2LL * a.taps * (a.b_tap_stride > 0 ? a.b_tap_stride : 0) >= lim)
        return FS2HIP_EINVAL;
      auto fits = [&](long long ld) { return 4LL * a.Mc * ld < lim; };
      if (!fits(a.splitk > 1 ? a.Nc : a.ldc) || (a.out_pre && !fits(a.ldpre)) || (a.epi == FS2_EPI_RESID && !fits(a.ldr)) ||
          (a.epi == FS2_EPI_DACT && !fits(a.ldaux)))
        return FS2HIP_EINVAL;
    }
    a.counters = nullptr;
    {
      const int per16 = (a.io_bf16 & 1) ? 8 : 4;  // elements per 16 bytes of the results
      p.staged = a.splitk > 1 ? (a.Nc % 4 == 0)
                              : ((a.ldc % per16) == 0 && (a.c_tap_stride % per16) == 0 &&
                                 (!a.out_pre || (a.ldpre % per16) == 0));
    }
    p.drop = fs2_make_drop(a.drop_p, a.drop_seed, a.drop_step);
    const int nzb = (a.shift_operand == 1 ? a.taps : 1) * a.splitk;
    int tb = a.tile;
    if (tb == 0) tb = (long long)((a.Mc + 127) / 128) * ((a.Nc + 127) / 128) * nzb >= 256 ? 20 : 23;
    if (tb >= 30 && nzb != 1) return FS2HIP_EINVAL;
    route = {ROUTE_B, tb, nzb};
    return 0;
  }
  // vector-load preconditions: the contiguous dimension of every operand is a multiple of 4
  // floats and rows start 16-byte aligned
  if ((a.lda % 4) || (a.ldb % 4)) return FS2HIP_EINVAL;
  if (((uintptr_t)a.A % 16) || ((uintptr_t)a.B % 16)) return FS2HIP_EINVAL;
  // (a reduction-major operand may have a row count that is not a multiple of 4 as long as its leading
  // dimension covers the rounded-up width: the float4 that straddles the edge stays inside the row)
  if (a.a_kcontig ? (a.R / (a.shift_operand == 0 ? a.taps : 1)) % 4 : a.lda < ((a.Mc + 3) / 4) * 4) return FS2HIP_EINVAL;
  if (a.b_kcontig ? (a.R / (a.shift_operand == 0 ? a.taps : 1)) % 4 : a.ldb < ((a.Nc + 3) / 4) * 4) return FS2HIP_EINVAL;
  if (a.operand_bf16 == 3) {
    // A and B hold bf16 (k-contiguous rows, leading dimensions and R in bf16 elements): from here on the reduction is
    // counted in 4-byte slots of two elements, which is all the loaders of core v2 need to know
    if (!a.a_kcontig || !a.b_kcontig || a.splitk != 1 || (a.R % 8) || (a.lda % 8) || (a.ldb % 8) || (a.b_tap_stride % 2) ||
        (a.taps > 1 && (a.shift_operand != 0 || (a.R / a.taps) % 64)))
      return FS2HIP_EINVAL;
    a.R /= 2;
    a.lda /= 2;
    a.ldb /= 2;
    a.b_tap_stride /= 2;
  } else if (a.operand_bf16 < 0 || a.operand_bf16 > 4) {
    return FS2HIP_EINVAL;
  }
  p.Rper = a.R;
  if (a.taps > 1) {
    if (a.T <= 0 || a.Mc <= 0) return FS2HIP_EINVAL;
    if (a.shift_operand == 0) {
      if (!a.a_kcontig || a.R % a.taps) return FS2HIP_EINVAL;
      p.Rper = a.R / a.taps;
      if (a.Mc % a.T) return FS2HIP_EINVAL;
    } else {
      if (a.a_kcontig || a.b_kcontig || a.R % a.T) return FS2HIP_EINVAL;
    }
  }
  if (a.splitk > 1 && (a.a_kcontig || a.b_kcontig || !a.workspace)) return FS2HIP_EINVAL;
  if (a.colsum && (a.a_kcontig || a.b_kcontig)) return FS2HIP_EINVAL;  // the bias gradient belongs to a weight gradient
  if (a.splitk <= 1) a.counters = nullptr;
  // (one arrival counter per output tile and tap; the smallest tile is 64 x 64)
  if (a.counters && (long long)a.taps * ((a.Mc + 63) / 64) * ((a.Nc + 63) / 64) > FS2_SPLITK_COUNTERS) return FS2HIP_EINVAL;
  if (a.epi == FS2_EPI_RESID && !a.resid) return FS2HIP_EINVAL;
  if (a.epi == FS2_EPI_DACT && !a.aux) return FS2HIP_EINVAL;
  {  // the epilogue addresses its tensors with 32-bit byte offsets (gemm_common.h): each must stay below 2 GiB
    auto fits = [&](long long ld) { return 4LL * a.Mc * ld < 0x7fffffffLL; };
    if (!fits(a.splitk > 1 ? a.Nc : a.ldc) || (a.out_pre && !fits(a.ldpre)) ||
        (a.epi == FS2_EPI_RESID && !fits(a.ldr)) || (a.epi == FS2_EPI_DACT && !fits(a.ldaux)))
      return FS2HIP_EINVAL;
  }
  p.drop = fs2_make_drop(a.drop_p, a.drop_seed, a.drop_step);
  int chunk = (a.R + a.splitk - 1) / a.splitk;
  p.r_chunk = ((chunk + BK - 1) / BK) * BK;
  const int nz = (a.shift_operand == 1 ? a.taps : 1) * a.splitk;
  static const int env_tile = getenv("FS2_GEMM_TILE") ? atoi(getenv("FS2_GEMM_TILE")) : 0;  // tuning aid
  int tile = a.tile ? a.tile : env_tile;
  // core v2 (direct-to-LDS, BK = 32) handles NT / NN / TN; core v1 (register-staged, BK = 16) additionally
  // needs conv taps aligned to its K-tile
  const bool v2_ok = a.a_kcontig || !a.b_kcontig;
  const bool v1_ok = !(a.taps > 1 && a.shift_operand == 0 && (p.Rper % BK));
  if (tile == 0) {
    // heuristic (the host autotuner normally picks): v2, 128x128 unless few tiles / a narrow or ragged N
    const bool narrow = a.Nc <= 64 || (a.Nc % 128 != 0 && a.Nc % 128 <= 64) ||
                        ((long long)((a.Mc + 127) / 128) * ((a.Nc + 127) / 128) * nz < 384);
    // conv taps that core v2 cannot keep uniform per K-tile run on its generic-decode kernel (tile 7 only)
    const bool odd_taps = a.taps > 1 && (a.shift_operand == 0 ? (p.Rper % 32) != 0 : a.T < 32);
    tile = v2_ok ? (odd_taps ? (v1_ok && !a.operand_bf16 && !a.colsum ? 3 : 7) : (narrow ? 5 : 4)) : (narrow ? 2 : 1);  // (core v1 is fp32 only)
  }
  if (a.operand_bf16 == 3 && (tile < 4 || tile > 9)) return FS2HIP_EINVAL;  // the one-tile-per-workgroup direct-to-LDS core only
  if (tile >= 4) {
    if (tile != 32 && !v2_ok) return FS2HIP_EINVAL;
    route = {ROUTE_V2, tile, nz};
    return 0;
  }
  if (!v1_ok || a.colsum) return FS2HIP_EINVAL;  // (the register-staged core does not sum columns)
  route = {ROUTE_V1, tile, nz};
  return 0;
}

int gemm_launch(GemmP& p, const GemmRoute& route, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  const int tile = route.tile, nz = route.nz;
  if (route.core == ROUTE_B) {
    if (tile == 33) return fs2_gemmws4_launch(p, s);
    if (tile >= 30) return fs2_gemmws_launch(p, tile, s);
    return fs2_gemmb_launch(p, tile, nz, s);
  }
  if (route.core == ROUTE_V2) {
    if (tile == 32) return fs2_gemmws32_launch(p, s);
    return tile >= 10 ? fs2_gemm2p_launch(p, tile, nz, s) : fs2_gemm2_launch(p, tile, nz, s);
  }
  if (tile == 3) {
    p.tiles_n = (a.Nc + 63) / 64;
    dim3 grid(((a.Mc + 63) / 64) * p.tiles_n, 1, nz);
    gemm_kernel<64, 64><<<grid, dim3(256), 0, s>>>(p);
  } else if (tile == 2) {
    p.tiles_n = (a.Nc + 63) / 64;
    dim3 grid(((a.Mc + 127) / 128) * p.tiles_n, 1, nz);
    gemm_kernel<128, 64><<<grid, dim3(256), 0, s>>>(p);
  } else {
    p.tiles_n = (a.Nc + 127) / 128;
    dim3 grid(((a.Mc + 127) / 128) * p.tiles_n, 1, nz);
    gemm_kernel<128, 128><<<grid, dim3(256), 0, s>>>(p);
  }
  FS2_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int fs2hip_gemm(const Fs2GemmArgs* args, void* stream) {
  if (!args) return FS2HIP_EINVAL;
  GemmP p;
  GemmRoute route;
  const int rc = gemm_prepare(args, p, route);
  if (rc != 0) return rc;
  return gemm_launch(p, route, (hipStream_t)stream);
}

extern "C" int fs2hip_gemm_grouped(const Fs2GemmArgs* args, int n, void* stream) {
  if (!args || n < 1 || n > FS2_GEMM_GROUP_MAX) return FS2HIP_EINVAL;
  if (n == 1) return fs2hip_gemm(args, stream);
  GemmPG g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  const bool stored = args[0].operand_bf16 == 4;
  int tile = args[0].tile;
  if (tile == 0) tile = stored ? 23 : 7;
  for (int i = 0; i < n; ++i) {
    Fs2GemmArgs m = args[i];
    // one kernel instance for all members: plain (tap-free) GEMMs of one orientation and operand type, on a tile the
    // grouped kernels are built for; no in-kernel split-K finish (the counters are per stream, not per member), no
    // dropout in the epilogue (not needed by any caller: the instance would have to carry it)
    if (m.operand_bf16 != args[0].operand_bf16 || (m.operand_bf16 != 0 && m.operand_bf16 != 4)) return FS2HIP_EINVAL;
    if ((m.a_kcontig != 0) != (args[0].a_kcontig != 0) || (m.b_kcontig != 0) != (args[0].b_kcontig != 0)) return FS2HIP_EINVAL;
    if (m.taps > 1 || m.counters || m.drop_p > 0.f) return FS2HIP_EINVAL;
    m.tile = tile;
    GemmRoute route;
    const int rc = gemm_prepare(&m, g.m[i], route);
    if (rc != 0) return rc;
    if (route.core != (stored ? ROUTE_B : ROUTE_V2) || route.tile != tile || route.nz != g.m[i].a.splitk) return FS2HIP_EINVAL;
  }
  return stored ? fs2_gemmb_launch_grouped(g, tile, (hipStream_t)stream) : fs2_gemm2_launch_grouped(g, tile, (hipStream_t)stream);
}
