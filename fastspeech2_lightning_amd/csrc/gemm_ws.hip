// Weights-stationary streaming GEMM on the bf16-storage core: C[M][N] = epi(A[M][K] . W[N][K]^T), both operands
// k-contiguous bf16, K = 256 (the Conformer's model width: every QKV / out / pointwise projection, the first
// feed-forward GEMM and -- through the transposed weight mirror -- their data gradients).
//
// Why another structure.  At the bf16 MFMA rate a tiled GEMM with K = 256 is bound by what a CU can pull in: a
// 128 x 64 tile re-reads its 64 KB of A rows for every one of the N / 64 column tiles and its slice of W for every
// one of the M / 128 row tiles (0.5 GB of L2 -> LDS traffic for the 43 008 x 1024 x 256 feed-forward GEMM, 20 MB of
// operands), four K-tiles of MFMAs never amortise a workgroup's prologue, and the load, MFMA and store phases of a
// tile add up instead of overlapping (phase ablation).  Here
//   * a workgroup (8 wavefronts, one per CU) keeps its slice of W -- 256 or 512 output columns x K -- in REGISTERS
//     as MFMA fragments for the whole launch (128 registers per lane at 64 columns per wavefront), read once from L2;
//   * A streams through LDS in row tiles of 32 rows x K (16 KB, LDS-DMA, a ring of three tiles, two in flight), read
//     by all eight wavefronts: per row tile and wavefront 16 ds_read_b128 and 16 x NB MFMAs, one barrier;
//   * the CU's ingest is A alone (8 bytes per cycle against ~14 from the Infinity Cache), results leave through a
//     wavefront-private LDS staging region as whole-row stores, and nothing is re-fetched per tile;
//   * the two wavefronts of a SIMD run half a period apart (wavefronts 0-3: MFMAs, then their epilogue; 4-7: the
//     previous tile's epilogue, then MFMAs), so one's SiLU / dropout vector work runs beside the other's MFMAs;
//   * vector-memory bookkeeping is by hand: operand loads, LDS-DMA pieces and stores retire in issue order, every wait
//     is a counted s_waitcnt vmcnt(N) and no compiler-visible load or LDS access exists inside the loop (hipcc would
//     drain the DMA ring with vmcnt(0) in front of each).  Every wavefront issues the same operations for every tile
//     (tiles past the end are issued with out-of-range offsets: zeros into a free stage, dropped stores), so the counts
//     are compile-time constants.
// Epilogues, dropout masks and rounding are those of gemm_bf16_core.h, element for element.
#include <type_traits>

#include "gemm_bf16_core.h"
#include "gemm_ws_util.h"

namespace {

constexpr int WS_THREADS = 512, WS_WAVES = 8, WS_NST = 3;
constexpr int WS_KT = 4;                     // K = 256: four 64-deep K-tile images per row tile
constexpr int WS_TILE_BYTES = WS_KT * 4096;  // [KT][32 rows][64 bf16]

// vector-memory operations of one row tile, per wavefront (compile-time: the waits are counted)
template <int NB, int EPI, bool OBF, bool TWO>
struct WsCounts {
  static constexpr int ES = OBF ? 2 : 4;
  static constexpr int ROWB = 32 * NB * ES;  // bytes of a wavefront's row segment
  static constexpr int LPR = ROWB / 16, RPP = 64 / LPR;
  static constexpr int STORES = (32 / RPP) * (TWO ? 2 : 1);
  static constexpr int LOADS = (EPI == FS2_EPI_RESID || EPI == FS2_EPI_DACT) ? 4 * NB : 0;
  static constexpr int DMA = WS_KT * 256 / WS_THREADS;  // LDS-DMA pieces per thread and row tile
};

// NB: 32-column blocks per wavefront (a workgroup covers 256 * NB output columns).
// EPI / ACT / OBF (bf16 results) / TWO (pre-activation output) / AUXB (bf16 act' operand) / DROP (dropout on):
// compile-time epilogue.
template <int NB, int EPI, int ACT, bool OBF, bool TWO, bool AUXB, bool DROP>
__global__ __launch_bounds__(WS_THREADS, (NB == 1 && OBF) ? 4 : 2) void gemmws_kernel(GemmP p, int n_slices, int n_streams,
                                                                                        int n_row_tiles) {
  typedef WsCounts<NB, EPI, OBF, TWO> CT;
  constexpr int ES = CT::ES, ROWB = CT::ROWB, RS = ROWB + 16, LPR = CT::LPR, RPP = CT::RPP;
  constexpr int STG = 32 * RS;  // staging bytes per wavefront
  constexpr bool DELAY_PRE = !(NB == 1 && OBF);
  constexpr int RING = WS_NST * WS_TILE_BYTES, BIAS_OFF = RING + WS_WAVES * STG;
  __shared__ __attribute__((aligned(16))) char lds[BIAS_OFF + 256 * NB * 4];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // workgroups that share a row stream (same A rows, different column slices) sit on one XCD: workgroups are dealt
  // round-robin over the XCDs, so the XCD is blockIdx & 7 and the index inside it blockIdx >> 3
  const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
  const int slice = bj % n_slices;
  const int stream = (bj / n_slices) * 8 + bx;
  const int ns0 = slice * 256 * NB;         // the workgroup's first output column
  const int nw0 = ns0 + wave * 32 * NB;     // this wavefront's
  const int cnt = stream < n_row_tiles ? (n_row_tiles - 1 - stream) / n_streams + 1 : 0;  // row tiles of this workgroup

  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const unsigned stg = lds0 + RING + wave * STG;

  // bias of the workgroup's columns -> LDS (before any DMA is in flight: these are ordinary accesses)
  for (int c = tid; c < 256 * NB; c += WS_THREADS)
    reinterpret_cast<float*>(lds + BIAS_OFF)[c] = (EPI >= 0 && a.bias && ns0 + c < a.Nc) ? a.bias[ns0 + c] : 0.f;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  __syncthreads();

  // ---- W fragments: registers for the whole launch -------------------------------------------------------------------
  // fragment (s, j): lane (l31, h) holds W[nw0 + 32 j + l31][16 s + 8 h .. + 7]; rows past Nc read zeros (range check)
  u32x4 wf[16][NB];
  {
    const u32x4 rw = ws_rsrc(a.B, (unsigned)a.Nc * (unsigned)a.ldb * 2u);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int voff = ((nw0 + 32 * j + l31) * a.ldb + 8 * h) * 2;
#define WS_W(S) ws_ld128<32 * (S)>(wf[S][j], rw, voff);
      WS_W(0) WS_W(1) WS_W(2) WS_W(3) WS_W(4) WS_W(5) WS_W(6) WS_W(7)
      WS_W(8) WS_W(9) WS_W(10) WS_W(11) WS_W(12) WS_W(13) WS_W(14) WS_W(15)
#undef WS_W
    }
  }

  // ---- A stream: LDS-DMA pieces of this thread ----------------------------------------------------------------------
  // piece q = it * 512 + tid of a row tile: K-tile q >> 8, row (q >> 3) & 31, 16-byte chunk q & 7 (swizzled on the
  // source side); the LDS image [KT][32][128 B] is piece-linear.  Rows past Mc read zeros (num_records = the operand).
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.Mc * a.lda * 2, 0x00020000);
  // (piece it * 512 + tid: the K-tile index grows by 2 per `it`, i.e. 256 bytes along the row)
  static_assert(CT::DMA == 2, "two pieces per thread");
  const int avoff = (((tid >> 3) & 31) * a.lda + (tid >> 8) * 64 + (((tid & 7) ^ ((tid >> 4) & 7)) << 3)) * 2;
  const int tile_stride = 32 * a.lda * 2;  // bytes between row tiles
  auto dma = [&](int k) {  // k-th row tile of this workgroup -> stage k % NST (k >= cnt: zeros into a stage nobody reads)
    const bool real = k < cnt;
    const int soff = real ? (stream + k * n_streams) * tile_stride : 0;
    char* dst = lds + (k % WS_NST) * WS_TILE_BYTES;
    const int vo = real ? avoff : B_OOB;
    // (the 256 bytes travel in the scalar offset: an instruction offset would move the LDS destination as well)
    b_dma16(ra, vo, soff, dst + (wave * 64) * 16);
    b_dma16(ra, vo, soff + 256, dst + (WS_THREADS + wave * 64) * 16);
  };

  // A fragment reads: K-step s = 4 kt + g reads chunk 2 g + h of K-tile kt
  unsigned ard[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) ard[g] = lds0 + l31 * 128 + (((2 * g + h) ^ ((l31 >> 1) & 7)) << 4);

  // ---- epilogue state ------------------------------------------------------------------------------------------------
  f32x16 acc[NB];
  const u32x4 rc = ws_rsrc(a.C, (unsigned)a.Mc * (unsigned)a.ldc * ES);
  const u32x4 rp = ws_rsrc(TWO ? (const void*)a.out_pre : (const void*)a.C, TWO ? (unsigned)a.Mc * (unsigned)a.ldpre * ES : 0u);
  constexpr int XES = (EPI == FS2_EPI_DACT && AUXB) ? 2 : 4;
  const u32x4 rx = EPI == FS2_EPI_RESID ? ws_rsrc(a.resid, (unsigned)a.Mc * (unsigned)a.ldr * 4u)
                   : EPI == FS2_EPI_DACT ? ws_rsrc(a.aux, (unsigned)a.Mc * (unsigned)a.ldaux * XES)
                                         : ws_rsrc(a.C, 0u);
  const int ldx = EPI == FS2_EPI_RESID ? a.ldr : a.ldaux;
  // residual / act' operand quads of the tile whose epilogue comes next: written by loads the compiler does not see,
  // straight into the registers they are used from (a copy made before the wait would copy stale registers)
  typedef typename std::conditional<XES == 4, u32x4, u32x2>::type xq_t;
  xq_t xq[NB][4];
  const int rr = lane / LPR, cc = lane % LPR;  // on the way out: row inside a pass, 16-byte chunk
  const int ncol = nw0 + cc * (16 / ES);
  const unsigned bias_rd = lds0 + BIAS_OFF + (wave * 32 * NB + 4 * h) * 4;

  // per-lane byte offsets inside a row tile (constants of the launch; a column past Nc carries the out-of-range
  // sentinel, rows past Mc fall to the resources' range checks): the tile's own offset is scalar
  int xvoff[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = nw0 + 32 * j + 4 * h;
    xvoff[j] = n < a.Nc ? (l31 * ldx + n) * XES : B_OOB;
  }
  const int cvoff = ncol + 16 / ES <= a.Nc ? (rr * a.ldc + ncol) * ES : B_OOB;
  const int pvoff = cvoff;  // (the launcher requires ldpre == ldc)
  auto load_x = [&](int k) {  // operand quads for the epilogue of the k-th row tile, issued ahead of its MFMAs
    if constexpr (CT::LOADS != 0) {
      const int soff = (stream + k * n_streams) * 32 * ldx * XES;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if constexpr (XES == 4) {
          ws_ld128s<0>(xq[j][0], rx, xvoff[j], soff); ws_ld128s<32>(xq[j][1], rx, xvoff[j], soff);
          ws_ld128s<64>(xq[j][2], rx, xvoff[j], soff); ws_ld128s<96>(xq[j][3], rx, xvoff[j], soff);
        } else {
          ws_ld64s<0>(xq[j][0], rx, xvoff[j], soff); ws_ld64s<16>(xq[j][1], rx, xvoff[j], soff);
          ws_ld64s<32>(xq[j][2], rx, xvoff[j], soff); ws_ld64s<48>(xq[j][3], rx, xvoff[j], soff);
        }
      }
    }
  };

  auto put = [&](int j, int t, const float (&v)[4]) {
    const unsigned ad = stg + l31 * RS + 4 * h * ES + (32 * j + 8 * t) * ES;
    if (OBF) {
      const u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      ws_dsw64(ad, w);
    } else {
      const f32x4 w = {v[0], v[1], v[2], v[3]};
      ws_dsw128(ad, __builtin_bit_cast(u32x4, w));
    }
  };
  // acc = alpha * acc + bias, in place; the bias quads come from LDS one quad ahead of their use (two rotating registers)
  auto add_bias = [&]() {
    if constexpr (EPI >= 0) {
      u32x4 bq[2];
      ws_dsr128<0>(bq[0], bias_rd);
#define WS_BIAS(Q_)                                                                                        \
  if constexpr ((Q_) + 1 < 4 * NB) ws_dsr128<32 * (((Q_) + 1) & 3)>(bq[((Q_) + 1) & 1], bias_rd + 128 * (((Q_) + 1) >> 2)); \
  if constexpr ((Q_) + 1 < 4 * NB) b_lds_wait<1>(); else b_lds_wait<0>();                                  \
  {                                                                                                        \
    asm volatile("" : "+v"(bq[(Q_) & 1]));                                                                 \
    const f32x4 b = __builtin_bit_cast(f32x4, bq[(Q_) & 1]);                                               \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                          \
      acc[(Q_) >> 2][4 * ((Q_) & 3) + e] = a.alpha * acc[(Q_) >> 2][4 * ((Q_) & 3) + e] + b[e];            \
  }
      WS_BIAS(0) WS_BIAS(1) WS_BIAS(2) WS_BIAS(3)
      if constexpr (NB == 2) { WS_BIAS(4) WS_BIAS(5) WS_BIAS(6) WS_BIAS(7) }
#undef WS_BIAS
    }
  };
  // the two halves of a flush: the staged rows are requested from LDS; later -- after other work has covered the LDS
  // round trip -- they are stored (LDS serves a wavefront's accesses in order: puts issued in between land behind the reads)
  auto flush_begin = [&](u32x4 (&w)[32 / RPP]) {
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ++ps) ws_dsr128<0>(w[ps], stg + (ps * RPP + rr) * RS + cc * 16);
  };
  auto flush_end = [&](u32x4 (&w)[32 / RPP], u32x4 r, int ld, int voff, int m0) {
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ++ps) {
      asm volatile("" : "+v"(w[ps]));
      ws_st128(w[ps], r, voff, (m0 + ps * RPP) * ld * ES);
    }
  };

  auto epilogue = [&](int k) {  // accumulators of the k-th row tile -> memory
    const int m0 = (stream + k * n_streams) * 32;
    const int m = m0 + l31;
    add_bias();
    u32x4 wpre[32 / RPP];
    if constexpr (TWO) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float v[4] = {acc[j][4 * t], acc[j][4 * t + 1], acc[j][4 * t + 2], acc[j][4 * t + 3]};
          put(j, t, v);
        }
      flush_begin(wpre);
      if constexpr (!DELAY_PRE) {  // (four wavefronts per SIMD: the others cover the round trip; the registers are worth more)
        b_lds_wait<0>();
        flush_end(wpre, rp, a.ldpre, pvoff, m0);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = nw0 + 32 * j + 8 * t + 4 * h;
        float q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = acc[j][4 * t + e];
        if constexpr (EPI == FS2_EPI_ACT) {
          if (OBF && TWO) {  // the activation sees what the backward pass will read back: the rounded pre-activation
            const unsigned w0 = pack_bf16x2(q[0], q[1]), w1 = pack_bf16x2(q[2], q[3]);
            q[0] = bf16_lo(w0); q[1] = bf16_hi(w0); q[2] = bf16_lo(w1); q[3] = bf16_hi(w1);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = act_b<ACT>(a.act, q[e]);
        } else if constexpr (EPI == FS2_EPI_DACT) {
          float x[4];
          if constexpr (XES == 2) {
            x[0] = bf16_lo(xq[j][t][0]); x[1] = bf16_hi(xq[j][t][0]); x[2] = bf16_lo(xq[j][t][1]); x[3] = bf16_hi(xq[j][t][1]);
          } else {  // (whole-vector cast: __builtin_bit_cast on a vector ELEMENT reads element 0 -- clang 19)
            const f32x4 xf = __builtin_bit_cast(f32x4, xq[j][t]);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = xf[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] *= dact_b<ACT>(a.act, x[e]);
        }
        if constexpr (EPI > 0 && DROP) {  // element index m * ldc + n, as everywhere else this mask is used; ldc is a
          float f[4];                     // multiple of 4 here, so a quad is two whole hash pairs
          fs2_drop_quad(drop, (unsigned)(m * a.ldc + n), f);
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] *= f[e];
        }
        if constexpr (EPI == FS2_EPI_RESID) {
          const f32x4 xf = __builtin_bit_cast(f32x4, xq[j][t]);
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = xf[e] + a.res_scale * q[e];
        }
        if constexpr (TWO && DELAY_PRE) {
          if (j == 0 && t == 1) {  // the pre-activation rows: their LDS reads have had two quads of arithmetic to come back
            b_lds_wait<0>();
            flush_end(wpre, rp, a.ldpre, pvoff, m0);
          }
        }
        put(j, t, q);
      }
    }
    u32x4 wout[32 / RPP];
    flush_begin(wout);
    b_lds_wait<0>();
    flush_end(wout, rc, a.ldc, cvoff, m0);
  };

  auto mfmas = [&](int k) {  // acc = A(row tile k) . W^T
    const unsigned sb = (k % WS_NST) * WS_TILE_BYTES;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // sixteen K-steps; the A fragment of step s + 2 is requested while step s is in the MFMAs (three rotating buffers)
    u32x4 af[3];
#define WS_RD(S_) ws_dsr128<((S_) >> 2) * 4096>(af[(S_) % 3], ard[(S_) & 3] + sb);
#define WS_MM(S_, PENDING)                                                                                       \
  b_lds_wait<PENDING>();                                                                                         \
  asm volatile("" : "+v"(af[(S_) % 3]));                                                                         \
  _Pragma("unroll") for (int j = 0; j < NB; ++j)                                                                 \
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[S_][j]),                       \
                                                     __builtin_bit_cast(bf16x8, af[(S_) % 3]), acc[j], 0, 0, 0);
    WS_RD(0) WS_RD(1)
#define WS_STEP(S_) WS_RD((S_) + 2) WS_MM(S_, 2)
    WS_STEP(0) WS_STEP(1) WS_STEP(2) WS_STEP(3) WS_STEP(4) WS_STEP(5) WS_STEP(6) WS_STEP(7)
    WS_STEP(8) WS_STEP(9) WS_STEP(10) WS_STEP(11) WS_STEP(12) WS_STEP(13)
    WS_MM(14, 1) WS_MM(15, 0)
#undef WS_STEP
#undef WS_RD
#undef WS_MM
  };

  // ---- the stream ----------------------------------------------------------------------------------------------------
  // Vector-memory operations of a wavefront in issue order (D = DMA pieces, L = operand loads, S = stores per tile):
  //   early wavefronts, interval i:  [wait DMA(i)] barrier | L(i) D(i+2) | MFMA(i) | wait L(i) | S(i)
  //   late  wavefronts, interval i:  [wait DMA(i)] barrier | wait L(i-1) | S(i-1) | L(i) D(i+2) | MFMA(i)
  // DMA(i), i >= 2, was issued in interval i - 2; younger than it at the top of interval i:
  //   early: S(i-2) L(i-1) D(i+1) S(i-1) = 2 S + L + D          late: S(i-2) L(i-1) D(i+1) = S + L + D
  // DMA(0) and DMA(1) are issued together ahead of the loop: at the top of intervals 0 and 1 at least D younger
  // operations exist (DMA(1); L(0) D(2) ...), so those two intervals wait with vmcnt(D).
  constexpr int D = CT::DMA, L = CT::LOADS, S = CT::STORES;
  static_assert(2 * S + L + D <= 63, "vmcnt range");
  ws_vmwait<0>();  // the W fragments
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(wf[s][j]));
  dma(0);
  dma(1);
  if (wave < 4) {
    for (int i = 0; i < cnt; ++i) {
      if (i < 2) ws_vmwait<D>(); else ws_vmwait<2 * S + L + D>();
      __builtin_amdgcn_s_barrier();
      load_x(i);
      dma(i + 2);
      mfmas(i);
      if (L) {
        ws_vmwait<D>();
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(xq[j][t]));
      }
      epilogue(i);
    }
  } else {
    for (int i = 0; i <= cnt; ++i) {
      if (i < cnt) {
        if (i < 2) ws_vmwait<D>(); else ws_vmwait<S + L + D>();
        __builtin_amdgcn_s_barrier();
      }
      if (i > 0) {
        if (L) {
          ws_vmwait<D>();  // younger than L(i-1): DMA(i+1) only
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(xq[j][t]));
        }
        epilogue(i - 1);
      }
      if (i < cnt) {
        load_x(i);
        dma(i + 2);
        mfmas(i);
      }
    }
  }
  ws_vmwait<0>();  // nothing of this wavefront may still be writing LDS or memory when the workgroup's LDS is released
}

#define WS_GO(NB_, EPI_, ACT_, OBF_, TWO_, AUXB_)                                                                       \
  do {                                                                                                                  \
    if ((EPI_) > 0 && p.drop.on)                                                                                        \
      gemmws_kernel<NB_, EPI_, ACT_, OBF_, TWO_, AUXB_, ((EPI_) > 0)><<<grid, block, 0, s>>>(p, n_slices, n_streams, n_row_tiles); \
    else                                                                                                                \
      gemmws_kernel<NB_, EPI_, ACT_, OBF_, TWO_, AUXB_, false><<<grid, block, 0, s>>>(p, n_slices, n_streams, n_row_tiles);   \
  } while (0)

template <int NB>
int launch_ws(GemmP& p, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return FS2HIP_EINVAL;
    n_cu = prop.multiProcessorCount;
  }
  const int n_slices = (a.Nc + 256 * NB - 1) / (256 * NB);
  // 32 columns per wavefront with bf16 results: 128 registers and 70 KB of LDS -- two workgroups per CU (four wavefronts
  // per SIMD: the LDS round trips and the arithmetic of one hide under the others')
  const int wgs_per_cu = (NB == 1 && (a.io_bf16 & 1)) ? 2 : 1;
  const int per_xcd = n_cu / 8 * wgs_per_cu;
  if (per_xcd < 1 || n_slices > per_xcd) return FS2HIP_EINVAL;
  const int n_row_tiles = (a.Mc + 31) / 32;
  int streams_per_xcd = per_xcd / n_slices;
  // few rows: no more streams than row tiles
  while (streams_per_xcd > 1 && (streams_per_xcd - 1) * 8 >= n_row_tiles) --streams_per_xcd;
  const int n_streams = streams_per_xcd * 8;
  dim3 grid(8 * streams_per_xcd * n_slices), block(WS_THREADS);
  const bool obf = (a.io_bf16 & 1) != 0, auxb = (a.io_bf16 & 2) != 0, two = a.out_pre != nullptr;
  // 64 columns per wavefront hold 128 registers of W: the instances that also carry operand quads or a second fp32
  // output do not fit the 256 registers of two wavefronts per SIMD (they would spill, and scratch traffic is
  // vector-memory traffic the counted waits know nothing about) -- those launches take the 32-column form or a tiled kernel
  if (NB == 2) {
    const bool drop_on = a.epi > 0 && p.drop.on;
    const bool fits = a.epi == FS2_EPI_STORE || (a.epi == FS2_EPI_ACT && (obf || !two)) ||
                      (a.epi == FS2_EPI_DACT && obf && auxb && drop_on && a.act == FS2_ACT_SILU);
    if (!fits) return FS2HIP_EINVAL;
  }
  switch (a.epi) {
    case FS2_EPI_ACT:
      if (a.act == FS2_ACT_SILU) {
        if (obf) { if (two) WS_GO(NB, FS2_EPI_ACT, FS2_ACT_SILU, true, true, false); else WS_GO(NB, FS2_EPI_ACT, FS2_ACT_SILU, true, false, false); }
        else { if (two) WS_GO(NB, FS2_EPI_ACT, FS2_ACT_SILU, false, true, false); else WS_GO(NB, FS2_EPI_ACT, FS2_ACT_SILU, false, false, false); }
      } else {
        if (obf) { if (two) WS_GO(NB, FS2_EPI_ACT, -1, true, true, false); else WS_GO(NB, FS2_EPI_ACT, -1, true, false, false); }
        else { if (two) WS_GO(NB, FS2_EPI_ACT, -1, false, true, false); else WS_GO(NB, FS2_EPI_ACT, -1, false, false, false); }
      }
      break;
    case FS2_EPI_RESID:
      if (obf) WS_GO(NB, FS2_EPI_RESID, -1, true, false, false); else WS_GO(NB, FS2_EPI_RESID, -1, false, false, false);
      break;
    case FS2_EPI_DACT:
      if (a.act == FS2_ACT_SILU) {
        if (obf) { if (auxb) WS_GO(NB, FS2_EPI_DACT, FS2_ACT_SILU, true, false, true); else WS_GO(NB, FS2_EPI_DACT, FS2_ACT_SILU, true, false, false); }
        else { if (auxb) WS_GO(NB, FS2_EPI_DACT, FS2_ACT_SILU, false, false, true); else WS_GO(NB, FS2_EPI_DACT, FS2_ACT_SILU, false, false, false); }
      } else {
        if (obf) { if (auxb) WS_GO(NB, FS2_EPI_DACT, -1, true, false, true); else WS_GO(NB, FS2_EPI_DACT, -1, true, false, false); }
        else { if (auxb) WS_GO(NB, FS2_EPI_DACT, -1, false, false, true); else WS_GO(NB, FS2_EPI_DACT, -1, false, false, false); }
      }
      break;
    default:
      if (obf) WS_GO(NB, 0, -1, true, false, false); else WS_GO(NB, 0, -1, false, false, false);
      break;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}
#undef WS_GO

}  // namespace

// tile ids 30 / 31: 64 / 32 output columns per wavefront (512 / 256 per workgroup).  Takes: forward orientation (both
// operands k-contiguous bf16), K = 256, no conv taps, no split, whole 16-byte chunks in every output row.
int fs2_gemmws_launch(GemmP& p, int tile, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  if (!a.a_kcontig || !a.b_kcontig || a.taps != 1 || a.splitk != 1 || a.R != 64 * WS_KT || a.colsum) return FS2HIP_EINVAL;
  const int per16 = (a.io_bf16 & 1) ? 8 : 4;
  if ((a.Nc % per16) || (a.ldc % per16) || ((uintptr_t)a.C % 16)) return FS2HIP_EINVAL;
  if (a.out_pre && (a.epi != FS2_EPI_ACT || a.ldpre != a.ldc || ((uintptr_t)a.out_pre % 16))) return FS2HIP_EINVAL;
  if ((long long)(a.Mc + 64) * a.lda * 2 >= 0x7fffffffLL || (long long)(a.Nc + 64) * a.ldb * 2 >= 0x7fffffffLL) return FS2HIP_EINVAL;
  if ((long long)(a.Mc + 64) * a.ldc * 4 >= 0x7fffffffLL) return FS2HIP_EINVAL;
  switch (tile) {
    case 30: return launch_ws<2>(p, s);
    case 31: return launch_ws<1>(p, s);
    default: return FS2HIP_EINVAL;
  }
}
