// Weights-stationary streaming GEMM on the bf16-storage core for K = 1024: C[M][N] = epi(A[M][1024] . W[N][1024]^T),
// both operands k-contiguous bf16 -- the Conformer's second feed-forward GEMM (N = 256: residual + dropout epilogue) and,
// through the transposed weight mirror, the data gradient of the first one (N = 256, bf16 result).
//
// The tiled kernels run these shapes at 300-540 TFLOP/s: a 128 x 128 tile pulls in 64 flop per operand byte, a CU takes
// in ~70 GB/s from L2, and at K = 1024 that is 4.5 TFLOP/s per CU, half of what its matrix pipes do; the 336 x 2 tiles
// fill the 256 CUs 2.6 times (a partly filled last round); and the residual epilogue (44 MB read, 44 MB written) runs
// after the tile's MFMAs, not beside anything.  Here, as in gemm_ws.hip (K = 256):
//   * a workgroup is FOUR wavefronts, one per SIMD, and owns 128 output columns: wavefront w keeps
//     W[32 w .. 32 w + 31][0 .. 1023] = 64 MFMA fragments in 256 registers for the whole launch -- 224 of them
//     accumulation registers (AGPRs, which the MFMA reads as an operand directly: inline-asm MFMAs, the allocator is not
//     asked) and 32 vector registers -- so a byte of A feeds 128 flop, and only A streams;
//   * A streams through LDS as chunks of 32 rows x 256 k (16 KB, LDS-DMA, a ring of eight chunks, six in flight = 96 KB
//     per CU): a row tile is four chunks = 64 MFMAs per wavefront (2 048 matrix-pipe cycles) behind one epilogue;
//   * workgroups walk row streams: no tile rounds, W is read once per workgroup, the two column slices of a row stream
//     sit on one XCD and share its L2;
//   * vector-memory bookkeeping is by hand (counted s_waitcnt vmcnt, no compiler-visible memory access in the loop), and
//     so are the MFMA hazards the compiler cannot see through inline asm (wait states after the last MFMA of a tile).
// One wavefront per SIMD: the epilogue's vector work and the DMA issue are not hidden behind another wavefront's MFMAs; what
// is bought is the register file for W.  (A two-role form -- eight wavefronts, the two of a SIMD splitting K and handing the
// accumulator over through LDS, so that one always feeds the matrix pipe -- was built, is bit-identical, and runs exactly as
// fast: tools/probes/gemm_ws4_two_role.hip.txt.  Whatever bounds this kernel at ~3.7 TB/s of operand + result traffic, it is
// not what a single wavefront per SIMD fails to overlap.)
// Epilogues, dropout masks and rounding are those of gemm_bf16_core.h, element for element.
#include "gemm_bf16_core.h"
#include "gemm_ws_util.h"

namespace {

constexpr int W4_THREADS = 256, W4_WAVES = 4;
constexpr int W4_NST = 8, W4_AHEAD = 6;      // ring stages; chunks in flight
constexpr int W4_CHUNK_BYTES = 4 * 4096;     // [4 K-tiles][32 rows][64 bf16]
constexpr int W4_NA = 56;                    // fragments 0 .. 55 live in AGPRs, 56 .. 63 in VGPRs

// the MFMA with its W operand in an accumulation register / a vector register; `first` starts the chain from zero
// (inline constant: no 16-register clear; early-clobber: the result registers must not be an operand's).  The A fragment
// comes from a ds_read the wavefront has waited for.
template <bool AG, bool FIRST>
__device__ __forceinline__ void w4_mfma(f32x16& acc, const u32x4& w, const u32x4& af) {
  if constexpr (AG) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(af));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(af));
  } else {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(af));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(af));
  }
}
template <int OFF>
__device__ __forceinline__ void w4_ldw_a(u32x4& v, u32x4 r, int voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=a"(v) : "v"(voff), "s"(r), "n"(OFF) : "memory");
}

template <int EPI, bool OBF>
struct W4Counts {
  static constexpr int ES = OBF ? 2 : 4;
  static constexpr int ROWB = 32 * ES;  // bytes of a wavefront's row segment
  static constexpr int LPR = ROWB / 16, RPP = 64 / LPR;
  static constexpr int STORES = 32 / RPP;
  static constexpr int LOADS = EPI == FS2_EPI_RESID ? 4 : 0;
  static constexpr int DMA = 4;  // LDS-DMA pieces per thread and chunk (1 024 pieces, 256 threads)
};

template <int EPI, bool OBF, bool DROP>
__global__ __launch_bounds__(W4_THREADS) void gemmws4_kernel(GemmP p, int n_slices, int n_streams, int n_row_tiles) {
  typedef W4Counts<EPI, OBF> CT;
  constexpr int ES = CT::ES, ROWB = CT::ROWB, RS = ROWB + 16, LPR = CT::LPR, RPP = CT::RPP;
  constexpr int STG = 32 * RS;
  constexpr int RING = W4_NST * W4_CHUNK_BYTES, BIAS_OFF = RING + W4_WAVES * STG;
  __shared__ __attribute__((aligned(16))) char lds[BIAS_OFF + 128 * 4];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;  // (XCD, index inside it): the slices of a row stream share an XCD
  const int slice = bj % n_slices;
  const int stream = (bj / n_slices) * 8 + bx;
  const int ns0 = slice * 128, nw0 = ns0 + wave * 32;
  const int cnt = stream < n_row_tiles ? (n_row_tiles - 1 - stream) / n_streams + 1 : 0;

  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const unsigned stg = lds0 + RING + wave * STG;
  if (tid < 128) reinterpret_cast<float*>(lds + BIAS_OFF)[tid] = (a.bias && ns0 + tid < a.Nc) ? a.bias[ns0 + tid] : 0.f;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  __syncthreads();

  // ---- W fragments: fragment s, lane (l31, h) holds W[nw0 + l31][16 s + 8 h .. + 7]; rows past Nc read zeros -----------
  u32x4 wa[W4_NA], wv[64 - W4_NA];
  {
    const u32x4 rw = ws_rsrc(a.B, (unsigned)a.Nc * (unsigned)a.ldb * 2u);
    const int voff = ((nw0 + l31) * a.ldb + 8 * h) * 2;
#define W4_WA(S) w4_ldw_a<32 * (S)>(wa[S], rw, voff);
#define W4_WA8(S) W4_WA(S) W4_WA((S) + 1) W4_WA((S) + 2) W4_WA((S) + 3) W4_WA((S) + 4) W4_WA((S) + 5) W4_WA((S) + 6) W4_WA((S) + 7)
    W4_WA8(0) W4_WA8(8) W4_WA8(16) W4_WA8(24) W4_WA8(32) W4_WA8(40) W4_WA8(48)
#undef W4_WA8
#undef W4_WA
#define W4_WV(S) ws_ld128<32 * (W4_NA + (S))>(wv[S], rw, voff);
    W4_WV(0) W4_WV(1) W4_WV(2) W4_WV(3) W4_WV(4) W4_WV(5) W4_WV(6) W4_WV(7)
#undef W4_WV
    static_assert(W4_NA == 56, "the fragment load lists above");
  }

  // ---- A stream: chunk c = 4 * (row tile index) + kq; piece it * 256 + tid: K-tile it, row (tid >> 3) & 31, 16-byte chunk
  // tid & 7 (swizzled on the source side); the 128 bytes per K-tile and the 512 per kq travel in the scalar offset ---------
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.Mc * a.lda * 2, 0x00020000);
  const int avoff = (((tid >> 3) & 31) * a.lda + (((tid & 7) ^ ((tid >> 4) & 7)) << 3)) * 2;
  const int tile_stride = 32 * a.lda * 2;
  auto dma = [&](int c) {  // chunks past the end: zeros into a stage nobody reads (the counts stay constants)
    const int k = c >> 2, kq = c & 3;
    const bool real = k < cnt;
    const int soff = real ? (stream + k * n_streams) * tile_stride + kq * 512 : 0;
    const int vo = real ? avoff : B_OOB;
    char* dst = lds + (c % W4_NST) * W4_CHUNK_BYTES + (wave * 64) * 16;
    b_dma16(ra, vo, soff, dst);
    b_dma16(ra, vo, soff + 128, dst + 256 * 16);
    b_dma16(ra, vo, soff + 256, dst + 512 * 16);
    b_dma16(ra, vo, soff + 384, dst + 768 * 16);
  };
  unsigned ard[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) ard[g] = lds0 + l31 * 128 + (((2 * g + h) ^ ((l31 >> 1) & 7)) << 4);

  // ---- epilogue state (gemm_ws.hip's, one 32-column block per wavefront) ---------------------------------------------------
  f32x16 acc;
  const u32x4 rc = ws_rsrc(a.C, (unsigned)a.Mc * (unsigned)a.ldc * ES);
  const u32x4 rx = EPI == FS2_EPI_RESID ? ws_rsrc(a.resid, (unsigned)a.Mc * (unsigned)a.ldr * 4u) : ws_rsrc(a.C, 0u);
  u32x4 xq[4];
  const int rr = lane / LPR, cc = lane % LPR;
  const int ncol = nw0 + cc * (16 / ES);
  const unsigned bias_rd = lds0 + BIAS_OFF + (wave * 32 + 4 * h) * 4;
  const int xvoff = nw0 + 4 * h < a.Nc ? (l31 * a.ldr + nw0 + 4 * h) * 4 : B_OOB;
  const int cvoff = ncol + 16 / ES <= a.Nc ? (rr * a.ldc + ncol) * ES : B_OOB;
  auto load_x = [&](int k) {
    if constexpr (CT::LOADS != 0) {
      const int soff = (stream + k * n_streams) * 32 * a.ldr * 4;
      ws_ld128s<0>(xq[0], rx, xvoff, soff); ws_ld128s<32>(xq[1], rx, xvoff, soff);
      ws_ld128s<64>(xq[2], rx, xvoff, soff); ws_ld128s<96>(xq[3], rx, xvoff, soff);
    }
  };
  auto put = [&](int t, const float (&v)[4]) {
    const unsigned ad = stg + l31 * RS + 4 * h * ES + 8 * t * ES;
    if (OBF) {
      const u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      ws_dsw64(ad, w);
    } else {
      const f32x4 w = {v[0], v[1], v[2], v[3]};
      ws_dsw128(ad, __builtin_bit_cast(u32x4, w));
    }
  };
  auto epilogue = [&](int k) {
    const int m0 = (stream + k * n_streams) * 32;
    const int m = m0 + l31;
    u32x4 bq[4];
    ws_dsr128<0>(bq[0], bias_rd); ws_dsr128<32>(bq[1], bias_rd); ws_dsr128<64>(bq[2], bias_rd); ws_dsr128<96>(bq[3], bias_rd);
    b_lds_wait<0>();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      asm volatile("" : "+v"(bq[t]));
      const f32x4 b = __builtin_bit_cast(f32x4, bq[t]);
      const int n = nw0 + 8 * t + 4 * h;
      float q[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = a.alpha * acc[4 * t + e] + b[e];
      if constexpr (EPI > 0 && DROP) {
        float f[4];
        fs2_drop_quad(drop, (unsigned)(m * a.ldc + n), f);
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] *= f[e];
      }
      if constexpr (EPI == FS2_EPI_RESID) {
        const f32x4 xf = __builtin_bit_cast(f32x4, xq[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = xf[e] + a.res_scale * q[e];
      }
      put(t, q);
    }
    u32x4 wout[32 / RPP];
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ++ps) ws_dsr128<0>(wout[ps], stg + (ps * RPP + rr) * RS + cc * 16);
    b_lds_wait<0>();
#pragma unroll
    for (int ps = 0; ps < 32 / RPP; ++ps) {
      asm volatile("" : "+v"(wout[ps]));
      ws_st128(wout[ps], rc, cvoff, (m0 + ps * RPP) * a.ldc * ES);
    }
  };

  // sixteen K-steps of chunk kq (fragments 16 kq .. 16 kq + 15); the A fragment of step s + 2 is requested while step s
  // is in the MFMAs (three rotating buffers)
  u32x4 af[3];
#define W4_RD(S_) ws_dsr128<((S_) >> 2) * 4096>(af[(S_) % 3], ard[(S_) & 3] + sb);
#define W4_MM(KQ_, S_, PENDING)                                                                     \
  b_lds_wait<PENDING>();                                                                            \
  asm volatile("" : "+v"(af[(S_) % 3]));                                                            \
  if constexpr (16 * (KQ_) + (S_) < W4_NA)                                                          \
    w4_mfma<true, (KQ_) == 0 && (S_) == 0>(acc, wa[(16 * (KQ_) + (S_)) < W4_NA ? 16 * (KQ_) + (S_) : 0], af[(S_) % 3]); \
  else                                                                                              \
    w4_mfma<false, false>(acc, wv[(16 * (KQ_) + (S_)) >= W4_NA ? 16 * (KQ_) + (S_) - W4_NA : 0], af[(S_) % 3]);
#define W4_STEP(KQ_, S_) W4_RD((S_) + 2) W4_MM(KQ_, S_, 2)
#define W4_CHUNK(KQ_)                                                                                                   \
  {                                                                                                                     \
    W4_RD(0) W4_RD(1)                                                                                                   \
    W4_STEP(KQ_, 0) W4_STEP(KQ_, 1) W4_STEP(KQ_, 2) W4_STEP(KQ_, 3) W4_STEP(KQ_, 4) W4_STEP(KQ_, 5) W4_STEP(KQ_, 6)       \
    W4_STEP(KQ_, 7) W4_STEP(KQ_, 8) W4_STEP(KQ_, 9) W4_STEP(KQ_, 10) W4_STEP(KQ_, 11) W4_STEP(KQ_, 12) W4_STEP(KQ_, 13)   \
    W4_MM(KQ_, 14, 1) W4_MM(KQ_, 15, 0)                                                                                 \
  }

  // ---- the stream ------------------------------------------------------------------------------------------------------------
  // Vector-memory operations of a wavefront in issue order (D = 4 DMA pieces per chunk, L = operand loads and S = stores per
  // row tile); step (i, kq) = chunk c = 4 i + kq:
  //   [wait DMA(c)] barrier | D(c + AHEAD) | kq == 0: L(i) | MFMAs | kq == 3: wait L(i), S(i)
  // DMA(c) was issued in step c - 6.  Younger than it at the top of step c: DMA(c + 1 .. c + 5) and the L and S groups of
  // the six steps c - 6 .. c - 1, i.e. one of each plus a second L when kq is 1 or 2 and a second S when kq is 0 or 1.
  // The first six chunks are waited for outright ahead of the loop (the counted waits of their steps then wait for nothing
  // that matters), and from chunk 6 on the counts hold.
  constexpr int D = CT::DMA, L = CT::LOADS, S = CT::STORES;
  static_assert(W4_AHEAD == 6 && W4_NST >= W4_AHEAD + 2, "the counts below");
  static_assert(5 * D + 2 * L + 2 * S <= 63, "vmcnt range");
  ws_vmwait<0>();  // the W fragments
#pragma unroll
  for (int s = 0; s < W4_NA; ++s) asm volatile("" : "+a"(wa[s]));
#pragma unroll
  for (int s = 0; s < 64 - W4_NA; ++s) asm volatile("" : "+v"(wv[s]));
#pragma unroll
  for (int c = 0; c < W4_AHEAD; ++c) dma(c);
  ws_vmwait<0>();
  for (int i = 0; i < cnt; ++i) {
#define W4_TOP(KQ_)                                                                                           \
  ws_vmwait<5 * D + L * (1 + ((KQ_) == 1 || (KQ_) == 2)) + S * (1 + ((KQ_) == 0 || (KQ_) == 1))>();            \
  __builtin_amdgcn_s_barrier();                                                                               \
  dma(4 * i + (KQ_) + W4_AHEAD);
    {
      W4_TOP(0)
      load_x(i);
      const unsigned sb = ((4 * i) % W4_NST) * W4_CHUNK_BYTES;
      W4_CHUNK(0)
    }
    {
      W4_TOP(1)
      const unsigned sb = ((4 * i + 1) % W4_NST) * W4_CHUNK_BYTES;
      W4_CHUNK(1)
    }
    {
      W4_TOP(2)
      const unsigned sb = ((4 * i + 2) % W4_NST) * W4_CHUNK_BYTES;
      W4_CHUNK(2)
    }
    {
      W4_TOP(3)
      const unsigned sb = ((4 * i + 3) % W4_NST) * W4_CHUNK_BYTES;
      W4_CHUNK(3)
    }
#undef W4_TOP
    // the last MFMA's result is read by vector instructions next: 8 passes -> wait states the compiler cannot count
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    asm volatile("" : "+v"(acc));
    if (L) {
      ws_vmwait<3 * D>();  // younger than L(i): the DMA pieces of steps kq = 1, 2, 3
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(xq[t]));
    }
    epilogue(i);
  }
#undef W4_CHUNK
#undef W4_STEP
#undef W4_MM
#undef W4_RD
  ws_vmwait<0>();  // nothing of this wavefront may still be writing LDS or memory when the workgroup's LDS is released
}

template <int EPI, bool OBF>
void go4(GemmP& p, dim3 grid, int n_slices, int n_streams, int n_row_tiles, hipStream_t s) {
  if (EPI > 0 && p.drop.on) gemmws4_kernel<EPI, OBF, (EPI > 0)><<<grid, dim3(W4_THREADS), 0, s>>>(p, n_slices, n_streams, n_row_tiles);
  else gemmws4_kernel<EPI, OBF, false><<<grid, dim3(W4_THREADS), 0, s>>>(p, n_slices, n_streams, n_row_tiles);
}

}  // namespace

// tile id 33: forward orientation (both operands k-contiguous bf16), K = 1024, no conv taps, no split, store or
// residual epilogue, whole 16-byte chunks in every output row
int fs2_gemmws4_launch(GemmP& p, hipStream_t s) {
  const Fs2GemmArgs& a = p.a;
  if (!a.a_kcontig || !a.b_kcontig || a.taps != 1 || a.splitk != 1 || a.R != 1024 || a.colsum || a.out_pre) return FS2HIP_EINVAL;
  if (a.epi != FS2_EPI_STORE && a.epi != FS2_EPI_RESID) return FS2HIP_EINVAL;
  const bool obf = (a.io_bf16 & 1) != 0;
  const int per16 = obf ? 8 : 4;
  if ((a.Nc % per16) || (a.ldc % per16) || ((uintptr_t)a.C % 16) || (a.lda % 8) || (a.ldb % 8) || ((uintptr_t)a.A % 16) ||
      ((uintptr_t)a.B % 16))
    return FS2HIP_EINVAL;
  if (a.epi == FS2_EPI_RESID && ((a.ldr % 4) || ((uintptr_t)a.resid % 16))) return FS2HIP_EINVAL;
  if ((long long)(a.Mc + 64) * a.lda * 2 >= 0x7fffffffLL || (long long)(a.Nc + 64) * a.ldb * 2 >= 0x7fffffffLL) return FS2HIP_EINVAL;
  if ((long long)(a.Mc + 64) * a.ldc * 4 >= 0x7fffffffLL || (long long)(a.Mc + 64) * a.ldr * 4 >= 0x7fffffffLL) return FS2HIP_EINVAL;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return FS2HIP_EINVAL;
    n_cu = prop.multiProcessorCount;
  }
  const int n_slices = (a.Nc + 127) / 128;
  const int per_xcd = n_cu / 8;  // one workgroup per CU (150 KB of LDS, 512 registers per lane)
  if (per_xcd < 1 || n_slices > per_xcd) return FS2HIP_EINVAL;
  const int n_row_tiles = (a.Mc + 31) / 32;
  int streams_per_xcd = per_xcd / n_slices;
  while (streams_per_xcd > 1 && (streams_per_xcd - 1) * 8 >= n_row_tiles) --streams_per_xcd;
  const int n_streams = streams_per_xcd * 8;
  dim3 grid(8 * streams_per_xcd * n_slices);
  if (a.epi == FS2_EPI_RESID) {
    if (obf) go4<FS2_EPI_RESID, true>(p, grid, n_slices, n_streams, n_row_tiles, s);
    else go4<FS2_EPI_RESID, false>(p, grid, n_slices, n_streams, n_row_tiles, s);
  } else {
    if (obf) go4<0, true>(p, grid, n_slices, n_streams, n_row_tiles, s);
    else go4<0, false>(p, grid, n_slices, n_streams, n_row_tiles, s);
  }
  FS2_LAUNCH_CHECK();
  return 0;
}
