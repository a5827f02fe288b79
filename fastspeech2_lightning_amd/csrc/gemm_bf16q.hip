// Deep-ring persistent form of the bf16-storage GEMM core (gemm_bf16_core.h).
//
// Measured on the one-tile and two-stage persistent kernels (tools/bench_gemm_bf16.py, tools/bench_epilogue_bf16.py):
// this model's GEMMs are 4 .. 16 K-tiles long and, at the bf16 MFMA rate, wait for memory -- with two LDS stages a
// workgroup has ONE K-tile in flight and pays a memory latency per K-tile (a 41 472 x 1024 x 256 GEMM takes 52 us
// whatever it stores).  Here ONE workgroup per CU owns almost all of the LDS as a ring of NST stages and keeps NST - 1
// K-tiles in flight at all times, across tile boundaries and under its epilogues:
//  * waits are COUNTED (`s_waitcnt vmcnt(N)`, N = DMA pieces issued after the tile that is needed, plus the epilogue's
//    stores when they were issued after it): vector-memory operations retire in issue order, so waiting for fewer
//    would drain the ring.  N is a run-time value here (it depends on where the stream and the last epilogue are), so
//    the wait is a switch over immediates;
//  * nothing in an epilogue may make the compiler wait for a vector load behind the ring: the bias lives in LDS (copied
//    once per launch), the dropout record is resolved before the first DMA, the staging of whole-row stores is inline
//    assembly (a compiler-visible LDS access behind an LDS-DMA costs a vmcnt(0)), and every store is issued
//    unconditionally (out-of-range offsets instead of branches) so that their number is exact.  Epilogues that need a
//    vector load (residual, act' operand) do drain the ring once per unit -- by then most of it has landed.
#include "gemm_bf16_core.h"

namespace {

__device__ __forceinline__ void q_wait_vmcnt_barrier(int n) {
  switch (n) {
#define FS2_W(K) case K: asm volatile("s_waitcnt vmcnt(" #K ")\n\ts_barrier" ::: "memory"); break;
    FS2_W(0) FS2_W(1) FS2_W(2) FS2_W(3) FS2_W(4) FS2_W(5) FS2_W(6) FS2_W(7) FS2_W(8) FS2_W(9)
    FS2_W(10) FS2_W(11) FS2_W(12) FS2_W(13) FS2_W(14) FS2_W(15) FS2_W(16) FS2_W(17) FS2_W(18) FS2_W(19)
    FS2_W(20) FS2_W(21) FS2_W(22) FS2_W(23) FS2_W(24) FS2_W(25) FS2_W(26) FS2_W(27) FS2_W(28) FS2_W(29)
    FS2_W(30) FS2_W(31) FS2_W(32) FS2_W(33) FS2_W(34) FS2_W(35) FS2_W(36) FS2_W(37) FS2_W(38) FS2_W(39)
    FS2_W(40) FS2_W(41) FS2_W(42) FS2_W(43) FS2_W(44) FS2_W(45) FS2_W(46) FS2_W(47) FS2_W(48) FS2_W(49)
    FS2_W(50) FS2_W(51) FS2_W(52) FS2_W(53) FS2_W(54) FS2_W(55) FS2_W(56) FS2_W(57) FS2_W(58) FS2_W(59)
    FS2_W(60) FS2_W(61) FS2_W(62)
#undef FS2_W
    default: asm volatile("s_waitcnt vmcnt(63)\n\ts_barrier" ::: "memory"); break;  // (the counter's maximum)
  }
}

// LDS staging in inline assembly (invisible to the compiler's wait-count pass)
template <int OFF>
__device__ __forceinline__ void q_wr64(unsigned addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void q_wr128(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void q_lds_drain() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int Q_SGB = 32 * (128 + 16);  // staging region of one wavefront: one 32 x 32 fp32 accumulator, padded rows
constexpr int Q_BIAS = 1024;            // bias columns kept in LDS

// One unit's epilogue: every 32 x 32 accumulator goes through the wavefront's staging region and out as whole rows.
// Issues exactly TM * TN * (OBF ? 2 : 4) stores per output tensor (+ loads for RESID / DACT).
template <int BM, int BN, int EPI, bool OBF, int ACT>
__device__ __forceinline__ void epilogue_q(const GemmP& p, const Fs2Drop& drop, const f32x16 (&acc)[BM / 64][BN / 64], void* Cv,
                                           int ldc, int m0, int n0, int wm, int wn, int lane, unsigned stg, unsigned bias_lds) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int ES = OBF ? 2 : 4;
  constexpr int ROWB = 32 * ES, RS = ROWB + 16, LPR = ROWB / 16, RPP = 64 / LPR, PASSES = 32 / RPP;
  const Fs2GemmArgs& a = p.a;
  const int l31 = lane & 31, h = lane >> 5;
  const bool aux_bf = (a.io_bf16 & 2) != 0;
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, a.Mc * ldc * ES, 0x00020000);
  const bool two = EPI == FS2_EPI_ACT && a.out_pre != nullptr;
  const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_pre, 0, two ? a.Mc * a.ldpre * ES : 0, 0x00020000);
  const int rr = lane / LPR, cc = lane % LPR;
  const unsigned wr_base = stg + l31 * RS + 4 * h * ES;
  const unsigned rd_base = stg + rr * RS + cc * 16;

#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (BM / 2) + 32 * i + l31;
    const bool rowok = m < a.Mc;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nj = n0 + wn * (BN / 2) + 32 * j;
      float v[4][4];
      // bias quads from LDS (zero beyond Nc, and when there is no bias)
      u32x4 bq[4];
      if (EPI >= 0) {
        const unsigned ba = bias_lds + (nj + 4 * h) * 4;
        b_rd128<0>(bq[0], ba);
        b_rd128<32>(bq[1], ba);
        b_rd128<64>(bq[2], ba);
        b_rd128<96>(bq[3], ba);
        q_lds_drain();
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 b = EPI >= 0 ? __builtin_bit_cast(f32x4, bq[t]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[t][e] = EPI >= 0 ? a.alpha * acc[i][j][4 * t + e] + b[e] : acc[i][j][4 * t + e];
      }
      auto stage_and_store = [&](__amdgpu_buffer_rsrc_t r, int ld) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (OBF) {
            const u32x2 w = {pack_bf16x2(v[t][0], v[t][1]), pack_bf16x2(v[t][2], v[t][3])};
            if (t == 0) q_wr64<0>(wr_base, w);
            else if (t == 1) q_wr64<8 * ES>(wr_base, w);
            else if (t == 2) q_wr64<16 * ES>(wr_base, w);
            else q_wr64<24 * ES>(wr_base, w);
          } else {
            const f32x4 f = {v[t][0], v[t][1], v[t][2], v[t][3]};
            const u32x4 w = __builtin_bit_cast(u32x4, f);
            if (t == 0) q_wr128<0>(wr_base, w);
            else if (t == 1) q_wr128<8 * ES>(wr_base, w);
            else if (t == 2) q_wr128<16 * ES>(wr_base, w);
            else q_wr128<24 * ES>(wr_base, w);
          }
        }
        u32x4 w[PASSES];
        b_rd128<0>(w[0], rd_base);
        b_rd128<RPP * RS>(w[1], rd_base);
        if constexpr (PASSES == 4) {
          b_rd128<2 * RPP * RS>(w[2], rd_base);
          b_rd128<3 * RPP * RS>(w[3], rd_base);
        }
        q_lds_drain();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int mm = m0 + wm * (BM / 2) + 32 * i + ps * RPP + rr;
          const int ncol = nj + cc * (16 / ES);
          const bool ok = mm < a.Mc && ncol + 16 / ES <= a.Nc;
          __builtin_amdgcn_raw_buffer_store_b128(w[ps], r, ok ? (mm * ld + ncol) * ES : B_OOB, 0, 0);
        }
      };
      if (two) {
        stage_and_store(rp, a.ldpre);
        if (OBF) {  // the activation sees what the backward pass will read back: the rounded pre-activation
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const unsigned w0 = pack_bf16x2(v[t][0], v[t][1]), w1 = pack_bf16x2(v[t][2], v[t][3]);
            v[t][0] = bf16_lo(w0); v[t][1] = bf16_hi(w0); v[t][2] = bf16_lo(w1); v[t][3] = bf16_hi(w1);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = nj + 8 * t + 4 * h;
        const bool ok = rowok && n < a.Nc;
        float(&q)[4] = v[t];
        if (EPI == FS2_EPI_ACT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = act_b<ACT>(a.act, q[e]);
        } else if (EPI == FS2_EPI_DACT) {
          float x[4];
          if (aux_bf) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.Mc * a.ldaux * 2, 0x00020000);
            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rx, ok ? (m * a.ldaux + n) * 2 : B_OOB, 0, 0);
            x[0] = bf16_lo(w[0]); x[1] = bf16_hi(w[0]); x[2] = bf16_lo(w[1]); x[3] = bf16_hi(w[1]);
          } else {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, a.Mc * a.ldaux * 4, 0x00020000);
            const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (m * a.ldaux + n) * 4 : B_OOB, 0, 0));
            x[0] = w[0]; x[1] = w[1]; x[2] = w[2]; x[3] = w[3];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] *= dact_b<ACT>(a.act, x[e]);
        }
        if (EPI > 0 && drop.on) {
          const unsigned long long idx = (unsigned long long)(unsigned)(m * ldc + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] *= fs2_drop_factor(drop, idx + e);
        }
        if (EPI == FS2_EPI_RESID) {
          const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0, a.Mc * a.ldr * 4, 0x00020000);
          const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (m * a.ldr + n) * 4 : B_OOB, 0, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = x[e] + a.res_scale * q[e];
        }
      }
      stage_and_store(rc, ldc);
    }
  }
}

template <int BM, int BN, bool AKC, bool BKC, int TAPS, int NST, bool COLSUM>
__global__ __launch_bounds__(256) void gemmbq_kernel(GemmP p, int nunits, int tiles) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int L = BM / 32 + BN / 32;  // LDS-DMA instructions per thread per K-tile
  static_assert((NST - 2) * L + 2 * TM * TN * 4 + TM <= 63, "the wait counts fit the counter");
  __shared__ __attribute__((aligned(16))) char lds[NST * STAGE + 4 * Q_SGB + Q_BIAS * 4];
  const Fs2GemmArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);  // (reads the device step counter: before the first DMA)
  {  // the bias, once per launch, zero-padded to whole tiles
    float* bl = reinterpret_cast<float*>(lds + NST * STAGE + 4 * Q_SGB);
    for (int i = tid; i < Q_BIAS; i += 256) bl[i] = (a.bias && i < a.Nc) ? a.bias[i] : 0.f;
  }
  __syncthreads();
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)lds;
  const unsigned stg = lds0 + NST * STAGE + wave * Q_SGB;
  const unsigned bias_lds = lds0 + NST * STAGE + 4 * Q_SGB;

  // stores (and nothing else) that an epilogue leaves in flight, per wavefront: exact
  const bool obf = (a.io_bf16 & 1) != 0 && a.splitk <= 1;
  const bool two = a.splitk <= 1 && a.epi == FS2_EPI_ACT && a.out_pre != nullptr;
  const bool cs_on = COLSUM && a.colsum != nullptr;
  const int S = TM * TN * (obf ? 2 : 4) * (two ? 2 : 1) + (cs_on ? TM : 0);

  f32x16 acc[TM][TN];
  f32x16 cs[TM];

  // ---- producer ------------------------------------------------------------------------------------------------------
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
  auto produce = [&](int slot) -> bool {
    if (u_p >= nunits) return false;
    char* At = lds + slot * STAGE;
    st.template issue<BM, BN>(p, At, At + A_BYTES, pa, pb, wave, tid);
    if (st.kt == nkt_p) {
      u_p += G;
      enter_unit();
    }
    return true;
  };

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

  // ring state: slot_c = slot of the next K-tile to consume, slot_p = next slot to fill, inflight = K-tiles issued and not
  // yet consumed (the one being computed included while it is), spend = how many of the coming waits still have the last
  // epilogue's stores behind the tile they wait for
  int slot_c = 0, slot_p = 0, inflight = 0, spend = 0;
  enter_unit();
  for (int i = 0; i < NST - 1; ++i)
    if (produce(slot_p)) {
      slot_p = slot_p + 1 == NST ? 0 : slot_p + 1;
      ++inflight;
    }
  for (int u_c = blockIdx.x; u_c < nunits; u_c += G) {
    const UnitB uc = decode_unit_b(p, u_c, nunits, tiles, BM, BN);
    clear();
    const bool do_cs = cs_on && uc.tile_n == 0 && uc.tapz == 0;
    for (int kt = 0; kt < uc.nkt; ++kt) {
      q_wait_vmcnt_barrier((inflight - 1) * L + (spend > 0 ? S : 0));  // the tile has landed for everybody; the slot
      if (spend > 0) --spend;                                          // consumed one iteration ago is free
      if (produce(slot_p)) {
        slot_p = slot_p + 1 == NST ? 0 : slot_p + 1;
        ++inflight;
      }
      compute_ktile_b<BM, BN, AKC, BKC, COLSUM>(acc, cs, rda, rdb, lds0 + slot_c * STAGE, lds0 + slot_c * STAGE + A_BYTES, do_cs);
      slot_c = slot_c + 1 == NST ? 0 : slot_c + 1;
      --inflight;
    }
    // ---- epilogue: S stores per wavefront, issued behind every K-tile now in flight ----------------------------------
    if (a.splitk > 1) {
      float* slab = a.workspace + ((long long)uc.split * a.taps + uc.tapz) * ((long long)a.Mc * a.Nc);
      epilogue_q<BM, BN, -1, false, -1>(p, drop, acc, slab, a.Nc, uc.m0, uc.n0, wm, wn, lane, stg, bias_lds);
    } else {
      char* C = (char*)a.C;
      if (a.shift_operand == 1) C += (long long)uc.tapz * a.c_tap_stride * (obf ? 2 : 4);
#define FS2_QEPI(E, A)                                                                                               \
  {                                                                                                                  \
    if (obf) epilogue_q<BM, BN, E, true, A>(p, drop, acc, C, a.ldc, uc.m0, uc.n0, wm, wn, lane, stg, bias_lds);      \
    else epilogue_q<BM, BN, E, false, A>(p, drop, acc, C, a.ldc, uc.m0, uc.n0, wm, wn, lane, stg, bias_lds);         \
  }
      if constexpr (COLSUM) {  // weight gradients: plain fp32 results
        FS2_QEPI(0, -1)
      } else {
        switch (a.epi) {
          case FS2_EPI_ACT:
            if (a.act == FS2_ACT_SILU) FS2_QEPI(FS2_EPI_ACT, FS2_ACT_SILU)
            else FS2_QEPI(FS2_EPI_ACT, -1)
            break;
          case FS2_EPI_RESID: FS2_QEPI(FS2_EPI_RESID, -1) break;
          case FS2_EPI_DACT:
            if (a.act == FS2_ACT_SILU) FS2_QEPI(FS2_EPI_DACT, FS2_ACT_SILU)
            else FS2_QEPI(FS2_EPI_DACT, -1)
            break;
          default: FS2_QEPI(0, -1) break;
        }
      }
#undef FS2_QEPI
    }
    if (cs_on) {  // every wavefront issues its TM stores (out of range for those that hold no sum): the count is exact
      const __amdgpu_buffer_rsrc_t rcs = __builtin_amdgcn_make_buffer_rsrc((void*)a.colsum, 0, a.splitk * a.Mc * 4, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = uc.m0 + wm * (BM / 2) + 32 * i + lane;
        const bool w = do_cs && wn == 0 && lane < 32 && m < a.Mc;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cs[i][0]), rcs, w ? (uc.split * a.Mc + m) * 4 : B_OOB, 0, 0);
      }
    }
    spend = inflight;
  }
}

template <int BM, int BN, int NST>
int launch_bq(GemmP& p, int nz, hipStream_t s) {
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
  // whole-row stores only (their number has to be exact), the bias inside the LDS copy, bf16 results in whole chunks
  if (!p.staged || p.tiles_n * BN > Q_BIAS || ((a.io_bf16 & 1) && a.splitk <= 1 && (a.Nc % 8))) return FS2HIP_EINVAL;
  const int tiles = p.tiles_m * p.tiles_n;
  const long long nunits_ll = (long long)tiles * nz;
  if (nunits_ll > 0x7fffffffLL) return FS2HIP_EINVAL;
  const int nunits = (int)nunits_ll;
  dim3 grid(nunits < n_cu ? nunits : n_cu), block(256);
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
    if (mode == BT_RED) gemmbq_kernel<BM, BN, true, true, BT_RED, NST, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else if (mode == BT_NONE) gemmbq_kernel<BM, BN, true, true, BT_NONE, NST, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else return FS2HIP_EINVAL;
  } else if (a.a_kcontig && !a.b_kcontig) {
    if (mode == BT_RED) gemmbq_kernel<BM, BN, true, false, BT_RED, NST, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else if (mode == BT_NONE) gemmbq_kernel<BM, BN, true, false, BT_NONE, NST, false><<<grid, block, 0, s>>>(p, nunits, tiles);
    else return FS2HIP_EINVAL;
  } else if (!a.a_kcontig && !a.b_kcontig) {
    if (a.epi != FS2_EPI_STORE || a.bias || (a.io_bf16 & 1)) return FS2HIP_EINVAL;
    if (mode == BT_ROWS) gemmbq_kernel<BM, BN, false, false, BT_ROWS, NST, true><<<grid, block, 0, s>>>(p, nunits, tiles);
    else if (mode == BT_NONE) gemmbq_kernel<BM, BN, false, false, BT_NONE, NST, true><<<grid, block, 0, s>>>(p, nunits, tiles);
    else return FS2HIP_EINVAL;
  } else {
    return FS2HIP_EINVAL;
  }
  FS2_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// tile ids: 26 = 128x128, ring of 4 (146 KiB of LDS, one workgroup per CU); 27 = 128x64, ring of 5
int fs2_gemmbq_launch(GemmP& p, int tile, int nz, hipStream_t s) {
  switch (tile) {
    case 26: return launch_bq<128, 128, 4>(p, nz, s);
    case 27: return launch_bq<128, 64, 5>(p, nz, s);
    default: return FS2HIP_EINVAL;
  }
}
