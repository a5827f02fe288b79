// Shared by the two GEMM cores (gemm.hip: register-staged BK=16; gemm2.hip: direct-to-LDS BK=32).
#pragma once
#include "common.h"

struct GemmP {
  Fs2GemmArgs a;
  int Rper;        // reduction length per tap (shift_operand == 0) or R
  int tiles_m;     // number of tiles along Mc
  int tiles_n;     // number of tiles along Nc
  int r_chunk;     // split-K chunk (multiple of the core's BK)
  Fs2Drop drop;
};

// accumulator tile -> global memory with the fused epilogue (bias, activation, residual, dropout, act')
// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5):
// one store instruction writes two full 128-byte row segments.
// EPI is a compile-time copy of a.epi (-1 = split-K slab: raw partial sums into the workspace), so that each
// variant is one straight run of stores -- the epilogue is executed once per tile, from a cold
// instruction cache, and must not be a chain of per-element branches.
template <int BM, int BN, int EPI>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], float* C,
                                                   int ldc, int m0, int n0, int wm, int wn, int lane) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const Fs2GemmArgs& a = p.a;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const int mrow = m0 + wm * (BM / 2) + 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
    if (n >= a.Nc) continue;
    const float bias = (EPI >= 0 && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // one 32x32 accumulator (16 values per lane) at a time; every run-time choice (activation kind,
      // dropout on/off, optional pre-activation output) is tested once per accumulator, not per element
      const int mb = mrow + i * 32;
      float v[16];
      bool in[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = acc[i][j][r];
        in[r] = mb + (r & 3) + 8 * (r >> 2) < a.Mc;
      }
#define FS2_ROW(r) ((long long)(mb + ((r) & 3) + 8 * ((r) >> 2)))
      if (EPI >= 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = a.alpha * v[r] + bias;
      }
      if (EPI == FS2_EPI_ACT) {
        if (a.out_pre) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (in[r]) a.out_pre[FS2_ROW(r) * a.ldpre + n] = v[r];
        }
        if (a.act == FS2_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fs2_act(FS2_ACT_RELU, v[r]);
        } else if (a.act == FS2_ACT_SILU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fs2_act(FS2_ACT_SILU, v[r]);
        } else if (a.act == FS2_ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fs2_act(FS2_ACT_TANH, v[r]);
        }
      } else if (EPI == FS2_EPI_DACT) {
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = in[r] ? a.aux[FS2_ROW(r) * a.ldaux + n] : 0.f;
        if (a.act == FS2_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] *= fs2_dact(FS2_ACT_RELU, x[r]);
        } else if (a.act == FS2_ACT_SILU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] *= fs2_dact(FS2_ACT_SILU, x[r]);
        } else if (a.act == FS2_ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] *= fs2_dact(FS2_ACT_TANH, x[r]);
        }
      }
      if (EPI > 0 && drop.on) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] *= fs2_drop_factor(drop, (unsigned long long)(FS2_ROW(r) * ldc + n));
      }
      if (EPI == FS2_EPI_RESID) {
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = in[r] ? a.resid[FS2_ROW(r) * a.ldr + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = x[r] + a.res_scale * v[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (in[r]) C[FS2_ROW(r) * ldc + n] = v[r];
#undef FS2_ROW
    }
  }
}

template <int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0,
                                              int wm, int wn, int lane, int split, int tapz) {
  const Fs2GemmArgs& a = p.a;
  float* C = a.C;
  if (a.splitk > 1) {
    C = a.workspace + ((long long)split * a.taps + tapz) * ((long long)a.Mc * a.Nc);
    gemm_epilogue_impl<BM, BN, -1>(p, acc, C, a.Nc, m0, n0, wm, wn, lane);
    return;
  }
  if (a.shift_operand == 1) C += (long long)tapz * a.c_tap_stride;
  switch (a.epi) {
    case FS2_EPI_ACT: gemm_epilogue_impl<BM, BN, FS2_EPI_ACT>(p, acc, C, a.ldc, m0, n0, wm, wn, lane); break;
    case FS2_EPI_RESID: gemm_epilogue_impl<BM, BN, FS2_EPI_RESID>(p, acc, C, a.ldc, m0, n0, wm, wn, lane); break;
    case FS2_EPI_DACT: gemm_epilogue_impl<BM, BN, FS2_EPI_DACT>(p, acc, C, a.ldc, m0, n0, wm, wn, lane); break;
    default: gemm_epilogue_impl<BM, BN, 0>(p, acc, C, a.ldc, m0, n0, wm, wn, lane); break;
  }
}

// XCD-aware workgroup order: consecutive work ids (which share the A rows of one M-tile) land on the
// same XCD / L2.  Dispatch deals workgroups round-robin over the 8 XCDs; bijective for any grid size.
__device__ __forceinline__ int fs2_xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// v2 core launcher (gemm2.hip); tile: 4 = 128x128 (3-stage ring), 5 = 128x64 (3), 6 = 64x64 (4),
// 7 = 64x64 (2), 8 = 128x64 (2), 9 = 128x128 (2)
int fs2_gemm2_launch(GemmP& p, int tile, int nz, hipStream_t s);
