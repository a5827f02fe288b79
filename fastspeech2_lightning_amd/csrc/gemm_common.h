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
// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
template <int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0,
                                              int wm, int wn, int lane, int split, int tapz) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const Fs2GemmArgs& a = p.a;
  float* C = a.C;
  if (a.splitk > 1) {
    C = a.workspace + ((long long)split * a.taps + tapz) * ((long long)a.Mc * a.Nc);
  } else if (a.shift_operand == 1) {
    C += (long long)tapz * a.c_tap_stride;
  }
  const int ldc = a.splitk > 1 ? a.Nc : a.ldc;
  const bool plain = a.splitk > 1;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
      if (n >= a.Nc) continue;
      const float bias = (!plain && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= a.Mc) continue;
        float v = acc[i][j][r];
        const long long o = (long long)m * ldc + n;
        if (plain) {
          C[o] = v;
          continue;
        }
        v = a.alpha * v + bias;
        switch (a.epi) {
          case FS2_EPI_ACT:
            if (a.out_pre) a.out_pre[(long long)m * a.ldpre + n] = v;
            v = fs2_act(a.act, v) * fs2_drop_factor(drop, (unsigned long long)o);
            break;
          case FS2_EPI_RESID:
            v = a.resid[(long long)m * a.ldr + n] + a.res_scale * (v * fs2_drop_factor(drop, (unsigned long long)o));
            break;
          case FS2_EPI_DACT:
            v = v * fs2_dact(a.act, a.aux[(long long)m * a.ldaux + n]) * fs2_drop_factor(drop, (unsigned long long)o);
            break;
          default: break;
        }
        C[o] = v;
      }
    }
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
