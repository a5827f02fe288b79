// Shared by the two GEMM cores (gemm.hip: register-staged BK=16; gemm2.hip: direct-to-LDS BK=32).
#pragma once
#include "common.h"

struct GemmP {
  Fs2GemmArgs a;
  int Rper;        // reduction length per tap (shift_operand == 0) or R
  int tiles_m;     // number of tiles along Mc
  int tiles_n;     // number of tiles along Nc
  int r_chunk;     // split-K chunk (multiple of the core's BK)
  int probe;       // measurement aid (FS2_GEMM_PROBE, bf16-storage one-tile kernels): bit 0 skip the MFMA phase, bit 1 skip
                   // the DMA, bit 2 skip the epilogue -- wrong results, timing only
  int staged;      // bf16-storage core: every output row starts 16-byte aligned (whole-row stores through LDS)
  Fs2Drop drop;
};

// Grouped launch (fs2hip_gemm_grouped): up to FS2_GEMM_GROUP_MAX independent GEMMs of ONE kernel instance (same core, tile,
// operand orientations, no conv taps) in one grid.  Member i owns the workgroups [start[i], start[i + 1]); inside them
// the member's own GemmP applies unchanged (shapes, split-K, epilogue and pointers may all differ).  The whole table
// travels as the kernel argument (2.4 KB): nothing to upload, and a recorded launch plan replays it as it stands.
struct GemmPG {
  int n;
  int start[FS2_GEMM_GROUP_MAX + 1];
  GemmP m[FS2_GEMM_GROUP_MAX];
};
__device__ __forceinline__ int fs2_group_member(const GemmPG& g, int bid) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < FS2_GEMM_GROUP_MAX; ++k) i += (k < g.n && bid >= g.start[k]) ? 1 : 0;
  return i;
}

// accumulator tile -> global memory with the fused epilogue (bias, activation, residual, dropout, act')
// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5):
// one store instruction writes two full 128-byte row segments.
//
// EPI is a compile-time copy of a.epi (-1 = split-K slab: raw partial sums into the workspace; -2: the same with
// device-coherent stores, see splitk_finish), so that each
// variant is one straight run of stores -- the epilogue is executed once per tile, from a cold
// instruction cache, and must not be a chain of per-element branches.
//
// Addressing: every tensor the epilogue touches goes through a raw buffer resource with ONE per-lane byte
// offset (the lane's column in the tile's first row) and a scalar byte offset per accumulator row, so the
// 16 addresses of an accumulator cost one VGPR instead of 32, and rows past Mc / columns past Nc are dropped by
// the buffer range check (num_records = Mc * ld * 4; a lane outside Nc carries the out-of-range offset)
// instead of 16 predicates.  Extents are < 2 GiB (checked by fs2hip_gemm).
constexpr int FS2_EPI_OOB = (int)0x80000000;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t fs2_epi_rsrc(const float* base, int rows, int ld) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rows * ld * 4, 0x00020000);
}
__device__ __forceinline__ float fs2_buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void fs2_buf_store(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
// device-coherent variants (sc1): the store is written through this XCD's L2, the load does not take a line another
// XCD may have made stale -- what a relaxed agent-scope atomic store / load compiles to on gfx942 / gfx950
constexpr int FS2_SC1 = 16;
__device__ __forceinline__ void fs2_buf_store_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, FS2_SC1);
}

template <int BM, int BN, int EPI>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], float* C,
                                                   int ldc, int m0, int n0, int wm, int wn, int lane) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const Fs2GemmArgs& a = p.a;
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const int mrow = m0 + wm * (BM / 2) + 4 * (lane >> 5);  // the lane's row of accumulator register 0, block i = 0
  const __amdgpu_buffer_rsrc_t rc = fs2_epi_rsrc(C, a.Mc, ldc);
#define FS2_ROWOFF(i, r) ((i) * 32 + ((r) & 3) + 8 * ((r) >> 2))  // compile-time row inside the wave tile
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
    const bool col = n < a.Nc;
    const float bias = (EPI >= 0 && a.bias && col) ? a.bias[n] : 0.f;
    const int vc = col ? (mrow * ldc + n) * 4 : FS2_EPI_OOB;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // one 32x32 accumulator (16 values per lane) at a time; every run-time choice (activation kind,
      // dropout on/off, optional pre-activation output) is tested once per accumulator, not per element
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
      if (EPI >= 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = a.alpha * v[r] + bias;
      }
      if (EPI == FS2_EPI_ACT) {
        if (a.out_pre) {
          const __amdgpu_buffer_rsrc_t rp = fs2_epi_rsrc(a.out_pre, a.Mc, a.ldpre);
          const int vp = col ? (mrow * a.ldpre + n) * 4 : FS2_EPI_OOB;
#pragma unroll
          for (int r = 0; r < 16; ++r) fs2_buf_store(rp, vp, FS2_ROWOFF(i, r) * a.ldpre * 4, v[r]);
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
        const __amdgpu_buffer_rsrc_t rx = fs2_epi_rsrc(a.aux, a.Mc, a.ldaux);
        const int vx = col ? (mrow * a.ldaux + n) * 4 : FS2_EPI_OOB;
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = fs2_buf_load(rx, vx, FS2_ROWOFF(i, r) * a.ldaux * 4);  // 0 outside
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
      if (EPI > 0 && drop.on) {  // element index m * ldc + n, as everywhere else this mask is used
        if ((ldc & 1) == 0) {
          // One hash serves a pair of neighbouring elements and in this layout the pair sits in neighbouring LANES
          // (same register), so written per element both lanes compute the same hash.  With ldc even the lane's
          // parity is the element's: for registers r, r + 1 (rows one apart) the even lane hashes the pair of row r,
          // the odd lane the pair of row r + 1, and they swap (DPP quad_perm [1, 0, 3, 2]): one hash per lane for
          // two elements -- the hash is most of the vector work of these epilogues, and fp32 vector work comes
          // straight out of the fp32 MFMAs' budget.
          const int par = lane & 1;
          const unsigned he = (unsigned)((mrow * ldc + n) + par * ldc);  // the lane's hashed element of register 0, block 0
          const unsigned sh = par << 4;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const uint32_t H = fs2_hash32_lo(drop.seed, (he + (unsigned)(FS2_ROWOFF(i, r) * ldc)) >> 1);
            const uint32_t Hn = (uint32_t)__builtin_amdgcn_mov_dpp((int)H, 0xB1, 0xF, 0xF, true);
            const uint32_t ha = par ? Hn : H, hb = par ? H : Hn;  // the pair of row r / of row r + 1
            v[r] *= ((ha >> sh) & 0xffffu) < drop.thresh ? 0.f : drop.scale;
            v[r + 1] *= ((hb >> sh) & 0xffffu) < drop.thresh ? 0.f : drop.scale;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            v[r] *= fs2_drop_factor(drop, (unsigned long long)((unsigned)(vc + FS2_ROWOFF(i, r) * ldc * 4) >> 2));
        }
      }
      if (EPI == FS2_EPI_RESID) {
        const __amdgpu_buffer_rsrc_t rx = fs2_epi_rsrc(a.resid, a.Mc, a.ldr);
        const int vx = col ? (mrow * a.ldr + n) * 4 : FS2_EPI_OOB;
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = fs2_buf_load(rx, vx, FS2_ROWOFF(i, r) * a.ldr * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = x[r] + a.res_scale * v[r];
      }
      if (EPI == -2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) fs2_buf_store_sc1(rc, vc, FS2_ROWOFF(i, r) * ldc * 4, v[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) fs2_buf_store(rc, vc, FS2_ROWOFF(i, r) * ldc * 4, v[r]);
      }
    }
  }
#undef FS2_ROWOFF
}

// Split-K without a second launch: every workgroup of an output tile, after its slab tile is written, bumps the
// tile's arrival counter; the one that finds splitk - 1 there is the last, sums the tile's slabs in slab order
// (the order fs2hip_reduce_slabs uses: the result is the same bit for bit, whichever workgroup ends up last) and
// writes C.  The eight XCDs' L2s are not coherent with each other, and the device-scope fences that would make
// ordinary stores visible write back / invalidate a whole L2 each (measured: +6.8 ms per step, the other workgroups
// of the XCD lose their operand tiles).  Instead the slab traffic itself is device-coherent: slabs are stored with
// sc1 (written through), the workgroup waits for its stores to be acknowledged (vmcnt 0) before the counter's
// device-scope atomic, and the last workgroup reads the slabs with sc1 loads.  The counter is re-armed for the next
// launch on the stream.
template <int BM, int BN>
__device__ __forceinline__ void splitk_finish(const GemmP& p, int m0, int n0, int tapz) {
  const Fs2GemmArgs& a = p.a;
  __shared__ int last_one;
  const int slot = (tapz * ((a.Mc + BM - 1) / BM) + m0 / BM) * p.tiles_n + n0 / BN;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's slab stores are acknowledged
  __syncthreads();                                   // ... and everybody's
  if (threadIdx.x == 0)
    last_one = __hip_atomic_fetch_add(a.counters + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.splitk - 1;
  __syncthreads();
  if (!last_one) return;
  const long long slab = (long long)a.Mc * a.Nc;
  const long long kstride = (long long)a.taps * slab;  // workspace[split][taps][Mc * Nc]
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(a.workspace + (long long)tapz * slab), 0, 0x7ffffffc, 0x00020000);
  float* C = a.C + (a.shift_operand == 1 ? (long long)tapz * a.c_tap_stride : 0);
  const bool vec = (a.Nc % 4) == 0 && (a.ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)a.workspace % 16) == 0;
  constexpr int C4 = BN / 4;
  for (int e = threadIdx.x; e < BM * C4; e += blockDim.x) {
    const int m = m0 + e / C4, n = n0 + (e % C4) * 4;
    if (m >= a.Mc || n >= a.Nc) continue;
    const int off = (m * a.Nc + n) * 4;  // (< 2 GiB: checked by fs2hip_gemm)
    if (vec) {
      // eight slabs in flight per thread (these loads miss every cache by construction), added in slab order
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < a.splitk; k0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(a.workspace + (long long)tapz * slab + (k0 + u) * kstride), 0, k0 + u < a.splitk ? 0x7ffffffc : 0, 0x00020000);
          v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, off, 0, FS2_SC1));  // (0 past the last slab)
        }
        if (k0 == 0) {
          t = v[0];
#pragma unroll
          for (int u = 1; u < 8; ++u)
            if (u < a.splitk) t += v[u];
        } else {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k0 + u < a.splitk) t += v[u];
        }
      }
      *reinterpret_cast<f32x4*>(C + (long long)m * a.ldc + n) = t;
    } else {
      for (int i = 0; i < 4 && n + i < a.Nc; ++i) {
        float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, off + 4 * i, 0, FS2_SC1));
        for (int k = 1; k < a.splitk; ++k) {
          const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(a.workspace + (long long)tapz * slab + k * kstride), 0, 0x7ffffffc, 0x00020000);
          t += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rk, off + 4 * i, 0, FS2_SC1));
        }
        C[(long long)m * a.ldc + n + i] = t;
      }
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(a.counters + slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, const f32x16 (&acc)[BM / 64][BN / 64], int m0, int n0,
                                              int wm, int wn, int lane, int split, int tapz) {
  const Fs2GemmArgs& a = p.a;
  float* C = a.C;
  if (a.splitk > 1) {
    C = a.workspace + ((long long)split * a.taps + tapz) * ((long long)a.Mc * a.Nc);
    if (a.counters) {
      gemm_epilogue_impl<BM, BN, -2>(p, acc, C, a.Nc, m0, n0, wm, wn, lane);
      splitk_finish<BM, BN>(p, m0, n0, tapz);
    } else {
      gemm_epilogue_impl<BM, BN, -1>(p, acc, C, a.Nc, m0, n0, wm, wn, lane);
    }
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
// persistent v2 core (gemm2p.hip); tile: 10 = 64x64, 11 = 128x64, 12 = 128x128 (all 2-stage); 13 / 14 = 128x128 /
// 128x64 with the last partial round of tiles cut along the reduction
int fs2_gemm2p_launch(GemmP& p, int tile, int nz, hipStream_t s);
// bf16-storage core (gemm_bf16.hip / gemm_bf16p.hip); tile: 20 = 128x128, 22 = 128x64, 23 = 64x64, 24 / 25 = persistent
// 128x128 / 128x64
int fs2_gemmb_launch(GemmP& p, int tile, int nz, hipStream_t s);
// weights-stationary streaming form of the bf16-storage core (gemm_ws.hip): forward orientation, K = 256; tile 30 / 31 =
// 512 / 256 output columns per workgroup
int fs2_gemmws_launch(GemmP& p, int tile, hipStream_t s);
// the same structure in exact fp32 (gemm_ws32.hip): tile 32
int fs2_gemmws32_launch(GemmP& p, hipStream_t s);
// bf16-storage core, K = 1024, tile id 33 (gemm_ws4.hip)
int fs2_gemmws4_launch(GemmP& p, hipStream_t s);
// grouped forms (GemmPG): fp32 core tiles 7 / 8, bf16-storage core tiles 22 / 23 / 26; every member prepared by
// fs2hip_gemm's own checks (p.a, Rper, drop, staged set; tiles_m / tiles_n / r_chunk / start are filled in here)
int fs2_gemm2_launch_grouped(GemmPG& g, int tile, hipStream_t s);
int fs2_gemmb_launch_grouped(GemmPG& g, int tile, hipStream_t s);
// finishes the reduction-split tail tiles of a persistent launch (reduce.hip)
int fs2_tail_fixup(const float* ws, int S, long long slab, float* C, int ldc, const float* bias, float alpha, int m0,
                   int Mc, int Nc, hipStream_t s);
