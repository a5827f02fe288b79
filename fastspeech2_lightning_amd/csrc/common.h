// Shared device helpers for the fs2hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fs2hip.h"

#define FS2_WAVE 64

#define FS2_LAUNCH_CHECK()                    \
  do {                                        \
    hipError_t e__ = hipGetLastError();       \
    if (e__ != hipSuccess) return (int)e__;   \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- stateless dropout: element idx keeps iff its 16-bit field of hash(seed, idx >> 1) >= p * 2^16 ----------
// One 32-bit hash serves TWO neighbouring elements (even idx: low half, odd idx: high half): fp32 vector
// instructions and fp32 MFMAs share the arithmetic on this chip (a vector instruction between two MFMAs is not
// hidden, it is added), and the hash is most of the vector work of the attention kernels and GEMM epilogues.
// The drop probability is p rounded to 1/65536.
// 32-bit two-round multiply-xorshift mixer (the per-kernel 64-bit seed enters before the first and between the
// two rounds, so that two dropout sites / steps are not index-permuted copies of one mask).  Integer
// multiplies are quarter rate on CDNA: a 64-bit splitmix (12 of them per element) made the GEMM epilogues and
// the attention kernels VALU-bound; this one costs two.
__device__ __forceinline__ uint32_t fs2_hash32(unsigned long long seed, unsigned long long idx) {
  const uint32_t hi = (uint32_t)(idx >> 32);
  uint32_t x = (uint32_t)idx ^ (uint32_t)seed ^ ((hi << 13) | (hi >> 19));
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= (uint32_t)(seed >> 32);
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
struct Fs2Drop {
  uint32_t thresh;  // drop iff the element's 16-bit hash field < thresh (0 .. 65536)
  float scale;      // 1/(1-p)
  unsigned long long seed;
  const unsigned long long* step;  // device-resident step counter mixed into the seed (may be null):
                                   // kernel arguments are frozen when a captured hipGraph replays
  bool on;
};
inline Fs2Drop fs2_make_drop(float p, unsigned long long seed, const unsigned long long* step = nullptr) {
  Fs2Drop d;
  d.on = p > 0.f;
  d.seed = seed;
  d.step = step;
  double t = (double)p * 65536.0 + 0.5;
  d.thresh = t >= 65536.0 ? 65536u : (uint32_t)t;
  d.scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  return d;
}
// call once at kernel entry
__device__ __forceinline__ Fs2Drop fs2_resolve_drop(Fs2Drop d) {
  if (d.on) {
    unsigned long long z = d.seed + (d.step ? (*d.step) * 0xD1B54A32D192ED03ull : 0ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;  // once per thread: spread site / step over both seed words
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    d.seed = z ^ (z >> 31);
  }
  return d;
}
__device__ __forceinline__ float fs2_drop_factor(const Fs2Drop& d, unsigned long long idx) {
  if (!d.on) return 1.f;
  const uint32_t h = fs2_hash32(d.seed, idx >> 1);
  return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) < d.thresh ? 0.f : d.scale;
}

// The same factors for FOUR consecutive elements idx .. idx + 3 with idx EVEN: the quad is two whole pairs, so two hashes
// serve it (written element by element the compiler cannot know that idx is even and hashes four times; the two
// integer multiplies of a hash are quarter-rate instructions, and the SiLU / dropout epilogues of the feed-forward
// GEMMs are bound by exactly this arithmetic).  idx < 2^32 (every tensor here is below 2 GiB).
__device__ __forceinline__ uint32_t fs2_hash32_lo(unsigned long long seed, uint32_t idx) {  // = fs2_hash32(seed, idx)
  uint32_t x = idx ^ (uint32_t)seed;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= (uint32_t)(seed >> 32);
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ void fs2_drop_quad(const Fs2Drop& d, uint32_t idx, float (&f)[4]) {
  const uint32_t h0 = fs2_hash32_lo(d.seed, idx >> 1), h1 = fs2_hash32_lo(d.seed, (idx >> 1) + 1u);
  f[0] = (h0 & 0xffffu) < d.thresh ? 0.f : d.scale;
  f[1] = (h0 >> 16) < d.thresh ? 0.f : d.scale;
  f[2] = (h1 & 0xffffu) < d.thresh ? 0.f : d.scale;
  f[3] = (h1 >> 16) < d.thresh ? 0.f : d.scale;
}

// ---- activations ------------------------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (about 1 ulp each): the epilogues that call this run once per output element
__device__ __forceinline__ float fs2_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fs2_act(int act, float x) {
  switch (act) {
    case FS2_ACT_RELU: return x > 0.f ? x : 0.f;
    case FS2_ACT_SILU: return x * fs2_sigmoid(x);
    case FS2_ACT_TANH: return tanhf(x);
    default: return x;
  }
}
// derivative with respect to the pre-activation x
__device__ __forceinline__ float fs2_dact(int act, float x) {
  switch (act) {
    case FS2_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case FS2_ACT_SILU: {
      float s = fs2_sigmoid(x);
      return s * (1.f + x * (1.f - s));
    }
    case FS2_ACT_TANH: {
      float t = tanhf(x);
      return 1.f - t * t;
    }
    default: return 1.f;
  }
}

// ---- wavefront reductions (64 lanes) ------------------------------------------------------
__device__ __forceinline__ float fs2_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float fs2_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- shared host-side launchers (reduce.hip) ----------------------------------------------------
// out0[c] = sum over rows of src[r*stride + c] for c < n0, out1[c - n0] for n0 <= c < n
int fs2_reduce_rows(const float* src, int rows, int n, long long stride, float* out0, int n0, float* out1,
                    hipStream_t s);
