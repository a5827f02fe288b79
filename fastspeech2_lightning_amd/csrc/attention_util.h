// Pieces shared by the attention kernel families (attention2.hip: fp32 storage; attention_bf16.hip: bf16 storage):
// compile-time loops, the lane-pair reductions, the dropout pair hash and the workgroup -> (row block, utterance, head)
// order.  Everything here has to stay bit-identical between the families: the keep mask of an element is a function of
// (seed, element index) only, whichever kernel regenerates it.
#pragma once
#include <utility>

#include "gemm_common.h"

namespace {

template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: the body gets the index as an integral_constant (ds_read immediates need constants)
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// max / sum over the lane pair (l, l ^ 32) -- the two halves of a row's 32 keys -- with v_permlane32_swap (vector
// pipe; ds_bpermute would be an LDS round trip in the one stretch of the tile that is not under MFMAs)
__device__ __forceinline__ float pair_max(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// fs2_drop_factor (common.h) for the products' weights, bit-identical to it: one hash per PAIR of neighbouring keys
// (steps t = 4a + r with r = 0,1 and r = 2,3 are neighbours; the mask rows are padded to an even length, so a pair
// never straddles a hash).  The element index stays below 2^32 (checked by the launcher).
struct PairHash {
  uint32_t thresh, s_lo, s_hi;
  __device__ __forceinline__ void setup(const Fs2Drop& d) {
    thresh = d.thresh; s_lo = (uint32_t)d.seed; s_hi = (uint32_t)(d.seed >> 32);
  }
  __device__ __forceinline__ uint32_t hash(uint32_t pair_idx) const {
    uint32_t x = pair_idx ^ s_lo;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= s_hi;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
  }
  template <int ODD>
  __device__ __forceinline__ bool keep(uint32_t h) const { return (ODD ? (h >> 16) : (h & 0xffffu)) >= thresh; }
};

// Work order.  A (batch, head) unit is `nblk` workgroups (its row blocks) whose cost grows with the utterance's
// length, and the lengths of a batch are ragged (0.66 .. 1 of the padded length): dealt in batch order, the long
// utterances that happen to come last set the kernel's time (measured: +27 % against the same work at uniform
// length).  Longest first instead: dispatch deals workgroup ids round-robin over the 8 XCDs, so workgroup id w is the
// (w >> 3)-th workgroup of XCD (w & 7); unit number (w >> 3) / nblk * 8 + (w & 7) in order of DEcreasing length goes
// there -- every XCD gets a long-to-short sequence of units, and the row blocks of a unit still share one XCD's L2.
// Needs B <= 64 (one lane per utterance ranks them) and B * H a multiple of 8; otherwise plain XCD-contiguous order.
// Also returns the unit's length lens[b].
template <class Args>
__device__ __forceinline__ void work_unit(const Args& p, int nblk, int& blk, int& b, int& h, int& len) {
  const int units = p.B * p.H;
  if (p.B <= 64 && (units & 7) == 0) {
    const int w = blockIdx.x, k = w >> 3;
    blk = k % nblk;
    const int u = (k / nblk) * 8 + (w & 7);  // rank of the unit; u / H = rank of the utterance
    const int lane = threadIdx.x & 63;
    const int mylen = lane < p.B ? p.lens[lane] : -1;
    int rank = 0;
    for (int j = 0; j < p.B; ++j) {
      const int lj = __builtin_amdgcn_readlane(mylen, j);
      rank += (lj > mylen || (lj == mylen && j < lane)) ? 1 : 0;
    }
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(lane < p.B && rank == u / p.H);
    b = __builtin_ctzll(hit);
    h = u % p.H;
    len = __builtin_amdgcn_readlane(mylen, b);  // (already here: no second, dependent load of lens[b])
  } else {
    const int wid = fs2_xcd_remap(blockIdx.x, gridDim.x);
    blk = wid % nblk;
    const int bh = wid / nblk;
    h = bh % p.H;
    b = bh / p.H;
    len = p.lens[b];
  }
  // (integer division runs on the vector pipe: without this the compiler keeps the quotients -- and every pointer,
  // buffer resource and DMA offset derived from them -- in vector registers and wraps each DMA in a waterfall loop)
  blk = __builtin_amdgcn_readfirstlane(blk);
  b = __builtin_amdgcn_readfirstlane(b);
  h = __builtin_amdgcn_readfirstlane(h);
  len = __builtin_amdgcn_readfirstlane(len);
}

}  // namespace
