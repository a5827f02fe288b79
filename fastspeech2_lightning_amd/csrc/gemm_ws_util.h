// Hand-counted vector-memory and LDS operations of the weights-stationary streaming GEMMs (gemm_ws.hip, gemm_ws4.hip):
// loads, stores and LDS accesses the compiler never sees -- their completion is waited for with counted s_waitcnt.
#pragma once
#include "gemm_bf16_core.h"

namespace {

__device__ __forceinline__ u32x4 ws_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
// loads / stores the compiler never sees: their completion is waited for by hand
template <int OFF>
__device__ __forceinline__ void ws_ld128(u32x4& v, u32x4 r, int voff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(v) : "v"(voff), "s"(r), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ws_ld64(u32x2& v, u32x4 r, int voff) {
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:%3" : "=v"(v) : "v"(voff), "s"(r), "n"(OFF) : "memory");
}
// A buffer store of more than 8 bytes must not be followed at once by a write of its data registers (the store reads them
// over the following cycles: two wait states on gfx950).  The compiler's hazard recogniser pads its own stores; it does not
// see these, and does reuse the registers in the very next instruction (seen on the chip: the second dword of some lanes
// came out overwritten) -- hence the s_nop inside the statement.  The row-tile offset travels in soffset, which makes the
// per-lane offset a constant of the launch.
__device__ __forceinline__ void ws_st128(u32x4 v, u32x4 r, int voff, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ws_ld128s(u32x4& v, u32x4 r, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(r), "s"(soff), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ws_ld64s(u32x2& v, u32x4 r, int voff, int soff) {
  asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(r), "s"(soff), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void ws_vmwait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}
__device__ __forceinline__ void ws_dsw64(unsigned addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void ws_dsw128(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ws_dsr128(u32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}

}  // namespace
