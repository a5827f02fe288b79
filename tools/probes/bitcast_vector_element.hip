// Compiler trap worth a probe: __builtin_bit_cast(float, v[k]) on an ext_vector_type ELEMENT expression yields element 0
// for every k (hipcc of ROCm 7.2: the generated code below stores exp2(s - v[0]) four times).  Cast the whole vector
// (__builtin_bit_cast(f32x4, v)) and index the result.   hipcc --offload-arch=gfx950 -O3 -S -o - bitcast_vector_element.hip
#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void b_rd128(u32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ab_pin(u32x4& v) { asm volatile("" : "+v"(v)); }
__global__ void k(float* out, const float* in) {
  __shared__ float sm[256];
  sm[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  u32x4 a0;
  unsigned a = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)sm + 32 * (threadIdx.x >> 5);
  b_rd128<0>(a0, a);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  ab_pin(a0);
  float s = in[300 + threadIdx.x];
  out[threadIdx.x * 4 + 0] = __builtin_amdgcn_exp2f(s - __builtin_bit_cast(float, a0[0]));
  out[threadIdx.x * 4 + 1] = __builtin_amdgcn_exp2f(s - __builtin_bit_cast(float, a0[1]));
  out[threadIdx.x * 4 + 2] = __builtin_amdgcn_exp2f(s - __builtin_bit_cast(float, a0[2]));
  out[threadIdx.x * 4 + 3] = __builtin_amdgcn_exp2f(s - __builtin_bit_cast(float, a0[3]));
}
