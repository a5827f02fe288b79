// micro-test of __builtin_amdgcn_global_load_lds (16 B per lane): LDS dest = wave-uniform base + lane*16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
__global__ void k(const float* __restrict__ src, const int* __restrict__ perm, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // each lane fetches the float4 number perm[tid] of src; it lands at lds[(wave*64 + lane)*4]
  const float* g = src + 4 * perm[threadIdx.x];
  __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(lds + wave * 256), 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
  std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  std::vector<int> p(256); for (int i = 0; i < 256; ++i) p[i] = (i * 37) % 1024;
  float *d, *o; int* dp;
  hipMalloc(&d, 4096 * 4); hipMalloc(&o, 1024 * 4); hipMalloc(&dp, 256 * 4);
  hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice); hipMemcpy(dp, p.data(), 256 * 4, hipMemcpyHostToDevice);
  k<<<1, 256>>>(d, dp, o);
  std::vector<float> r(1024); hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) for (int e = 0; e < 4; ++e) if (r[t * 4 + e] != (float)(4 * p[t] + e)) ++bad;
  printf("glds test: %d mismatches (first: %f %f %f %f | %f)\n", bad, r[0], r[1], r[2], r[3], r[4]);
  return bad != 0;
}
