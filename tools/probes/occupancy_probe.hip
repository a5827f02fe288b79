// How many 256-thread workgroups does a CU really hold as a function of static LDS?  hipOccupancy API beside a timing
// census: 512 workgroups (2 per CU) that each spin for a fixed number of cycles -- one spin long if two fit per CU,
// two if only one does.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS_FLOATS>
__global__ __launch_bounds__(256) void k(float* out, long long cycles) {
  __shared__ float buf[LDS_FLOATS];
  buf[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  float v = buf[(threadIdx.x * 7) % LDS_FLOATS];
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) out[0] = v;
}
template <int N>
void probe(float* d) {
  int nb = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<N>, 256, 0);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<N><<<512, 256>>>(d, 100000);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<N><<<512, 256>>>(d, 1000000);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("static LDS %6d B: API says %d blocks/CU; 512 blocks x 1M-cycle spin took %.3f ms (1M cycles = %.3f ms at 2.4 GHz)\n", N * 4,
         nb, ms, 1e6 / 2.4e6);
}
int main() {
  float* d;
  (void)hipMalloc(&d, 1024);
  probe<4096>(d); probe<8192>(d); probe<12288>(d); probe<16384>(d); probe<20480>(d);
  return 0;
}
