// fp32 MFMA issue rate on gfx950: v_mfma_f32_16x16x4_f32 against v_mfma_f32_32x32x2_f32, operands in registers,
// NCHAIN independent accumulator chains per wavefront, 1 or 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCHAIN>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NCHAIN];
  for (int i = 0; i < NCHAIN; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NCHAIN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NCHAIN; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NCHAIN>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NCHAIN];
  for (int i = 0; i < NCHAIN; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NCHAIN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NCHAIN; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F>
double tfl(F launch, double flops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return flops / (ms * 1e-3) / 1e12;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * wps;  // 4 wavefronts per workgroup: one workgroup per CU = 1 wavefront per SIMD
    printf("%d wavefront(s) per SIMD\n", wps);
#define RUN16(N) printf("  16x16x4  %d chain(s): %6.1f TFLOP/s\n", N, tfl([&] { k16<N><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, 2048.0 * 8 * N * iters * 4.0 * blocks));
#define RUN32(N) printf("  32x32x2  %d chain(s): %6.1f TFLOP/s\n", N, tfl([&] { k32<N><<<blocks, 256>>>(out, iters, 1.f, 2.f); }, 4096.0 * 8 * N * iters * 4.0 * blocks));
    RUN16(1) RUN16(2) RUN16(4) RUN32(1) RUN32(2) RUN32(4)
  }
  return 0;
}
