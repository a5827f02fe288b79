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
// co-execution: in every workgroup of 8 wavefronts (two per SIMD) wavefronts 0-3 run the MFMA loop and 4-7 a
// packed-fp32 FMA loop (v_pk_fma_f32, 16 independent accumulators); MODE 0 = both, 1 = MFMA wavefronts only (the
// others exit), 2 = vector wavefronts only
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void kmix(float* out, int iters, float a, float b) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {  // wavefront w sits on SIMD w % 4: 0-3 = one MFMA wavefront per SIMD, 4-7 = one vector wavefront per SIMD
    if (MODE == 2) return;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][15];
  } else {
    if (MODE == 1) return;
    f32x2 acc[16], x = {a, b}, y = {b, a};
    for (int i = 0; i < 16; ++i) acc[i] = (f32x2){0.f, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(x, y, acc[i]);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
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
  {
    const int blocks = 256;  // one 8-wavefront workgroup per CU
    const double mf = 4096.0 * 8 * 2 * iters * 4.0 * blocks;        // 4 MFMA wavefronts per workgroup
    const double vf = 2.0 * 2 * 64 * 8 * 16 * (double)iters * 4.0 * blocks;  // 4 vector wavefronts: 2 lanes-worth x 2 flops x 64 lanes
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return (double)ms * 1e-3; };
    double t1 = time([&] { kmix<1><<<blocks, 512>>>(out, iters, 1.f, 2.f); });
    double t2 = time([&] { kmix<2><<<blocks, 512>>>(out, iters, 1.f, 2.f); });
    double t0 = time([&] { kmix<0><<<blocks, 512>>>(out, iters, 1.f, 2.f); });
    printf("one MFMA + one vector wavefront per SIMD:\n");
    printf("  MFMA wavefronts alone   : %6.1f TFLOP/s (%.2f ms)\n", mf / t1 / 1e12, t1 * 1e3);
    printf("  v_pk_fma_f32 alone      : %6.1f TFLOP/s (%.2f ms)\n", vf / t2 / 1e12, t2 * 1e3);
    printf("  both in one launch      : %6.1f TFLOP/s in total (%.2f ms; MFMA part %.1f, vector part %.1f if each ran for the whole launch)\n",
           (mf + vf) / t0 / 1e12, t0 * 1e3, mf / t0 / 1e12, vf / t0 / 1e12);
  }
  return 0;
}
