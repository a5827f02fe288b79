// How much of the fp32 matrix pipe does ONE dependent accumulation chain per wavefront fill?  (the weights-stationary
// fp32 GEMM, csrc/gemm_ws32.hip, runs 128 MFMAs per row tile into a single 32x32 accumulator.)  One or two wavefronts
// per SIMD, 1 / 2 / 4 independent chains each, v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/chain_probe.hip -o tools/probes/chain_probe && tools/probes/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCHAIN, int THREADS>
__global__ __launch_bounds__(THREADS) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NCHAIN];
  for (int i = 0; i < NCHAIN; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32 / NCHAIN; ++r)
#pragma unroll
      for (int i = 0; i < NCHAIN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NCHAIN; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int NCHAIN, int THREADS>
__global__ __launch_bounds__(THREADS) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NCHAIN];
  for (int i = 0; i < NCHAIN; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / NCHAIN; ++r)
#pragma unroll
      for (int i = 0; i < NCHAIN; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NCHAIN; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <class K>
static void run(const char* name, K kern, int threads, int cus, float* out, double flops_per_iter_per_wave) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 40000;
  hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, out, 1000, 1.0f, 0.5f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double tf = flops_per_iter_per_wave * iters * (threads / 64) * cus / (ms * 1e-3) / 1e12;
  printf("%-44s %8.3f ms %8.2f TFLOP/s  %.3f of 157.3\n", name, ms, tf, tf / 157.3);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)cus * 512 * 4);
  const double f32 = 32.0 * 2 * 32 * 32 * 2, f16 = 64.0 * 2 * 16 * 16 * 4;
  run("32x32x2, 1 wave/SIMD, 1 chain", k32<1, 256>, 256, cus, out, f32);
  run("32x32x2, 1 wave/SIMD, 2 chains", k32<2, 256>, 256, cus, out, f32);
  run("32x32x2, 1 wave/SIMD, 4 chains", k32<4, 256>, 256, cus, out, f32);
  run("32x32x2, 2 waves/SIMD, 1 chain", k32<1, 512>, 512, cus, out, f32);
  run("32x32x2, 2 waves/SIMD, 2 chains", k32<2, 512>, 512, cus, out, f32);
  run("16x16x4, 1 wave/SIMD, 1 chain", k16<1, 256>, 256, cus, out, f16);
  run("16x16x4, 1 wave/SIMD, 2 chains", k16<2, 256>, 256, cus, out, f16);
  run("16x16x4, 1 wave/SIMD, 4 chains", k16<4, 256>, 256, cus, out, f16);
  run("16x16x4, 2 waves/SIMD, 1 chain", k16<1, 512>, 512, cus, out, f16);
  run("16x16x4, 2 waves/SIMD, 4 chains", k16<4, 512>, 512, cus, out, f16);
  return 0;
}
