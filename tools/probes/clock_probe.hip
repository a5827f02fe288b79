// Which clock does the chip hold under sustained fp32 MFMA load?  (DESIGN.md carried two claims --
// "105 TFLOP/s IS the loop at ~2.2 GHz" and "143 TFLOP/s on whole rounds, no throttling" -- that cannot both stand.)
// Every CU runs 4 wavefronts (one per SIMD) of a pure v_mfma_f32_32x32x2_f32 loop, 4 independent accumulator chains;
// each wavefront stamps the shader-clock counter (clock64 = s_memtime) and the 100 MHz real-time counter
// (wall_clock64 = s_memrealtime) around the loop.  core clock = d(clock64) / d(wall time); TFLOP/s from the host's
// events.  Runs of ~2 ms, ~20 ms, ~200 ms and ~2 s.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/clock_probe.hip -o tools/probes/clock_probe && tools/probes/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* stamps, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = c1 - c0;
    stamps[2 * w + 1] = w1 - w0;
  }
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, waves = cus * 4;
  float* out;
  long long* st;
  hipMalloc(&out, (size_t)cus * 256 * 4);
  hipMalloc(&st, (size_t)waves * 2 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("device: %s, %d CUs, clockRate %.0f MHz (reported)\n", p.name, cus, p.clockRate / 1e3);
  printf("%10s %10s %12s %14s %16s %14s\n", "iters", "ms", "TFLOP/s", "core GHz", "cycles/MFMA", "frac of 157.3");
  const int sweep[] = {2000, 20000, 200000, 2000000};
  for (int rep = 0; rep < 2; ++rep)
    for (int iters : sweep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(256), 0, 0, out, st, iters, 1.0f, 0.5f);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(waves * 2);
      hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int w = 0; w < waves; ++w) {
        cyc += (double)h[2 * w];
        wall += (double)h[2 * w + 1];
      }
      cyc /= waves;
      wall /= waves;                                  // ticks of 10 ns
      const double mfmas = (double)iters * 32;         // per wavefront
      const double flops = mfmas * waves * 2.0 * 32 * 32 * 2;
      const double tf = flops / (ms * 1e-3) / 1e12;
      printf("%10d %10.3f %12.2f %14.3f %16.2f %14.3f\n", iters, ms, tf, cyc / (wall * 10.0), cyc / mfmas, tf / 157.3);
    }
  return 0;
}
