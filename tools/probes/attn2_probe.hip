// Stand-alone harness for the second-generation attention kernels: times them and, built with -DFS2_ATTN_STAMPS,
// prints where a wavefront's cycles go (phase sums from s_memtime stamps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFS2_ATTN_STAMPS -I include -I fastspeech2_lightning_amd/csrc \
//         tools/probes/attn2_probe.hip -o tools/probes/attn2_probe && ./tools/probes/attn2_probe [T] [len]
#include "../../fastspeech2_lightning_amd/csrc/attention2.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int B = 32, H = 2, HD = 128, D = H * HD;
  const int T = argc > 1 ? atoi(argv[1]) : 648, len = argc > 2 ? atoi(argv[2]) : T;
  const float pdrop = argc > 3 ? atof(argv[3]) : 0.f;
  const int planes = argc > 4 ? atoi(argv[4]) : 0;
  std::vector<float> h((size_t)B * T * 3 * D);
  unsigned x = 12345;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) & 0xffff) / 32768.f - 1.f; }
  std::vector<int> lens(B, len);
  float *qkv, *o, *lse; int* dl; long long* st;
  (void)hipMalloc(&qkv, h.size() * 4); (void)hipMalloc(&o, (size_t)B * T * D * 4); (void)hipMalloc(&lse, (size_t)B * H * T * 4);
  const size_t nst = (size_t)((T + 63) / 64) * H * B * 4 * 8;
  (void)hipMalloc(&dl, B * 4); (void)hipMalloc(&st, nst * 8);
  (void)hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dl, lens.data(), B * 4, hipMemcpyHostToDevice);
  Attn2Args a{qkv, dl, B, T, H, HD, 1.f / sqrtf((float)HD), fs2_make_drop(pdrop, 777), planes, st};
  for (int i = 0; i < 300; ++i) fs2_attn2_fwd(a, o, lse, 0);
  (void)hipDeviceSynchronize();
  (void)hipMemset(st, 0, nst * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  const int N = 50;
  for (int i = 0; i < N; ++i) fs2_attn2_fwd(a, o, lse, 0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fl = 4.0 * B * H * (double)T * len * HD;
  printf("fwd T=%d len=%d drop=%.1f: %.1f us  %.1f TFLOP/s\n", T, len, pdrop, ms / N * 1e3, fl / (ms / N * 1e-3) / 1e12);
  std::vector<long long> all(nst); (void)hipMemcpy(all.data(), st, nst * 8, hipMemcpyDeviceToHost);
  long long hs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < nst; ++i) hs[i & 7] += all[i];
  const double waves = (double)((T + 63) / 64) * H * B * 4 * N;
  const char* names[8] = {"loop/prologue", "wait X", "K.Q^T", "row max", "wait Y", "P.V", "K dma issue", ""};
  double tot = 0; for (int i = 0; i < 7; ++i) tot += hs[i];
  if (tot > 0) for (int i = 0; i < 7; ++i) printf("  %-14s %9.0f cycles/wave (%4.1f %%)\n", names[i], hs[i] / waves, 100.0 * hs[i] / tot);
  // ---- backward
  float *dout, *dqkv, *aux;
  (void)hipMalloc(&dout, (size_t)B * T * D * 4); (void)hipMalloc(&dqkv, h.size() * 4); (void)hipMalloc(&aux, ((size_t)B * H * T * 2 + 4) * 4);
  (void)hipMemcpy(dout, qkv, (size_t)B * T * D * 4, hipMemcpyDeviceToDevice);
  for (int which = 0; which < 2; ++which) {
    Attn2Args ab = a;
    ab.stamps = nullptr;
    for (int i = 0; i < 30; ++i) fs2_attn2_bwd(ab, o, dout, lse, aux, dqkv, 0);
    (void)hipDeviceSynchronize();
    (void)hipMemset(st, 0, nst * 8);
    ab.stamps = st;
    (void)hipEventRecord(e0);
    for (int i = 0; i < N; ++i) fs2_attn2_bwd_one(ab, dout, aux, dqkv, which, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double prods = which == 0 ? 3 : 4;
    printf("%s: %.1f us  %.1f TFLOP/s on %g products\n", which == 0 ? "dQ" : "dK/dV", ms / N * 1e3, prods / 2 * fl / (ms / N * 1e-3) / 1e12, prods);
    (void)hipMemcpy(all.data(), st, nst * 8, hipMemcpyDeviceToHost);
    long long hb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < nst; ++i) hb[i & 7] += all[i];
    const char* nb[8] = {"loop", "wait+barrier", "dot 1 (S)", "dot 2 (dP)", "first weight", "grad stream", "", ""};
    double tb = 0; for (int i = 0; i < 6; ++i) tb += hb[i];
    if (tb > 0) for (int i = 0; i < 6; ++i) printf("  %-14s %9.0f cycles/wave (%4.1f %%)\n", nb[i], hb[i] / waves, 100.0 * hb[i] / tb);
  }
  return 0;
}
