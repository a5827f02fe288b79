// Where does the attention forward spend its time?  The production kernel with parts switched off (results are
// meaningless; only the durations matter).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I fastspeech2_lightning_amd/csrc
//   FLAGS bit 0: no S = K Q^T products      bit 1: no softmax arithmetic (exp, hash, max/sum shuffles)
//         bit 2: no P V products            bit 3: K/V tile staged once, no loads / LDS writes in the loop
//         bit 4: no barriers in the loop (implies bit 3)
#include "../../fastspeech2_lightning_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>

namespace {
template <int HD, int FLAGS>
__global__ __launch_bounds__(256, 2) void probe_fwd(AttnP p, float* __restrict__ o, float* __restrict__ lse) {
  constexpr bool BF = false;
  constexpr int LDT = HD + 4, NJ = HD / 16;
  __shared__ __attribute__((aligned(16))) float Ks[64 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[64 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z, T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = blockIdx.x * 64 + wave * 16 + c;
  const int len = p.lens[b];
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const float* base = p.qkv + (long long)b * T * ld;
  Own<HD, BF> qr;
  qr.load([&](int j) {
    float4 v = q < T ? *reinterpret_cast<const float4*>(base + (long long)q * ld + h * HD + 16 * j + 4 * g)
                     : make_float4(0, 0, 0, 0);
    return make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
  });
  f32x4 oacc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) oacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const int kend = min(T, len);
  const unsigned long long rowidx = ((unsigned long long)(b * p.H + h) * T + q) * T;
  RowRegs<HD> kreg, vreg;
  fetch_rows<HD>(kreg, base, ld, D + h * HD, 0, T, tid);
  fetch_rows<HD>(vreg, base, ld, 2 * D + h * HD, 0, T, tid);
  if constexpr ((FLAGS & 32) != 0) {  // stagger: the workgroup in the odd wave slots starts half a tile period later
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_ID.WAVE_ID
    if (slot & 1) {
      __builtin_amdgcn_s_sleep(127);
      if constexpr ((FLAGS & 64) != 0) __builtin_amdgcn_s_sleep(127);
    }
  }
  if constexpr ((FLAGS & 24) != 0) {
    commit_rows<HD>(Ks, kreg, tid);
    commit_rows<HD>(Vs, vreg, tid);
    __syncthreads();
  }
  for (int key0 = 0; key0 < kend; key0 += 64) {
    if constexpr ((FLAGS & 24) == 0) {
      __syncthreads();
      commit_rows<HD>(Ks, kreg, tid);
      commit_rows<HD>(Vs, vreg, tid);
      __syncthreads();
      if (key0 + 64 < kend) {
        fetch_rows<HD>(kreg, base, ld, D + h * HD, key0 + 64, T, tid);
        fetch_rows<HD>(vreg, base, ld, 2 * D + h * HD, key0 + 64, T, tid);
      }
    } else if constexpr ((FLAGS & 16) == 0) {
      __syncthreads();
      __syncthreads();
    }
    f32x4 s[4];
    float mx = -INFINITY;
    if constexpr (FLAGS & 1) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[kt] = (f32x4){0.01f * c, 0.02f * g, 0.03f, 0.001f * key0};
    } else {
      dot_tiles<HD, BF, 4>(Ks, qr, 0, c, g, s);
    }
    float alpha = 1.f;
    if constexpr ((FLAGS & 2) == 0) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int key = key0 + 16 * kt + 4 * g + r;
          if (key >= len) s[kt][r] = -INFINITY;
          mx = fmaxf(mx, s[kt][r]);
        }
      }
      mx = xor_max16_32(mx);
      const float mnew = fmaxf(m, mx);
      alpha = __expf(m - mnew);
      float rs = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __expf(s[kt][r] - mnew);
          rs += pv;
          s[kt][r] = pv * fs2_drop_factor(drop, rowidx + (unsigned long long)(key0 + 16 * kt + 4 * g + r));
        }
      rs = xor_sum16_32(rs);
      l = l * alpha + rs;
      m = mnew;
    } else {
      l += s[0][0];
    }
    if constexpr ((FLAGS & 4) == 0) {
      W2<BF> pw[2];
      pw[0].set(s[0], s[1]);
      pw[1].set(s[2], s[3]);
#pragma unroll
      for (int dt = 0; dt < NJ; ++dt) {
        oacc[dt] *= alpha;
        oacc[dt] = acc_pair<HD, BF>(Vs, 1, pw[1], dt, c, g, acc_pair<HD, BF>(Vs, 0, pw[0], dt, c, g, oacc[dt]));
      }
    } else {
#pragma unroll
      for (int dt = 0; dt < NJ; ++dt) oacc[dt] = oacc[dt] * alpha + s[dt & 3];
    }
  }
  if (q < T) {
    const float inv = 1.f / (l + 1.f);
    float* orow = o + ((long long)b * T + q) * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt)
      *reinterpret_cast<float4*>(orow + 16 * dt + 4 * g) =
          make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
    if (g == 0) lse[((long long)b * p.H + h) * T + q] = m + l;
  }
}

// 32 keys per tile instead of 64: half the LDS (33.8 KB) and fewer live registers -> three workgroups per CU
template <int HD, int WGS>
__global__ __launch_bounds__(256, WGS) void probe_fwd32(AttnP p, float* __restrict__ o, float* __restrict__ lse) {
  constexpr bool BF = false;
  constexpr int LDT = HD + 4, NJ = HD / 16, F4 = HD / 4, NLD = (32 * F4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDT];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z, T = p.T, D = p.H * HD, ld = 3 * D;
  const int q = blockIdx.x * 64 + wave * 16 + c;
  const int len = p.lens[b];
  const Fs2Drop drop = fs2_resolve_drop(p.drop);
  const float* base = p.qkv + (long long)b * T * ld;
  Own<HD, BF> qr;
  qr.load([&](int j) {
    float4 v = q < T ? *reinterpret_cast<const float4*>(base + (long long)q * ld + h * HD + 16 * j + 4 * g)
                     : make_float4(0, 0, 0, 0);
    return make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
  });
  f32x4 oacc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) oacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const int kend = min(T, len);
  const unsigned long long rowidx = ((unsigned long long)(b * p.H + h) * T + q) * T;
  float4 kreg[NLD], vreg[NLD];
  auto fetch = [&](float4 (&regs)[NLD], int col, int row0) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      int idx = tid + it * 256, r = idx / F4, c4 = idx % F4, row = row0 + r;
      regs[it] = (idx < 32 * F4 && row < T) ? *reinterpret_cast<const float4*>(base + (long long)row * ld + col + c4 * 4)
                                             : make_float4(0, 0, 0, 0);
    }
  };
  auto commit = [&](float* dst, const float4 (&regs)[NLD]) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      int idx = tid + it * 256, r = idx / F4, c4 = idx % F4;
      if (idx < 32 * F4) *reinterpret_cast<float4*>(dst + r * LDT + c4 * 4) = regs[it];
    }
  };
  fetch(kreg, D + h * HD, 0);
  fetch(vreg, 2 * D + h * HD, 0);
  for (int key0 = 0; key0 < kend; key0 += 32) {
    __syncthreads();
    commit(Ks, kreg);
    commit(Vs, vreg);
    __syncthreads();
    if (key0 + 32 < kend) {
      fetch(kreg, D + h * HD, key0 + 32);
      fetch(vreg, 2 * D + h * HD, key0 + 32);
    }
    f32x4 s[2];
    float mx = -INFINITY;
    dot_tiles<HD, BF, 2>(Ks, qr, 0, c, g, s);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = key0 + 16 * kt + 4 * g + r;
        if (key >= len) s[kt][r] = -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = xor_max16_32(mx);
    const float mnew = fmaxf(m, mx);
    const float alpha = __expf(m - mnew);
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = __expf(s[kt][r] - mnew);
        rs += pv;
        s[kt][r] = pv * fs2_drop_factor(drop, rowidx + (unsigned long long)(key0 + 16 * kt + 4 * g + r));
      }
    rs = xor_sum16_32(rs);
    l = l * alpha + rs;
    m = mnew;
    W2<BF> pw;
    pw.set(s[0], s[1]);
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt) {
      oacc[dt] *= alpha;
      oacc[dt] = acc_pair<HD, BF>(Vs, 0, pw, dt, c, g, oacc[dt]);
    }
  }
  if (q < T) {
    const float inv = 1.f / l;
    float* orow = o + ((long long)b * T + q) * D + h * HD;
#pragma unroll
    for (int dt = 0; dt < NJ; ++dt)
      *reinterpret_cast<float4*>(orow + 16 * dt + 4 * g) =
          make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv);
    if (g == 0) lse[((long long)b * p.H + h) * T + q] = m + logf(l);
  }
}

template <int WGS>
float run32(AttnP p, float* o, float* lse, dim3 grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) probe_fwd32<128, WGS><<<grid, dim3(256)>>>(p, o, lse);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) probe_fwd32<128, WGS><<<grid, dim3(256)>>>(p, o, lse);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}

template <int FLAGS>
float run(AttnP p, float* o, float* lse, dim3 grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) probe_fwd<128, FLAGS><<<grid, dim3(256)>>>(p, o, lse);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) probe_fwd<128, FLAGS><<<grid, dim3(256)>>>(p, o, lse);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}
}  // namespace

int main() {
  const int B = 32, T = 648, H = 2, HD = 128, D = H * HD;
  std::vector<float> hq((size_t)B * T * 3 * D);
  unsigned st = 12345;
  for (auto& v : hq) { st = st * 1664525u + 1013904223u; v = ((st >> 8) & 0xffff) / 65536.f - 0.5f; }
  std::vector<int> hl(B, T);
  float *qkv, *o, *lse; int* lens;
  hipMalloc(&qkv, hq.size() * 4); hipMalloc(&o, (size_t)B * T * D * 4); hipMalloc(&lse, (size_t)B * H * T * 4); hipMalloc(&lens, B * 4);
  hipMemcpy(qkv, hq.data(), hq.size() * 4, hipMemcpyHostToDevice); hipMemcpy(lens, hl.data(), B * 4, hipMemcpyHostToDevice);
  AttnP p{qkv, lens, B, T, H, 1.f / sqrtf((float)HD), fs2_make_drop(0.1f, 777ull, nullptr)};
  dim3 grid((T + 63) / 64, H, B);
  printf("full kernel                         %7.1f us\n", run<0>(p, o, lse, grid));
  printf("no S products                       %7.1f us\n", run<1>(p, o, lse, grid));
  printf("no softmax arithmetic               %7.1f us\n", run<2>(p, o, lse, grid));
  printf("no PV products                      %7.1f us\n", run<4>(p, o, lse, grid));
  printf("no S, no PV (softmax + staging)     %7.1f us\n", run<5>(p, o, lse, grid));
  printf("no staging in the loop              %7.1f us\n", run<8>(p, o, lse, grid));
  printf("no staging, no barriers             %7.1f us\n", run<16>(p, o, lse, grid));
  printf("products only (no softmax/staging)  %7.1f us\n", run<18>(p, o, lse, grid));
  printf("S only                              %7.1f us\n", run<22>(p, o, lse, grid));
  printf("PV only                             %7.1f us\n", run<19>(p, o, lse, grid));
  printf("staging only                        %7.1f us\n", run<7>(p, o, lse, grid));
  printf("full kernel, staggered start (3.4us)%7.1f us\n", run<32>(p, o, lse, grid));
  printf("full kernel, staggered start (6.8us)%7.1f us\n", run<96>(p, o, lse, grid));
  printf("full kernel again                   %7.1f us\n", run<0>(p, o, lse, grid));
  printf("32-key tiles, 2 workgroups per CU   %7.1f us\n", run32<2>(p, o, lse, grid));
  printf("32-key tiles, 3 workgroups per CU   %7.1f us\n", run32<3>(p, o, lse, grid));
  return 0;
}
