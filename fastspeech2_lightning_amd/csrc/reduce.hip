// Row reductions of dense [rows][n] fp32 matrices (bias gradients, split-K slabs, per-workgroup
// partial sums of the norm / conv kernels).  HBM-bound; deterministic (fixed summation order).
//
//   wide   : [M][N] with M large, N <= 1024: a workgroup sweeps whole rows (every wavefront
//            instruction is a contiguous >= 1 KiB piece) and leaves one partial row per 128 rows;
//   small  : few hundred rows, n <= 16384: 64 columns x 16 row lanes per 1024-thread workgroup;
//   slabs  : few slabs of a large n (split-K): column-parallel float4.
#include "common.h"

namespace {

constexpr int WIDE_ROWS = 64;

// NACC accumulators per thread: 1 = sum(x); 2 = (sum x, sum x*x)
__global__ __launch_bounds__(256) void colsum_wide_kernel(const float* __restrict__ x, int ldx, int M, int N,
                                                           float* __restrict__ partial, int tpr, int rpi) {
  __shared__ float4 red[256];
  const int tid = threadIdx.x;
  const int rl = tid / tpr, c4 = tid - rl * tpr;
  const bool active = rl < rpi;
  const int r0 = blockIdx.x * WIDE_ROWS, r1 = min(M, r0 + WIDE_ROWS);
  float4 acc = make_float4(0, 0, 0, 0);
  if (active) {
    int r = r0 + rl;
    for (; r + 3 * rpi < r1; r += 4 * rpi) {  // four independent loads in flight per thread
      const float* q = x + (long long)r * ldx + c4 * 4;
      float4 v0 = *reinterpret_cast<const float4*>(q);
      float4 v1 = *reinterpret_cast<const float4*>(q + (long long)rpi * ldx);
      float4 v2 = *reinterpret_cast<const float4*>(q + 2LL * rpi * ldx);
      float4 v3 = *reinterpret_cast<const float4*>(q + 3LL * rpi * ldx);
      acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
      acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; r < r1; r += rpi) {
      float4 v = *reinterpret_cast<const float4*>(x + (long long)r * ldx + c4 * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (rl == 0) {
    for (int l = 1; l < rpi; ++l) {
      float4 v = red[l * tpr + c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + (long long)blockIdx.x * N + c4 * 4) = acc;
  }
}

// generic strided fallback (N not a multiple of 4 or wider than 1024)
__global__ __launch_bounds__(256) void colsum_strided_kernel(const float* __restrict__ x, int ldx, int M, int N,
                                                              float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * WIDE_ROWS, r1 = min(M, r0 + WIDE_ROWS);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) s += x[(long long)r * ldx + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N)
    partial[(long long)blockIdx.y * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(1024) void reduce_rows_small_kernel(const float* __restrict__ src, int rows, int n,
                                                                  long long stride, float* __restrict__ out0, int n0,
                                                                  float* __restrict__ out1) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < n) {
    int r = rl;
    for (; r + 48 < rows; r += 64) {
      float a = src[(long long)r * stride + c], b = src[(long long)(r + 16) * stride + c];
      float d = src[(long long)(r + 32) * stride + c], e = src[(long long)(r + 48) * stride + c];
      s += (a + b) + (d + e);
    }
    for (; r < rows; r += 16) s += src[(long long)r * stride + c];
  }
  red[rl][lane] = s;
  __syncthreads();
  if (rl == 0 && c < n) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][lane];
    if (c < n0) out0[c] = t; else out1[c - n0] = t;
  }
}

// Several independent row reductions in one launch (blockIdx.y = job): the second stage of the bias-gradient
// column sums and of the LayerNorm parameter gradients.  Their only consumer is the optimizer (or the
// data-parallel bucket exchange), so the host collects them and finishes up to FS2_REDUCE_MAX_JOBS at a time
// instead of paying one 4 us launch each (~200 per step).
struct ReduceJobs {
  Fs2ReduceJob j[FS2_REDUCE_MAX_JOBS];
};
__global__ __launch_bounds__(1024) void reduce_rows_multi_kernel(ReduceJobs jobs) {
  __shared__ float red[16][64];
  const Fs2ReduceJob& jb = jobs.j[blockIdx.y];
  const int lane = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  if ((int)(blockIdx.x * 64) >= jb.n) return;  // uniform per workgroup
  const float* __restrict__ src = jb.src;
  const long long stride = jb.stride;
  const int rows = jb.rows;
  float s = 0.f;
  if (c < jb.n) {
    int r = rl;
    for (; r + 48 < rows; r += 64) {
      float a = src[(long long)r * stride + c], b = src[(long long)(r + 16) * stride + c];
      float d = src[(long long)(r + 32) * stride + c], e = src[(long long)(r + 48) * stride + c];
      s += (a + b) + (d + e);
    }
    for (; r < rows; r += 16) s += src[(long long)r * stride + c];
  }
  red[rl][lane] = s;
  __syncthreads();
  if (rl == 0 && c < jb.n) {
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += red[l][lane];
    if (c < jb.n0) jb.out0[c] = t; else jb.out1[c - jb.n0] = t;
  }
}

template <bool VEC>
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                    int nslabs, long long stride) {
  if (VEC) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long step = (long long)gridDim.x * blockDim.x * 4;
    for (; i < n; i += step) {
      if (i + 3 < n) {
        float4 s = *reinterpret_cast<const float4*>(slabs + i);
        for (int k = 1; k < nslabs; ++k) {
          float4 v = *reinterpret_cast<const float4*>(slabs + k * stride + i);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + i) = s;
      } else {
        for (long long e = i; e < n; ++e) {
          float s = slabs[e];
          for (int k = 1; k < nslabs; ++k) s += slabs[k * stride + e];
          out[e] = s;
        }
      }
    }
  } else {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
      float s = slabs[i];
      for (int k = 1; k < nslabs; ++k) s += slabs[k * stride + i];
      out[i] = s;
    }
  }
}

// Many split-K finishes in one launch.  The weight gradients are only read by the optimizer / the data-parallel bucket
// exchange, so their slab sums need not follow each GEMM as a launch of their own (87 per bf16-mixed step, 5.6 us each,
// most of it launch and drain): job blockIdx.y is summed by the workgroups of its grid row.
struct SlabJobs {
  Fs2SlabJob j[FS2_REDUCE_MAX_JOBS];
};
__global__ __launch_bounds__(256) void reduce_slabs_multi_kernel(SlabJobs jobs) {
  const Fs2SlabJob& jb = jobs.j[blockIdx.y];
  const float* __restrict__ slabs = jb.slabs;
  float* __restrict__ out = jb.out;
  const long long n = jb.n, stride = jb.stride;
  const int nslabs = jb.nslabs;
  if (jb.vec) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
      float4 s = reinterpret_cast<const float4*>(slabs)[i];
      int k = 1;
      for (; k + 1 < nslabs; k += 2) {
        const float4 a = reinterpret_cast<const float4*>(slabs + k * stride)[i];
        const float4 b = reinterpret_cast<const float4*>(slabs + (k + 1) * stride)[i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      }
      if (k < nslabs) {
        const float4 a = reinterpret_cast<const float4*>(slabs + k * stride)[i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      }
      reinterpret_cast<float4*>(out)[i] = s;
    }
  } else {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
      float s = slabs[i];
      for (int k = 1; k < nslabs; ++k) s += slabs[k * stride + i];
      out[i] = s;
    }
  }
}

// C[m][n] = alpha * sum_s slab_s[m - m0][n] + bias[n] for the rows [m0, Mc) whose tiles a persistent GEMM cut
// along the reduction (gemm2p.hip, tiles 13/14)
__global__ __launch_bounds__(256) void tail_fixup_kernel(const float* __restrict__ ws, int S, long long slab,
                                                         float* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                         float alpha, int m0, int Mc, int Nc) {
  const int n4 = Nc >> 2;
  const long long total = (long long)(Mc - m0) * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / n4), c4 = (int)(i - (long long)r * n4);
    const float* src = ws + (long long)r * Nc + c4 * 4;
    float4 acc = *reinterpret_cast<const float4*>(src);
    for (int k = 1; k < S; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(src + k * slab);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float4 b = make_float4(0, 0, 0, 0);
    if (bias) b = *reinterpret_cast<const float4*>(bias + c4 * 4);
    float* dst = C + (long long)(m0 + r) * ldc + c4 * 4;
    *reinterpret_cast<float4*>(dst) =
        make_float4(alpha * acc.x + b.x, alpha * acc.y + b.y, alpha * acc.z + b.z, alpha * acc.w + b.w);
  }
}

}  // namespace

// internal (declared in gemm_common.h)
int fs2_tail_fixup(const float* ws, int S, long long slab, float* C, int ldc, const float* bias, float alpha, int m0,
                   int Mc, int Nc, hipStream_t s) {
  const long long total = (long long)(Mc - m0) * (Nc >> 2);
  if (total <= 0) return 0;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  tail_fixup_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(ws, S, slab, C, ldc, bias, alpha, m0, Mc, Nc);
  FS2_LAUNCH_CHECK();
  return 0;
}

// internal (declared in common.h): out0[c] for c < n0, out1[c - n0] otherwise
int fs2_reduce_rows(const float* src, int rows, int n, long long stride, float* out0, int n0, float* out1,
                    hipStream_t s) {
  if (rows <= 0 || n <= 0) return FS2HIP_EINVAL;
  reduce_rows_small_kernel<<<dim3((n + 63) / 64), dim3(1024), 0, s>>>(src, rows, n, stride, out0, n0, out1);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_reduce_rows_multi(const Fs2ReduceJob* jobs, int njobs, void* stream) {
  if (njobs <= 0) return 0;
  if (!jobs || njobs > FS2_REDUCE_MAX_JOBS) return FS2HIP_EINVAL;
  ReduceJobs arg;
  int nmax = 0;
  for (int i = 0; i < njobs; ++i) {
    const Fs2ReduceJob& j = jobs[i];
    if (!j.src || !j.out0 || j.rows <= 0 || j.n <= 0 || j.n0 < 0 || (j.n0 < j.n && !j.out1)) return FS2HIP_EINVAL;
    arg.j[i] = j;
    nmax = j.n > nmax ? j.n : nmax;
  }
  reduce_rows_multi_kernel<<<dim3((nmax + 63) / 64, njobs), dim3(1024), 0, (hipStream_t)stream>>>(arg);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_reduce_slabs(const float* slabs, float* out, long long n, int nslabs,
                                   long long slab_stride, void* stream) {
  if (n <= 0) return 0;
  if (nslabs < 1) return FS2HIP_EINVAL;
  if (n <= 16384 && nslabs >= 8)
    return fs2_reduce_rows(slabs, nslabs, (int)n, slab_stride, out, (int)n, nullptr, (hipStream_t)stream);
  const bool vec = (slab_stride % 4) == 0 && ((uintptr_t)slabs % 16) == 0 && ((uintptr_t)out % 16) == 0;
  long long blocks = ((vec ? n / 4 : n) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  if (vec)
    reduce_slabs_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(slabs, out, n, nslabs, slab_stride);
  else
    reduce_slabs_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(slabs, out, n, nslabs, slab_stride);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_reduce_slabs_multi(const Fs2SlabJob* jobs, int njobs, void* stream) {
  if (njobs <= 0) return 0;
  if (!jobs || njobs > FS2_REDUCE_MAX_JOBS) return FS2HIP_EINVAL;
  SlabJobs arg;
  long long nmax = 0;
  for (int i = 0; i < njobs; ++i) {
    Fs2SlabJob j = jobs[i];
    if (!j.slabs || !j.out || j.n <= 0 || j.nslabs < 1 || (j.nslabs > 1 && j.stride < j.n)) return FS2HIP_EINVAL;
    // (the same summation order as fs2hip_reduce_slabs: slab 0, 1, 2, ... per element)
    j.vec = (j.n % 4) == 0 && (j.stride % 4) == 0 && ((uintptr_t)j.slabs % 16) == 0 && ((uintptr_t)j.out % 16) == 0;
    arg.j[i] = j;
    nmax = j.n > nmax ? j.n : nmax;
  }
  long long blocks = (nmax / 4 + 255) / 256;
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  reduce_slabs_multi_kernel<<<dim3((unsigned)blocks, njobs), dim3(256), 0, (hipStream_t)stream>>>(arg);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_colsum_rows(int M) { return (M + WIDE_ROWS - 1) / WIDE_ROWS; }

extern "C" int fs2hip_colsum(const float* x, int ldx, int M, int N, float* partial, float* out, void* stream) {
  if (M <= 0 || N <= 0) return FS2HIP_EINVAL;
  const int gy = fs2hip_colsum_rows(M);
  hipStream_t s = (hipStream_t)stream;
  if ((N % 4) == 0 && N <= 1024 && (ldx % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)partial % 16) == 0) {
    const int tpr = N / 4, rpi = 256 / tpr;
    colsum_wide_kernel<<<dim3(gy), dim3(256), 0, s>>>(x, ldx, M, N, partial, tpr, rpi);
  } else {
    colsum_strided_kernel<<<dim3((N + 63) / 64, gy), dim3(256), 0, s>>>(x, ldx, M, N, partial);
  }
  FS2_LAUNCH_CHECK();
  if (!out) return 0;  // partial sums only: the caller finishes them with fs2hip_reduce_rows_multi
  return fs2_reduce_rows(partial, gy, N, N, out, N, nullptr, s);
}
