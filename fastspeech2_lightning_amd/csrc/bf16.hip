// bf16 operand storage for the "bf16-mixed" GEMMs (Fs2GemmArgs.operand_bf16 == 3): casts of fp32 tensors to the
// k-contiguous bf16 rows the GEMM core reads -- weights once per step (as stored, and transposed for the data-gradient
// GEMM), activations where their producer does not emit the bf16 copy itself.  HBM-bound: 4 B read + 2 B written per
// element.  Rounding is to nearest even (v_cvt_pk_bf16_f32), as in the register-rounding mode.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                         long long n8) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
    const f32x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    reinterpret_cast<bf16x8_t*>(dst)[i] = __builtin_convertvector(v, bf16x8_t);
  }
}

// out[(b, t)][tap * C + c] = bf16(x[(b, t + dir * (tap - (taps - 1) / 2))][c]), zero outside the utterance's [0, T): the
// rows of a k-tap 'same' convolution over time laid side by side, so that a convolution whose channel count is not whole
// 64-deep K-tiles per tap (the PostNet's 80 mel bins, fs2/layers.py:143-212) is ONE plain GEMM with K = taps * C on the
// bf16-storage core instead of a per-piece-decoded launch on the fp32-operand tiles.  dir = +1: forward (pairs with the
// weight as [taps * Cin][Cout]); dir = -1: the data gradient (pairs with the weight as stored, [taps * Cout][Cin] read
// reduction-major).  Eight elements (one 16-byte store) per thread; C % 8 == 0 keeps a piece inside one tap.
template <bool XB>
__global__ __launch_bounds__(256) void im2col_taps_kernel(const void* __restrict__ x, __bf16* __restrict__ out, int T,
                                                           int C, int taps, int dir, long long n8) {
  const int c8 = C / 8, w8 = taps * c8, pad = (taps - 1) / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / w8;
    const int j = (int)(i - row * w8), tap = j / c8, c = (j - tap * c8) * 8;
    const int t = (int)(row % T), ts = t + dir * (tap - pad);
    bf16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ts >= 0 && ts < T) {
      const long long src = (row + (ts - t)) * C + c;
      if (XB) {
        v = *reinterpret_cast<const bf16x8_t*>((const __bf16*)x + src);
      } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>((const float*)x + src);
        const f32x4 b = *reinterpret_cast<const f32x4*>((const float*)x + src + 4);
        const f32x8_t f = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        v = __builtin_convertvector(f, bf16x8_t);
      }
    }
    reinterpret_cast<bf16x8_t*>(out)[i] = v;
  }
}

// dst[c][r] = bf16(src[r][c]) for a [rows][cols] matrix: 64 x 64 tiles through LDS (padded against bank conflicts),
// reads and writes both contiguous.  dst rows are ld_dst elements apart; the pad columns rows..ld_dst-1 are zeroed.
__global__ __launch_bounds__(256) void transpose_cast_bf16_kernel(const float* __restrict__ src, int rows, int cols,
                                                                   int ld_src, __bf16* __restrict__ dst, int ld_dst) {
  __shared__ float tile[64][65];
  src += (long long)blockIdx.z * rows * ld_src;  // (batch of matrices: the taps of a convolution weight)
  dst += (long long)blockIdx.z * cols * ld_dst;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(long long)r * ld_src + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < ld_dst) dst[(long long)c * ld_dst + r] = (__bf16)tile[tx][i];
  }
}

// several matrices in one launch (the transposed bf16 mirrors of the K = 256 data-gradient weights, once per step)
struct TransposeJobs {
  Fs2TransposeJob j[FS2_TRANSPOSE_MAX_JOBS];
};
__global__ __launch_bounds__(256) void transpose_cast_bf16_multi_kernel(TransposeJobs jobs) {
  __shared__ float tile[64][65];
  const Fs2TransposeJob& jb = jobs.j[blockIdx.y];
  const int tc = (jb.cols + 63) / 64, tr = (jb.rows + 63) / 64;
  if ((int)blockIdx.x >= tc * tr) return;  // uniform per workgroup
  const int r0 = ((int)blockIdx.x / tc) * 64, c0 = ((int)blockIdx.x % tc) * 64;
  const float* __restrict__ src = jb.src;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < jb.rows && c < jb.cols) ? src[(long long)r * jb.cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < jb.cols && r < jb.rows) {
      if (jb.fp32_out) ((float*)jb.dst)[(long long)c * jb.rows + r] = tile[tx][i];
      else ((__bf16*)jb.dst)[(long long)c * jb.rows + r] = (__bf16)tile[tx][i];
    }
  }
}

}  // namespace

extern "C" int fs2hip_transpose_cast_bf16_multi(const Fs2TransposeJob* jobs, int njobs, void* stream) {
  if (njobs <= 0) return 0;
  if (!jobs || njobs > FS2_TRANSPOSE_MAX_JOBS) return FS2HIP_EINVAL;
  TransposeJobs arg;
  int tmax = 0;
  for (int i = 0; i < njobs; ++i) {
    const Fs2TransposeJob& j = jobs[i];
    if (!j.src || !j.dst || j.rows <= 0 || j.cols <= 0) return FS2HIP_EINVAL;
    arg.j[i] = j;
    const int t = ((j.rows + 63) / 64) * ((j.cols + 63) / 64);
    tmax = t > tmax ? t : tmax;
  }
  transpose_cast_bf16_multi_kernel<<<dim3(tmax, njobs), dim3(256), 0, (hipStream_t)stream>>>(arg);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_cast_bf16(const float* src, void* dst, long long n, void* stream) {
  if (n <= 0 || (n % 8) || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16)) return FS2HIP_EINVAL;
  const long long n8 = n / 8;
  long long blocks = (n8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  cast_bf16_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(src, (__bf16*)dst, n8);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_im2col_taps(const void* x, int x_bf16, void* out_bf16, int B, int T, int C, int taps, int dir,
                                  void* stream) {
  if (B <= 0 || T <= 0 || C <= 0 || (C % 8) || taps < 1 || !(taps & 1) || (dir != 1 && dir != -1)) return FS2HIP_EINVAL;
  if (((uintptr_t)x % 16) || ((uintptr_t)out_bf16 % 16)) return FS2HIP_EINVAL;
  const long long n8 = (long long)B * T * taps * (C / 8);
  long long blocks = (n8 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipStream_t s = (hipStream_t)stream;
  if (x_bf16) im2col_taps_kernel<true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (__bf16*)out_bf16, T, C, taps, dir, n8);
  else im2col_taps_kernel<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, (__bf16*)out_bf16, T, C, taps, dir, n8);
  FS2_LAUNCH_CHECK();
  return 0;
}

extern "C" int fs2hip_transpose_cast_bf16(const float* src, int rows, int cols, int ld_src, void* dst, int ld_dst,
                                          int batch, void* stream) {
  if (rows <= 0 || cols <= 0 || ld_src < cols || ld_dst < rows || batch < 1 || batch > 65535) return FS2HIP_EINVAL;
  dim3 grid((cols + 63) / 64, (ld_dst + 63) / 64, batch);
  transpose_cast_bf16_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>(src, rows, cols, ld_src, (__bf16*)dst, ld_dst);
  FS2_LAUNCH_CHECK();
  return 0;
}
