// probe: v_mfma_f32_16x16x16_bf16 on gfx950 -- operand layout, and destination overlapping a source
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// mode 0: builtin (compiler may overlap dst and srcA); mode 1: asm with early-clobber destination
template <int MODE>
__global__ void k(float* out, const float* A, const float* B) {
  int l = threadIdx.x;
  int r = l & 15, g = l >> 4;
  uint2 ua = make_uint2(cvt_pk_bf16(A[r * 16 + 4 * g], A[r * 16 + 4 * g + 1]), cvt_pk_bf16(A[r * 16 + 4 * g + 2], A[r * 16 + 4 * g + 3]));
  uint2 ub = make_uint2(cvt_pk_bf16(B[(4 * g) * 16 + r], B[(4 * g + 1) * 16 + r]), cvt_pk_bf16(B[(4 * g + 2) * 16 + r], B[(4 * g + 3) * 16 + r]));
  f32x4 acc = {0, 0, 0, 0};
  if (MODE == 2) {  // compiler-generated conversions (8 wide, as in the GEMM), builtin MFMA
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x8 v = {A[r * 16 + 4 * g], A[r * 16 + 4 * g + 1], A[r * 16 + 4 * g + 2], A[r * 16 + 4 * g + 3],
               B[(4 * g) * 16 + r], B[(4 * g + 1) * 16 + r], B[(4 * g + 2) * 16 + r], B[(4 * g + 3) * 16 + r]};
    bf16x8 w = __builtin_convertvector(v, bf16x8);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_shufflevector(w, w, 0, 1, 2, 3), __builtin_shufflevector(w, w, 4, 5, 6, 7), acc, 0, 0, 0);
  } else if (MODE == 3) {  // asm conversions, each followed by a wait state
    unsigned a0, a1, b0, b1;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 1" : "=v"(a0) : "v"(A[r * 16 + 4 * g]), "v"(A[r * 16 + 4 * g + 1]));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 1" : "=v"(a1) : "v"(A[r * 16 + 4 * g + 2]), "v"(A[r * 16 + 4 * g + 3]));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 1" : "=v"(b0) : "v"(B[(4 * g) * 16 + r]), "v"(B[(4 * g + 1) * 16 + r]));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2\n\ts_nop 1" : "=v"(b1) : "v"(B[(4 * g + 2) * 16 + r]), "v"(B[(4 * g + 3) * 16 + r]));
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4, make_uint2(a0, a1)), __builtin_bit_cast(bf16x4, make_uint2(b0, b1)), acc, 0, 0, 0);
  } else if (MODE == 4) {  // asm conversions, builtin MFMA, operands kept alive behind it (no destination overlap)
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4, ua), __builtin_bit_cast(bf16x4, ub), acc, 0, 0, 0);
    if (out[300] == 123.f) { out[301] = __uint_as_float(ua.x ^ ua.y ^ ub.x ^ ub.y); }
  } else if (MODE == 0) {
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(bf16x4, ua), __builtin_bit_cast(bf16x4, ub), acc, 0, 0, 0);
  } else {
    f32x2 fa = __builtin_bit_cast(f32x2, ua), fb = __builtin_bit_cast(f32x2, ub);
    asm volatile("s_nop 4\n\tv_mfma_f32_16x16x16_bf16 %0, %1, %2, 0\n\ts_nop 15" : "=&v"(acc) : "v"(fa), "v"(fb));
  }
  for (int e = 0; e < 4; ++e) out[(4 * g + e) * 16 + r] = acc[e];  // C row = 4g+e, col = l&15
}
int main() {
  float hA[256], hB[256], hC[256], ref[256];
  float *dout, *dA, *dB;
  hipMalloc(&dout, sizeof hC + 1024); hipMemset(dout, 0, sizeof hC + 1024); hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB);
  for (int pat = 0; pat < 4; ++pat) {
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      hA[i * 16 + j] = pat == 0 ? 1.f : pat == 1 ? (float)i : pat == 2 ? 1.f : (float)((i * 16 + j) * 7 % 13 - 6);   // A[i][k]
      hB[i * 16 + j] = pat == 0 ? 1.f : pat == 1 ? 1.f : pat == 2 ? (float)j : (float)((i * 16 + j) * 5 % 11 - 5);  // B[k][j]
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 5; ++mode) {
      if (mode == 0) k<0><<<1, 64>>>(dout, dA, dB); else if (mode == 1) k<1><<<1, 64>>>(dout, dA, dB);
      else if (mode == 2) k<2><<<1, 64>>>(dout, dA, dB); else if (mode == 3) k<3><<<1, 64>>>(dout, dA, dB); else k<4><<<1, 64>>>(dout, dA, dB);
      hipMemcpy(hC, dout, sizeof hC, hipMemcpyDeviceToHost);
      double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hC[i] - ref[i]));
      printf("pattern %d mode %d: max err %g   C[0][0..3] = %g %g %g %g  C[1][0]=%g C[5][0]=%g (ref %g %g %g %g | %g %g)\n", pat, mode, err, hC[0], hC[1], hC[2], hC[3],
             hC[16], hC[80], ref[0], ref[1], ref[2], ref[3], ref[16], ref[80]);
    }
  }
  return 0;
}
