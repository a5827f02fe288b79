// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS for out-of-range lanes (or leave LDS untouched)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* src, float* out, int soff, int nrec) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  lds[threadIdx.x] = -7.f; lds[threadIdx.x + 64] = -7.f; lds[threadIdx.x + 128] = -7.f; lds[threadIdx.x + 192] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
  int voff = threadIdx.x * 16;  // all in-range lanes stay inside the 16 KiB allocation in both passes
  if (threadIdx.x == 5) voff = 0x80000000;
  if (threadIdx.x == 9) voff = nrec;        // first byte out of range
  if (threadIdx.x == 11) voff = nrec - 8;   // partially out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n s_barrier" ::: "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = lds[threadIdx.x * 4 + j];
}
int main() {
  float *src, *out; float h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (float)i + 1.f;
  hipMalloc(&src, sizeof(h)); hipMalloc(&out, sizeof(o));
  hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass) {
    int soff = pass ? 4096 : 0, nrec = pass ? 2048 : 16384;  // pass 1: soffset beyond nrec -> is soffset range-checked?
    k<<<1, 64>>>(src, out, soff, nrec);
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    printf("pass %d (soff %d nrec %d): lane0 %g %g | lane5 (sentinel) %g %g %g %g | lane9 (==nrec) %g %g | lane11 (partial) %g %g %g %g | lane 12 %g\n",
           pass, soff, nrec, o[0], o[1], o[20], o[21], o[22], o[23], o[36], o[37], o[44], o[45], o[46], o[47], o[48]);
  }
  return 0;
}
