// Is the SGPR offset of a raw buffer access part of the range check on gfx950?  (hipcc --offload-arch=gfx950 -O2 -o /tmp/bsr ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* buf, int num_records_bytes, int voff, int soff, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, num_records_bytes, 0x00020000);
  out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 777.f), r, voff, soff, 0);
}
int main() {
  float *buf, *out;
  hipMalloc(&buf, 4096); hipMalloc(&out, 16);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  struct { int voff, soff; } cases[] = {{0, 0}, {252, 0}, {256, 0}, {0, 256}, {128, 128}, {128, 192}, {252, 4}};
  for (auto c : cases) {
    hipMemcpy(buf, h, 4096, hipMemcpyHostToDevice);
    k<<<1, 1>>>(buf, 256, c.voff, c.soff, out);  // num_records = 256 bytes = 64 floats
    float o, hb[1024];
    hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb, buf, 4096, hipMemcpyDeviceToHost);
    int idx = (c.voff + c.soff) / 4;
    printf("num_records 256 B, voffset %3d, soffset %3d (byte %3d): load -> %6.1f, store %s\n", c.voff, c.soff, c.voff + c.soff, o,
           hb[idx] == 777.f ? "LANDED" : "dropped");
  }
  return 0;
}
