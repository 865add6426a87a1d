// probe: does a buffer_load ... lds with an out-of-range offset write ZEROS into LDS (or leave it untouched)?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  unsigned *d, *o, h[256], ho[256];
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1024u, o);
  hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 8; ++l) printf("lane %d: %x %x %x %x\n", l, ho[l * 4], ho[l * 4 + 1], ho[l * 4 + 2], ho[l * 4 + 3]);
  return 0;
}
