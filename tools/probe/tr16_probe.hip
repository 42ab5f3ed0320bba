// What does ds_read_b64_tr_b16 return?  Each lane passes the LDS address of 4 contiguous 16-bit values
// (lane i of a 16-lane group: element 4*i of that group's 64-element block); print what each lane gets.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short buf[256];
  for (int i = threadIdx.x; i < 256; i += 64) buf[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  short* p = buf + (l >> 4) * 64 + (l & 15) * 4;
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512); short h[256];
  k<<<1, 64>>>(d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0;
}
