#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(h4* out, const h4* in) {
  __shared__ __align__(16) _Float16 buf[64*4*2];
  const unsigned l = threadIdx.x;
  *reinterpret_cast<h4*>(buf + 4*l) = in[l];
  __builtin_amdgcn_wave_barrier();
  fp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(buf + 4*l));
  h4 o; __builtin_memcpy(&o, &r, 8);
  out[l] = o;
}
int main() {
  h4 hin[64], hout[64];
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) hin[l][j] = (_Float16)(l * 4 + j);
  h4 *din, *dout; hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dout, din);
  hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) printf("lane %2d: %3d %3d %3d %3d\n", l, (int)hout[l][0], (int)hout[l][1], (int)hout[l][2], (int)hout[l][3]);
  return 0;
}
