#include <hip/hip_runtime.h>
__device__ __forceinline__ float xor16(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const bool odd_row = (threadIdx.x >> 4) & 1;
  return __builtin_bit_cast(float, odd_row ? r[0] : r[1]);
}
__device__ __forceinline__ float xor32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const bool hi = (threadIdx.x >> 5) & 1;
  return __builtin_bit_cast(float, hi ? r[0] : r[1]);
}
__device__ __forceinline__ float xor4(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true);
  t = __builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, true);
  return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float xor8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}
__global__ void k(float* out, const float* in) {
  float v = in[threadIdx.x];
  out[threadIdx.x] = xor16(v);
  out[64 + threadIdx.x] = xor32(v);
  out[128 + threadIdx.x] = xor4(v);
  out[192 + threadIdx.x] = xor8(v);
  out[256 + threadIdx.x] = __shfl_xor(v, 16);
}
int main() {
  float *in, *out; hipMalloc(&in, 256); hipMalloc(&out, 5 * 256);
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = i;
  hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, in);
  float o[320]; hipMemcpy(o, out, 1280, hipMemcpyDeviceToHost);
  int bad = 0;
  const int x[5] = {16, 32, 4, 8, 16};
  for (int t = 0; t < 5; ++t) for (int i = 0; i < 64; ++i) if (o[t * 64 + i] != (float)(i ^ x[t])) { ++bad; if (bad < 10) printf("t%d lane %d got %g want %d\n", t, i, o[t*64+i], i ^ x[t]); }
  printf("bad=%d\n", bad);
  return bad != 0;
}
