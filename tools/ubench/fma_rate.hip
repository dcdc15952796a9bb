#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_scalar(float* out, float a, float b, int iters) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_packed(float* out, float a, float b, int iters) {
  f2 x[8];
  for (int i = 0; i < 8; ++i) { x[i].x = threadIdx.x * 0.001f + i; x[i].y = x[i].x + 0.5f; }
  f2 av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
  const int iters = 4096;
  for (int wpb = 1; wpb <= 8; wpb *= 2) {   // waves per SIMD: blocks of 256 thr = 1 wave/SIMD each; wpb blocks per CU
    int grid = 256 * wpb;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; ++variant) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        else hipLaunchKernelGGL(k_packed, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * 256 * iters * 16 * 2;
      printf("waves/SIMD=%d %s: %.3f ms  %.1f TFLOP/s\n", wpb, variant ? "v_pk_fma_f32" : "v_fma_f32  ", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
