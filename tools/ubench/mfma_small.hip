// f32 MFMA forms for skinny products (K = 1 or 4) on gfx950: operand / result lane maps and issue rates, alone and
// beside VALU work.  What the MLP stages of the step need to know before they lean on the matrix pipe
// (csrc/qc_mlp.hip: H = 50 hidden units, n = 4 circuit wires, six derivative channels).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_small mfma_small.hip && ./mfma_small
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// ---- lane maps, by exact small-integer data
__global__ void k_map_4x4x1(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
__global__ void k_map_16x16x4(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

// ---- issue rates: V = 0: 4x4x1 on 8 independent accumulators; 1: 16x16x4 on 8; 2: 4x4x1 + 4 v_fma per MFMA;
// 3: 16x16x4 + 16 v_fma per MFMA; 4: the v_fma alone (4 per slot); 5: 4x4x1 dependent chain on ONE accumulator
template <int V>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters) {
  f4 acc[8];
  float va[16];
  for (int i = 0; i < 8; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) va[i] = threadIdx.x * 1e-3f + i;
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (V == 0 || V == 2) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
      if constexpr (V == 1 || V == 3) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      if constexpr (V == 5) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0);
      if constexpr (V == 2 || V == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(va[(i * 4 + j) & 15]) : "v"(a), "v"(b));
      }
      if constexpr (V == 3) {
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(va[j]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += va[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
static void rate(const char* name, float* out, double mfma_per_it, double flop_per_mfma) {
  const int iters = 4000, blocks = 256 * 2;   // 2 blocks of 4 waves per CU: 2 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<V>), dim3(blocks), dim3(256), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<V>), dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double slots = (double)iters * 8;   // MFMA slots (or VALU groups) per wave
  const double ns_per_slot_per_simd = ms * 1e6 / (slots * 2);   // 2 waves per SIMD share it
  printf("%-44s %8.3f ms  %7.2f ns per slot per SIMD", name, ms, ns_per_slot_per_simd);
  if (mfma_per_it > 0) printf("  -> %6.1f TFLOP/s on the matrix pipe", flop_per_mfma / ns_per_slot_per_simd * 1024 / 1e3);
  printf("\n");
}

int main() {
  float *a, *b, *d;
  hipMalloc(&a, 64 * 4);
  hipMalloc(&b, 64 * 4);
  hipMalloc(&d, 256 * 4 * 4 * 512);
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; ++l) {
    ha[l] = (float)(l + 1);          // A value identifies its lane
    hb[l] = (float)(1 << (l & 3)) * (1 + (l >> 2) * 0);   // B = 1, 2, 4, 8 by lane & 3
  }
  hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_map_4x4x1, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
  printf("v_mfma_f32_4x4x1_16b_f32: A[l] = l + 1, B[l] = 1 << (l & 3); D[lane][reg] =\n");
  for (int l = 0; l < 64; ++l) printf("%s lane %2d: %6.0f %6.0f %6.0f %6.0f%s", (l & 1) ? "   |" : "", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3], (l & 1) ? "\n" : "");
  // 16x16x4: A[l] = 1 + (l & 15) (row), scaled by 100^(l >> 4) would overflow exactness; use two probes
  for (int l = 0; l < 64; ++l) {
    ha[l] = (float)(1 + (l & 15)) * ((l >> 4) == 0 ? 1.f : 0.f);   // only k = 0 contributes: A[row][0] = row + 1
    hb[l] = (float)(100 * (1 + (l & 15))) * ((l >> 4) == 0 ? 1.f : 0.f);   // B[0][col] = 100 (col + 1)
  }
  hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_map_16x16x4, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, sizeof(hd), hipMemcpyDeviceToHost);
  printf("v_mfma_f32_16x16x4_f32, k = 0 only: D = (row + 1) * 100 (col + 1); lanes 0, 1, 16, 17, 63:\n");
  const int ls[5] = {0, 1, 16, 17, 63};
  for (int l : ls) printf("  lane %2d: %7.0f %7.0f %7.0f %7.0f\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
  rate<0>("4x4x1_16b, 8 accumulators", d, 1, 512);
  rate<5>("4x4x1_16b, one accumulator (dependent)", d, 1, 512);
  rate<1>("16x16x4, 8 accumulators", d, 1, 2048);
  rate<4>("4 v_fma_f32 per slot, no MFMA", d, 0, 0);
  rate<2>("4x4x1_16b + 4 v_fma_f32 per slot", d, 1, 512);
  rate<3>("16x16x4 + 16 v_fma_f32 per slot", d, 1, 2048);
  return 0;
}
