// Issue rate of scalar vs packed fp32 VALU instructions on gfx950, per SIMD, as a function of resident waves.
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
// Each variant is a loop of 32 independent instructions on 32 accumulators (no back-to-back dependencies), written
// in inline asm so the compiler cannot repack them.  Output: wave-instructions per ns per SIMD and flop rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template <int V>
__global__ void __launch_bounds__(256) k(float* out, float sa, float sb, int iters) {
  f2 acc[16];
  const float t0 = threadIdx.x * 1e-3f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f2){t0 + i, t0 - i};
  f2 m = {1.0001f + t0 * 1e-6f, 0.9999f}, a = {0.5f, 0.25f};
  f2 sp = {sa, sb};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (V == 0) {   // 2 x v_fma_f32, VGPR operands
        asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x), "v"(a.x));
      } else if constexpr (V == 1) {   // 2 x v_fma_f32 with an SGPR multiplier
        asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(acc[i].x), "+v"(acc[i].y) : "s"(sa), "v"(a.x));
      } else if constexpr (V == 2) {   // 2 x v_pk_fma_f32, VGPR operands (4 FMAs)
        asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(m), "v"(a));
      } else if constexpr (V == 3) {   // 2 x v_pk_fma_f32, SGPR pair broadcast (op_sel_hi) as multiplier
        asm volatile("v_pk_fma_f32 %0, %2, %0, %3 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %2, %1, %3 op_sel_hi:[0,1,1]"
                     : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "s"(sp), "v"(a));
      } else if constexpr (V == 4) {   // 2 x v_pk_fma_f32 with a half swap + negate on a VGPR source
        asm volatile("v_pk_fma_f32 %0, %2, %0, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n"
                     "v_pk_fma_f32 %1, %2, %1, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                     : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(m), "v"(a));
      } else if constexpr (V == 5) {   // 2 x v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(m));
      } else if constexpr (V == 6) {   // 2 x v_mul_f32
        asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x));
      } else if constexpr (V == 7) {   // 2 x v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(a));
      } else if constexpr (V == 8) {   // dependent chain: v_pk_fma_f32 feeding the next one (latency / hazard)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[0]) : "v"(m), "v"(a));
      } else if constexpr (V == 9) {   // dependent chain: v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(acc[0].x) : "v"(m.x), "v"(a.x));
      } else if constexpr (V == 10) {  // v_add_f32 with a DPP row permute + a plain v_fma
        asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fma_f32 %1, %1, %2, %3"
                     : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x), "v"(a.x));
      } else if constexpr (V == 11) {  // v_mov_b32 x 2
        asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x));
      } else if constexpr (V == 12) {  // 2 x v_fmac_f32 (VOP2 encoding: dst += a * b), VGPR operands
        asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x), "v"(a.x));
      } else if constexpr (V == 13) {  // 2 x v_fmac_f32 with an SGPR multiplier
        asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3" : "+v"(acc[i].x), "+v"(acc[i].y) : "s"(sa), "v"(a.x));
      } else if constexpr (V == 14) {  // 2 x v_fma_f32 accumulate form (dst == src2), VOP3 encoding
        asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %2, %3, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x), "v"(a.x));
      } else if constexpr (V == 15) {  // 2 x v_pk_fma_f32 accumulate form
        asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(m), "v"(a));
      } else if constexpr (V == 16) {  // 2 x v_add_f32
        asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(a.x));
      } else if constexpr (V == 17) {  // v_mul_f32 + v_fmac_f32: one real half of a complex multiply-add, scalar form
        asm volatile("v_mul_f32 %0, %2, %0\n v_fmac_f32 %0, %3, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(m.x), "v"(a.x));
      } else if constexpr (V == 18) {  // v_pk_mul_f32 + v_pk_fma_f32: the packed form of the same (two halves at once)
        asm volatile("v_pk_mul_f32 %0, %2, %0\n v_pk_fma_f32 %0, %3, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                     : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(m), "v"(a));
      } else if constexpr (V == 19) {  // s_nop 0 between packed instructions
        asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n s_nop 0\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(acc[i]), "+v"(acc[(i + 8) & 15]) : "v"(m), "v"(a));
      }
    }
  }
  f2 s = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <int V>
void run(const char* name, double flops_per_pair, float* out) {
  const int iters = 2048;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-44s", name);
  for (int wps = 1; wps <= 8; wps *= 2) {   // 256-thread blocks = 1 wave per SIMD each; wps blocks per CU
    const int grid = 256 * wps;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.9999f, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double instr_per_simd = (double)wps * iters * 32;       // wave-instructions issued per SIMD
    printf("  w%d: %6.3f inst/ns/SIMD %6.1f TF", wps, instr_per_simd / (best * 1e6),
           flops_per_pair * 16 * iters * (double)grid * 256 / (best * 1e-3) / 1e12);
  }
  printf("\n");
}

__global__ void k_clock(long long* o) {
  const long long c0 = clock64(), w0 = wall_clock64();
  float x = threadIdx.x;
  for (int i = 0; i < 200000; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  const long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { o[0] = c1 - c0; o[1] = w1 - w0; o[2] = (long long)x; }
}
int main() {
  {
    long long* o; hipMalloc(&o, 64); long long h[3];
    hipLaunchKernelGGL(k_clock, dim3(2048), dim3(256), 0, 0, o);
    hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("clock64 ticks %lld, wall_clock64 (100 MHz) ticks %lld -> clock64 runs at %.1f MHz (all CUs busy)\n", h[0], h[1], 100.0 * h[0] / h[1]);
  }
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run<0>("v_fma_f32 (VGPR)", 4, out);
  run<1>("v_fma_f32 (SGPR multiplier)", 4, out);
  run<2>("v_pk_fma_f32 (VGPR)", 8, out);
  run<3>("v_pk_fma_f32 (SGPR pair, op_sel_hi bcast)", 8, out);
  run<4>("v_pk_fma_f32 (op_sel swap + neg_lo)", 8, out);
  run<5>("v_pk_mul_f32", 4, out);
  run<6>("v_mul_f32", 2, out);
  run<7>("v_pk_add_f32", 4, out);
  run<8>("v_pk_fma_f32 dependent chain", 8.0 / 16, out);
  run<9>("v_fma_f32 dependent chain", 4.0 / 16, out);
  run<10>("v_add_f32_dpp + v_fma_f32", 3, out);
  run<11>("v_mov_b32", 0, out);
  run<12>("v_fmac_f32 (VOP2, VGPR)", 4, out);
  run<13>("v_fmac_f32 (VOP2, SGPR multiplier)", 4, out);
  run<14>("v_fma_f32 accumulate (VOP3, dst = src2)", 4, out);
  run<15>("v_pk_fma_f32 accumulate (dst = src2)", 8, out);
  run<16>("v_add_f32", 2, out);
  run<17>("v_mul_f32 + v_fmac_f32 (scalar c*a + s*b)", 3, out);
  run<18>("v_pk_mul_f32 + v_pk_fma_f32 (packed same)", 6, out);
  run<19>("v_pk_fma_f32, s_nop 0 between (3 instr)", 8, out);
  return 0;
}
