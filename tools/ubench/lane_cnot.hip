// Lane-by-lane check of the masked DPP forms of a CNOT whose control and target are lane bits of one 16-lane row
// (lane_cnot<CB, TB> of csrc/qc_circuit_wave_kernels.h): out[lane] = bit CB of lane set ? in[lane ^ (1 << TB)] : in[lane].
//   hipcc --offload-arch=gfx950 -O2 -o lane_cnot tools/ubench/lane_cnot.hip && ./lane_cnot
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <type_traits>
#include <utility>

namespace {
typedef float wf2 __attribute__((ext_vector_type(2)));
}
// the two helpers under test, textually the ones of the kernel header
__host__ __device__ constexpr bool lane_cnot_ok(int CB, int TB) {
  if (CB < 0 || TB < 0 || CB > 3 || TB > 3 || CB == TB) return false;
  if (CB <= 1 && TB <= 1) return true;
  return CB >= 2;
}
template <int CB, int TB>
__device__ __forceinline__ float lane_cnot(float v) {
  const int u = __builtin_bit_cast(int, v);
  if constexpr (CB <= 1 && TB <= 1) {
    constexpr int ctrl = CB == 1 ? (0 | (1 << 2) | (3 << 4) | (2 << 6)) : (0 | (3 << 2) | (2 << 4) | (1 << 6));
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, u, ctrl, 0xF, 0xF, false));
  } else {
    constexpr int bank = CB == 2 ? 0xA : 0xC;
    if constexpr (TB == 0) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, u, 0xB1, 0xF, bank, false));
    } else if constexpr (TB == 1) {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, u, 0x4E, 0xF, bank, false));
    } else if constexpr (TB == 2) {
      const int t = __builtin_amdgcn_update_dpp(0, u, 0x141, 0xF, 0xF, true);
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, t, 0x1B, 0xF, bank, false));
    } else {
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(u, u, 0x128, 0xF, bank, false));
    }
  }
}

template <int CB, int TB>
__global__ void k(float* out, const float* in) {
  if constexpr (lane_cnot_ok(CB, TB)) out[threadIdx.x] = lane_cnot<CB, TB>(in[threadIdx.x]);
}

template <int CB, int TB>
int one(float* d_in, float* d_out) {
  if constexpr (!lane_cnot_ok(CB, TB)) {
    return 0;
  } else {
    hipLaunchKernelGGL((k<CB, TB>), dim3(1), dim3(64), 0, 0, d_out, d_in);
    float o[64];
    (void)hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
      const int want = ((i >> CB) & 1) ? (i ^ (1 << TB)) : i;
      if (o[i] != (float)want) {
        if (++bad < 4) printf("  CB=%d TB=%d lane %d got %g want %d\n", CB, TB, i, o[i], want);
      }
    }
    printf("control bit %d, target bit %d: %s\n", CB, TB, bad ? "WRONG" : "ok");
    return bad;
  }
}

int main() {
  float *in, *out;
  (void)hipMalloc(&in, 256);
  (void)hipMalloc(&out, 256);
  float h[64];
  for (int i = 0; i < 64; ++i) h[i] = (float)i;
  (void)hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
  int bad = 0;
  bad += one<1, 0>(in, out) + one<0, 1>(in, out);
  bad += one<2, 0>(in, out) + one<2, 1>(in, out) + one<2, 3>(in, out);
  bad += one<3, 0>(in, out) + one<3, 1>(in, out) + one<3, 2>(in, out);
  printf("bad=%d\n", bad);
  return bad != 0;
}
