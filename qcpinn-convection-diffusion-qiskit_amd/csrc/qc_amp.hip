// Amplitude encoding (reference nn/DVQuantumLayer.py:177-180: AmplitudeEmbedding(x, normalize=True,
// pad_with=0.0)): the n features of a point, zero-padded to 2^n and L2-normalised, ARE the initial
// statevector (amplitude k = feature k, real).  Its derivative channels along the collocation
// coordinates are the jets of u(a) = a / |a|:
//   u    = a / r
//   u_k  = a_k / r - a p_k / r^3                                   p_k  = a . a_k
//   u_kk = a_kk / r - 2 a_k p_k / r^3 - a (q_kk / r^3 - 3 p_k^2 / r^5)   q_kk = a_k . a_k + a . a_kk
// k_amp_fwd maps the angle-jet tensor [NCH][n][B] produced by the pre-network to these "initial
// amplitude jets" (same layout); the circuit kernels then start from them instead of the RX product
// state, and return the cotangents w.r.t. them, which k_amp_bwd pulls back to the pre-network output.
#include "qc_internal.h"

namespace {

template <int NCH>
__global__ void __launch_bounds__(256) k_amp_fwd(const float* __restrict__ a, float* __restrict__ u, int n, int64_t B) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= B) return;
  auto A = [&](int c, int j) { return a[((int64_t)c * n + j) * B + p]; };
  float r2 = 0.f, pk[3] = {0.f, 0.f, 0.f}, qk[2] = {0.f, 0.f};
  for (int j = 0; j < n; ++j) {
    const float a0 = A(0, j);
    r2 += a0 * a0;
    if constexpr (NCH == 6) {
      const float at = A(1, j), ax = A(2, j), ay = A(3, j);
      pk[0] += a0 * at;
      pk[1] += a0 * ax;
      pk[2] += a0 * ay;
      qk[0] += ax * ax + a0 * A(4, j);
      qk[1] += ay * ay + a0 * A(5, j);
    }
  }
  const float i1 = rsqrtf(r2), i3 = i1 * i1 * i1, i5 = i3 * i1 * i1;
  for (int j = 0; j < n; ++j) {
    const float a0 = A(0, j);
    u[(int64_t)j * B + p] = a0 * i1;
    if constexpr (NCH == 6) {
#pragma unroll
      for (int k = 0; k < 3; ++k) u[((int64_t)(1 + k) * n + j) * B + p] = A(1 + k, j) * i1 - a0 * pk[k] * i3;
#pragma unroll
      for (int k = 0; k < 2; ++k)
        u[((int64_t)(4 + k) * n + j) * B + p] = A(4 + k, j) * i1 - 2.f * A(2 + k, j) * pk[1 + k] * i3 -
                                                a0 * (qk[k] * i3 - 3.f * pk[1 + k] * pk[1 + k] * i5);
    }
  }
}

// reverse of k_amp_fwd: ub = cotangents of the initial-amplitude jets -> ab = cotangents of the angle jets
template <int NCH>
__global__ void __launch_bounds__(256) k_amp_bwd(const float* __restrict__ a, const float* __restrict__ ub,
                                                 float* __restrict__ ab, int n, int64_t B) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= B) return;
  auto A = [&](int c, int j) { return a[((int64_t)c * n + j) * B + p]; };
  auto U = [&](int c, int j) { return ub[((int64_t)c * n + j) * B + p]; };
  float r2 = 0.f, pk[3] = {0.f, 0.f, 0.f}, qk[2] = {0.f, 0.f};
  float d0a = 0.f;                    // U0 . a
  float dka[3] = {0.f, 0.f, 0.f};     // U_k . a
  float dkk[3] = {0.f, 0.f, 0.f};     // U_k . a_k
  float ekk[2] = {0.f, 0.f};          // U_kk . a_kk
  float ek[2] = {0.f, 0.f};           // U_kk . a_k
  float ea[2] = {0.f, 0.f};           // U_kk . a
  for (int j = 0; j < n; ++j) {
    const float a0 = A(0, j);
    r2 += a0 * a0;
    d0a += U(0, j) * a0;
    if constexpr (NCH == 6) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float ak = A(1 + k, j), uk = U(1 + k, j);
        pk[k] += a0 * ak;
        dka[k] += uk * a0;
        dkk[k] += uk * ak;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float ak = A(2 + k, j), akk = A(4 + k, j), ukk = U(4 + k, j);
        qk[k] += ak * ak + a0 * akk;
        ekk[k] += ukk * akk;
        ek[k] += ukk * ak;
        ea[k] += ukk * a0;
      }
    }
  }
  const float i1 = rsqrtf(r2), i2 = i1 * i1, i3 = i1 * i2, i5 = i3 * i2, i7 = i5 * i2;
  float s1 = d0a, s3 = 0.f, s5 = 0.f;   // cotangents of 1/r, 1/r^3, 1/r^5
  float pb[3] = {0.f, 0.f, 0.f}, qb[2] = {0.f, 0.f};
  if constexpr (NCH == 6) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      s1 += dkk[k];
      pb[k] += -dka[k] * i3;
      s3 += -dka[k] * pk[k];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float p1 = pk[1 + k];
      s1 += ekk[k];
      pb[1 + k] += -2.f * ek[k] * i3 + 6.f * ea[k] * p1 * i5;
      s3 += -2.f * ek[k] * p1 - ea[k] * qk[k];
      s5 += 3.f * ea[k] * p1 * p1;
      qb[k] = -ea[k] * i3;
    }
  }
  const float radial = -i3 * s1 - 3.f * i5 * s3 - 5.f * i7 * s5;   // d(1/r^m)/da = -m a / r^(m+2)
  for (int j = 0; j < n; ++j) {
    const float a0 = A(0, j);
    float g0 = U(0, j) * i1 + a0 * radial;
    if constexpr (NCH == 6) {
      float gk[3], gkk[2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float ak = A(1 + k, j), uk = U(1 + k, j);
        gk[k] = uk * i1 + pb[k] * a0;
        g0 += -uk * pk[k] * i3 + pb[k] * ak;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float ak = A(2 + k, j), akk = A(4 + k, j), ukk = U(4 + k, j);
        const float p1 = pk[1 + k];
        gkk[k] = ukk * i1 + qb[k] * a0;
        gk[1 + k] += -2.f * ukk * p1 * i3 + 2.f * qb[k] * ak;
        g0 += -ukk * (qk[k] * i3 - 3.f * p1 * p1 * i5) + qb[k] * akk;
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) ab[((int64_t)(1 + k) * n + j) * B + p] = gk[k];
#pragma unroll
      for (int k = 0; k < 2; ++k) ab[((int64_t)(4 + k) * n + j) * B + p] = gkk[k];
    }
    ab[(int64_t)j * B + p] = g0;
  }
}

}  // namespace

int qc_amp_fwd_launch(const float* a, float* u, int n, int64_t B, int nch, hipStream_t st) {
  if (nch == 6) hipLaunchKernelGGL((k_amp_fwd<6>), dim3(qc_ceil_div(B, 256)), dim3(256), 0, st, a, u, n, B);
  else hipLaunchKernelGGL((k_amp_fwd<1>), dim3(qc_ceil_div(B, 256)), dim3(256), 0, st, a, u, n, B);
  return QC_OK;
}

int qc_amp_bwd_launch(const float* a, const float* ub, float* ab, int n, int64_t B, int nch, hipStream_t st) {
  if (nch == 6) hipLaunchKernelGGL((k_amp_bwd<6>), dim3(qc_ceil_div(B, 256)), dim3(256), 0, st, a, ub, ab, n, B);
  else hipLaunchKernelGGL((k_amp_bwd<1>), dim3(qc_ceil_div(B, 256)), dim3(256), 0, st, a, ub, ab, n, B);
  return QC_OK;
}
