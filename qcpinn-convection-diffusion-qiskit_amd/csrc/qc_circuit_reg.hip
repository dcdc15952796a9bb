// Variational-circuit kernels, "reg" family: 2 <= n <= 5 qubits, one lane = one statevector
// held entirely in VGPRs (qc_gates.h).  Replaces the PennyLane default.qubit simulation behind
// DVQuantumLayer.forward (reference nn/DVQuantumLayer.py:151-154,176-214) and the torch
// double-backward through it that nn/pde.py:59-70 + loss.backward() trigger.
//
// Four kernels per qubit count:
//   value_fwd : angles[n][B]            -> <Z_w>[n][B]                       (1 channel)
//   value_bwd : angles, cot[n][B]       -> d_angles[n][B], d_theta partial rows
//   jets_fwd  : angle jets[6][n][B]     -> <Z_w> jets[6][n][B]               (6 channels)
//   jets_bwd  : angle jets, cot jets    -> d(angle jets)[6][n][B], d_theta partial rows
// Channels: 0 value, 1 d/dt, 2 d/dx, 3 d/dy, 4 d2/dx2, 5 d2/dy2 (forward-mode derivatives of the
// circuit output w.r.t. the collocation coordinates; the gates are linear so every channel runs
// the same gate program on its own initial vector).  In the jet kernels a block is 6 waves x 64
// points: wave c carries channel c of the block's 64 points, channels meet through LDS only
// where the bilinear <Z> forms and their cotangents couple them.
//
// Batch-minor ([feature][B]) layouts make every global access 64 consecutive floats per wave.
#include "qc_gates.h"
#include "qc_internal.h"

namespace {

template <int N>
__device__ __forceinline__ void run_program_fwd(SV<N> (&v)[1], const QcGate* __restrict__ prog,
                                                const QcTrig* __restrict__ trig,
                                                const float* __restrict__ umat, int n_gates) {
  for (int g = 0; g < n_gates; ++g) {
    const QcGate gt = prog[g];
    const QcTrig tr = trig[g];
    qc_apply_gate<N, 1, false>(v, gt, tr.c, tr.s, umat);
  }
}

// Reverse sweep over the program for one (chi, lam) pair: accumulates Im<lam|G|chi> per
// parameter slot into acc[slot] (LDS, one row per wave), then un-applies the gate on both.
template <int N>
__device__ __forceinline__ void run_program_bwd(SV<N> (&cl)[2], const QcGate* __restrict__ prog,
                                                const QcTrig* __restrict__ trig,
                                                const float* __restrict__ umat, int n_gates,
                                                float* __restrict__ acc_wave, int lane) {
  for (int g = n_gates - 1; g >= 0; --g) {
    const QcGate gt = prog[g];
    const QcTrig tr = trig[g];
    if (gt.op != QC_U4 && gt.slot >= 0) {
      const float gr = qc_wave_sum_to_lane63(qc_gate_grad<N>(cl[1], cl[0], gt));
      if (lane == 63) acc_wave[gt.slot] += gr;
    }
    qc_apply_gate<N, 2, true>(cl, gt, tr.c, tr.s, umat);
  }
}

template <int N>
__device__ __forceinline__ void load_sincos(float (&ca)[N], float (&sa)[N], const float* __restrict__ a,
                                            int64_t B, int64_t p) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const float h = 0.5f * a[(int64_t)w * B + p];
    sincosf(h, &sa[w], &ca[w]);
  }
}

// ================================================================== value channel only
template <int N>
__global__ void __launch_bounds__(256) k_value_fwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, int n_gates,
                                                   const float* __restrict__ angles, float* __restrict__ expval,
                                                   int64_t B) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t pc = p < B ? p : B - 1;
  float ca[N], sa[N], zero[N];
#pragma unroll
  for (int w = 0; w < N; ++w) zero[w] = 0.f;
  load_sincos<N>(ca, sa, angles, B, pc);
  float P0[1 << N], P1[1 << N], P2[1 << N];
  qc_embed_series<N, 0>(P0, P1, P2, ca, sa, zero, zero);
  SV<N> v[1];
  qc_phase_load<N>(v[0], P0);
  run_program_fwd<N>(v, prog, trig, umat, n_gates);
  float t[1 << N], q[N];
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) t[k] = v[0].re[k] * v[0].re[k] + v[0].im[k] * v[0].im[k];
  qc_signed_sums<N>(q, t);
  if (p < B) {
#pragma unroll
    for (int w = 0; w < N; ++w) expval[(int64_t)w * B + p] = q[w];
  }
}

template <int N>
__global__ void __launch_bounds__(256) k_value_bwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, int n_gates, int n_params,
                                                   const float* __restrict__ angles, const float* __restrict__ cot,
                                                   float* __restrict__ d_angles, float* __restrict__ part,
                                                   int64_t part_stride, int64_t row0, int64_t B) {
  extern __shared__ float smem[];  // [4 waves][n_params]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4 * n_params; i += 256) smem[i] = 0.f;
  __syncthreads();

  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  float ca[N], sa[N], zero[N];
#pragma unroll
  for (int w = 0; w < N; ++w) zero[w] = 0.f;
  load_sincos<N>(ca, sa, angles, B, pc);
  float P0[1 << N], P1[1 << N], P2[1 << N];
  qc_embed_series<N, 0>(P0, P1, P2, ca, sa, zero, zero);
  SV<N> cl[2];  // [0] = chi, [1] = lambda
  {
    SV<N> v[1];
    qc_phase_load<N>(v[0], P0);
    run_program_fwd<N>(v, prog, trig, umat, n_gates);
    cl[0] = v[0];
  }
  float qb[N];
#pragma unroll
  for (int w = 0; w < N; ++w) qb[w] = live ? cot[(int64_t)w * B + pc] : 0.f;
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) {
    float d = 0.f;
#pragma unroll
    for (int w = 0; w < N; ++w) d += ((k >> (N - 1 - w)) & 1) ? -qb[w] : qb[w];
    cl[1].re[k] = d * cl[0].re[k];
    cl[1].im[k] = d * cl[0].im[k];
  }
  run_program_bwd<N>(cl, prog, trig, umat, n_gates, smem + wave * n_params, lane);
  float T[N];
  qc_embed_ip<N>(T, cl[1], P0);
  if (live) {
#pragma unroll
    for (int w = 0; w < N; ++w) d_angles[(int64_t)w * B + p] = T[w];
  }
  __syncthreads();
  // one partial row per wave = per 64-point tile (same tiling as the MLP kernels)
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile * 64 < B)
    for (int i = lane; i < n_params; i += 64) part[(row0 + tile) * part_stride + i] = smem[wave * n_params + i];
}

// ================================================================== six derivative channels
// Builds the initial vector of channel `ch` for this lane's point.  P0/P1/P2 are left holding
// the embedding series of the channel's direction (needed again by the backward kernel).
template <int N>
__device__ __forceinline__ void build_channel(SV<N>& v, float (&P0)[1 << N], float (&P1)[1 << N],
                                              float (&P2)[1 << N], int ch, const float* __restrict__ ajets,
                                              int64_t B, int64_t pc) {
  float ca[N], sa[N], da[N], dda[N];
  load_sincos<N>(ca, sa, ajets, B, pc);
  const int dirch = ch == 0 ? 0 : (ch <= 3 ? ch : ch - 2);  // channel holding the first derivative
#pragma unroll
  for (int w = 0; w < N; ++w) {
    da[w] = ch >= 1 ? ajets[((int64_t)dirch * N + w) * B + pc] : 0.f;
    dda[w] = ch >= 4 ? ajets[((int64_t)ch * N + w) * B + pc] : 0.f;
  }
  if (ch == 0) {
    qc_embed_series<N, 0>(P0, P1, P2, ca, sa, da, dda);
    qc_phase_load<N>(v, P0);
  } else if (ch <= 3) {
    qc_embed_series<N, 1>(P0, P1, P2, ca, sa, da, dda);
    qc_phase_load<N>(v, P1);
  } else {
    qc_embed_series<N, 2>(P0, P1, P2, ca, sa, da, dda);
    qc_phase_load<N>(v, P2);
  }
}

template <int N>
__global__ void __launch_bounds__(384) k_jets_fwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates,
                                                  const float* __restrict__ ajets, float* __restrict__ qjets,
                                                  int64_t B) {
  constexpr int A2 = 2 << N;                 // floats per statevector
  __shared__ float s_chi0[A2 * 64];          // [amp*2+{re,im}][lane]
  __shared__ float s_sq[2 * N * 64];         // 2<chi_k|Z_w|chi_k> for k = x, y
  const int lane = threadIdx.x & 63;
  const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index = channel (scalar)
  const int64_t p = (int64_t)blockIdx.x * 64 + lane;
  const int64_t pc = p < B ? p : B - 1;

  SV<N> v[1];
  float P0[1 << N], P1[1 << N], P2[1 << N];
  build_channel<N>(v[0], P0, P1, P2, ch, ajets, B, pc);
  run_program_fwd<N>(v, prog, trig, umat, n_gates);

  float t[1 << N], q[N];
  if (ch == 0) {
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      s_chi0[(2 * k) * 64 + lane] = v[0].re[k];
      s_chi0[(2 * k + 1) * 64 + lane] = v[0].im[k];
      t[k] = v[0].re[k] * v[0].re[k] + v[0].im[k] * v[0].im[k];
    }
    qc_signed_sums<N>(q, t);
  } else if (ch == 2 || ch == 3) {
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) t[k] = 2.f * (v[0].re[k] * v[0].re[k] + v[0].im[k] * v[0].im[k]);
    qc_signed_sums<N>(q, t);
#pragma unroll
    for (int w = 0; w < N; ++w) s_sq[((ch - 2) * N + w) * 64 + lane] = q[w];
  }
  __syncthreads();
  if (ch != 0) {
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      const float r0 = s_chi0[(2 * k) * 64 + lane], i0 = s_chi0[(2 * k + 1) * 64 + lane];
      t[k] = 2.f * (r0 * v[0].re[k] + i0 * v[0].im[k]);
    }
    qc_signed_sums<N>(q, t);
    if (ch >= 4) {
#pragma unroll
      for (int w = 0; w < N; ++w) q[w] += s_sq[((ch - 4) * N + w) * 64 + lane];
    }
  }
  if (p < B) {
#pragma unroll
    for (int w = 0; w < N; ++w) qjets[((int64_t)ch * N + w) * B + p] = q[w];
  }
}

template <int N>
__global__ void __launch_bounds__(384) k_jets_bwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates, int n_params,
                                                  const float* __restrict__ ajets, const float* __restrict__ qbar,
                                                  float* __restrict__ abar, float* __restrict__ part,
                                                  int64_t part_stride, int64_t row0, int64_t B) {
  constexpr int A2 = 2 << N;
  extern __shared__ float smem[];
  float* s_chi = smem;                       // [6][A2][64]; later reused as [6 waves][3][N][64]
  float* s_acc = smem + 6 * A2 * 64;         // [6 waves][n_params]
  const int lane = threadIdx.x & 63;
  const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 6 * n_params; i += 384) s_acc[i] = 0.f;

  const int64_t p = (int64_t)blockIdx.x * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;

  SV<N> cl[2];
  float P0[1 << N], P1[1 << N], P2[1 << N];
  {
    SV<N> v[1];
    build_channel<N>(v[0], P0, P1, P2, ch, ajets, B, pc);
    run_program_fwd<N>(v, prog, trig, umat, n_gates);
    cl[0] = v[0];
  }
  float* mine = s_chi + ch * A2 * 64;
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) {
    mine[(2 * k) * 64 + lane] = cl[0].re[k];
    mine[(2 * k + 1) * 64 + lane] = cl[0].im[k];
  }
  __syncthreads();

  // ---- cotangent of this channel's final state (bilinear <Z> forms)
  auto dvec = [&](int c, float (&d)[1 << N]) {
    float qb[N];
#pragma unroll
    for (int w = 0; w < N; ++w) qb[w] = live ? qbar[((int64_t)c * N + w) * B + pc] : 0.f;
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < N; ++w) s += ((k >> (N - 1 - w)) & 1) ? -qb[w] : qb[w];
      d[k] = s;
    }
  };
  float d[1 << N];
  if (ch == 0) {
    dvec(0, d);
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      cl[1].re[k] = d[k] * cl[0].re[k];
      cl[1].im[k] = d[k] * cl[0].im[k];
    }
    for (int c = 1; c < QC_NCH; ++c) {
      dvec(c, d);
      const float* other = s_chi + c * A2 * 64;
#pragma unroll
      for (int k = 0; k < (1 << N); ++k) {
        cl[1].re[k] = fmaf(d[k], other[(2 * k) * 64 + lane], cl[1].re[k]);
        cl[1].im[k] = fmaf(d[k], other[(2 * k + 1) * 64 + lane], cl[1].im[k]);
      }
    }
  } else {
    dvec(ch, d);
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      cl[1].re[k] = d[k] * s_chi[(2 * k) * 64 + lane];
      cl[1].im[k] = d[k] * s_chi[(2 * k + 1) * 64 + lane];
    }
    if (ch == 2 || ch == 3) {
      dvec(ch + 2, d);
#pragma unroll
      for (int k = 0; k < (1 << N); ++k) {
        cl[1].re[k] = fmaf(2.f * d[k], cl[0].re[k], cl[1].re[k]);
        cl[1].im[k] = fmaf(2.f * d[k], cl[0].im[k], cl[1].im[k]);
      }
    }
  }
  __syncthreads();  // everyone is done reading s_chi

  run_program_bwd<N>(cl, prog, trig, umat, n_gates, s_acc + ch * n_params, lane);

  // ---- cotangents of the angle jets: Im<Lambda| X_w |phi> against the embedding series
  float* buf = s_chi + ch * 3 * N * 64;  // [3][N][64] per wave
  float T[N];
  if (ch == 0) {
    qc_embed_ip<N>(T, cl[1], P0);
#pragma unroll
    for (int w = 0; w < N; ++w) buf[(0 * N + w) * 64 + lane] = T[w];
  } else if (ch <= 3) {
    qc_embed_ip<N>(T, cl[1], P1);
#pragma unroll
    for (int w = 0; w < N; ++w) buf[(0 * N + w) * 64 + lane] = T[w];
    qc_embed_ip<N>(T, cl[1], P0);
#pragma unroll
    for (int w = 0; w < N; ++w) buf[(1 * N + w) * 64 + lane] = T[w];
  } else {
    qc_embed_ip<N>(T, cl[1], P2);
#pragma unroll
    for (int w = 0; w < N; ++w) buf[(0 * N + w) * 64 + lane] = T[w];
    qc_embed_ip<N>(T, cl[1], P1);
#pragma unroll
    for (int w = 0; w < N; ++w) buf[(1 * N + w) * 64 + lane] = 2.f * T[w];
    qc_embed_ip<N>(T, cl[1], P0);
#pragma unroll
    for (int w = 0; w < N; ++w) buf[(2 * N + w) * 64 + lane] = T[w];
  }
  __syncthreads();
  auto at = [&](int wv, int slot, int w) { return s_chi[((wv * 3 + slot) * N + w) * 64 + lane]; };
#pragma unroll
  for (int w = 0; w < N; ++w) {
    float r;
    if (ch == 0)
      r = ((at(0, 0, w) + at(1, 0, w)) + (at(2, 0, w) + at(3, 0, w))) + (at(4, 0, w) + at(5, 0, w));
    else if (ch == 1)
      r = at(1, 1, w);
    else if (ch <= 3)
      r = at(ch, 1, w) + at(ch + 2, 1, w);
    else
      r = at(ch, 2, w);
    if (live) abar[((int64_t)ch * N + w) * B + p] = r;
  }
  for (int i = threadIdx.x; i < n_params; i += 384) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 6; ++wv) s += s_acc[wv * n_params + i];
    part[(row0 + blockIdx.x) * part_stride + i] = s;
  }
}

}  // namespace

// ------------------------------------------------------------------ host-side launchers (called from qc_api.hip)
#define QC_DISPATCH_N(n, CALL)        \
  switch (n) {                        \
    case 2: { CALL(2) } break;        \
    case 3: { CALL(3) } break;        \
    case 4: { CALL(4) } break;        \
    case 5: { CALL(5) } break;        \
    default: return QC_ERR_UNSUPPORTED; \
  }

int qc_reg_value_fwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* angles,
                     float* expval, int64_t B, hipStream_t st) {
  const int grid = qc_ceil_div(B, 256);
#define CALL(NN) \
  hipLaunchKernelGGL(k_value_fwd<NN>, dim3(grid), dim3(256), 0, st, pg->d_gates, trig, umat, pg->n_gates, angles, expval, B);
  QC_DISPATCH_N(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}

int qc_reg_value_bwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* angles,
                     const float* cot, float* d_angles, float* part, int64_t part_stride, int64_t row0,
                     int64_t B, hipStream_t st) {
  const int grid = qc_ceil_div(B, 256);
  const size_t sh = (size_t)4 * pg->n_params * sizeof(float);
#define CALL(NN)                                                                                        \
  hipLaunchKernelGGL(k_value_bwd<NN>, dim3(grid), dim3(256), sh, st, pg->d_gates, trig, umat, pg->n_gates, \
                     pg->n_params, angles, cot, d_angles, part, part_stride, row0, B);
  QC_DISPATCH_N(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}

int qc_reg_jets_fwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                    float* qjets, int64_t B, hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
#define CALL(NN) \
  hipLaunchKernelGGL(k_jets_fwd<NN>, dim3(grid), dim3(384), 0, st, pg->d_gates, trig, umat, pg->n_gates, ajets, qjets, B);
  QC_DISPATCH_N(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}

size_t qc_reg_jets_bwd_lds(int n, int n_params) {
  return ((size_t)6 * (2u << n) * 64 + (size_t)6 * n_params) * sizeof(float);
}

int qc_reg_jets_bwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                    const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0,
                    int64_t B, hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
  const size_t sh = qc_reg_jets_bwd_lds(pg->n_qubits, pg->n_params);
  if (sh > 160 * 1024) return QC_ERR_UNSUPPORTED;
#define CALL(NN)                                                                                            \
  {                                                                                                         \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      hipFuncSetAttribute(reinterpret_cast<const void*>(&k_jets_bwd<NN>),                                   \
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                          \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL(k_jets_bwd<NN>, dim3(grid), dim3(384), sh, st, pg->d_gates, trig, umat, pg->n_gates, \
                       pg->n_params, ajets, qbar, abar, part, part_stride, row0, B);                              \
  }
  QC_DISPATCH_N(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}
