// Register-resident statevector gate library ("reg" family, n <= 5 qubits).
//
// One lane owns one complete statevector: 2^N complex amplitudes in 2*2^N VGPRs.  Every gate
// is an index-paired amplitude update with compile-time register indices; the gate PROGRAM is
// run-time data (wave-uniform), so the interpreter dispatches through a scalar `switch` on
// (opcode, bit) into fully unrolled bodies.  No LDS, no cross-lane traffic, no MFMA.
//
// Conventions (reference nn/DVQuantumLayer.py:176-214 on PennyLane default.qubit):
//   amplitude index bit (N-1-w) <-> wire w;  RX(t)=exp(-itX/2), RY(t)=exp(-itY/2),
//   RZ(t)=diag(e^{-it/2},e^{+it/2});  controlled gates act where the control bit is 1;
//   a two-wire unitary on bits (hb, lb) uses row/col index 2*bit_hb + bit_lb.
#pragma once
#include "qc_common.h"

template <int N>
struct SV {
  float re[1 << N];
  float im[1 << N];
};

// insert a zero bit at position B into k
__host__ __device__ constexpr int qc_ins0(int k, int B) { return ((k >> B) << (B + 1)) | (k & ((1 << B) - 1)); }
__host__ __device__ constexpr int qc_popc(int k) { return k == 0 ? 0 : (k & 1) + qc_popc(k >> 1); }

// ------------------------------------------------------------------ single-bit gates
// `s` carries the sign: the adjoint of a rotation is the same body with s -> -s.
template <int N, int B>
__device__ __forceinline__ void g_rx(SV<N>& v, float c, float s) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const float ar = v.re[i0], ai = v.im[i0], br = v.re[i1], bi = v.im[i1];
    v.re[i0] = fmaf(s, bi, c * ar);
    v.im[i0] = fmaf(-s, br, c * ai);
    v.re[i1] = fmaf(s, ai, c * br);
    v.im[i1] = fmaf(-s, ar, c * bi);
  }
}
template <int N, int B>
__device__ __forceinline__ void g_ry(SV<N>& v, float c, float s) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const float ar = v.re[i0], ai = v.im[i0], br = v.re[i1], bi = v.im[i1];
    v.re[i0] = fmaf(-s, br, c * ar);
    v.im[i0] = fmaf(-s, bi, c * ai);
    v.re[i1] = fmaf(s, ar, c * br);
    v.im[i1] = fmaf(s, ai, c * bi);
  }
}
template <int N, int B>
__device__ __forceinline__ void g_rz(SV<N>& v, float c, float s) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const float ar = v.re[i0], ai = v.im[i0], br = v.re[i1], bi = v.im[i1];
    v.re[i0] = fmaf(s, ai, c * ar);   // * (c - i s)
    v.im[i0] = fmaf(-s, ar, c * ai);
    v.re[i1] = fmaf(-s, bi, c * br);  // * (c + i s)
    v.im[i1] = fmaf(s, br, c * bi);
  }
}
template <int N, int B>
__device__ __forceinline__ void g_h(SV<N>& v) {
  const float r = 0.70710678118654752440f;
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const float ar = v.re[i0], ai = v.im[i0], br = v.re[i1], bi = v.im[i1];
    v.re[i0] = (ar + br) * r;
    v.im[i0] = (ai + bi) * r;
    v.re[i1] = (ar - br) * r;
    v.im[i1] = (ai - bi) * r;
  }
}

// ------------------------------------------------------------------ controlled gates (control bit CB = 1)
template <int CB, int TB>
__host__ __device__ constexpr int qc_ctl_base(int k) {
  return (CB < TB) ? qc_ins0(qc_ins0(k, CB), TB) : qc_ins0(qc_ins0(k, TB), CB);
}
template <int N, int CB, int TB>
__device__ __forceinline__ void g_cnot(SV<N>& v) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    const float ar = v.re[i0], ai = v.im[i0];
    v.re[i0] = v.re[i1];
    v.im[i0] = v.im[i1];
    v.re[i1] = ar;
    v.im[i1] = ai;
  }
}
template <int N, int CB, int TB>
__device__ __forceinline__ void g_crx(SV<N>& v, float c, float s) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    const float ar = v.re[i0], ai = v.im[i0], br = v.re[i1], bi = v.im[i1];
    v.re[i0] = fmaf(s, bi, c * ar);
    v.im[i0] = fmaf(-s, br, c * ai);
    v.re[i1] = fmaf(s, ai, c * br);
    v.im[i1] = fmaf(-s, ar, c * bi);
  }
}
template <int N, int CB, int TB>
__device__ __forceinline__ void g_crz(SV<N>& v, float c, float s) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    const float ar = v.re[i0], ai = v.im[i0], br = v.re[i1], bi = v.im[i1];
    v.re[i0] = fmaf(s, ai, c * ar);
    v.im[i0] = fmaf(-s, ar, c * ai);
    v.re[i1] = fmaf(-s, bi, c * br);
    v.im[i1] = fmaf(s, br, c * bi);
  }
}

// ------------------------------------------------------------------ fixed two-wire unitary
// `u` points at 32 floats: row-major 4x4, (re, im) interleaved; wave-uniform address.
template <int N, int HB, int LB>
__device__ __forceinline__ void g_u4(SV<N>& v, const float* __restrict__ u) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int b = qc_ctl_base<HB, LB>(k);
    const int idx[4] = {b, b | (1 << LB), b | (1 << HB), b | (1 << HB) | (1 << LB)};
    float xr[4], xi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xr[j] = v.re[idx[j]];
      xi[j] = v.im[idx[j]];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float yr = 0.f, yi = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ur = u[(r * 4 + j) * 2], ui = u[(r * 4 + j) * 2 + 1];
        yr = fmaf(ur, xr[j], yr);
        yr = fmaf(-ui, xi[j], yr);
        yi = fmaf(ur, xi[j], yi);
        yi = fmaf(ui, xr[j], yi);
      }
      v.re[idx[r]] = yr;
      v.im[idx[r]] = yi;
    }
  }
}

// ------------------------------------------------------------------ generator inner products
// Im <lam| G |chi> for the gate generators: what the adjoint sweep accumulates per parameter.
template <int N, int B>
__device__ __forceinline__ float ip_x(const SV<N>& l, const SV<N>& x) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) {
    const int j = k ^ (1 << B);
    acc = fmaf(l.re[k], x.im[j], acc);
    acc = fmaf(-l.im[k], x.re[j], acc);
  }
  return acc;
}
template <int N, int B>
__device__ __forceinline__ float ip_z(const SV<N>& l, const SV<N>& x) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) {
    const float t = l.re[k] * x.im[k] - l.im[k] * x.re[k];
    acc += ((k >> B) & 1) ? -t : t;
  }
  return acc;
}
template <int N, int B>
__device__ __forceinline__ float ip_y(const SV<N>& l, const SV<N>& x) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    acc -= l.re[i0] * x.re[i1] + l.im[i0] * x.im[i1];
    acc += l.re[i1] * x.re[i0] + l.im[i1] * x.im[i0];
  }
  return acc;
}
template <int N, int CB, int TB>
__device__ __forceinline__ float ip_cx(const SV<N>& l, const SV<N>& x) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    acc = fmaf(l.re[i0], x.im[i1], acc);
    acc = fmaf(-l.im[i0], x.re[i1], acc);
    acc = fmaf(l.re[i1], x.im[i0], acc);
    acc = fmaf(-l.im[i1], x.re[i0], acc);
  }
  return acc;
}
template <int N, int CB, int TB>
__device__ __forceinline__ float ip_cz(const SV<N>& l, const SV<N>& x) {
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    acc += l.re[i0] * x.im[i0] - l.im[i0] * x.re[i0];
    acc -= l.re[i1] * x.im[i1] - l.im[i1] * x.re[i1];
  }
  return acc;
}

// ------------------------------------------------------------------ run-time dispatch
// K statevectors go through the same gate inside one switch arm (amortises the scalar branch).
// ADJ = apply the adjoint (reverse sweep).  `umat` = base of the U4 table:
// [slot][fwd|adj][32 floats].

#define QC_CASE1(BIT, BODY) \
  case BIT:                 \
    if constexpr (BIT < N) { BODY(BIT) } \
    break;

template <int N, int K, bool ADJ>
__device__ __forceinline__ void qc_apply_gate(SV<N> (&v)[K], const QcGate g, const float c, const float s_in,
                                              const float* __restrict__ umat) {
  const float s = ADJ ? -s_in : s_in;
  switch (g.op) {
    case QC_RX:
#define B_(BIT) _Pragma("unroll") for (int q = 0; q < K; ++q) g_rx<N, BIT>(v[q], c, s);
      switch (g.ba) { QC_CASE1(0, B_) QC_CASE1(1, B_) QC_CASE1(2, B_) QC_CASE1(3, B_) QC_CASE1(4, B_) }
#undef B_
      break;
    case QC_RY:
#define B_(BIT) _Pragma("unroll") for (int q = 0; q < K; ++q) g_ry<N, BIT>(v[q], c, s);
      switch (g.ba) { QC_CASE1(0, B_) QC_CASE1(1, B_) QC_CASE1(2, B_) QC_CASE1(3, B_) QC_CASE1(4, B_) }
#undef B_
      break;
    case QC_RZ:
#define B_(BIT) _Pragma("unroll") for (int q = 0; q < K; ++q) g_rz<N, BIT>(v[q], c, s);
      switch (g.ba) { QC_CASE1(0, B_) QC_CASE1(1, B_) QC_CASE1(2, B_) QC_CASE1(3, B_) QC_CASE1(4, B_) }
#undef B_
      break;
    case QC_H:
#define B_(BIT) _Pragma("unroll") for (int q = 0; q < K; ++q) g_h<N, BIT>(v[q]);
      switch (g.ba) { QC_CASE1(0, B_) QC_CASE1(1, B_) QC_CASE1(2, B_) QC_CASE1(3, B_) QC_CASE1(4, B_) }
#undef B_
      break;
    case QC_CNOT:
    case QC_CRX:
    case QC_CRZ: {
      if constexpr (N >= 2) {
        const int code = g.ba * 8 + g.bb;
#define C2_(CB, TB)                                                                \
  case (CB * 8 + TB):                                                              \
    if constexpr (CB < N && TB < N && CB != TB) {                                  \
      _Pragma("unroll") for (int q = 0; q < K; ++q) {                              \
        if (g.op == QC_CNOT) g_cnot<N, CB, TB>(v[q]);                              \
        else if (g.op == QC_CRX) g_crx<N, CB, TB>(v[q], c, s);                     \
        else g_crz<N, CB, TB>(v[q], c, s);                                         \
      }                                                                            \
    }                                                                              \
    break;
#define C2ROW_(CB) C2_(CB, 0) C2_(CB, 1) C2_(CB, 2) C2_(CB, 3) C2_(CB, 4)
        switch (code) { C2ROW_(0) C2ROW_(1) C2ROW_(2) C2ROW_(3) C2ROW_(4) }
#undef C2ROW_
#undef C2_
      }
      break;
    }
    case QC_U4: {
      if constexpr (N >= 4) {
        const float* u = umat + (g.slot * 2 + (ADJ ? 1 : 0)) * 32;
        if (g.slot == 0) {
#pragma unroll
          for (int q = 0; q < K; ++q) g_u4<N, N - 1, N - 2>(v[q], u);
        } else {
#pragma unroll
          for (int q = 0; q < K; ++q) g_u4<N, N - 3, N - 4>(v[q], u);
        }
      }
      break;
    }
    default:
      break;
  }
}

// Im<lam|G|chi> of gate g's generator (0 for non-parametric gates), both vectors taken at the
// OUTPUT side of the gate.
template <int N>
__device__ __forceinline__ float qc_gate_grad(const SV<N>& lam, const SV<N>& chi, const QcGate g) {
  float r = 0.f;
  switch (g.op) {
    case QC_RX:
#define B_(BIT) r = ip_x<N, BIT>(lam, chi);
      switch (g.ba) { QC_CASE1(0, B_) QC_CASE1(1, B_) QC_CASE1(2, B_) QC_CASE1(3, B_) QC_CASE1(4, B_) }
#undef B_
      break;
    case QC_RY:
#define B_(BIT) r = ip_y<N, BIT>(lam, chi);
      switch (g.ba) { QC_CASE1(0, B_) QC_CASE1(1, B_) QC_CASE1(2, B_) QC_CASE1(3, B_) QC_CASE1(4, B_) }
#undef B_
      break;
    case QC_RZ:
#define B_(BIT) r = ip_z<N, BIT>(lam, chi);
      switch (g.ba) { QC_CASE1(0, B_) QC_CASE1(1, B_) QC_CASE1(2, B_) QC_CASE1(3, B_) QC_CASE1(4, B_) }
#undef B_
      break;
    case QC_CRX:
    case QC_CRZ: {
      if constexpr (N >= 2) {
        const int code = g.ba * 8 + g.bb;
#define C2_(CB, TB)                                                                              \
  case (CB * 8 + TB):                                                                            \
    if constexpr (CB < N && TB < N && CB != TB) {                                                \
      r = (g.op == QC_CRX) ? ip_cx<N, CB, TB>(lam, chi) : ip_cz<N, CB, TB>(lam, chi);           \
    }                                                                                            \
    break;
#define C2ROW_(CB) C2_(CB, 0) C2_(CB, 1) C2_(CB, 2) C2_(CB, 3) C2_(CB, 4)
        switch (code) { C2ROW_(0) C2ROW_(1) C2ROW_(2) C2ROW_(3) C2ROW_(4) }
#undef C2ROW_
#undef C2_
      }
      break;
    }
    default:
      break;
  }
  return r;
}

// ------------------------------------------------------------------ embedded product state and its jets
// The embedding RX(a_w) on every wire applied to |0..0> is a product state:
//   phi[k] = (-i)^{popcount(k)} * prod_w (bit_w(k) ? sin(a_w/2) : cos(a_w/2)).
// Along one input direction with first/second derivative of the angles (da, dda) each wire's
// 2-vector is a truncated Taylor series  W0 + e W1 + e^2/2 W2  with real magnitudes
//   W0=[c,s], W1=(da/2)[-s,c], W2=(dda/2)[-s,c]-(da^2/4)[c,s];
// the product series (P0,P1,P2) gives phi, d phi, d2 phi.  ORDER = highest series needed.
template <int N, int ORDER>
__device__ __forceinline__ void qc_embed_series(float (&P0)[1 << N], float (&P1)[1 << N], float (&P2)[1 << N],
                                                const float (&ca)[N], const float (&sa)[N],
                                                const float (&da)[N], const float (&dda)[N]) {
  P0[0] = 1.f;
  P1[0] = 0.f;
  P2[0] = 0.f;
#pragma unroll
  for (int w = 0; w < N; ++w) {  // wire w becomes the next-lower index bit: wire 0 ends up MSB
    const float c = ca[w], s = sa[w];
    const float w1_0 = -0.5f * da[w] * s, w1_1 = 0.5f * da[w] * c;
    const float q = 0.25f * da[w] * da[w];
    const float w2_0 = -0.5f * dda[w] * s - q * c, w2_1 = 0.5f * dda[w] * c - q * s;
#pragma unroll
    for (int i = (1 << w) - 1; i >= 0; --i) {
      const float p0 = P0[i], p1 = P1[i], p2 = P2[i];
      P0[2 * i] = p0 * c;
      P0[2 * i + 1] = p0 * s;
      if constexpr (ORDER >= 1) {
        P1[2 * i] = fmaf(p0, w1_0, p1 * c);
        P1[2 * i + 1] = fmaf(p0, w1_1, p1 * s);
      }
      if constexpr (ORDER >= 2) {
        P2[2 * i] = fmaf(p0, w2_0, fmaf(2.f * p1, w1_0, p2 * c));
        P2[2 * i + 1] = fmaf(p0, w2_1, fmaf(2.f * p1, w1_1, p2 * s));
      }
    }
  }
}

// real magnitudes -> complex amplitudes with the (-i)^{popcount} phase
template <int N>
__device__ __forceinline__ void qc_phase_load(SV<N>& v, const float (&P)[1 << N]) {
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) {
    const int ph = qc_popc(k) & 3;
    v.re[k] = (ph == 0) ? P[k] : (ph == 2 ? -P[k] : 0.f);
    v.im[k] = (ph == 1) ? -P[k] : (ph == 3 ? P[k] : 0.f);
  }
}

// T[w] = Im <lam| X_w |phi>,  phi given by real magnitudes P (phase applied on the fly).
template <int N>
__device__ __forceinline__ void qc_embed_ip(float (&T)[N], const SV<N>& lam, const float (&P)[1 << N]) {
  SV<N> phi;
  qc_phase_load<N>(phi, P);
#pragma unroll
  for (int w = 0; w < N; ++w) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      const int j = k ^ (1 << (N - 1 - w));
      acc = fmaf(lam.re[k], phi.im[j], acc);
      acc = fmaf(-lam.im[k], phi.re[j], acc);
    }
    T[w] = acc;
  }
}

// ---- embedding cotangents in the pulled-back frame
// The embedding U = prod_w RX(a_w) commutes with every X_w, so Im<Lam| X_w U |kappa> = Im<mu| X_w |kappa>
// with mu = U^dagger Lam, and the channel states are U applied to SPARSE vectors (G_v = -i X_v / 2):
//   kappa_0  = |0>,   kappa_s = sum_v da_v G_v |0>,   kappa_ss = [(sum_v da_v G_v)^2 + sum_v dda_v G_v] |0>,
// supported on basis states of weight <= 1 / <= 2.  Un-applying the n rotations on Lam (qc_unembed) and
// reading a few amplitudes of mu replaces rebuilding the product series and dense inner products.
template <int N, int W = 0>
__device__ __forceinline__ void qc_unembed(SV<N>& v, const float (&ca)[N], const float (&sa)[N]) {
  if constexpr (W < N) {
    g_rx<N, N - 1 - W>(v, ca[W], -sa[W]);
    qc_unembed<N, W + 1>(v, ca, sa);
  }
}
// T[w] = Im <mu| X_w |kappa_0>
template <int N>
__device__ __forceinline__ void qc_pull_ip0(float (&T)[N], const SV<N>& mu) {
#pragma unroll
  for (int w = 0; w < N; ++w) T[w] = -mu.im[1 << (N - 1 - w)];
}
// T[w] = Im <mu| X_w |kappa_s>
template <int N>
__device__ __forceinline__ void qc_pull_ip1(float (&T)[N], const SV<N>& mu, const float (&da)[N]) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const int bw = 1 << (N - 1 - w);
    float acc = da[w] * mu.re[0];
#pragma unroll
    for (int v = 0; v < N; ++v)
      if (v != w) acc = fmaf(da[v], mu.re[bw | (1 << (N - 1 - v))], acc);
    T[w] = -0.5f * acc;
  }
}
// T[w] = Im <mu| X_w |kappa_ss>
template <int N>
__device__ __forceinline__ void qc_pull_ip2(float (&T)[N], const SV<N>& mu, const float (&da)[N],
                                            const float (&dda)[N]) {
  float S = 0.f;
#pragma unroll
  for (int v = 0; v < N; ++v) S = fmaf(da[v], da[v], S);
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const int bw = 1 << (N - 1 - w);
    float a = S * mu.im[bw];
    float b = dda[w] * mu.re[0];
#pragma unroll
    for (int u = 0; u < N; ++u) {
      if (u != w) b = fmaf(dda[u], mu.re[bw | (1 << (N - 1 - u))], b);
#pragma unroll
      for (int v = u + 1; v < N; ++v)
        a = fmaf(2.f * da[u] * da[v], mu.im[(1 << (N - 1 - u)) ^ (1 << (N - 1 - v)) ^ bw], a);
    }
    T[w] = 0.25f * a - 0.5f * b;
  }
}

// <Z_w> style signed sums: out[w] = sum_k t[k] * (1 - 2 bit_{N-1-w}(k))
template <int N>
__device__ __forceinline__ void qc_signed_sums(float (&out)[N], const float (&t)[1 << N]) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) acc += ((k >> (N - 1 - w)) & 1) ? -t[k] : t[k];
    out[w] = acc;
  }
}
