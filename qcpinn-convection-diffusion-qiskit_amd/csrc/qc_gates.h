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

// ---- packed fp32 arithmetic.
// The gfx950 VALU issues a wave64 v_fma_f32 in 4 cycles and a v_pk_fma_f32 (two fp32 lanes per 64-bit register pair)
// in the same 4 (MI355X_MICROARCH.md, per-instruction cycle constants): scalar fp32 code tops out at HALF the 157 TF
// vector peak.  An amplitude is therefore ONE register pair (re, im), and every gate update is written on pairs:
//   (x + i y)(a)      = x a + {-y, y} swap(a)        (swap = the two halves exchanged: an op_sel modifier, free)
//   Im(conj(l) a)     = lo - hi of l * swap(a)
// so a complex multiply-add is 2 packed instructions instead of 4 scalar ones, whatever bit the gate acts on; the
// (cos, sin) / (re, im) coefficient pairs of the trig and Haar tables are adjacent in memory and reach the packed
// instructions as SGPR pairs with broadcast / negate modifiers (no repacking).  The layout is also the (re, im)
// record of the statevectors in LDS and HBM (n >= 9 family, final-state hand-off), so loads and stores are 64-bit.
typedef float qf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ qf2 qc_swp(const qf2 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ qf2 qc_dup(const float a) { return (qf2){a, a}; }
__device__ __forceinline__ qf2 qc_pm(const float a) { return (qf2){a, -a}; }     // {a, -a}
__device__ __forceinline__ qf2 qc_mp(const float a) { return (qf2){-a, a}; }     // {-a, a}
__device__ __forceinline__ qf2 qc_pk_fma(const qf2 a, const qf2 b, const qf2 c) { return __builtin_elementwise_fma(a, b, c); }
// z * a for z = x + i y
__device__ __forceinline__ qf2 qc_cmul(const float x, const float y, const qf2 a) { return qc_pk_fma(qc_mp(y), qc_swp(a), qc_dup(x) * a); }
// acc + z * a
__device__ __forceinline__ qf2 qc_cfma(const float x, const float y, const qf2 a, const qf2 acc) {
  return qc_pk_fma(qc_mp(y), qc_swp(a), qc_pk_fma(qc_dup(x), a, acc));
}

// M real numbers as M/2 register pairs (embedding magnitudes, diagonals, probabilities); element k = half (k & 1) of
// pair k >> 1.  operator[] keeps element-wise code (compile-time indices after unrolling) readable.
template <int M>
struct QcPk {
  static_assert(M >= 2 && M % 2 == 0, "packed arrays hold an even number of floats");
  qf2 p[M / 2];
  struct Ref {
    qf2& q;
    const int h;
    __device__ __forceinline__ operator float() const { return h ? q.y : q.x; }
    __device__ __forceinline__ Ref& operator=(const float x) {
      if (h) q.y = x;
      else q.x = x;
      return *this;
    }
    __device__ __forceinline__ Ref& operator=(const Ref& o) { return *this = (float)o; }
  };
  __device__ __forceinline__ Ref operator[](const int k) { return Ref{p[k >> 1], k & 1}; }
  __device__ __forceinline__ float operator[](const int k) const { return (k & 1) ? p[k >> 1].y : p[k >> 1].x; }
};

template <int N>
struct SV {
  qf2 a[1 << N];   // a[k] = (re, im) of amplitude k
};

// insert a zero bit at position B into k
__host__ __device__ constexpr int qc_ins0(int k, int B) { return ((k >> B) << (B + 1)) | (k & ((1 << B) - 1)); }
__host__ __device__ constexpr int qc_popc(int k) { return k == 0 ? 0 : (k & 1) + qc_popc(k >> 1); }

// ------------------------------------------------------------------ single-bit gates
// `s` carries the sign: the adjoint of a rotation is the same body with s -> -s.
template <int N, int B>
__device__ __forceinline__ void g_rx(SV<N>& v, float c, float s) {   // [[c, -i s], [-i s, c]]
  const qf2 cv = qc_dup(c), sv = qc_pm(s);
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const qf2 a = v.a[i0], b = v.a[i1];
    v.a[i0] = qc_pk_fma(sv, qc_swp(b), cv * a);   // c a - i s b = c a + s (b.im, -b.re)
    v.a[i1] = qc_pk_fma(sv, qc_swp(a), cv * b);
  }
}
template <int N, int B>
__device__ __forceinline__ void g_ry(SV<N>& v, float c, float s) {   // [[c, -s], [s, c]]
  const qf2 cv = qc_dup(c), sv = qc_dup(s);
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const qf2 a = v.a[i0], b = v.a[i1];
    v.a[i0] = qc_pk_fma(-sv, b, cv * a);
    v.a[i1] = qc_pk_fma(sv, a, cv * b);
  }
}
template <int N, int B>
__device__ __forceinline__ void g_rz(SV<N>& v, float c, float s) {   // diag(c - i s, c + i s)
  const qf2 cv = qc_dup(c), sv = qc_pm(s);
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const qf2 a = v.a[i0], b = v.a[i1];
    v.a[i0] = qc_pk_fma(sv, qc_swp(a), cv * a);
    v.a[i1] = qc_pk_fma(-sv, qc_swp(b), cv * b);
  }
}
template <int N, int B>
__device__ __forceinline__ void g_h(SV<N>& v) {
  const qf2 r = qc_dup(0.70710678118654752440f);
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const qf2 a = v.a[i0], b = v.a[i1];
    v.a[i0] = (a + b) * r;
    v.a[i1] = (a - b) * r;
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
    const qf2 a = v.a[i0];
    v.a[i0] = v.a[i1];
    v.a[i1] = a;
  }
}
template <int N, int CB, int TB>
__device__ __forceinline__ void g_crx(SV<N>& v, float c, float s) {
  const qf2 cv = qc_dup(c), sv = qc_pm(s);
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    const qf2 a = v.a[i0], b = v.a[i1];
    v.a[i0] = qc_pk_fma(sv, qc_swp(b), cv * a);
    v.a[i1] = qc_pk_fma(sv, qc_swp(a), cv * b);
  }
}
template <int N, int CB, int TB>
__device__ __forceinline__ void g_crz(SV<N>& v, float c, float s) {
  const qf2 cv = qc_dup(c), sv = qc_pm(s);
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    const qf2 a = v.a[i0], b = v.a[i1];
    v.a[i0] = qc_pk_fma(sv, qc_swp(a), cv * a);
    v.a[i1] = qc_pk_fma(-sv, qc_swp(b), cv * b);
  }
}

// ------------------------------------------------------------------ fixed two-wire unitary
// `u` points at 32 floats: row-major 4x4, (re, im) interleaved; wave-uniform address.  The four output rows are
// accumulated side by side (consecutive packed instructions never depend on each other).
template <int N, int HB, int LB, class UP = const float* __restrict__>
__device__ __forceinline__ void g_u4(SV<N>& v, UP u) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int b = qc_ctl_base<HB, LB>(k);
    const int idx[4] = {b, b | (1 << LB), b | (1 << HB), b | (1 << HB) | (1 << LB)};
    qf2 x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = v.a[idx[j]];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = qc_dup(u[(r * 4) * 2]) * x[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = qc_pk_fma(qc_mp(u[(r * 4) * 2 + 1]), qc_swp(x[0]), y[r]);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = qc_pk_fma(qc_dup(u[(r * 4 + j) * 2]), x[j], y[r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[r] = qc_pk_fma(qc_mp(u[(r * 4 + j) * 2 + 1]), qc_swp(x[j]), y[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v.a[idx[r]] = y[r];
  }
}

// ------------------------------------------------------------------ generator inner products
// Im <lam| G |chi> for the gate generators: what the adjoint sweep accumulates per parameter.
// Im(conj(l) y) = l.re y.im - l.im y.re = (lo - hi) of l * swap(y); sums run on pairs, one subtraction at the end.
template <int N, int B>
__device__ __forceinline__ float ip_x(const SV<N>& l, const SV<N>& x) {
  qf2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    acc0 = qc_pk_fma(l.a[i0], qc_swp(x.a[i1]), acc0);
    acc1 = qc_pk_fma(l.a[i1], qc_swp(x.a[i0]), acc1);
  }
  const qf2 acc = acc0 + acc1;
  return acc.x - acc.y;
}
template <int N, int B>
__device__ __forceinline__ float ip_z(const SV<N>& l, const SV<N>& x) {
  qf2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    acc0 = qc_pk_fma(l.a[i0], qc_swp(x.a[i0]), acc0);
    acc1 = qc_pk_fma(l.a[i1], qc_swp(x.a[i1]), acc1);
  }
  const qf2 acc = acc0 - acc1;
  return acc.x - acc.y;
}
template <int N, int B>
__device__ __forceinline__ float ip_y(const SV<N>& l, const SV<N>& x) {   // Re(conj(l_1) x_0 - conj(l_0) x_1)
  qf2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    acc0 = qc_pk_fma(l.a[i1], x.a[i0], acc0);
    acc1 = qc_pk_fma(l.a[i0], x.a[i1], acc1);
  }
  const qf2 acc = acc0 - acc1;
  return acc.x + acc.y;
}
template <int N, int CB, int TB>
__device__ __forceinline__ float ip_cx(const SV<N>& l, const SV<N>& x) {
  qf2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    acc0 = qc_pk_fma(l.a[i0], qc_swp(x.a[i1]), acc0);
    acc1 = qc_pk_fma(l.a[i1], qc_swp(x.a[i0]), acc1);
  }
  const qf2 acc = acc0 + acc1;
  return acc.x - acc.y;
}
template <int N, int CB, int TB>
__device__ __forceinline__ float ip_cz(const SV<N>& l, const SV<N>& x) {
  qf2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < (1 << (N - 2)); ++k) {
    const int i0 = qc_ctl_base<CB, TB>(k) | (1 << CB), i1 = i0 | (1 << TB);
    acc0 = qc_pk_fma(l.a[i0], qc_swp(x.a[i0]), acc0);
    acc1 = qc_pk_fma(l.a[i1], qc_swp(x.a[i1]), acc1);
  }
  const qf2 acc = acc0 - acc1;
  return acc.x - acc.y;
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
__device__ __forceinline__ void qc_embed_series(QcPk<(1 << N)>& P0, QcPk<(1 << N)>& P1, QcPk<(1 << N)>& P2,
                                                const float (&ca)[N], const float (&sa)[N],
                                                const float (&da)[N], const float (&dda)[N]) {
  // level w holds 2^(w+1) magnitudes as 2^w pairs; the next wire turns every HALF h of pair i into pair 2i + h
  // (one packed multiply by {c, s} with the half broadcast), wire 0 ends up the most significant index bit
  {
    const float c = ca[0], s = sa[0];
    P0.p[0] = (qf2){c, s};
    if constexpr (ORDER >= 1) P1.p[0] = (qf2){-0.5f * da[0] * s, 0.5f * da[0] * c};
    if constexpr (ORDER >= 2) {
      const float q = 0.25f * da[0] * da[0];
      P2.p[0] = (qf2){-0.5f * dda[0] * s - q * c, 0.5f * dda[0] * c - q * s};
    }
  }
#pragma unroll
  for (int w = 1; w < N; ++w) {
    const float c = ca[w], s = sa[w];
    const qf2 cs = {c, s};
    const qf2 w1 = {-0.5f * da[w] * s, 0.5f * da[w] * c};
    const float q = 0.25f * da[w] * da[w];
    const qf2 w2 = {-0.5f * dda[w] * s - q * c, 0.5f * dda[w] * c - q * s};
#pragma unroll
    for (int i = (1 << (w - 1)) - 1; i >= 0; --i) {
      const qf2 p0 = P0.p[i];
      qf2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
      if constexpr (ORDER >= 1) p1 = P1.p[i];
      if constexpr (ORDER >= 2) p2 = P2.p[i];
#pragma unroll
      for (int h = 1; h >= 0; --h) {
        const qf2 b0 = qc_dup(h ? p0.y : p0.x);
        P0.p[2 * i + h] = b0 * cs;
        if constexpr (ORDER >= 1) {
          const qf2 b1 = qc_dup(h ? p1.y : p1.x);
          P1.p[2 * i + h] = qc_pk_fma(b0, w1, b1 * cs);
          if constexpr (ORDER >= 2) {
            const qf2 b2 = qc_dup(h ? p2.y : p2.x);
            P2.p[2 * i + h] = qc_pk_fma(b0, w2, qc_pk_fma(b1 + b1, w1, b2 * cs));
          }
        }
      }
    }
  }
}

// real magnitudes -> complex amplitudes with the (-i)^{popcount} phase
template <int N>
__device__ __forceinline__ void qc_phase_load(SV<N>& v, const QcPk<(1 << N)>& P) {
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) {
    const int ph = qc_popc(k) & 3;
    const qf2 m = ph == 0 ? (qf2){1.f, 0.f} : ph == 1 ? (qf2){0.f, -1.f} : ph == 2 ? (qf2){-1.f, 0.f} : (qf2){0.f, 1.f};
    v.a[k] = qc_dup(P[k]) * m;
  }
}

// T[w] = Im <lam| X_w |phi>,  phi given by real magnitudes P (phase applied on the fly).
template <int N>
__device__ __forceinline__ void qc_embed_ip(float (&T)[N], const SV<N>& lam, const QcPk<(1 << N)>& P) {
  SV<N> phi;
  qc_phase_load<N>(phi, P);
#pragma unroll
  for (int w = 0; w < N; ++w) {
    qf2 acc = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) acc = qc_pk_fma(lam.a[k], qc_swp(phi.a[k ^ (1 << (N - 1 - w))]), acc);
    T[w] = acc.x - acc.y;
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
  for (int w = 0; w < N; ++w) T[w] = -mu.a[1 << (N - 1 - w)].y;
}
// T[w] = Im <mu| X_w |kappa_s>
template <int N>
__device__ __forceinline__ void qc_pull_ip1(float (&T)[N], const SV<N>& mu, const float (&da)[N]) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const int bw = 1 << (N - 1 - w);
    float acc = da[w] * mu.a[0].x;
#pragma unroll
    for (int v = 0; v < N; ++v)
      if (v != w) acc = fmaf(da[v], mu.a[bw | (1 << (N - 1 - v))].x, acc);
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
    float a = S * mu.a[bw].y;
    float b = dda[w] * mu.a[0].x;
#pragma unroll
    for (int u = 0; u < N; ++u) {
      if (u != w) b = fmaf(dda[u], mu.a[bw | (1 << (N - 1 - u))].x, b);
#pragma unroll
      for (int v = u + 1; v < N; ++v)
        a = fmaf(2.f * da[u] * da[v], mu.a[(1 << (N - 1 - u)) ^ (1 << (N - 1 - v)) ^ bw].y, a);
    }
    T[w] = 0.25f * a - 0.5f * b;
  }
}

// <Z_w> style signed sums: out[w] = sum_k t[k] * (1 - 2 bit_{N-1-w}(k))
template <int N>
__device__ __forceinline__ void qc_signed_sums(float (&out)[N], const QcPk<(1 << N)>& t) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    const int b = N - 1 - w;   // index bit of wire w; bit 0 = the two halves of a pair
    qf2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (1 << (N - 1)); ++j) {
      if (b >= 1 && ((j >> (b >= 1 ? b - 1 : 0)) & 1)) acc -= t.p[j];
      else acc += t.p[j];
    }
    out[w] = b == 0 ? acc.x - acc.y : acc.x + acc.y;
  }
}

// d[k] = sum_w q[w] (1 - 2 bit_{N-1-w}(k)): the diagonal of sum_w q_w Z_w, built by doubling (pair-index bit m = wire N-2-m)
template <int N>
__device__ __forceinline__ void qc_sign_sums(QcPk<(1 << N)>& d, const float (&q)[N]) {
  d.p[0] = (qf2){q[N - 1], -q[N - 1]};
#pragma unroll
  for (int m = 0; m < N - 1; ++m) {
    const qf2 qw = qc_dup(q[N - 2 - m]);
#pragma unroll
    for (int j = 0; j < (1 << m); ++j) {
      d.p[j | (1 << m)] = d.p[j] - qw;
      d.p[j] = d.p[j] + qw;
    }
  }
}
