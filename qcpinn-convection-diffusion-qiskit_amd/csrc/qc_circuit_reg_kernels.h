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
#pragma once
#include "qc_gates.h"
#include "qc_internal.h"

#include <stdlib.h>

#include <utility>

namespace {

// ------------------------------------------------------------------ program policies
// DynProg<N>: the gate program is run-time data (scalar switch per gate, any ansatz).
template <int NQ>
struct DynProg {
  static constexpr int N = NQ;
  __device__ static __forceinline__ void fwd(SV<NQ> (&v)[1], const QcGate* __restrict__ prog,
                                             const QcTrig* __restrict__ trig, const float* __restrict__ umat,
                                             int n_gates, int absorb) {
    for (int g = absorb ? NQ : 0; g < n_gates; ++g) {
      const QcGate gt = prog[g];
      const QcTrig tr = trig[g];
      qc_apply_gate<NQ, 1, false>(v, gt, tr.c, tr.s, umat);
    }
  }
  // Reverse sweep for one (chi, lam) pair: accumulates Im<lam|G|chi> per parameter slot into
  // acc_wave[slot] (LDS, one row per wave), then un-applies the gate on both.
  __device__ static __forceinline__ void bwd(SV<NQ> (&cl)[2], const QcGate* __restrict__ prog,
                                             const QcTrig* __restrict__ trig, const float* __restrict__ umat,
                                             int n_gates, float* __restrict__ acc_wave, int lane, int absorb) {
    const int g_first = absorb ? NQ : 0;
    for (int g = n_gates - 1; g >= g_first; --g) {
      const QcGate gt = prog[g];
      const QcTrig tr = trig[g];
      if (gt.op != QC_U4 && gt.slot >= 0) {
        const float gr = qc_wave_sum_to_lane63(qc_gate_grad<NQ>(cl[1], cl[0], gt));
        if (lane == 63) acc_wave[gt.slot] += gr;
      }
      qc_apply_gate<NQ, 2, true>(cl, gt, tr.c, tr.s, umat);
    }
  }
};

// StatProg<SP>: the gate list is a compile-time constant (SP::g[], generated from circuits.py by
// gen_static.py), so the whole circuit is straight-line code: no dispatch, no register shuffling
// at control-flow merges, trig table entries at constant offsets (scalar loads).
struct SGate {
  int op, ba, bb, slot;
};

template <int NQ, int K, bool ADJ, int OP, int BA, int BB, int SLOT>
__device__ __forceinline__ void qc_static_gate(SV<NQ> (&v)[K], const float c, const float s_in,
                                               const float* __restrict__ umat) {
  const float s = ADJ ? -s_in : s_in;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    if constexpr (OP == QC_RX) g_rx<NQ, BA>(v[q], c, s);
    else if constexpr (OP == QC_RY) g_ry<NQ, BA>(v[q], c, s);
    else if constexpr (OP == QC_RZ) g_rz<NQ, BA>(v[q], c, s);
    else if constexpr (OP == QC_H) g_h<NQ, BA>(v[q]);
    else if constexpr (OP == QC_CNOT) g_cnot<NQ, BA, BB>(v[q]);
    else if constexpr (OP == QC_CRX) g_crx<NQ, BA, BB>(v[q], c, s);
    else if constexpr (OP == QC_CRZ) g_crz<NQ, BA, BB>(v[q], c, s);
    else if constexpr (OP == QC_U4) g_u4<NQ, BA, BB>(v[q], umat + (SLOT * 2 + (ADJ ? 1 : 0)) * 32);
  }
}

template <int NQ, int OP, int BA, int BB>
__device__ __forceinline__ float qc_static_grad(const SV<NQ>& lam, const SV<NQ>& chi) {
  if constexpr (OP == QC_RX) return ip_x<NQ, BA>(lam, chi);
  else if constexpr (OP == QC_RY) return ip_y<NQ, BA>(lam, chi);
  else if constexpr (OP == QC_RZ) return ip_z<NQ, BA>(lam, chi);
  else if constexpr (OP == QC_CRX) return ip_cx<NQ, BA, BB>(lam, chi);
  else if constexpr (OP == QC_CRZ) return ip_cz<NQ, BA, BB>(lam, chi);
  else return 0.f;
}

template <class SP>
struct StatProg {
  static constexpr int N = SP::N;
  // The (cos, sin) pairs of every parametric gate are fetched before the first gate (they are wave-uniform:
  // scalar loads into SGPRs, batched by the compiler), so no gate starts by waiting on a scalar load.
  struct Trig {
    float c[SP::G], s[SP::G];
  };
  template <int I>
  __device__ static __forceinline__ void load_one(Trig& t, const QcTrig* __restrict__ trig) {
    constexpr SGate g = SP::g[I];
    t.c[I] = 1.f;
    t.s[I] = 0.f;
    if constexpr (g.op != QC_U4 && g.slot >= 0 && !fused(I)) {
      t.c[I] = trig[I].c;
      t.s[I] = trig[I].s;
    }
  }
  template <int... Is>
  __device__ static __forceinline__ void load_all(Trig& t, const QcTrig* __restrict__ trig,
                                                  std::integer_sequence<int, Is...>) {
    (load_one<Is>(t, trig), ...);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- runs of >= 2 consecutive diagonal gates (RZ / CRZ) are ONE table multiply (tables: qc_fill_diag_tables)
  static constexpr bool is_diag(int i) { return i >= 0 && i < SP::G && (SP::g[i].op == QC_RZ || SP::g[i].op == QC_CRZ); }
  static constexpr int run_begin(int i) { while (is_diag(i - 1)) --i; return i; }
  static constexpr int run_end(int i) { while (is_diag(i)) ++i; return i; }          // one past the last gate
  static constexpr bool fused(int i) { return is_diag(i) && run_end(i) - run_begin(i) >= 2; }
  static constexpr int run_ordinal(int i) {
    int r = 0;
    for (int g = 0; g < run_begin(i);) {
      if (!is_diag(g)) { ++g; continue; }
      const int e = run_end(g);
      if (e - g >= 2) ++r;
      g = e;
    }
    return r;
  }
  static_assert(run_ordinal(SP::G) <= QC_MAX_DIAG_RUNS, "more fused diagonal runs than the host records (qc_find_diag_runs)");
  template <int I, bool ADJ, int K>
  __device__ static __forceinline__ void apply_table(SV<N> (&v)[K], const QcTrig* __restrict__ trig) {
    const QcTrig* tab = trig + SP::G + run_ordinal(I) * (1 << N);
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      const float dr = tab[k].c, di = ADJ ? -tab[k].s : tab[k].s;   // adjacent in the table: one SGPR pair
#pragma unroll
      for (int q = 0; q < K; ++q) v[q].a[k] = qc_cmul(dr, di, v[q].a[k]);
    }
  }
  // gradient terms of the gates H, H+1, ..., E-1 of one run from t[k] = Im(conj(lam_k) chi_k) (invariant under
  // the other diagonal gates of the run): signed by the target bit, masked by the control bit
  template <int H, int E>
  __device__ static __forceinline__ void run_grads(const QcPk<(1 << N)>& t, float (&gacc)[SP::P > 0 ? SP::P : 1]) {
    if constexpr (H < E) {
      constexpr SGate g = SP::g[H];
      constexpr bool ctl = g.op == QC_CRZ;
      constexpr int tb = ctl ? g.bb : g.ba;
      constexpr int cb = ctl ? g.ba : -1;
      qf2 a2 = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < (1 << (N - 1)); ++j) {
        if (cb >= 1 && !((j >> (cb >= 1 ? cb - 1 : 0)) & 1)) continue;
        if (tb >= 1 && ((j >> (tb >= 1 ? tb - 1 : 0)) & 1)) a2 -= t.p[j];
        else a2 += t.p[j];
      }
      // target on bit 0: the halves carry opposite signs; control on bit 0: only the upper halves count
      const float acc = tb == 0 ? a2.x - a2.y : (cb == 0 ? a2.y : a2.x + a2.y);
      if constexpr (g.slot >= 0) gacc[g.slot] += qc_wave_sum_to_lane63(acc);
      run_grads<H + 1, E>(t, gacc);
    }
  }
  template <int I>
  __device__ static __forceinline__ void fwd_one(SV<N> (&v)[1], const Trig& t, const float* __restrict__ umat,
                                                 int absorb, const QcTrig* __restrict__ trig) {
    constexpr SGate g = SP::g[I];
    if constexpr (I < N) {
      if (absorb) return;   // leading RX layer folded into the embedding angles
    }
    if constexpr (fused(I)) {
      if constexpr (I == run_begin(I)) {
        apply_table<I, false, 1>(v, trig);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      qc_static_gate<N, 1, false, g.op, g.ba, g.bb, g.slot>(v, t.c[I], t.s[I], umat);
      __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from interleaving whole gates (register pressure)
    }
  }
  template <int... Is>
  __device__ static __forceinline__ void fwd_all(SV<N> (&v)[1], const Trig& t, const float* __restrict__ umat,
                                                 int absorb, const QcTrig* __restrict__ trig,
                                                 std::integer_sequence<int, Is...>) {
    (fwd_one<Is>(v, t, umat, absorb, trig), ...);
  }
  __device__ static __forceinline__ void fwd(SV<N> (&v)[1], const QcGate* __restrict__, const QcTrig* __restrict__ trig,
                                             const float* __restrict__ umat, int, int absorb) {
    Trig t;
    load_all(t, trig, std::make_integer_sequence<int, SP::G>{});
    fwd_all(v, t, umat, absorb, trig, std::make_integer_sequence<int, SP::G>{});
  }
  // Reverse sweep.  Parameter slots are compile-time constants here, so the wave totals of the
  // gradient terms stay in registers (gacc[slot], valid in lane 63) and reach LDS once, after the
  // sweep, instead of one LDS read-modify-write round trip per gate.
  template <int J>
  __device__ static __forceinline__ void bwd_one(SV<N> (&cl)[2], const Trig& t, const float* __restrict__ umat,
                                                 float (&gacc)[SP::P > 0 ? SP::P : 1], int absorb,
                                                 const QcTrig* __restrict__ trig) {
    constexpr int I = SP::G - 1 - J;
    constexpr SGate g = SP::g[I];
    if constexpr (I < N) {
      if (absorb) return;
    }
    if constexpr (fused(I)) {
      if constexpr (I == run_end(I) - 1) {   // first gate of the run met by the reverse sweep
        QcPk<(1 << N)> tk;   // Im(conj(lam_k) chi_k), two k per pair for the signed sums
#pragma unroll
        for (int k = 0; k < (1 << N); ++k) {
          const qf2 m = cl[1].a[k] * qc_swp(cl[0].a[k]);
          tk[k] = m.x - m.y;
        }
        run_grads<run_begin(I), run_end(I)>(tk, gacc);
        apply_table<I, true, 2>(cl, trig);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if constexpr (g.op != QC_U4 && g.slot >= 0)
        gacc[g.slot] += qc_wave_sum_to_lane63(qc_static_grad<N, g.op, g.ba, g.bb>(cl[1], cl[0]));
      qc_static_gate<N, 2, true, g.op, g.ba, g.bb, g.slot>(cl, t.c[I], t.s[I], umat);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  template <int... Js>
  __device__ static __forceinline__ void bwd_all(SV<N> (&cl)[2], const Trig& t, const float* __restrict__ umat,
                                                 float (&gacc)[SP::P > 0 ? SP::P : 1], int absorb,
                                                 const QcTrig* __restrict__ trig, std::integer_sequence<int, Js...>) {
    (bwd_one<Js>(cl, t, umat, gacc, absorb, trig), ...);
  }
  __device__ static __forceinline__ void bwd(SV<N> (&cl)[2], const QcGate* __restrict__, const QcTrig* __restrict__ trig,
                                             const float* __restrict__ umat, int, float* __restrict__ acc_wave,
                                             int lane, int absorb) {
    float gacc[SP::P > 0 ? SP::P : 1];
#pragma unroll
    for (int k = 0; k < (SP::P > 0 ? SP::P : 1); ++k) gacc[k] = 0.f;
    Trig t;
    load_all(t, trig, std::make_integer_sequence<int, SP::G>{});
    bwd_all(cl, t, umat, gacc, absorb, trig, std::make_integer_sequence<int, SP::G>{});
#pragma unroll
    for (int k = 0; k < SP::P; ++k)
      if (lane == 63) acc_wave[k] += gacc[k];   // lane 63 holds the wave totals
  }
};

// half-angle (cos, sin) of wire w's embedding rotation for one point
__device__ __forceinline__ void qc_wire_sincos(float& c, float& s, const float* __restrict__ a, int64_t B, int64_t p, int w,
                                               const QcTrig* __restrict__ trig, int absorb) {
  // absorb: gate w is RX(theta_w) on wire w, applied right after RX(a_w): one rotation by a_w + theta_w
  const float h = 0.5f * (a[(int64_t)w * B + p] + (absorb ? trig[w].th : 0.f));
  sincosf(h, &s, &c);
}
// `cs` != nullptr: the block already holds them in LDS as [cos | sin][N][64] (each of the jet kernels' six waves
// needs the same 2N values for its 64 points: waves 0..N-1 compute one wire each instead of all six computing all)
template <int N>
__device__ __forceinline__ void load_sincos(float (&ca)[N], float (&sa)[N], const float* __restrict__ a,
                                            int64_t B, int64_t p, const QcTrig* __restrict__ trig, int absorb,
                                            const float* cs = nullptr) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    if (cs != nullptr) {
      ca[w] = cs[w * 64 + (threadIdx.x & 63)];
      sa[w] = cs[(N + w) * 64 + (threadIdx.x & 63)];
    } else {
      qc_wire_sincos(ca[w], sa[w], a, B, p, w, trig, absorb);
    }
  }
}

// Embedding series of channel `ch`'s direction for this lane's point: P0 = phi, P1 = d phi,
// P2 = d2 phi (only as far as the channel needs).
template <int N>
__device__ __forceinline__ void channel_series(QcPk<(1 << N)>& P0, QcPk<(1 << N)>& P1, QcPk<(1 << N)>& P2,
                                               int ch, const float* __restrict__ ajets, int64_t B, int64_t pc,
                                               const QcTrig* __restrict__ trig, int absorb, const float* cs = nullptr) {
  float ca[N], sa[N], da[N], dda[N];
  load_sincos<N>(ca, sa, ajets, B, pc, trig, absorb, cs);
  const int dirch = ch == 0 ? 0 : (ch <= 3 ? ch : ch - 2);  // channel holding the first derivative
#pragma unroll
  for (int w = 0; w < N; ++w) {
    da[w] = ch >= 1 ? ajets[((int64_t)dirch * N + w) * B + pc] : 0.f;
    dda[w] = ch >= 4 ? ajets[((int64_t)ch * N + w) * B + pc] : 0.f;
  }
  if (ch == 0) qc_embed_series<N, 0>(P0, P1, P2, ca, sa, da, dda);
  else if (ch <= 3) qc_embed_series<N, 1>(P0, P1, P2, ca, sa, da, dda);
  else qc_embed_series<N, 2>(P0, P1, P2, ca, sa, da, dda);
}

// Initial vector of channel `ch`.  amp != 0: amplitude encoding — `ajets` then holds the jets of the
// (normalised, zero-padded) initial amplitudes themselves (qc_amp.hip): amplitude k = feature k, real.
template <int N>
__device__ __forceinline__ void build_channel(SV<N>& v, int ch, const float* __restrict__ ajets, int64_t B,
                                              int64_t pc, int amp, const QcTrig* __restrict__ trig, int absorb,
                                              const float* cs = nullptr) {
  if (amp) {
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      v.a[k].x = k < N ? ajets[((int64_t)ch * N + k) * B + pc] : 0.f;
      v.a[k].y = 0.f;
    }
    return;
  }
  QcPk<(1 << N)> P0, P1, P2;
  channel_series<N>(P0, P1, P2, ch, ajets, B, pc, trig, absorb, cs);
  if (ch == 0) qc_phase_load<N>(v, P0);
  else if (ch <= 3) qc_phase_load<N>(v, P1);
  else qc_phase_load<N>(v, P2);
}

// Hides a pointer's provenance from the optimiser: loads through the result are not merged with
// earlier loads, so values recomputed from them do not stay live in registers in between.
__device__ __forceinline__ const float* qc_launder(const float* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// ================================================================== value channel only
template <class PG, int WPB>
__device__ __forceinline__ void k_value_fwd_body(const int64_t bid, const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, int n_gates,
                                                   const float* __restrict__ angles, float* __restrict__ expval,
                                                   int64_t B, int amp) {
  constexpr int N = PG::N;
  const int64_t p = (int64_t)bid * (64 * WPB) + threadIdx.x;
  const int64_t pc = p < B ? p : B - 1;
  SV<N> v[1];
  build_channel<N>(v[0], 0, angles, B, pc, amp & 1, trig, amp >> 1);
  PG::fwd(v, prog, trig, umat, n_gates, amp >> 1);
  QcPk<(1 << N)> t;
  float q[N];
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) {
    const qf2 m = v[0].a[k] * v[0].a[k];
    t[k] = m.x + m.y;
  }
  qc_signed_sums<N>(q, t);
  if (p < B) {
#pragma unroll
    for (int w = 0; w < N; ++w) expval[(int64_t)w * B + p] = q[w];
  }
}

template <class PG>
__global__ void __launch_bounds__(256) k_value_fwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, int n_gates,
                                                   const float* __restrict__ angles, float* __restrict__ expval,
                                                   int64_t B, int amp) {
  k_value_fwd_body<PG, 4>(blockIdx.x, prog, trig, umat, n_gates, angles, expval, B, amp);
}

template <class PG, int WPB>
__device__ __forceinline__ void k_value_bwd_body(const int64_t bid, const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, int n_gates, int n_params,
                                                   const float* __restrict__ angles, const float* __restrict__ cot,
                                                   float* __restrict__ d_angles, float* __restrict__ part,
                                                   int64_t part_stride, int64_t row0, int64_t B, int amp) {
  constexpr int N = PG::N;
  extern __shared__ float smem[];  // [WPB waves][n_params]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < WPB * n_params; i += 64 * WPB) smem[i] = 0.f;
  __syncthreads();

  const int64_t p = (int64_t)bid * (64 * WPB) + threadIdx.x;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  SV<N> cl[2];  // [0] = chi, [1] = lambda
  {
    SV<N> v[1];
    build_channel<N>(v[0], 0, angles, B, pc, amp & 1, trig, amp >> 1);
    PG::fwd(v, prog, trig, umat, n_gates, amp >> 1);
    cl[0] = v[0];
  }
  float qb[N];
#pragma unroll
  for (int w = 0; w < N; ++w) qb[w] = live ? cot[(int64_t)w * B + pc] : 0.f;
  {
    QcPk<(1 << N)> d;
    qc_sign_sums<N>(d, qb);
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) cl[1].a[k] = qc_dup(d[k]) * cl[0].a[k];
  }
  PG::bwd(cl, prog, trig, umat, n_gates, smem + wave * n_params, lane, amp >> 1);
  float T[N];
  if (amp & 1) {   // d L / d(initial amplitude k) = 2 Re Lambda_k  (the initial amplitudes are real)
#pragma unroll
    for (int w = 0; w < N; ++w) T[w] = 2.f * cl[1].a[w].x;
  } else {
    QcPk<(1 << N)> Q0, Q1, Q2;
    channel_series<N>(Q0, Q1, Q2, 0, qc_launder(angles), B, pc, trig, amp >> 1);
    qc_embed_ip<N>(T, cl[1], Q0);
  }
  if (live) {
#pragma unroll
    for (int w = 0; w < N; ++w) d_angles[(int64_t)w * B + p] = T[w];
  }
  if (amp >> 1) {   // folded RX layer: d L / d theta_w = sum over points of d L / d angle_w
#pragma unroll
    for (int w = 0; w < N; ++w) {
      const float tot = qc_wave_sum_to_lane63(live ? T[w] : 0.f);
      if (lane == 63) smem[wave * n_params + prog[w].slot] += tot;
    }
  }
  __syncthreads();
  // one partial row per wave = per 64-point tile (same tiling as the MLP kernels)
  const int64_t tile = (int64_t)bid * WPB + wave;
  if (tile * 64 < B)
    for (int i = lane; i < n_params; i += 64) part[(row0 + tile) * part_stride + i] = smem[wave * n_params + i];
}

template <class PG>
__global__ void __launch_bounds__(256) k_value_bwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, int n_gates, int n_params,
                                                   const float* __restrict__ angles, const float* __restrict__ cot,
                                                   float* __restrict__ d_angles, float* __restrict__ part,
                                                   int64_t part_stride, int64_t row0, int64_t B, int amp) {
  k_value_bwd_body<PG, 4>(blockIdx.x, prog, trig, umat, n_gates, n_params, angles, cot, d_angles, part, part_stride, row0, B, amp);
}

// Final-state store handed from k_jets_fwd to k_jets_bwd: tile-major, so the 6 x A2 x 64 floats one
// block touches are one contiguous 6*A2*256-byte region (DRAM-page friendly), not A2 streams B apart.
// Units are amplitudes = (re, im) register pairs: 8 bytes per lane, one global_load/store_dwordx2 = 512 contiguous
// bytes per wave.
template <int A2>
__device__ __forceinline__ int64_t qc_chi_pair(int ch, int k, int64_t p) {
  return ((((p >> 6) * 6 + ch) * (A2 / 2) + k) << 6) | (p & 63);
}
template <int N>
__device__ __forceinline__ void qc_chi_write(float* __restrict__ chi_store, int ch, int64_t p, const SV<N>& v) {
  qf2* cs = reinterpret_cast<qf2*>(chi_store);
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) cs[qc_chi_pair<(2 << N)>(ch, k, p)] = v.a[k];
}
template <int N>
__device__ __forceinline__ void qc_chi_read(SV<N>& v, const float* __restrict__ chi_store, int ch, int64_t p) {
  const qf2* cs = reinterpret_cast<const qf2*>(chi_store);
#pragma unroll
  for (int k = 0; k < (1 << N); ++k) v.a[k] = cs[qc_chi_pair<(2 << N)>(ch, k, p)];
}

// ================================================================== six derivative channels
template <class PG>
__device__ __forceinline__ void k_jets_fwd_body(const int64_t bid, const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates,
                                                  const float* __restrict__ ajets, float* __restrict__ qjets,
                                                  int64_t B, float* __restrict__ chi_store, int amp) {
  constexpr int N = PG::N;
  constexpr int A2 = 2 << N;                 // floats per statevector
  __shared__ float s_chi0[A2 * 64];          // [amp*2+{re,im}][lane]
  __shared__ float s_sq[2 * N * 64];         // 2<chi_k|Z_w|chi_k> for k = x, y
  const int lane = threadIdx.x & 63;
  const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave index = channel (scalar)
  const int64_t p = (int64_t)bid * 64 + lane;
  const int64_t pc = p < B ? p : B - 1;

  __shared__ float s_cs[2 * N * 64];         // half-angle cos | sin of the block's 64 points, one wire per wave
  if (!(amp & 1)) {
    if (ch < N) qc_wire_sincos(s_cs[ch * 64 + lane], s_cs[(N + ch) * 64 + lane], ajets, B, pc, ch, trig, amp >> 1);
    __syncthreads();
  }
  SV<N> v[1];
  build_channel<N>(v[0], ch, ajets, B, pc, amp & 1, trig, amp >> 1, s_cs);
  PG::fwd(v, prog, trig, umat, n_gates, amp >> 1);
  if (chi_store != nullptr && p < B) qc_chi_write<N>(chi_store, ch, p, v[0]);   // final states for the adjoint kernel

  QcPk<(1 << N)> t;
  float q[N];
  qf2* s_x0 = reinterpret_cast<qf2*>(s_chi0);   // [amplitude][lane]
  if (ch == 0) {
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      s_x0[k * 64 + lane] = v[0].a[k];
      const qf2 m = v[0].a[k] * v[0].a[k];
      t[k] = m.x + m.y;
    }
    qc_signed_sums<N>(q, t);
  } else if (ch == 2 || ch == 3) {
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      const qf2 m = v[0].a[k] * v[0].a[k];
      t[k] = m.x + m.y;
    }
    qc_signed_sums<N>(q, t);
#pragma unroll
    for (int w = 0; w < N; ++w) s_sq[((ch - 2) * N + w) * 64 + lane] = 2.f * q[w];
  }
  __syncthreads();
  if (ch != 0) {
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) {
      const qf2 m = s_x0[k * 64 + lane] * v[0].a[k];   // Re(conj(chi_0) chi_c)
      t[k] = m.x + m.y;
    }
    qc_signed_sums<N>(q, t);
#pragma unroll
    for (int w = 0; w < N; ++w) q[w] *= 2.f;
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

template <class PG>
__global__ void __launch_bounds__(384) k_jets_fwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates,
                                                  const float* __restrict__ ajets, float* __restrict__ qjets,
                                                  int64_t B, float* __restrict__ chi_store, int amp) {
  k_jets_fwd_body<PG>(blockIdx.x, prog, trig, umat, n_gates, ajets, qjets, B, chi_store, amp);
}

#ifndef QC_JB_WAVES
#define QC_JB_WAVES 3
#endif
// LOAD: the final states come from chi_store (written by k_jets_fwd in the same step) instead of being
// recomputed from the angle jets: 12 instead of 18 circuit-equivalents per point.
template <class PG, bool LOAD>
__device__ __forceinline__ void k_jets_bwd_body(const int64_t bid, const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates, int n_params,
                                                  const float* __restrict__ ajets, const float* __restrict__ qbar,
                                                  float* __restrict__ abar, float* __restrict__ part,
                                                  int64_t part_stride, int64_t row0, int64_t B,
                                                  const float* __restrict__ chi_store, int amp) {
  constexpr int N = PG::N;
  constexpr int A2 = 2 << N;
  extern __shared__ float smem[];
  constexpr int XCH = LOAD ? 6 * 3 * N * 64 : 6 * A2 * 64;   // exchange region (floats)
  float* s_chi = smem;                       // !LOAD: [6][A2][64]; always reused as [6 waves][3][N][64] at the end
  float* s_acc = smem + XCH;                 // [6 waves][n_params]
  const int lane = threadIdx.x & 63;
  const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* s_cs = s_acc + 6 * n_params;        // [cos | sin][N][64] of the embedding half-angles, for the tail
  for (int i = threadIdx.x; i < 6 * n_params; i += 384) s_acc[i] = 0.f;
  const int64_t p = (int64_t)bid * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  if (!(amp & 1) && ch < N)
    qc_wire_sincos(s_cs[ch * 64 + lane], s_cs[(N + ch) * 64 + lane], ajets, B, pc, ch, trig, amp >> 1);
  if constexpr (LOAD) __syncthreads();   // (the !LOAD path has its barrier after the state exchange)


  SV<N> cl[2];
  if constexpr (LOAD) {
    qc_chi_read<N>(cl[0], chi_store, ch, pc);
  } else {
    SV<N> v[1];
    build_channel<N>(v[0], ch, ajets, B, pc, amp & 1, trig, amp >> 1);
    PG::fwd(v, prog, trig, umat, n_gates, amp >> 1);
    cl[0] = v[0];
  }
  // The other channels' final states: through LDS when this launch computed them, straight from the
  // forward kernel's store (L2-resident) when LOAD -- then no exchange buffer and no barrier is needed
  // before the sweep, and the block's LDS shrinks to the small tail buffers.
  if constexpr (!LOAD) {
    qf2* mine = reinterpret_cast<qf2*>(s_chi) + ch * (A2 / 2) * 64;
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) mine[k * 64 + lane] = cl[0].a[k];
    __syncthreads();
  }
  // amplitude k of channel c's final state
  auto other = [&](int c, int k) -> qf2 {
    return LOAD ? reinterpret_cast<const qf2*>(chi_store)[qc_chi_pair<A2>(c, k, pc)]
                : reinterpret_cast<const qf2*>(s_chi)[(c * (A2 / 2) + k) * 64 + lane];
  };

  // ---- cotangent of this channel's final state (bilinear <Z> forms)
  auto dvec = [&](int c, QcPk<(1 << N)>& d) {
    float qb[N];
#pragma unroll
    for (int w = 0; w < N; ++w) qb[w] = live ? qbar[((int64_t)c * N + w) * B + pc] : 0.f;
    qc_sign_sums<N>(d, qb);
  };
  QcPk<(1 << N)> d;
  if (ch == 0) {
    dvec(0, d);
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) cl[1].a[k] = qc_dup(d[k]) * cl[0].a[k];
#pragma unroll 1
    for (int c = 1; c < QC_NCH; ++c) {   // one channel at a time: keeps the live set at one vector
      dvec(c, d);
#pragma unroll
      for (int k = 0; k < (1 << N); ++k) cl[1].a[k] = qc_pk_fma(qc_dup(d[k]), other(c, k), cl[1].a[k]);
    }
  } else {
    dvec(ch, d);
#pragma unroll
    for (int k = 0; k < (1 << N); ++k) cl[1].a[k] = qc_dup(d[k]) * other(0, k);
    if (ch == 2 || ch == 3) {
      dvec(ch + 2, d);
#pragma unroll
      for (int k = 0; k < (1 << N); ++k) cl[1].a[k] = qc_pk_fma(qc_dup(2.f * d[k]), cl[0].a[k], cl[1].a[k]);
    }
  }
  if constexpr (!LOAD) __syncthreads();  // everyone is done reading s_chi

  PG::bwd(cl, prog, trig, umat, n_gates, s_acc + ch * n_params, lane, amp >> 1);

  if (amp & 1) {
    // amplitude encoding: the channel's initial vector IS the input jet, so its cotangent is read off
    // the swept lambda directly: d L / d u_c[k] = 2 Re Lambda_c[k] (no coupling between channels here)
    if (live) {
#pragma unroll
      for (int w = 0; w < N; ++w) abar[((int64_t)ch * N + w) * B + p] = 2.f * cl[1].a[w].x;
    }
    __syncthreads();
  } else {
    // ---- cotangents of the angle jets: Im<Lambda| X_w |phi_c>, evaluated in the frame pulled back through
    // the embedding (qc_gates.h): un-apply the n embedding rotations on Lambda, then sparse reads
    const float* aj = qc_launder(ajets);
    float ca[N], sa[N], da[N], dda[N];
    load_sincos<N>(ca, sa, aj, B, pc, trig, amp >> 1, s_cs);
    qc_unembed<N>(cl[1], ca, sa);
    const int dirch = ch == 0 ? 0 : (ch <= 3 ? ch : ch - 2);  // channel holding the first derivative
#pragma unroll
    for (int w = 0; w < N; ++w) {
      da[w] = ch >= 1 ? aj[((int64_t)dirch * N + w) * B + pc] : 0.f;
      dda[w] = ch >= 4 ? aj[((int64_t)ch * N + w) * B + pc] : 0.f;
    }
    float* buf = s_chi + ch * 3 * N * 64;  // [3][N][64] per wave
    float T[N];
    if (ch == 0) {
      qc_pull_ip0<N>(T, cl[1]);
  #pragma unroll
      for (int w = 0; w < N; ++w) buf[(0 * N + w) * 64 + lane] = T[w];
    } else if (ch <= 3) {
      qc_pull_ip1<N>(T, cl[1], da);
  #pragma unroll
      for (int w = 0; w < N; ++w) buf[(0 * N + w) * 64 + lane] = T[w];
      qc_pull_ip0<N>(T, cl[1]);
  #pragma unroll
      for (int w = 0; w < N; ++w) buf[(1 * N + w) * 64 + lane] = T[w];
    } else {
      qc_pull_ip2<N>(T, cl[1], da, dda);
  #pragma unroll
      for (int w = 0; w < N; ++w) buf[(0 * N + w) * 64 + lane] = T[w];
      qc_pull_ip1<N>(T, cl[1], da);
  #pragma unroll
      for (int w = 0; w < N; ++w) buf[(1 * N + w) * 64 + lane] = 2.f * T[w];
      qc_pull_ip0<N>(T, cl[1]);
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
      if ((amp >> 1) && ch == 0) {   // folded RX layer: d L / d theta_w = sum over points of d L / d angle_w
        const float tot = qc_wave_sum_to_lane63(live ? r : 0.f);
        if (lane == 63) s_acc[prog[w].slot] += tot;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n_params; i += 384) {
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 6; ++wv) s += s_acc[wv * n_params + i];
    part[(row0 + bid) * part_stride + i] = s;
  }
}

template <class PG, bool LOAD>
__global__ void __launch_bounds__(384, QC_JB_WAVES) k_jets_bwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates, int n_params,
                                                  const float* __restrict__ ajets, const float* __restrict__ qbar,
                                                  float* __restrict__ abar, float* __restrict__ part,
                                                  int64_t part_stride, int64_t row0, int64_t B,
                                                  const float* __restrict__ chi_store, int amp) {
  k_jets_bwd_body<PG, LOAD>(blockIdx.x, prog, trig, umat, n_gates, n_params, ajets, qbar, abar, part, part_stride, row0, B, chi_store, amp);
}

// ---- the fused step's adjoint sweep, second form: block = 3 waves on one 64-point tile, wave = TWO channels
// handled one after the other ({t, value}, {x, xx}, {y, yy}).  Against the six-wave form above:
//   * 5 blocks per CU (15 waves at <= 128 VGPRs) hold the whole grid of BASELINE config 2 (1 024 residual + 228
//     value blocks <= 1 280 slots) in ONE round instead of 2.2 rounds of 512;
//   * no straggler: every wave forms its share D_a chi_a + D_b chi_b of the value channel's cotangent from the two
//     final states it loads anyway and leaves it in LDS; the value channel's wave adds the three shares when its
//     turn comes (second in its wave), instead of 160 extra global loads and five extra D-vector builds up front;
//   * 2 block barriers instead of 3, per-gate wave reductions accumulate over both channels in registers of the
//     same wave, cross-wave traffic in the tail shrinks to the one row that sums over all channels.
template <class PG>
__device__ __forceinline__ void k_jets_bwd2_body(const int64_t bid, const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, int n_gates, int n_params,
                                                   const float* __restrict__ ajets, const float* __restrict__ qbar,
                                                   float* __restrict__ abar, float* __restrict__ part,
                                                   int64_t part_stride, int64_t row0, int64_t B,
                                                   const float* __restrict__ chi_store, int amp, int stagger = 0) {
  constexpr int N = PG::N;
  constexpr int A2 = 2 << N;
  constexpr int NA = 1 << N;
  // diagnostic (timing only, results wrong): bit 7 of the flags = every block reads the final states of tile 0, i.e.
  // from cache - what the stage would take if its 56 MB of final states did not have to come from HBM / Infinity Cache
  const bool same_tile = (amp & 0x80) != 0;
  amp &= 0x7f;
  extern __shared__ float smem[];
  float* s_l0 = smem;                          // [3 waves][A2][64]: shares of lam_0
  float* s_t0 = s_l0 + 3 * A2 * 64;            // [2][N][64]: waves 1, 2: their channels' terms of abar[0]
  float* s_cs = s_t0 + 2 * N * 64;             // [cos | sin][N][64] of the embedding half-angles
  float* s_acc = s_cs + 2 * N * 64;            // [3 waves][n_params]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t p = (int64_t)bid * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  for (int i = threadIdx.x; i < 3 * n_params; i += 192) s_acc[i] = 0.f;
  // every second tile can start late (stagger x 8 128 cycles): the whole grid is resident in one round, so all waves
  // load at the same time and then all sweep at the same time (HBM and VALU take turns instead of overlapping)
  if (stagger > 0 && (bid & 1)) {
#pragma unroll 1
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  // wires wv, wv + 3, ... of the embedding half-angles
  for (int w = wv; w < N; w += 3) qc_wire_sincos(s_cs[w * 64 + lane], s_cs[(N + w) * 64 + lane], ajets, B, pc, w, trig, amp >> 1);
  // the point index as an opaque value: addresses derived from it are recomputed at their use instead of being hoisted
  // above the two-channel loop and held (or spilled) for the whole kernel
  auto pcf = [&]() {
    int64_t v = pc;
    asm volatile("" : "+v"(v));
    return v;
  };
  const int cha = wv == 0 ? 1 : wv + 1;        // first channel of this wave: t, x, y
  const int chb = wv == 0 ? 0 : wv + 3;        // second: value, xx, yy

  const qf2* chi2 = reinterpret_cast<const qf2*>(chi_store);
  qf2* s_l02 = reinterpret_cast<qf2*>(s_l0);     // [3 waves][amplitude][64]
  auto pcs = [&]() { return same_tile ? (int64_t)(threadIdx.x & 63) : pcf(); };   // point index for final-state reads
  auto load_chi = [&](SV<N>& v, int c) { qc_chi_read<N>(v, chi_store, c, pcs()); };
  auto dvec = [&](int c, QcPk<NA>& d) {
    float qb[N];
    const int64_t pq = pcf();
#pragma unroll
    for (int w = 0; w < N; ++w) qb[w] = live ? qbar[((int64_t)c * N + w) * B + pq] : 0.f;
    qc_sign_sums<N>(d, qb);
  };

  SV<N> cl[2];
  QcPk<NA> d;
  {
    // this wave's share of lam_0 = sum_c D_c chi_c (to LDS), and lam of its first channel; the second channel's final
    // state and chi_0 stream through (they are reloaded when their turn comes) to keep the live set at two vectors
    QcPk<NA> da_;
    qf2* mine = s_l02 + wv * NA * 64;
    load_chi(cl[0], cha);
    dvec(cha, da_);
    dvec(chb, d);
    if (wv == 0) {            // second channel = the value channel: lam_t = D_t chi_0
      load_chi(cl[1], 0);
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        mine[k * 64 + lane] = qc_pk_fma(qc_dup(da_[k]), cl[0].a[k], qc_dup(d[k]) * cl[1].a[k]);
        cl[1].a[k] *= qc_dup(da_[k]);
      }
    } else {                  // lam_x = D_x chi_0 + 2 D_xx chi_x (the same for y)
      const int64_t pqs = pcs();
#pragma unroll
      for (int k = 0; k < NA; ++k)
        mine[k * 64 + lane] = qc_pk_fma(qc_dup(da_[k]), cl[0].a[k], qc_dup(d[k]) * chi2[qc_chi_pair<A2>(chb, k, pqs)]);
#pragma unroll
      for (int k = 0; k < NA; ++k)
        cl[1].a[k] = qc_pk_fma(qc_dup(2.f * d[k]), cl[0].a[k], qc_dup(da_[k]) * chi2[qc_chi_pair<A2>(0, k, pqs)]);
    }
  }
  __syncthreads();   // shares of lam_0, s_cs, s_acc

  float t_own[N];    // this wave's channels' terms of abar[0] (against their own embedded state)
  float t_a1[N];     // first channel, against phi: abar[cha]'s first term
#pragma unroll
  for (int w = 0; w < N; ++w) t_own[w] = t_a1[w] = 0.f;
  const float* aj = qc_launder(ajets);

#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const int ch = half == 0 ? cha : chb;
    if (half == 1) {
      load_chi(cl[0], chb);
      if (wv == 0) {          // lam_0: the three shares
#pragma unroll
        for (int k = 0; k < NA; ++k)
          cl[1].a[k] = (s_l02[k * 64 + lane] + s_l02[(NA + k) * 64 + lane]) + s_l02[(2 * NA + k) * 64 + lane];
      } else {                // lam_xx = D_xx chi_0
        dvec(chb, d);
        const int64_t pqs = pcs();
#pragma unroll
        for (int k = 0; k < NA; ++k) cl[1].a[k] = qc_dup(d[k]) * chi2[qc_chi_pair<A2>(0, k, pqs)];
      }
    }
    PG::bwd(cl, prog, trig, umat, n_gates, s_acc + wv * n_params, lane, amp >> 1);

    // cotangents of the angle jets in the frame pulled back through the embedding (qc_gates.h)
    float ca[N], sa[N], da[N], dda[N];
    const int64_t pq = pcf();
    load_sincos<N>(ca, sa, aj, B, pq, trig, amp >> 1, s_cs);
    qc_unembed<N>(cl[1], ca, sa);
    const int dirch = ch == 0 ? 0 : (ch <= 3 ? ch : ch - 2);
#pragma unroll
    for (int w = 0; w < N; ++w) {
      da[w] = ch >= 1 ? aj[((int64_t)dirch * N + w) * B + pq] : 0.f;
      dda[w] = ch >= 4 ? aj[((int64_t)ch * N + w) * B + pq] : 0.f;
    }
    float T[N];
    if (ch == 0) {
      qc_pull_ip0<N>(T, cl[1]);
#pragma unroll
      for (int w = 0; w < N; ++w) t_own[w] += T[w];
    } else if (ch <= 3) {
      qc_pull_ip1<N>(T, cl[1], da);
#pragma unroll
      for (int w = 0; w < N; ++w) t_own[w] += T[w];
      qc_pull_ip0<N>(T, cl[1]);
#pragma unroll
      for (int w = 0; w < N; ++w) t_a1[w] = T[w];
    } else {
      qc_pull_ip2<N>(T, cl[1], da, dda);
#pragma unroll
      for (int w = 0; w < N; ++w) t_own[w] += T[w];
      qc_pull_ip1<N>(T, cl[1], da);
#pragma unroll
      for (int w = 0; w < N; ++w) t_a1[w] = fmaf(2.f, T[w], t_a1[w]);   // abar[x] = ip0(mu_x) + 2 ip1(mu_xx)
      qc_pull_ip0<N>(T, cl[1]);
      if (live) {
#pragma unroll
        for (int w = 0; w < N; ++w) abar[((int64_t)ch * N + w) * B + p] = T[w];   // abar[xx] = ip0(mu_xx)
      }
    }
  }
  if (live) {
#pragma unroll
    for (int w = 0; w < N; ++w) abar[((int64_t)cha * N + w) * B + p] = t_a1[w];
  }
  if (wv != 0) {
#pragma unroll
    for (int w = 0; w < N; ++w) s_t0[((wv - 1) * N + w) * 64 + lane] = t_own[w];
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int w = 0; w < N; ++w) {
      const float r = (t_own[w] + s_t0[w * 64 + lane]) + s_t0[(N + w) * 64 + lane];
      if (live) abar[(int64_t)w * B + p] = r;
      if (amp >> 1) {   // folded RX layer: d L / d theta_w = sum over points of d L / d angle_w
        const float tot = qc_wave_sum_to_lane63(live ? r : 0.f);
        if (lane == 63) s_acc[prog[w].slot] += tot;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_params; i += 192)
    part[(row0 + bid) * part_stride + i] = (s_acc[i] + s_acc[n_params + i]) + s_acc[2 * n_params + i];
}

// ---- residual + value tiles in one launch (see qc_mlp.hip): blocks [0, n_val) run the value-channel kernel on
// 6 x 64 boundary / initial points, the rest the six-channel kernel on a 64-point residual tile
template <class PG>
__global__ void __launch_bounds__(384) k_circ_fwd_both(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                       const float* __restrict__ umat, int n_gates,
                                                       const float* __restrict__ ajets, float* __restrict__ qjets, int64_t Br,
                                                       float* __restrict__ chi_store, const float* __restrict__ angles,
                                                       float* __restrict__ expval, int64_t Bv, int amp, int n_val) {
  if ((int)blockIdx.x >= n_val) k_jets_fwd_body<PG>(blockIdx.x - n_val, prog, trig, umat, n_gates, ajets, qjets, Br, chi_store, amp);
  else k_value_fwd_body<PG, 6>(blockIdx.x, prog, trig, umat, n_gates, angles, expval, Bv, amp);
}

template <class PG>
__global__ void __launch_bounds__(384, QC_JB_WAVES) k_circ_bwd_both(
    const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig, const float* __restrict__ umat, int n_gates, int n_params,
    const float* __restrict__ ajets, const float* __restrict__ qbar, float* __restrict__ abar, int64_t row0_r, int64_t Br,
    const float* __restrict__ chi_store, const float* __restrict__ angles, const float* __restrict__ cot,
    float* __restrict__ d_angles, int64_t row0_v, int64_t Bv, float* __restrict__ part, int64_t part_stride, int amp, int n_val) {
  if ((int)blockIdx.x >= n_val)
    k_jets_bwd_body<PG, true>(blockIdx.x - n_val, prog, trig, umat, n_gates, n_params, ajets, qbar, abar, part, part_stride, row0_r, Br,
                              chi_store, amp);
  else
    k_value_bwd_body<PG, 6>(blockIdx.x, prog, trig, umat, n_gates, n_params, angles, cot, d_angles, part, part_stride,
                            row0_v, Bv, amp);
}

// the same stage with 3-wave blocks (k_jets_bwd2_body); angle encoding only (the merged step's precondition)
template <class PG>
__global__ void __launch_bounds__(192, 4) k_circ_bwd_both2(
    const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig, const float* __restrict__ umat, int n_gates, int n_params,
    const float* __restrict__ ajets, const float* __restrict__ qbar, float* __restrict__ abar, int64_t row0_r, int64_t Br,
    const float* __restrict__ chi_store, const float* __restrict__ angles, const float* __restrict__ cot,
    float* __restrict__ d_angles, int64_t row0_v, int64_t Bv, float* __restrict__ part, int64_t part_stride, int amp, int n_val) {
  const int stagger = (amp >> 8) & 0xff;   // (launcher: bits 8.. of the flags word)
  amp &= 0xff;
  if ((int)blockIdx.x >= n_val)
    k_jets_bwd2_body<PG>(blockIdx.x - n_val, prog, trig, umat, n_gates, n_params, ajets, qbar, abar, part, part_stride, row0_r, Br,
                         chi_store, amp, stagger);
  else
    k_value_bwd_body<PG, 3>(blockIdx.x, prog, trig, umat, n_gates, n_params, angles, cot, d_angles, part, part_stride,
                            row0_v, Bv, amp);
}

}  // namespace

// bit 0: amplitude encoding; bit 1: leading RX layer folded into the embedding (QC_NO_ABSORB=1 disables)
static inline int qc_embed_flags(const qc_program* pg) {
  static const bool off = [] { const char* e = getenv("QC_NO_ABSORB"); return e && e[0] == '1'; }();
  const int absorb = (pg->lead_rx && !pg->amplitude && !off) ? 2 : 0;
  return (pg->amplitude ? 1 : 0) | absorb;
}

// ------------------------------------------------------------------ launch helpers (per policy)
struct QcRegLaunchers {
  int (*value_fwd)(const qc_program*, const QcTrig*, const float*, const float*, float*, int64_t, hipStream_t);
  int (*value_bwd)(const qc_program*, const QcTrig*, const float*, const float*, const float*, float*, float*,
                   int64_t, int64_t, int64_t, hipStream_t);
  int (*jets_fwd)(const qc_program*, const QcTrig*, const float*, const float*, float*, int64_t, float*, hipStream_t);
  int (*jets_bwd)(const qc_program*, const QcTrig*, const float*, const float*, const float*, float*, float*,
                  int64_t, int64_t, int64_t, const float*, hipStream_t);
  // residual (six channels, chi_store required) + value tiles in one launch
  int (*circ_fwd_both)(const qc_program*, const QcTrig*, const float*, const float*, float*, int64_t, float*, const float*,
                       float*, int64_t, hipStream_t);
  int (*circ_bwd_both)(const qc_program*, const QcTrig*, const float*, const float*, const float*, float*, int64_t, int64_t,
                       const float*, const float*, const float*, float*, int64_t, int64_t, float*, int64_t, hipStream_t);
};

template <class PG>
struct RegLaunch {
  static int value_fwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* angles,
                       float* expval, int64_t B, hipStream_t st) {
    hipLaunchKernelGGL(k_value_fwd<PG>, dim3(qc_ceil_div(B, 256)), dim3(256), 0, st, pg->d_gates, trig, umat,
                       pg->n_gates, angles, expval, B, qc_embed_flags(pg));
    return QC_OK;
  }
  static int value_bwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* angles,
                       const float* cot, float* d_angles, float* part, int64_t part_stride, int64_t row0,
                       int64_t B, hipStream_t st) {
    const size_t sh = (size_t)4 * pg->n_params * sizeof(float);
    hipLaunchKernelGGL(k_value_bwd<PG>, dim3(qc_ceil_div(B, 256)), dim3(256), sh, st, pg->d_gates, trig, umat,
                       pg->n_gates, pg->n_params, angles, cot, d_angles, part, part_stride, row0, B, qc_embed_flags(pg));
    return QC_OK;
  }
  static int jets_fwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                      float* qjets, int64_t B, float* chi_store, hipStream_t st) {
    hipLaunchKernelGGL(k_jets_fwd<PG>, dim3(qc_ceil_div(B, 64)), dim3(384), 0, st, pg->d_gates, trig, umat,
                       pg->n_gates, ajets, qjets, B, chi_store, qc_embed_flags(pg));
    return QC_OK;
  }
  static int jets_bwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                      const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B,
                      const float* chi_store, hipStream_t st) {
    const size_t sh = ((size_t)6 * (2u << PG::N) * 64 + (size_t)6 * pg->n_params + 2 * PG::N * 64) * sizeof(float);
    const size_t sh_load = ((size_t)6 * 3 * PG::N * 64 + (size_t)6 * pg->n_params + 2 * PG::N * 64) * sizeof(float);
    if (sh > 160 * 1024) return QC_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_jets_bwd<PG, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_jets_bwd<PG, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    if (chi_store != nullptr)
      hipLaunchKernelGGL((k_jets_bwd<PG, true>), dim3(qc_ceil_div(B, 64)), dim3(384), sh_load, st, pg->d_gates, trig, umat,
                         pg->n_gates, pg->n_params, ajets, qbar, abar, part, part_stride, row0, B, chi_store, qc_embed_flags(pg));
    else
      hipLaunchKernelGGL((k_jets_bwd<PG, false>), dim3(qc_ceil_div(B, 64)), dim3(384), sh, st, pg->d_gates, trig, umat,
                         pg->n_gates, pg->n_params, ajets, qbar, abar, part, part_stride, row0, B, chi_store, qc_embed_flags(pg));
    return QC_OK;
  }
  static int circ_fwd_both(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                           int64_t Br, float* chi_store, const float* angles, float* expval, int64_t Bv, hipStream_t st) {
    const int nr = qc_ceil_div(Br, 64), nv = qc_ceil_div(Bv, 384);
    hipLaunchKernelGGL(k_circ_fwd_both<PG>, dim3(nr + nv), dim3(384), 0, st, pg->d_gates, trig, umat, pg->n_gates, ajets,
                       qjets, Br, chi_store, angles, expval, Bv, qc_embed_flags(pg), nv);
    return QC_OK;
  }
  static int circ_bwd_both(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                           const float* qbar, float* abar, int64_t row0_r, int64_t Br, const float* chi_store,
                           const float* angles, const float* cot, float* d_angles, int64_t row0_v, int64_t Bv, float* part,
                           int64_t part_stride, hipStream_t st) {
    static const bool six = [] { const char* e = getenv("QC_BWD6"); return e && e[0] == '1'; }();   // A/B: the six-wave form
    static const int stagger = [] { const char* e = getenv("QC_BWD2_STAGGER"); return e ? atoi(e) & 0xff : 0; }();
    if (!six && !pg->amplitude) {
      const int nr = qc_ceil_div(Br, 64), nv = qc_ceil_div(Bv, 192);
      // QC_BWD2_LDS_KB=k (diagnostic, A/B): k KB of unused dynamic LDS per block lower the blocks per CU, so that the grid
      // runs in more than one round and the load phases of later blocks overlap the sweeps of earlier ones
      static const int same_tile = [] { const char* e = getenv("QC_BWD2_SAMETILE"); return (e && e[0] == '1') ? 0x80 : 0; }();
      static const size_t pad = [] { const char* e = getenv("QC_BWD2_LDS_KB"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();
      const size_t sh = ((size_t)3 * (2u << PG::N) * 64 + (size_t)4 * PG::N * 64 + (size_t)3 * pg->n_params) * sizeof(float) + pad;
      if (pad) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_circ_bwd_both2<PG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k_circ_bwd_both2<PG>, dim3(nr + nv), dim3(192), sh, st, pg->d_gates, trig, umat, pg->n_gates,
                         pg->n_params, ajets, qbar, abar, row0_r, Br, chi_store, angles, cot, d_angles, row0_v, Bv, part,
                         part_stride, qc_embed_flags(pg) | (stagger << 8) | same_tile, nv);
      return QC_OK;
    }
    const int nr = qc_ceil_div(Br, 64), nv = qc_ceil_div(Bv, 384);
    const size_t sh = ((size_t)6 * 3 * PG::N * 64 + (size_t)6 * pg->n_params + 2 * PG::N * 64) * sizeof(float);   // >= the value blocks' 6 rows
    hipLaunchKernelGGL(k_circ_bwd_both<PG>, dim3(nr + nv), dim3(384), sh, st, pg->d_gates, trig, umat, pg->n_gates,
                       pg->n_params, ajets, qbar, abar, row0_r, Br, chi_store, angles, cot, d_angles, row0_v, Bv, part,
                       part_stride, qc_embed_flags(pg), nv);
    return QC_OK;
  }
  static constexpr QcRegLaunchers table() {
    return {&value_fwd, &value_bwd, &jets_fwd, &jets_bwd, &circ_fwd_both, &circ_bwd_both};
  }
};
