// Variational-circuit kernels, "hbm" family (9 <= n <= 20 qubits), COMPILE-TIME stage programs.
//
// Same algorithm, memory layout and side buffers as the run-time plan interpreter of qc_circuit_hbm2.hip (read its
// header first); replaces the same reference code (nn/DVQuantumLayer.py:151-154,176-214 behind DVQuantumLayer.forward,
// and the torch double-backward through it, nn/pde.py:59-70 + loss.backward()).  Here the execution plan of ONE gate
// program (qc_hbm2_plan.h, produced at build time by gen/h2_plan_tool from the gate list circuits.py lowers) is a
// constexpr record `PL`, and a kernel is instantiated per (plan, stage, channel count, direction):
//   * gate kind, register bits, trig-table offsets, LDS offsets and the local -> global index deposits are constants:
//     a round is straight-line packed-fp32 code on 2^RB register pairs, no plan records, no `switch`, no scalar
//     bookkeeping per gate;
//   * parameter-gradient inner products Im<lam|G|chi> accumulate in one register per gate ACROSS the channel loop and
//     are wave-reduced once per block (the interpreter reduces per gate and channel);
//   * the first adjoint stage un-embeds the register bits of its last round in that round (one LDS round trip fewer);
//   * <Z> sums: one wave reduction for the total serves every index bit that is constant across the wave.
// The host (qc_circuit_hbm2.hip, h2_group) launches these through H2sLaunchers when the caller's program matches a
// generated plan exactly (qc_h2s_match); every other program keeps the interpreter.
#pragma once
#include "qc_h2_shared.h"

#include <type_traits>

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void h2s_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    h2s_for<I + 1, N>(f);
  }
}

// OR over the source bits k < NB of ((v >> k) & 1) << M::at(k); runs of consecutive (k, at(k)) move as one field
template <class M, int NB>
constexpr int h2s_run_len(int k) {
  int len = 1;
  while (k + len < NB && M::at(k + len) == M::at(k) + len) ++len;
  return len;
}
template <class M, int NB, int K = 0>
__device__ __forceinline__ int h2s_deposit(int v) {
  if constexpr (K >= NB) {
    return 0;
  } else {
    constexpr int len = h2s_run_len<M, NB>(K);
    const int f = (K == 0 && M::at(0) == 0) ? (v & ((1 << len) - 1)) : (((v >> K) & ((1 << len) - 1)) << M::at(K));
    return f | h2s_deposit<M, NB, K + len>(v);
  }
}

// ---- compile-time geometry of a stage and of its rounds
template <class PL, int S>
struct H2sStage {
  static constexpr H2Stage sd = PL::stages[S];
  static constexpr int RB = PL::RB, R = 1 << RB, nloc = sd.nloc, TS = 1 << nloc, NT = TS >> RB, LBITS = nloc - RB;
  static constexpr int ntau = 1 << sd.ngb;
  // linear mapping: thread bit k = local position k, register bit j = local position LBITS + j
  struct LinG { static constexpr int at(int k) { return PL::stages[S].lb[k]; } };
  static constexpr int lin_dep(int q) {
    int d = 0;
    for (int j = 0; j < RB; ++j) d |= ((q >> j) & 1) << sd.lb[LBITS + j];
    return d;
  }
  static constexpr int lin_sw(int q) { return h2_swz<RB>(q << LBITS); }
  struct TauG { static constexpr int at(int k) { return PL::stages[S].gb[k]; } };
  // local positions 0..5 / 6..nloc-1 -> global bits (the two 64-entry factor tables of the generated state)
  struct LoG { static constexpr int at(int k) { return PL::stages[S].lb[k]; } };
  struct HiG { static constexpr int at(int k) { return PL::stages[S].lb[6 + k]; } };
};

template <class PL, int S, int RI>
struct H2sRound {
  using ST = H2sStage<PL, S>;
  static constexpr H2Round rd = PL::rounds[PL::stages[S].r0 + RI];
  static constexpr int RB = PL::RB;
  static constexpr bool is_reg(int p) {
    for (int j = 0; j < RB; ++j)
      if (PL::rounds[PL::stages[S].r0 + RI].rb[j] == p) return true;
    return false;
  }
  static constexpr int lpos(int k) {   // k-th local position outside the register set
    int seen = 0;
    for (int p = 0; p < ST::nloc; ++p)
      if (!is_reg(p)) {
        if (seen == k) return p;
        ++seen;
      }
    return 0;
  }
  struct LaneL { static constexpr int at(int k) { return lpos(k); } };
  struct LaneG { static constexpr int at(int k) { return PL::stages[S].lb[lpos(k)]; } };
  static constexpr int roff(int q) {
    int o = 0;
    for (int j = 0; j < RB; ++j) o |= ((q >> j) & 1) << rd.rb[j];
    return o;
  }
  static constexpr int sr(int q) { return h2_swz<RB>(roff(q)); }
  static constexpr int dr(int q) {
    int d = 0;
    for (int j = 0; j < RB; ++j) d |= ((q >> j) & 1) << ST::sd.lb[rd.rb[j]];
    return d;
  }
  static constexpr bool direct = rd.kind == H2_ROUND_GATES && rd.rb[0] >= 4;   // may talk to HBM in its own mapping
};

// ---- runs of RZ gates on register bits inside a round: ONE multiply by a 2^RB-entry phase record
// A round often carries RZ(b) for each of its register bits back to back (after the RX layer of an ansatz).  They are
// diagonal and commute: their product is one unit phase per register combination q, the same in every lane - a record
// of 2^RB (re, im) pairs built per parameter update by k_h2s_round_phases and read here as scalar operands.  In the
// adjoint sweep t_q = Im(conj(lam_q) chi_q) is invariant under the run, so every gate's gradient is a signed sum of the
// same t (as for the big phase tables).  Runs are numbered in plan order; the host mirrors this enumeration
// (h2s_enumerate_rz_runs in qc_circuit_hbm2.hip).
template <class PL>
struct H2sRz {
  static constexpr bool rz1(int g) { return PL::gates[g].kind == H2_K_REG1 && PL::gates[g].op == QC_RZ; }
  static constexpr int round_of(int g) {
    for (int r = 0; r < PL::NROUNDS; ++r)
      if (PL::rounds[r].kind == H2_ROUND_GATES && g >= PL::rounds[r].g0 && g < PL::rounds[r].g0 + PL::rounds[r].ng) return r;
    return 0;
  }
  static constexpr int begin_of(int g) {
    const int lo = PL::rounds[round_of(g)].g0;
    int b = g;
    while (b - 1 >= lo && rz1(b - 1)) --b;
    return b;
  }
  static constexpr int len_of(int b) {
    const int hi = PL::rounds[round_of(b)].g0 + PL::rounds[round_of(b)].ng;
    int e = b;
    while (e < hi && rz1(e)) ++e;
    return e - b;
  }
  static constexpr bool fused(int g) { return rz1(g) && len_of(begin_of(g)) >= 2; }
  static constexpr int ordinal(int b) {   // number of fused runs that begin before gate b
    int k = 0;
    for (int g = 0; g < b;) {
      if (rz1(g)) {
        const int len = len_of(g);
        if (len >= 2) ++k;
        g += len;
      } else {
        ++g;
      }
    }
    return k;
  }
};

// multiply (x, y) by (-i)^k, k a compile-time constant
template <int K>
__device__ __forceinline__ qf2 h2s_rot(const qf2 a) {
  if constexpr ((K & 3) == 0) return a;
  else if constexpr ((K & 3) == 1) return (qf2){a.y, -a.x};
  else if constexpr ((K & 3) == 2) return (qf2){-a.x, -a.y};
  else return (qf2){-a.y, a.x};
}

// ---- one gate of a round, everything about it a compile-time constant
template <class PL, int GI, int RB, int KV, bool BWD, int NPA, class DR>
__device__ __forceinline__ void h2s_gate(SV<RB> (&v)[KV], float (&gacc)[NPA], const H2Args& A, const int alane, DR dr) {
  constexpr H2Gate hg = PL::gates[GI];
  constexpr int R = 1 << RB;
  float c_ = 1.f, s_ = 0.f;
  if constexpr (hg.op != QC_U4 && hg.slot >= 0) {
    const auto* tr = h2_const(A.trig + hg.gi);
    c_ = tr->c;
    s_ = tr->s;
  }
  const float sg = BWD ? -s_ : s_;
  if constexpr (hg.kind == H2_K_REG1) {
    constexpr int TQ = hg.tq;
    if constexpr (BWD && hg.pidx >= 0) {
      if constexpr (hg.op == QC_RX) gacc[hg.pidx] += ip_x<RB, TQ>(v[KV - 1], v[0]);
      else if constexpr (hg.op == QC_RY) gacc[hg.pidx] += ip_y<RB, TQ>(v[KV - 1], v[0]);
      else if constexpr (hg.op == QC_RZ) gacc[hg.pidx] += ip_z<RB, TQ>(v[KV - 1], v[0]);
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if constexpr (hg.op == QC_RX) g_rx<RB, TQ>(v[k], c_, sg);
      else if constexpr (hg.op == QC_RY) g_ry<RB, TQ>(v[k], c_, sg);
      else if constexpr (hg.op == QC_RZ) g_rz<RB, TQ>(v[k], c_, sg);
      else if constexpr (hg.op == QC_H) g_h<RB, TQ>(v[k]);
    }
  } else if constexpr (hg.kind == H2_K_REG2) {
    constexpr int TQ = hg.tq, CQ = hg.cq;
    if constexpr (BWD && hg.pidx >= 0) {
      if constexpr (hg.op == QC_CRX) gacc[hg.pidx] += ip_cx<RB, CQ, TQ>(v[KV - 1], v[0]);
      else if constexpr (hg.op == QC_CRZ) gacc[hg.pidx] += ip_cz<RB, CQ, TQ>(v[KV - 1], v[0]);
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
      if constexpr (hg.op == QC_CNOT) g_cnot<RB, CQ, TQ>(v[k]);
      else if constexpr (hg.op == QC_CRX) g_crx<RB, CQ, TQ>(v[k], c_, sg);
      else if constexpr (hg.op == QC_CRZ) g_crz<RB, CQ, TQ>(v[k], c_, sg);
    }
  } else if constexpr (hg.kind == H2_K_PRED) {
    // control = a lane or non-local bit: the target update runs in every lane with the identity where the control is 0
    constexpr int TQ = hg.tq;
    const bool on = (alane >> hg.cbit) & 1;
    if constexpr (hg.op == QC_CNOT) {
#pragma unroll
      for (int k = 0; k < KV; ++k) {
#pragma unroll
        for (int h = 0; h < (R >> 1); ++h) {
          const int i0 = qc_ins0(h, TQ), i1 = i0 | (1 << TQ);
          const qf2 a = v[k].a[i0], b = v[k].a[i1];
          v[k].a[i0] = on ? b : a;
          v[k].a[i1] = on ? a : b;
        }
      }
    } else {   // CRX
      if constexpr (BWD && hg.pidx >= 0) {
        const float g = ip_x<RB, TQ>(v[KV - 1], v[0]);
        gacc[hg.pidx] += on ? g : 0.f;
      }
      const float cl = on ? c_ : 1.f, sl = on ? sg : 0.f;
#pragma unroll
      for (int k = 0; k < KV; ++k) g_rx<RB, TQ>(v[k], cl, sl);
    }
  } else if constexpr (hg.kind == H2_K_PHASE) {
    float grad = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int a = alane | dr(q);
      const bool on = hg.cbit < 0 || ((a >> (hg.cbit < 0 ? 0 : hg.cbit)) & 1);
      const bool hi = (a >> hg.tbit) & 1;
      if constexpr (BWD) {
        const qf2 m = v[KV - 1].a[q] * qc_swp(v[0].a[q]);
        const float tv = m.x - m.y;
        grad += on ? (hi ? -tv : tv) : 0.f;
      }
      const float sq = on ? (hi ? sg : -sg) : 0.f, cq = on ? c_ : 1.f;   // multiply by cq + i sq
#pragma unroll
      for (int k = 0; k < KV; ++k) v[k].a[q] = qc_cmul(cq, sq, v[k].a[q]);
    }
    if constexpr (BWD && hg.pidx >= 0) gacc[hg.pidx] += grad;
  } else if constexpr (hg.kind == H2_K_U4) {
    const auto* um = h2_const(A.umat + (hg.slot * 2 + (BWD ? 1 : 0)) * 32);
#pragma unroll
    for (int k = 0; k < KV; ++k) g_u4<RB, hg.tq, hg.cq>(v[k], um);
  }
  // The gradient accumulators are read once, after the channel loop.  Left free, the optimiser SINKS every inner product
  // Im<lam|G|chi> down to that use (or to the end of the loop body, behind its inner branches) and keeps every
  // intermediate version of the 2^RB amplitude pairs alive for it: hundreds of registers, kilobytes of scratch per lane.
  // An empty asm that "modifies" the accumulator pins the sum to the gate it belongs to.
  if constexpr (BWD && hg.pidx >= 0) asm volatile("" : "+v"(gacc[hg.pidx]));
}

// the fused RZ run that begins at gate B0 (see H2sRz)
template <class PL, int B0, int RB, int KV, bool BWD, int NPA>
__device__ __forceinline__ void h2s_rz_run(SV<RB> (&v)[KV], float (&gacc)[NPA], const H2Args& A) {
  constexpr int R = 1 << RB, LEN = H2sRz<PL>::len_of(B0);
  const auto* P = h2_const(A.rph + H2sRz<PL>::ordinal(B0) * R);
  if constexpr (BWD) {
    float t[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const qf2 m = v[KV - 1].a[q] * qc_swp(v[0].a[q]);
      t[q] = m.x - m.y;
    }
    h2s_for<0, LEN>([&](auto J) {
      constexpr H2Gate hg = PL::gates[B0 + J];
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < R; ++q) sum += ((q >> hg.tq) & 1) ? -t[q] : t[q];
      gacc[hg.pidx] += sum;
      asm volatile("" : "+v"(gacc[hg.pidx]));   // pinned to this gate (see h2s_gate)
    });
  }
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const float pr = P[q].re, pi = BWD ? -P[q].im : P[q].im;
#pragma unroll
    for (int k = 0; k < KV; ++k) v[k].a[q] = qc_cmul(pr, pi, v[k].a[q]);
  }
}

// phase records of the fused RZ runs: out[run][q] = prod_j (c_j -+ i s_j) by bit tq_j of q.  desc per run:
// {len, gi_0, tq_0, gi_1, tq_1, ...} padded to 1 + 2 * 8 ints
constexpr int H2S_PH_DESC = 17;
__global__ void k_h2s_round_phases(const int* __restrict__ desc, int n_runs, int R, const QcTrig* __restrict__ trig,
                                   Cplx* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_runs * R) return;
  const int run = i / R, q = i % R;
  const int* d = desc + run * H2S_PH_DESC;
  double zr = 1.0, zi = 0.0;
  for (int j = 0; j < d[0]; ++j) {
    const QcTrig tr = trig[d[1 + 2 * j]];
    const double c = tr.c, sg = ((q >> d[2 + 2 * j]) & 1) ? (double)tr.s : -(double)tr.s;   // bit 0: c - i s, bit 1: c + i s
    const double nr = zr * c - zi * sg, ni = zr * sg + zi * c;
    zr = nr;
    zi = ni;
  }
  out[i] = {(float)zr, (float)zi};
}

// MODE 0: forward, MODE 1: backward (chi and lam).  Block = (point, tile of 2^nloc amplitudes), NT = 2^(nloc - RB)
// threads; mappings (LINEAR / ROUND) as in the interpreter kernel.
template <class PL, int S, int NCH, int MODE>
__global__ void __launch_bounds__((H2sStage<PL, S>::NT), (H2sStage<PL, S>::NT >= 512 ? 4 : (MODE == 1 ? 2 : 3)))
    k_h2s_stage(const H2Args A) {
  using ST = H2sStage<PL, S>;
  constexpr H2Stage sd = ST::sd;
  constexpr bool BWD = MODE == 1, FIRST = S == 0, LAST = S == PL::NSTAGES - 1;
  constexpr int RB = ST::RB, R = ST::R, n = PL::N, nloc = ST::nloc, TS = ST::TS, NT = ST::NT, LBITS = ST::LBITS;
  constexpr int NW = (NT + 63) / 64, ntau = ST::ntau, NR = sd.nr, NP = sd.np, NTAB = sd.ntab;
  constexpr int NPA = NP > 0 ? NP : 1;
  constexpr bool TABC = BWD && NTAB > 0;
  constexpr int KV = BWD ? 2 : 1;
  constexpr int64_t N = (int64_t)1 << n;
  static_assert(nloc >= 9 && nloc % RB == 0 && NTAB <= 2, "tile geometry");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  Cplx* t0 = reinterpret_cast<Cplx*>(smem_raw);          // chi tile
  Cplx* t1 = t0 + (BWD ? TS : 0);                        // lam tile (backward)
  __shared__ float s_tabA[64][3], s_tabB[64][3];         // series factors of the tile (forward generation)
  __shared__ float s_D[6][128];                          // backward: D_c = g + A[l & 63] + B[l >> 6]
  __shared__ float s_Dg[6];
  __shared__ float s_g[8][NPA];                          // per-wave sums of the in-round gate gradients
  __shared__ float s_red[8][H2_XW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tau = blockIdx.x % ntau;
  const int64_t pt = blockIdx.x / ntau;                  // point index within this launch
  const int64_t p = A.p_first + pt;
  const bool live = p < A.B;
  const int64_t tile64 = pt >> 6;
  const int t = (int)(pt & 63);
  const int64_t blk = (int64_t)pt * ntau + tau;          // row of this block in the partial buffers
  const int64_t nblk = A.pt_stride * ntau;

  if (!live) {   // ragged tail: the folds read every row
    if constexpr (BWD) {
      for (int i = tid; i < NP; i += NT) A.gpart[(size_t)i * nblk + blk] = 0.f;
      for (int k = 0; k < NTAB; ++k)
        for (int i = tid; i < A.nc; i += NT) A.dpart[((size_t)k * nblk + blk) * A.nc + i] = 0.f;
    }
    return;
  }
  const int abase = h2s_deposit<typename ST::TauG, sd.ngb>(tau);   // global index of the tile (n <= 20: 32 bits)
  // The thread index as an opaque value: addresses derived from it are recomputed where they are used instead of
  // being hoisted out of the channel loop and held in registers for the whole kernel.
  auto ftid = [&]() {
    int v = tid;
    asm volatile("" : "+v"(v));
    return v;
  };
  const float* wdp = A.wd + (size_t)pt * n * 8;
  Cplx* slot = A.store + (size_t)tile64 * A.slot_elems;
  auto chi_of = [&](int c) { return slot + ((size_t)c * 64 + t) * N; };
  auto lam_of = [&](int c) { return slot + ((size_t)(NCH + c) * 64 + t) * N; };
  // Final states of a multi-stage plan live in the LAM half of the slot (free during the forward pass): the chi half keeps
  // the state before the last stage, which is what the adjoint of the stage before it reads - the last adjoint stage
  // overwrites the final states with its cotangents in place and does not write an un-computed copy of chi back
  // (one state transfer of eight less per channel).
  constexpr bool FINL = LAST && !FIRST;
  auto fin_of = [&](int c) { return FINL ? lam_of(c) : chi_of(c); };

  // registers that live across the channel loop
  qf2 x0[!BWD && LAST ? R : 1];       // forward, last stage: final value-channel tile
  qf2 l0acc[BWD && LAST ? R : 1];     // backward, last stage: lam_0 = sum_c D_c chi_c
  float tacc0[TABC ? R : 1], tacc1[TABC && NTAB > 1 ? R : 1];   // backward: t = sum_c Im(conj lam chi) per table
  float gacc[NPA];                    // backward: Im<lam|G|chi> per in-round parametric gate, summed over the channels
  if constexpr (BWD && LAST) {
#pragma unroll
    for (int q = 0; q < R; ++q) l0acc[q] = (qf2){0.f, 0.f};
  }
  if constexpr (TABC) {
#pragma unroll
    for (int q = 0; q < R; ++q) tacc0[q] = 0.f;
    if constexpr (NTAB > 1) {
#pragma unroll
      for (int q = 0; q < R; ++q) tacc1[q] = 0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < NPA; ++i) gacc[i] = 0.f;

  if constexpr (BWD && LAST) {
    // D_c[k] = sum_w (+-) qbar[c][w][p] by index bit n-1-w: constant part (non-local bits) + two 64-entry tables
    for (int i = tid; i < NCH * 128; i += NT) {
      const int c = i >> 7, e = i & 127, half = e >> 6, vv = e & 63;
      float s = 0.f;
      h2s_for<0, nloc>([&](auto J) {
        constexpr int j = J;
        constexpr int j0 = j < 6 ? 0 : 6;
        const float qb = A.qbar[((int64_t)c * n + (n - 1 - sd.lb[j])) * A.B + p];
        if ((j < 6) == (half == 0)) s += ((vv >> (j - j0)) & 1) ? -qb : qb;
      });
      s_D[c][e] = s;
    }
    if (tid < NCH) {
      float s = 0.f;
      h2s_for<0, sd.ngb>([&](auto J) {
        constexpr int j = J;
        const float qb = A.qbar[((int64_t)tid * n + (n - 1 - sd.gb[j])) * A.B + p];
        s += ((tau >> j) & 1) ? -qb : qb;
      });
      s_Dg[tid] = s;
    }
    H2_SYNC();
  }
  auto Dval = [&](int c, int l) { return s_Dg[c] + s_D[c][l & 63] + s_D[c][64 + (l >> 6)]; };

  // which rounds talk to HBM directly
  constexpr bool gen = !BWD && FIRST;
  constexpr int RF = BWD ? NR - 1 : 0, RL = BWD ? 0 : NR - 1;   // first / last executed round
  constexpr bool has_rounds = NR > 0;
  constexpr bool din = has_rounds && PL::rounds[sd.r0 + (has_rounds ? RF : 0)].kind == H2_ROUND_GATES &&
                       (gen || PL::rounds[sd.r0 + (has_rounds ? RF : 0)].rb[0] >= 4);
  constexpr bool dout = has_rounds && PL::rounds[sd.r0 + (has_rounds ? RL : 0)].kind == H2_ROUND_GATES &&
                        PL::rounds[sd.r0 + (has_rounds ? RL : 0)].rb[0] >= 4 && !(BWD && FIRST);
  constexpr bool uses_lds = !(din && dout && NR == 1);

  // forward generation: series factor tables of channel c (non-local bits folded into table A)
  auto gen_tables = [&](int c) {
    const int ord = c == 0 ? 0 : (c <= 3 ? 1 : 2);
    const int dsel = c == 0 ? 0 : (c <= 3 ? c - 1 : c - 3);      // direction: t, x, y
    const int ddsel = c >= 4 ? c - 4 : 0;
    auto step = [&](float& P0, float& P1, float& P2, int bit_pos, int bitval) {
      const float* w8 = wdp + (size_t)(n - 1 - bit_pos) * 8;
      const float cw = w8[0], sw = w8[1];
      const float da = ord >= 1 ? w8[2 + dsel] : 0.f, dda = ord >= 2 ? w8[5 + ddsel] : 0.f;
      const float w0 = bitval ? sw : cw, e = bitval ? cw : -sw;
      const float w1 = 0.5f * da * e, w2 = 0.5f * dda * e - 0.25f * da * da * w0;
      const float p0 = P0, p1 = P1, p2 = P2;
      P0 = p0 * w0;
      P1 = p0 * w1 + p1 * w0;
      P2 = p0 * w2 + 2.f * p1 * w1 + p2 * w0;
    };
    for (int e_ = tid; e_ < 128; e_ += NT) {
      const int half = e_ >> 6, vv = e_ & 63;
      float P0 = 1.f, P1 = 0.f, P2 = 0.f;
      if (half == 0) {
        h2s_for<0, sd.ngb>([&](auto J) { step(P0, P1, P2, sd.gb[J], (tau >> J) & 1); });
        h2s_for<0, 6>([&](auto J) { step(P0, P1, P2, sd.lb[J], (vv >> J) & 1); });
        s_tabA[vv][0] = P0; s_tabA[vv][1] = P1; s_tabA[vv][2] = P2;
      } else {
        h2s_for<6, nloc>([&](auto J) { step(P0, P1, P2, sd.lb[J], (vv >> (J - 6)) & 1); });
        s_tabB[vv][0] = P0; s_tabB[vv][1] = P1; s_tabB[vv][2] = P2;
      }
    }
  };
  // magnitude of the generated amplitude at local index l (real; the phase (-i)^popcount is applied by the caller)
  auto gen_mag = [&](int c, int l) {
    const int ord = c == 0 ? 0 : (c <= 3 ? 1 : 2);
    const float a0 = s_tabA[l & 63][0], a1 = s_tabA[l & 63][1], a2 = s_tabA[l & 63][2];
    const float b0 = s_tabB[l >> 6][0], b1 = s_tabB[l >> 6][1], b2 = s_tabB[l >> 6][2];
    return ord == 0 ? a0 * b0 : (ord == 1 ? a1 * b0 + a0 * b1 : a0 * b2 + 2.f * a1 * b1 + a2 * b0);
  };
  auto base_phase = [&](int a) {   // (-i)^popcount(a)
    const int ph = __popc((unsigned)a) & 3;
    return (qf2){ph == 0 ? 1.f : (ph == 2 ? -1.f : 0.f), ph == 1 ? -1.f : (ph == 3 ? 1.f : 0.f)};
  };
  // backward, last stage: cotangent of channel c's final state at local index l from (chi_c, chi_0) [DESIGN.md §3]:
  // lam_0 = sum_c D_c chi_c, lam_t = D_t chi_0, lam_x = D_x chi_0 + 2 D_xx chi_x, lam_xx = D_xx chi_0 (same for y)
  auto build_lam = [&](int c, int l, qf2 x, qf2 xv, qf2& acc) {
    qf2 y;
    if (c == 0) {
      const float d = Dval(0, l);
      y = qc_pk_fma(qc_dup(d), x, acc);
    } else {
      const float d = Dval(c, l);
      y = qc_dup(d) * xv;
      if (c == 2 || c == 3) y = qc_pk_fma(qc_dup(2.f * Dval(c + 2, l)), x, y);
      acc = qc_pk_fma(qc_dup(d), x, acc);
    }
    return y;
  };
  // forward, last stage: <Z> sums of this tile against the value channel's final tile, per index bit.  fin[q] sits at
  // local index lb_ | (q's bits at the local positions RBP::at(j)); LK::at(k) = local position of thread bit k.
  auto expval = [&](int c, const qf2 (&fin)[R], auto rbp_tag, auto lk_tag) {
    using RBP = decltype(rbp_tag);
    using LK = decltype(lk_tag);
    float tot = 0.f, sq = 0.f, qs[RB], qq[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) qs[j] = qq[j] = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      constexpr int XQ = (!BWD && LAST) ? R : 1;
      const int xq = q < XQ ? q : 0;
      if (c == 0) x0[xq] = fin[q];
      const qf2 pr = (c == 0 ? fin[q] : x0[xq]) * fin[q], p2 = fin[q] * fin[q];
      const float w = c == 0 ? pr.x + pr.y : 2.f * (pr.x + pr.y);
      const float w2 = 2.f * (p2.x + p2.y);
      tot += w;
      sq += w2;
#pragma unroll
      for (int j = 0; j < RB; ++j)
        if ((q >> j) & 1) {
          qs[j] += w;
          qq[j] += w2;
        }
    }
    for (int pass = 0; pass < ((c == 2 || c == 3) ? 2 : 1); ++pass) {
      const float T_ = pass ? sq : tot;
      const float Tw = qc_wave_sum_to_lane63(T_);   // serves every bit that is constant across the wave
      h2s_for<0, n>([&](auto Bt) {
        constexpr int b = Bt;                        // global index bit
        constexpr int wb = sd.where[b];              // local position, or -(1 + j) for the non-local bit gb[j]
        float wv;
        if constexpr (wb < 0) {
          wv = ((tau >> (-wb - 1)) & 1) ? -Tw : Tw;
        } else {
          constexpr int jr = [] { for (int j = 0; j < RB; ++j) if (RBP::at(j) == wb) return j; return -1; }();
          if constexpr (jr >= 0) {
            wv = qc_wave_sum_to_lane63(T_ - 2.f * (pass ? qq[jr] : qs[jr]));
          } else {
            constexpr int k = [] { for (int kk = 0; kk < LBITS; ++kk) if (LK::at(kk) == wb) return kk; return -1; }();
            static_assert(k >= 0, "index bit neither register nor thread bit");
            if constexpr (k >= 6) wv = ((wave >> (k - 6)) & 1) ? -Tw : Tw;
            else wv = qc_wave_sum_to_lane63(((lane >> k) & 1) ? -T_ : T_);
          }
        }
        if (lane == 63) s_red[wave][b] = wv;
      });
      H2_SYNC();
      if (tid < n) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += s_red[w][tid];
        const int ch8 = pass ? 6 + (c - 2) : c;
        A.xpart[(((size_t)ch8 * A.pt_stride + pt) * ntau + tau) * H2_XW + tid] = s;
      }
      H2_SYNC();
    }
  };
  // backward, last stage: every cotangent lam_c needs the value channel's final amplitudes chi_0 at the thread's own
  // indices (the same in every channel): read them once, keep them in registers (the channel loop would re-read
  // the tile five times from HBM)
  qf2 xz[BWD && LAST ? R : 1];
  if constexpr (BWD && LAST) {
    const Cplx* g0 = fin_of(0);
    if constexpr (din) {
      using RD0 = H2sRound<PL, S, RF>;
      const int al0 = abase | h2s_deposit<typename RD0::LaneG, LBITS>(tid);
      h2s_for<0, R>([&](auto Q) { xz[Q] = *reinterpret_cast<const qf2*>(g0 + RD0::dr(Q) + al0); });
    } else {
      const int td0 = abase | h2s_deposit<typename ST::LinG, LBITS>(tid);
      h2s_for<0, R>([&](auto Q) { xz[Q] = *reinterpret_cast<const qf2*>(g0 + (td0 | ST::lin_dep(Q))); });
    }
  }
  // backward, last stage, direct loads: the NEXT channel's final amplitudes are requested while this channel's sweep runs
  // (two waves per SIMD do not cover an HBM round trip on their own: load, sweep, store, load ... left the memory
  // system idle during every sweep)
  constexpr bool PREF = BWD && LAST && din && NCH > 1;
  // the other adjoint stages read (chi, lam) of a channel: the same prefetch, twice the registers
  constexpr bool PREF2 = BWD && !LAST && din && NCH > 1;
  qf2 nxt[PREF || PREF2 ? R : 1], nxl[PREF2 ? R : 1];
  if constexpr (PREF || PREF2) {
    using RD0 = H2sRound<PL, S, RF>;
    const int al0 = abase | h2s_deposit<typename RD0::LaneG, LBITS>(tid);
    const Cplx* g1 = PREF ? fin_of(1) : chi_of(1);   // the sweep's first channel
    h2s_for<0, R>([&](auto Q) { nxt[Q] = *reinterpret_cast<const qf2*>(g1 + RD0::dr(Q) + al0); });
    if constexpr (PREF2) {
      const Cplx* l1 = lam_of(1);
      h2s_for<0, R>([&](auto Q) { nxl[Q] = *reinterpret_cast<const qf2*>(l1 + RD0::dr(Q) + al0); });
    }
  }
  struct LinRBP { static constexpr int at(int j) { return H2sStage<PL, S>::LBITS + j; } };
  struct LinLK { static constexpr int at(int k) { return k; } };

  // ------------------------------------------------------------------ channel loop
  for (int ci = 0; ci < NCH; ++ci) {
    // backward: the value channel last (its cotangent needs every other channel's final state)
    const int c = BWD ? (ci + 1 < NCH ? ci + 1 : 0) : ci;
    if constexpr (gen) {
      gen_tables(c);
      H2_SYNC();
    }
    // ---------------- load phase through LDS (linear mapping) unless the first round reads HBM itself
    if constexpr (!din) {
      const int tg = ftid();
      const int tsw = h2_swz<RB>(tg), tdp = abase | h2s_deposit<typename ST::LinG, LBITS>(tg);
      const Cplx* g = (BWD && LAST) ? fin_of(c) : chi_of(c);
      if constexpr (!BWD) {
        const qf2 bp = gen ? base_phase(tdp) : (qf2){0.f, 0.f};
        h2s_for<0, R>([&](auto Q) {
          constexpr int q = Q;
          qf2 x;
          if constexpr (gen) x = qc_dup(gen_mag(c, tg | (q << LBITS))) * h2s_rot<qc_popc(ST::lin_dep(q))>(bp);
          else x = *reinterpret_cast<const qf2*>(g + (tdp | ST::lin_dep(q)));
          *reinterpret_cast<qf2*>(t0 + (tsw ^ ST::lin_sw(q))) = x;
        });
      } else {
        const Cplx* gl = lam_of(c);
        h2s_for<0, R>([&](auto Q) {
          constexpr int q = Q;
          const int a = tdp | ST::lin_dep(q);
          qf2 x, y;
          if constexpr (LAST) {
            x = c == 0 ? xz[q] : *reinterpret_cast<const qf2*>(g + a);
            y = build_lam(c, tg | (q << LBITS), x, xz[q], l0acc[q]);
          } else {
            x = *reinterpret_cast<const qf2*>(g + a);
            y = *reinterpret_cast<const qf2*>(gl + a);
          }
          *reinterpret_cast<qf2*>(t0 + (tsw ^ ST::lin_sw(q))) = x;
          *reinterpret_cast<qf2*>(t1 + (tsw ^ ST::lin_sw(q))) = y;
        });
      }
      H2_SYNC();
    }

    // ---------------- rounds
    h2s_for<0, NR>([&](auto RI_) {
      constexpr int ri = RI_;
      constexpr int rix = BWD ? NR - 1 - ri : ri;
      using RD = H2sRound<PL, S, rix>;
      constexpr H2Round rd = RD::rd;
      if constexpr (rd.kind == H2_ROUND_TABLE) {   // a table without a gate round to ride on: element-wise, linear mapping
        const Cplx* tab = A.tabs + (size_t)rd.table * N;
        const int tg = ftid();
        const int tsw = h2_swz<RB>(tg), tdp = abase | h2s_deposit<typename ST::LinG, LBITS>(tg);
        h2s_for<0, R>([&](auto Q) {
          constexpr int q = Q;
          const qf2 ph = *reinterpret_cast<const qf2*>(tab + (tdp | ST::lin_dep(q)));
          qf2* p0 = reinterpret_cast<qf2*>(t0 + (tsw ^ ST::lin_sw(q)));
          if constexpr (!BWD) {
            *p0 = qc_cmul(ph.x, ph.y, *p0);
          } else {
            qf2* p1 = reinterpret_cast<qf2*>(t1 + (tsw ^ ST::lin_sw(q)));
            const qf2 x = *p0, y = *p1;
            const qf2 m = y * qc_swp(x);   // Im(conj(lam) chi) = lo - hi, invariant under the run's gates
            if constexpr (TABC) {
              if constexpr (rd.tslot == 0) tacc0[q] += m.x - m.y;
              else if constexpr (NTAB > 1) tacc1[q] += m.x - m.y;
            }
            *p0 = qc_cmul(ph.x, -ph.y, x);
            *p1 = qc_cmul(ph.x, -ph.y, y);
          }
        });
        H2_SYNC();
      } else {
        // gate round: per-thread parts of the mapping (the per-q parts are compile-time constants)
        const int tv = ftid();
        const int lbase = h2s_deposit<typename RD::LaneL, LBITS>(tv);
        const int sl = h2_swz<RB>(lbase);
        const int alane = abase | h2s_deposit<typename RD::LaneG, LBITS>(tv);   // global index, register bits clear
        constexpr bool first_r = ri == 0, last_r = ri == NR - 1;
        SV<RB> v[KV];
        if constexpr (first_r && din) {
          const Cplx* g = (BWD && LAST) ? fin_of(c) : chi_of(c);
          if constexpr (!BWD) {
            if constexpr (gen) {
              const qf2 bp = base_phase(alane);
              h2s_for<0, R>([&](auto Q) {
                constexpr int q = Q;
                v[0].a[q] = qc_dup(gen_mag(c, lbase | RD::roff(q))) * h2s_rot<qc_popc(RD::dr(q))>(bp);
              });
            } else {
              h2s_for<0, R>([&](auto Q) {
                constexpr int q = Q;
                v[0].a[q] = *reinterpret_cast<const qf2*>(g + RD::dr(q) + alane);
              });
            }
          } else {
            const Cplx* gl = lam_of(c);
            h2s_for<0, R>([&](auto Q) {
              constexpr int q = Q;
              qf2 x, y;
              if constexpr (LAST) {
                if constexpr (PREF) x = c == 0 ? xz[q] : nxt[q];
                else x = c == 0 ? xz[q] : *reinterpret_cast<const qf2*>(g + RD::dr(q) + alane);
                y = build_lam(c, lbase | RD::roff(q), x, xz[q], l0acc[q]);
              } else if constexpr (PREF2) {
                x = nxt[q];
                y = nxl[q];
              } else {
                x = *reinterpret_cast<const qf2*>(g + RD::dr(q) + alane);
                y = *reinterpret_cast<const qf2*>(gl + RD::dr(q) + alane);
              }
              v[0].a[q] = x;
              v[1].a[q] = y;
            });
            if constexpr (PREF) {
              if (ci + 2 < NCH) {   // channels run 1, 2, ..., NCH - 1, 0: request channel ci + 2 now
                const Cplx* gn = fin_of(ci + 2);
                h2s_for<0, R>([&](auto Q) { nxt[Q] = *reinterpret_cast<const qf2*>(gn + RD::dr(Q) + alane); });
              }
            }
            if constexpr (PREF2) {
              if (ci + 1 < NCH) {   // next channel of the order 1, 2, ..., NCH - 1, 0
                const int cn = ci + 2 < NCH ? ci + 2 : 0;
                const Cplx* gn = chi_of(cn);
                const Cplx* ln = lam_of(cn);
                h2s_for<0, R>([&](auto Q) { nxt[Q] = *reinterpret_cast<const qf2*>(gn + RD::dr(Q) + alane); });
                h2s_for<0, R>([&](auto Q) { nxl[Q] = *reinterpret_cast<const qf2*>(ln + RD::dr(Q) + alane); });
              }
            }
          }
        } else {
          h2s_for<0, R>([&](auto Q) {
            constexpr int q = Q;
            v[0].a[q] = *reinterpret_cast<const qf2*>(t0 + (sl ^ RD::sr(q)));
            if constexpr (BWD) v[1].a[q] = *reinterpret_cast<const qf2*>(t1 + (sl ^ RD::sr(q)));
          });
        }
        // a diagonal table riding on this round: forward before the gates (pre) / after them (post); backward mirrored,
        // with t = Im(conj(lam) chi) accumulated per amplitude at the table's output side
        auto table_here = [&](auto TB, auto TSL) {
          constexpr int table = TB, tslot = TSL;
          const Cplx* tab = A.tabs + (size_t)table * N;
          h2s_for<0, R>([&](auto Q) {
            constexpr int q = Q;
            const qf2 ph = *reinterpret_cast<const qf2*>(tab + RD::dr(q) + alane);
            if constexpr (!BWD) {
              v[0].a[q] = qc_cmul(ph.x, ph.y, v[0].a[q]);
            } else {
              const qf2 m = v[1].a[q] * qc_swp(v[0].a[q]);   // Im(conj(lam) chi) = lo - hi
              if constexpr (TABC) {
                if constexpr (tslot == 0) tacc0[q] += m.x - m.y;
                else if constexpr (NTAB > 1) tacc1[q] += m.x - m.y;
              }
#pragma unroll
              for (int k = 0; k < 2; ++k) v[k].a[q] = qc_cmul(ph.x, -ph.y, v[k].a[q]);
            }
          });
        };
        if constexpr (!BWD && rd.tab_pre >= 0) table_here(std::integral_constant<int, rd.tab_pre>{}, std::integral_constant<int, rd.ts_pre>{});
        if constexpr (BWD && rd.tab_post >= 0) table_here(std::integral_constant<int, rd.tab_post>{}, std::integral_constant<int, rd.ts_post>{});
#ifndef H2_ABLATE_GATES
        h2s_for<0, rd.ng>([&](auto GI_) {
          constexpr int gi = rd.g0 + (BWD ? rd.ng - 1 - GI_ : (int)GI_);
          if constexpr (H2sRz<PL>::fused(gi)) {
            // a run of RZ gates on register bits: applied as one phase record where the sweep first meets it
            constexpr int b0 = H2sRz<PL>::begin_of(gi);
            if constexpr (gi == (BWD ? b0 + H2sRz<PL>::len_of(b0) - 1 : b0)) h2s_rz_run<PL, b0, RB, KV, BWD, NPA>(v, gacc, A);
          } else {
            h2s_gate<PL, gi, RB, KV, BWD, NPA>(v, gacc, A, alane, [](int q) { return RD::dr(q); });
          }
        });
#endif
        if constexpr (!BWD && rd.tab_post >= 0) table_here(std::integral_constant<int, rd.tab_post>{}, std::integral_constant<int, rd.ts_post>{});
        if constexpr (BWD && rd.tab_pre >= 0) table_here(std::integral_constant<int, rd.tab_pre>{}, std::integral_constant<int, rd.ts_pre>{});

        if constexpr (last_r && dout) {
          // straight to HBM in this round's mapping
          if constexpr (!BWD) {
            if constexpr (LAST) {
              qf2 fin[R];
#pragma unroll
              for (int q = 0; q < R; ++q) fin[q] = v[0].a[q];
              struct RBP { static constexpr int at(int j) { return H2sRound<PL, S, rix>::rd.rb[j]; } };
              expval(c, fin, RBP{}, typename RD::LaneL{});
            }
            if (!LAST || A.keep_final) {
              Cplx* g = LAST ? fin_of(c) : chi_of(c);
              h2s_for<0, R>([&](auto Q) {
                constexpr int q = Q;
                *reinterpret_cast<qf2*>(g + RD::dr(q) + alane) = v[0].a[q];
              });
            }
          } else {
            Cplx* g = chi_of(c);
            Cplx* gl = lam_of(c);
            h2s_for<0, R>([&](auto Q) {
              constexpr int q = Q;
              if constexpr (!FINL) *reinterpret_cast<qf2*>(g + RD::dr(q) + alane) = v[0].a[q];   // (FINL: the chi half has it)
              *reinterpret_cast<qf2*>(gl + RD::dr(q) + alane) = v[1].a[q];
            });
          }
        } else {
          if constexpr (BWD && FIRST && last_r && rd.rb[0] % RB == 0 && rd.rb[RB - 1] == rd.rb[0] + RB - 1) {
            // the adjoint sweep ends here: un-embed the register bits of this round (RX^dagger with the point's angles)
            // before lam goes back to LDS; the group loop below skips this register group
            h2s_for<0, RB>([&](auto J) {
              constexpr int j = J;
              const float* w8 = wdp + (size_t)(n - 1 - sd.lb[rd.rb[j]]) * 8;
              g_rx<RB, j>(v[1], h2_unif(w8[0]), -h2_unif(w8[1]));
            });
          }
          h2s_for<0, R>([&](auto Q) {
            constexpr int q = Q;
            if constexpr (!(BWD && FIRST && last_r)) *reinterpret_cast<qf2*>(t0 + (sl ^ RD::sr(q))) = v[0].a[q];
            if constexpr (BWD) *reinterpret_cast<qf2*>(t1 + (sl ^ RD::sr(q))) = v[1].a[q];
          });
          H2_SYNC();
        }
      }
    });

    // ---------------- store phase through LDS (linear mapping) unless the last round wrote HBM itself
    if constexpr (!dout) {
      const int tg = ftid();
      const int tsw = h2_swz<RB>(tg), tdp = abase | h2s_deposit<typename ST::LinG, LBITS>(tg);
      if constexpr (!BWD) {
        if constexpr (LAST) {
          qf2 fin[R];
          h2s_for<0, R>([&](auto Q) { fin[Q] = *reinterpret_cast<const qf2*>(t0 + (tsw ^ ST::lin_sw(Q))); });
          expval(c, fin, LinRBP{}, LinLK{});
          if (A.keep_final) {
            Cplx* g = fin_of(c);
            h2s_for<0, R>([&](auto Q) { *reinterpret_cast<qf2*>(g + (tdp | ST::lin_dep(Q))) = fin[Q]; });
          }
        } else {
          Cplx* g = chi_of(c);
          h2s_for<0, R>([&](auto Q) {
            *reinterpret_cast<qf2*>(g + (tdp | ST::lin_dep(Q))) = *reinterpret_cast<const qf2*>(t0 + (tsw ^ ST::lin_sw(Q)));
          });
        }
      } else {
        if constexpr (!FIRST) {
          Cplx* g = chi_of(c);
          Cplx* gl = lam_of(c);
          h2s_for<0, R>([&](auto Q) {
            if constexpr (!FINL)
              *reinterpret_cast<qf2*>(g + (tdp | ST::lin_dep(Q))) = *reinterpret_cast<const qf2*>(t0 + (tsw ^ ST::lin_sw(Q)));
            *reinterpret_cast<qf2*>(gl + (tdp | ST::lin_dep(Q))) = *reinterpret_cast<const qf2*>(t1 + (tsw ^ ST::lin_sw(Q)));
          });
        } else {
          // un-embed lam on the local wires (RX^dagger with this point's angles), then keep the amplitudes of weight <= 3
          constexpr bool fused_grp = has_rounds && PL::rounds[sd.r0].kind == H2_ROUND_GATES && PL::rounds[sd.r0].rb[0] % RB == 0 &&
                                     PL::rounds[sd.r0].rb[RB - 1] == PL::rounds[sd.r0].rb[0] + RB - 1;
          constexpr int skip_grp = fused_grp ? PL::rounds[sd.r0].rb[0] / RB : -1;
          h2s_for<0, nloc / RB>([&](auto G_) {
            constexpr int grp = G_;
            if constexpr (grp != skip_grp) {
              struct GL { static constexpr int at(int k) { return k < decltype(G_)::value * PL::RB ? k : k + PL::RB; } };
              const int lb2 = h2s_deposit<GL, LBITS>(ftid());
              const int sl2 = h2_swz<RB>(lb2);
              SV<RB> u;
              h2s_for<0, R>([&](auto Q) { u.a[Q] = *reinterpret_cast<const qf2*>(t1 + (sl2 ^ h2_swz<RB>(Q << (grp * RB)))); });
              h2s_for<0, RB>([&](auto J) {
                constexpr int j = J;
                const float* w8 = wdp + (size_t)(n - 1 - sd.lb[grp * RB + j]) * 8;
                g_rx<RB, j>(u, h2_unif(w8[0]), -h2_unif(w8[1]));
              });
              h2s_for<0, R>([&](auto Q) { *reinterpret_cast<qf2*>(t1 + (sl2 ^ h2_swz<RB>(Q << (grp * RB)))) = u.a[Q]; });
              H2_SYNC();
            }
          });
          Cplx* xo = A.xi + (((size_t)c * A.pt_stride + pt) * ntau + tau) * A.nx;
          for (int j = tid; j < A.nx; j += NT) xo[j] = t1[h2_swz<RB>(A.sparse_idx[j])];
        }
      }
    }
    if constexpr (uses_lds || gen) H2_SYNC();   // the next channel overwrites the tile / the series tables
  }

  if constexpr (BWD) {
    // in-round gate gradients of this block: one wave reduction per gate, then across the waves
    if constexpr (NP > 0) {
      h2s_for<0, NP>([&](auto I) {
        const float tot = qc_wave_sum_to_lane63(gacc[I]);
        if (lane == 63) s_g[wave][I] = tot;
      });
      H2_SYNC();
      for (int i = tid; i < NP; i += NT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += s_g[w][i];
        A.gpart[(size_t)i * nblk + blk] = s;
      }
    }
    // diagonal tables: Walsh-Hadamard transform of t over the local bits, coefficients of weight <= 2
    if constexpr (TABC) {
      float* tf = reinterpret_cast<float*>(smem_raw);
      h2s_for<0, NTAB>([&](auto K_) {
        constexpr int k = K_;
        H2_SYNC();
        {
          // t was accumulated in the mapping of the round the table rides on
          using RT = H2sRound<PL, S, sd.tab_round[k]>;
          if constexpr (RT::rd.kind == H2_ROUND_TABLE) {
            const int tsw = h2_swz<RB>(ftid());
            h2s_for<0, R>([&](auto Q) {
              if constexpr (k == 0) tf[tsw ^ ST::lin_sw(Q)] = tacc0[Q];
              else tf[tsw ^ ST::lin_sw(Q)] = tacc1[Q];
            });
          } else {
            const int sl = h2_swz<RB>(h2s_deposit<typename RT::LaneL, LBITS>(ftid()));
            h2s_for<0, R>([&](auto Q) {
              if constexpr (k == 0) tf[sl ^ RT::sr(Q)] = tacc0[Q];
              else tf[sl ^ RT::sr(Q)] = tacc1[Q];
            });
          }
        }
        H2_SYNC();
        h2s_for<0, nloc / RB>([&](auto G_) {
          constexpr int grp = G_;
          struct GL { static constexpr int at(int kk) { return kk < decltype(G_)::value * PL::RB ? kk : kk + PL::RB; } };
          const int sl2 = h2_swz<RB>(h2s_deposit<GL, LBITS>(ftid()));
          float u[R];
          h2s_for<0, R>([&](auto Q) { u[Q] = tf[sl2 ^ h2_swz<RB>(Q << (grp * RB))]; });
#pragma unroll
          for (int j = 0; j < RB; ++j) {
#pragma unroll
            for (int q = 0; q < R; ++q)
              if (!((q >> j) & 1)) {
                const float a = u[q], b = u[q | (1 << j)];
                u[q] = a + b;
                u[q | (1 << j)] = a - b;
              }
          }
          h2s_for<0, R>([&](auto Q) { tf[sl2 ^ h2_swz<RB>(Q << (grp * RB))] = u[Q]; });
          H2_SYNC();
        });
        for (int j = tid; j < A.nc; j += NT) A.dpart[((size_t)k * nblk + blk) * A.nc + j] = tf[h2_swz<RB>(A.wht_idx[j])];
      });
    }
  }
}

}  // namespace

// ---- host side: one launcher table per generated plan
struct H2sLaunchers {
  int n_stages;
  // stage launch: nch in {1, 6}, mode 0 forward / 1 backward; returns false for a combination that was not built
  bool (*stage)(int nch, int mode, int stage, const H2Args& A, int64_t npts64, hipStream_t st);
};

template <class PL, int S, int NCH, int MODE>
static void h2s_launch_one(const H2Args& A, int64_t npts64, hipStream_t st) {
  using ST = H2sStage<PL, S>;
  const size_t sh = sizeof(Cplx) * (size_t)ST::TS * (MODE == 1 ? 2 : 1);
  // the attribute is per device: set before every launch (cheap) rather than once per process
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_h2s_stage<PL, S, NCH, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const unsigned grid = (unsigned)(npts64 << ST::sd.ngb);
  hipLaunchKernelGGL((k_h2s_stage<PL, S, NCH, MODE>), dim3(grid), dim3(ST::NT), sh, st, A);
}

template <class PL, int S = 0>
static bool h2s_dispatch(int nch, int mode, int stage, const H2Args& A, int64_t npts64, hipStream_t st) {
  if constexpr (S >= PL::NSTAGES) {
    return false;
  } else {
    if (stage != S) return h2s_dispatch<PL, S + 1>(nch, mode, stage, A, npts64, st);
    if (nch == 6 && mode == 0) h2s_launch_one<PL, S, 6, 0>(A, npts64, st);
    else if (nch == 6 && mode == 1) h2s_launch_one<PL, S, 6, 1>(A, npts64, st);
    else if (nch == 1 && mode == 0) h2s_launch_one<PL, S, 1, 0>(A, npts64, st);
    else if (nch == 1 && mode == 1) h2s_launch_one<PL, S, 1, 1>(A, npts64, st);
    else return false;
    return true;
  }
}

template <class PL>
struct H2sLaunch {
  static H2sLaunchers table() { return {PL::NSTAGES, &h2s_dispatch<PL, 0>}; }
};
