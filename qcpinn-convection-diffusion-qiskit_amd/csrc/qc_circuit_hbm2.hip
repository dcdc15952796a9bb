// Variational-circuit kernels, "hbm" family, staged execution (plan: qc_hbm2_plan.h): 9 <= n <= 20 qubits,
// statevectors resident in HBM, angle encoding.  Replaces the PennyLane default.qubit simulation behind
// DVQuantumLayer.forward (reference nn/DVQuantumLayer.py:151-154,176-214) and the torch double-backward through it
// (nn/pde.py:59-70 + loss.backward()) for the 16-qubit cross_mesh configuration.
//
// Memory: every 64-point tile owns one SLOT [chi: nch][64][2^n] [lam: nch][64][2^n] (complex64).  A launch covers ALL
// tiles that fit the workspace; a block = (point, tile of 2^nloc amplitudes) and walks the channels of its point one
// after the other, so that what couples the channels stays on chip:
//   forward   stage 0 GENERATES the embedded product state and its derivative series in LDS (no init pass);
//             the last stage reduces <Z_w> and its jets against the value channel's tile held in registers (no
//             read-back pass) and leaves the final states in the slot for the adjoint sweep;
//   backward  the last stage builds the cotangents lam_c from the final states on load (lam_0 accumulates in
//             registers over the channel loop), every stage un-applies its rounds on (chi, lam) and accumulates
//             Im<lam|G|chi> per gate; diagonal tables: t = sum_c Im(conj lam chi) in registers, one Walsh-Hadamard
//             transform per tile, coefficients of weight <= 2; stage 0 finally un-embeds lam (RX^dagger of the
//             local wires) and writes only its amplitudes of weight <= 3: the cotangents of the angle jets are sparse
//             inner products in that frame (qc_gates.h, "pulled-back frame").
// HBM traffic per channel-evaluation at n = 16 cross_mesh (2 stages): forward 1 write + 1 read + 1 write (kept
// final state), backward 1 read + 2 writes + 2 reads  ->  8 state transfers for forward + adjoint against the
// SURVEY §8(d) minimum of 4; the round-1 kernels moved ~56.
#include "qc_circuit_h2s_kernels.h"

#include <stdlib.h>
#include <string.h>

#include <vector>

namespace {

// ---- device copy of the plan
struct H2Dev {
  H2Plan plan;
  H2Round* d_rounds = nullptr;
  H2Gate* d_gates = nullptr;
  H2DiagGate* d_dgates = nullptr;
  int* d_sparse = nullptr;      // local indices of weight <= 3 (first stage's tile)
  int* d_rank = nullptr;        // inverse: local index -> position in d_sparse, or -1
  int* d_postab = nullptr;      // k_h2_abar: positions of mu[k] in a point's [tau][rank] copy (pos1 | pos2 | pos3 | pair u | pair v)
  int* d_whtidx = nullptr;      // local masks of weight 0, 1, 2 (coefficient order of the table gradients)
  int* d_pslots = nullptr;      // per stage: parameter slot of in-round parametric gate pidx; offsets in pslot_off
  std::vector<int> pslot_off;
  int nx = 0, nc = 0;
  const H2sLaunchers* stat = nullptr;   // compile-time stage programs of exactly this plan (gen/qc_static_h2_*.hip), or null
  int n_ph = 0;                 // fused RZ runs of the compile-time programs (H2sRz), their gate lists on the device
  int* d_phdesc = nullptr;
};

// ---------------------------------------------------------------- per-point, per-wire embedding data
// wd[pt][w][8] = {cos, sin of (a_w + theta_w)/2, da_t, da_x, da_y, dda_xx, dda_yy, 0}, pt = point within the launch group
template <int NCH>
__global__ void k_h2_wiredata(const float* __restrict__ ajets, int64_t B, int64_t p_first, int64_t npts, int n,
                              float* __restrict__ wd, const QcTrig* __restrict__ trig, int absorb) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npts * n) return;
  const int64_t p = p_first + i / n;
  const int w = (int)(i % n);
  float* o = wd + (size_t)i * 8;
  const float a = ajets[(int64_t)w * B + p] + (absorb ? trig[w].th : 0.f);
  float s, c;
  sincosf(0.5f * a, &s, &c);
  o[0] = c;
  o[1] = s;
#pragma unroll
  for (int k = 0; k < 6; ++k) o[2 + k] = 0.f;
  if constexpr (NCH == 6) {
    for (int k = 0; k < 3; ++k) o[2 + k] = ajets[((int64_t)(1 + k) * n + w) * B + p];
    for (int k = 0; k < 2; ++k) o[5 + k] = ajets[((int64_t)(4 + k) * n + w) * B + p];
  }
}

// phase table of one diagonal run: tab[k] = prod_g phase_g(k), accumulated in double
__global__ void __launch_bounds__(256) k_h2_diag_table(const H2DiagGate* __restrict__ dg, int ng,
                                                       const QcTrig* __restrict__ trig, int n, Cplx* __restrict__ tab) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= ((int64_t)1 << n)) return;
  double zr = 1.0, zi = 0.0;
  for (int i = 0; i < ng; ++i) {
    const H2DiagGate g = dg[i];
    if (g.bc >= 0 && !((k >> g.bc) & 1)) continue;
    const QcTrig tr = trig[g.gi];
    const double c = tr.c, s = ((k >> g.bt) & 1) ? (double)tr.s : -(double)tr.s;   // bit 0: c - i s, bit 1: c + i s
    const double nr = zr * c - zi * s, ni = zr * s + zi * c;
    zr = nr;
    zi = ni;
  }
  tab[k] = {(float)zr, (float)zi};
}

// ---------------------------------------------------------------- stage kernel (run-time plan)
__device__ __forceinline__ H2Round h2_load_round(const H2Round* p) {
  const auto* c = h2_const(p);
  H2Round r;
  r.kind = c->kind;
  r.nrb = c->nrb;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.rb[j] = c->rb[j];
  r.g0 = c->g0;
  r.ng = c->ng;
  r.table = c->table;
  r.tslot = c->tslot;
  r.tab_pre = c->tab_pre;
  r.ts_pre = c->ts_pre;
  r.tab_post = c->tab_post;
  r.ts_post = c->ts_post;
  return r;
}
__device__ __forceinline__ H2Gate h2_load_gate(const H2Gate* p) {
  const auto* c = h2_const(p);
  H2Gate g;
  g.op = c->op;
  g.kind = c->kind;
  g.tq = c->tq;
  g.cq = c->cq;
  g.tbit = c->tbit;
  g.cbit = c->cbit;
  g.gi = c->gi;
  g.slot = c->slot;
  g.pidx = c->pidx;
  return g;
}

// one-bit X (the CNOT body under a predicate)
template <int N, int B>
__device__ __forceinline__ void g_x(SV<N>& v) {
#pragma unroll
  for (int k = 0; k < (1 << (N - 1)); ++k) {
    const int i0 = qc_ins0(k, B), i1 = i0 | (1 << B);
    const qf2 a = v.a[i0];
    v.a[i0] = v.a[i1];
    v.a[i1] = a;
  }
}

template <int N, int K>
__device__ __forceinline__ void h2_apply_x(SV<N> (&v)[K], int tq) {
  switch (tq) {
// (the empty asm with a per-case immediate keeps the optimiser from merging the cases into one body with
// run-time register indices, which would move the amplitude arrays to scratch memory)
#define HX_(BIT) case BIT: if constexpr (BIT < N) { _Pragma("unroll") for (int q = 0; q < K; ++q) g_x<N, BIT>(v[q]); } asm volatile("" ::"n"(BIT)); break;
    HX_(0) HX_(1) HX_(2) HX_(3)
#undef HX_
    default: break;
  }
}

template <int N, int K, class UP>
__device__ __forceinline__ void h2_apply_u4(SV<N> (&v)[K], int hq, int lq, UP u) {
  if constexpr (N >= 2) {
    switch (hq * 4 + lq) {
#define HU_(HB, LB)                                                                       \
  case (HB * 4 + LB):                                                                     \
    if constexpr (HB < N && LB < N && HB != LB) {                                         \
      _Pragma("unroll") for (int q = 0; q < K; ++q) g_u4<N, HB, LB, UP>(v[q], u);         \
    }                                                                                     \
    asm volatile("" ::"n"(HB * 4 + LB));                                                  \
    break;
      HU_(0, 1) HU_(0, 2) HU_(0, 3) HU_(1, 0) HU_(1, 2) HU_(1, 3) HU_(2, 0) HU_(2, 1) HU_(2, 3) HU_(3, 0) HU_(3, 1) HU_(3, 2)
#undef HU_
      default: break;
    }
  }
}

// MODE 0: forward.  MODE 1: backward (chi and lam).  RB register bits per round, NT = 2^(nloc - RB) threads.
//
// Amplitude mappings.  LINEAR: thread tid owns local indices tid | (q << LBITS), q < 2^RB (coalesced HBM access).
// ROUND: thread owns lbase | roff[q], lbase = tid spread over the positions outside the round's register set.  The LDS
// swizzle and the local -> global index deposit are GF(2)-linear / OR-linear over disjoint bit sets, so every
// address is (per-thread part, once per phase) XOR / OR (per-q part, wave-uniform scalars).
// A stage whose first executed round keeps the low four local positions as lane bits takes its input straight from
// HBM (or generates it) in that round's mapping, and the last one stores straight to HBM: a one-round stage touches
// LDS only for reductions.
// ROLE (what the stage may need, so that dead per-thread state costs no registers): bit 0 = may be the plan's last
// stage (forward: value-channel tile for the <Z> sums; backward: lam_0 accumulator), bit 1 = may carry diagonal tables
// (backward: t accumulators).
template <int RB, int NCH, int MODE, int ROLE>
__global__ void __launch_bounds__(RB == 3 ? 512 : 256, RB == 3 ? 4 : 2) k_h2_stage(const H2Args A) {
  constexpr bool BWD = MODE == 1;
  constexpr bool LASTC = (ROLE & 1) != 0, TABC = (ROLE & 2) != 0;
  constexpr int R = 1 << RB;
  constexpr int KV = BWD ? 2 : 1;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const H2Stage& sd = A.sd;
  const int nloc = sd.nloc, TS = 1 << nloc, NT = TS >> RB, LBITS = nloc - RB;
  Cplx* t0 = reinterpret_cast<Cplx*>(smem_raw);          // chi tile
  Cplx* t1 = t0 + (BWD ? TS : 0);                        // lam tile (backward)
  __shared__ int s_depA[64], s_depB[64];                 // local index -> global offset, low / high 6 positions
  __shared__ float s_tabA[64][3], s_tabB[64][3];         // series factors of the tile (forward generation)
  __shared__ float s_D[6][128];                          // backward: D_c = g + A[l & 63] + B[l >> 6]
  __shared__ float s_Dg[6];
  __shared__ float s_g[8][H2_MAXP];                      // per-wave sums of the in-round gate gradients
  __shared__ float s_red[8][H2_XW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NW = (NT + 63) >> 6;
  const int n = A.n;
  const int64_t N = (int64_t)1 << n;
  const int ntau = 1 << sd.ngb;
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
      for (int i = tid; i < sd.np; i += NT) A.gpart[(size_t)i * nblk + blk] = 0.f;
      for (int k = 0; k < sd.ntab; ++k)
        for (int i = tid; i < A.nc; i += NT) A.dpart[((size_t)k * nblk + blk) * A.nc + i] = 0.f;
    }
    return;
  }
  int abase = 0;   // global index of the tile (amplitude indices fit 32 bits: n <= 20)
  for (int j = 0; j < sd.ngb; ++j)
    if ((tau >> j) & 1) abase |= 1 << sd.gb[j];
  if (tid < 64) {
    int a = 0, b = 0;
    for (int j = 0; j < 6 && j < nloc; ++j)
      if ((tid >> j) & 1) a |= 1 << sd.lb[j];
    for (int j = 6; j < nloc; ++j)
      if ((tid >> (j - 6)) & 1) b |= 1 << sd.lb[j];
    s_depA[tid] = a;
    s_depB[tid] = b;
  }
  if constexpr (BWD) {
    for (int i = tid; i < 8 * H2_MAXP; i += NT) (&s_g[0][0])[i] = 0.f;
  }
  auto dep = [&](int l) { return s_depA[l & 63] | s_depB[l >> 6]; };
  // The thread index as an opaque value: addresses derived from it are recomputed where they are used instead of
  // being hoisted out of the channel loop and held in ~100 VGPRs for the whole kernel.
  auto ftid = [&]() {
    int v = tid;
    asm volatile("" : "+v"(v));
    return v;
  };
  const float* wdp = A.wd + (size_t)pt * n * 8;
  Cplx* slot = A.store + (size_t)tile64 * A.slot_elems;
  auto chi_of = [&](int c) { return slot + ((size_t)c * 64 + t) * N; };
  auto lam_of = [&](int c) { return slot + ((size_t)(NCH + c) * 64 + t) * N; };
  // final states of a multi-stage plan live in the lam half of the slot; the chi half keeps the state before the last
  // stage for the adjoint of the stage before it (see k_h2s_stage)
  const bool finl = A.last && !A.first;
  auto fin_of = [&](int c) { return finl ? lam_of(c) : chi_of(c); };

  // linear mapping, per-q parts (wave-uniform)
  int lin_sw[R], lin_dep[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    lin_sw[q] = h2_swz<RB>(q << LBITS);
    int d = 0;
#pragma unroll
    for (int j = 0; j < RB; ++j) d |= ((q >> j) & 1) << sd.lb[LBITS + j];
    lin_dep[q] = d;
  }

  // registers that live across the channel loop
  Cplx x0[!BWD && LASTC ? R : 1];     // forward, last stage: final value-channel tile
  Cplx l0acc[BWD && LASTC ? R : 1];   // backward, last stage: lam_0 = sum_c D_c chi_c
  float tacc0[BWD && TABC ? R : 1], tacc1[BWD && TABC ? R : 1];   // backward: t = sum_c Im(conj lam chi) per table
  if constexpr (BWD && LASTC) {
#pragma unroll
    for (int q = 0; q < R; ++q) l0acc[q] = {0.f, 0.f};
  }
  if constexpr (BWD && TABC) {
#pragma unroll
    for (int q = 0; q < R; ++q) tacc0[q] = tacc1[q] = 0.f;
  }
  H2_SYNC();

  if constexpr (BWD && LASTC) {
    if (A.last) {
      // D_c[k] = sum_w (+-) qbar[c][w][p] by index bit n-1-w: constant part (non-local bits) + two 64-entry tables
      for (int i = tid; i < NCH * 128; i += NT) {
        const int c = i >> 7, e = i & 127, half = e >> 6, v = e & 63;
        float s = 0.f;
        const int j0 = half ? 6 : 0, j1 = half ? nloc : (nloc < 6 ? nloc : 6);
        for (int j = j0; j < j1; ++j) {
          const float qb = A.qbar[((int64_t)c * n + (n - 1 - sd.lb[j])) * A.B + p];
          s += ((v >> (j - j0)) & 1) ? -qb : qb;
        }
        s_D[c][e] = s;
      }
      if (tid < NCH) {
        float s = 0.f;
        for (int j = 0; j < sd.ngb; ++j) {
          const float qb = A.qbar[((int64_t)tid * n + (n - 1 - sd.gb[j])) * A.B + p];
          s += ((tau >> j) & 1) ? -qb : qb;
        }
        s_Dg[tid] = s;
      }
      H2_SYNC();
    }
  }
  auto Dval = [&](int c, int l) { return s_Dg[c] + s_D[c][l & 63] + s_D[c][64 + (l >> 6)]; };

  // which rounds talk to HBM directly
  const bool gen = !BWD && A.first;
  bool din = false, dout = false;
  if (sd.nr > 0) {
    const H2Round* rf = A.rounds + sd.r0 + (BWD ? sd.nr - 1 : 0);
    const H2Round* rl = A.rounds + sd.r0 + (BWD ? 0 : sd.nr - 1);
    din = h2_uni(rf->kind) == H2_ROUND_GATES && (gen || h2_uni(rf->rb[0]) >= 4);
    dout = h2_uni(rl->kind) == H2_ROUND_GATES && h2_uni(rl->rb[0]) >= 4 && !(BWD && A.first);
  }
  const bool uses_lds = !(din && dout && sd.nr == 1);

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
      const int half = e_ >> 6, v = e_ & 63;
      float P0 = 1.f, P1 = 0.f, P2 = 0.f;
      if (half == 0) {
        for (int j = 0; j < sd.ngb; ++j) step(P0, P1, P2, sd.gb[j], (tau >> j) & 1);
        for (int j = 0; j < 6 && j < nloc; ++j) step(P0, P1, P2, sd.lb[j], (v >> j) & 1);
        s_tabA[v][0] = P0; s_tabA[v][1] = P1; s_tabA[v][2] = P2;
      } else {
        for (int j = 6; j < nloc; ++j) step(P0, P1, P2, sd.lb[j], (v >> (j - 6)) & 1);
        s_tabB[v][0] = P0; s_tabB[v][1] = P1; s_tabB[v][2] = P2;
      }
    }
  };
  auto gen_amp = [&](int c, int l, int a) {   // local index l, global index a
    if (A.amp) {   // amplitude encoding (nn/DVQuantumLayer.py:177-180): basis state a carries feature a (real), a < n
      Cplx v = {0.f, 0.f};
      if (a < n) v.re = A.ajets[((int64_t)c * n + a) * A.B + p];
      return v;
    }
    const int ord = c == 0 ? 0 : (c <= 3 ? 1 : 2);
    const float a0 = s_tabA[l & 63][0], a1 = s_tabA[l & 63][1], a2 = s_tabA[l & 63][2];
    const float b0 = s_tabB[l >> 6][0], b1 = s_tabB[l >> 6][1], b2 = s_tabB[l >> 6][2];
    const float m = ord == 0 ? a0 * b0 : (ord == 1 ? a1 * b0 + a0 * b1 : a0 * b2 + 2.f * a1 * b1 + a2 * b0);
    const int ph = __popc((unsigned)a) & 3;
    Cplx v;
    v.re = ph == 0 ? m : (ph == 2 ? -m : 0.f);
    v.im = ph == 1 ? -m : (ph == 3 ? m : 0.f);
    return v;
  };
  // backward, last stage: cotangent of channel c's final state at local index l from (chi_c, chi_0) [DESIGN.md §3]:
  // lam_0 = sum_c D_c chi_c, lam_t = D_t chi_0, lam_x = D_x chi_0 + 2 D_xx chi_x, lam_xx = D_xx chi_0 (same for y)
  auto build_lam = [&](int c, int l, Cplx x, Cplx xv, Cplx& acc) {
    Cplx y;
    if (c == 0) {
      const float d = Dval(0, l);
      y = {acc.re + d * x.re, acc.im + d * x.im};
    } else {
      const float d = Dval(c, l);
      y = {d * xv.re, d * xv.im};
      if (c == 2 || c == 3) {
        const float d2 = 2.f * Dval(c + 2, l);
        y.re += d2 * x.re;
        y.im += d2 * x.im;
      }
      acc.re += d * x.re;
      acc.im += d * x.im;
    }
    return y;
  };
  // forward, last stage: <Z> sums of this tile against the value channel's final tile, per index bit.
  // fin[q] at local index lb_ | (q's bits at positions rbp[]), lb_ = this thread's lane part.
  auto expval = [&](int c, const Cplx (&fin)[R], int lb_, const int (&rbp)[RB]) {
    float tot = 0.f, sq = 0.f, qs[RB], qq[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) qs[j] = qq[j] = 0.f;
#pragma unroll
    for (int q = 0; q < R; ++q) {
      constexpr int XQ = (!BWD && LASTC) ? R : 1;
      const int xq = q < XQ ? q : 0;
      if (c == 0) x0[xq] = fin[q];
      const float w = c == 0 ? fin[q].re * fin[q].re + fin[q].im * fin[q].im
                             : 2.f * (x0[xq].re * fin[q].re + x0[xq].im * fin[q].im);
      const float w2 = 2.f * (fin[q].re * fin[q].re + fin[q].im * fin[q].im);
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
      for (int b = 0; b < n; ++b) {   // b = global index bit
        const int wb = sd.where[b];   // local position, or -(1 + j) for the non-local bit gb[j]
        const float T_ = pass ? sq : tot;
        float mine;
        if (wb < 0) {
          mine = ((tau >> (-wb - 1)) & 1) ? -T_ : T_;
        } else {
          float part = 0.f;
          bool isreg = false;
#pragma unroll
          for (int j = 0; j < RB; ++j)
            if (wb == rbp[j]) {
              part = pass ? qq[j] : qs[j];
              isreg = true;
            }
          mine = isreg ? T_ - 2.f * part : (((lb_ >> wb) & 1) ? -T_ : T_);
        }
        const float wv = qc_wave_sum_to_lane63(mine);
        if (lane == 63) s_red[wave][b] = wv;
      }
      H2_SYNC();
      if (tid < n) {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += s_red[w][tid];
        const int ch8 = pass ? 6 + (c - 2) : c;
        A.xpart[(((size_t)ch8 * A.pt_stride + pt) * ntau + tau) * H2_XW + tid] = s;
      }
      H2_SYNC();
    }
  };
  // per-thread part of a round mapping: positions outside the register set rb[] take the bits of the thread index
  auto round_lbase = [&](const int (&rb)[4]) {
    bool contig = true;
#pragma unroll
    for (int j = 1; j < RB; ++j) contig = contig && rb[j] == rb[0] + j;
    const int tv = ftid();
    if (contig) {
      const int lo = rb[0];
      return (tv & ((1 << lo) - 1)) | ((tv >> lo) << (lo + RB));
    }
    int regmask = 0, lbase = 0, tb = tv;
#pragma unroll
    for (int j = 0; j < RB; ++j) regmask |= 1 << rb[j];
    for (int pos = 0; pos < nloc; ++pos)
      if (!((regmask >> pos) & 1)) {
        lbase |= (tb & 1) << pos;
        tb >>= 1;
      }
    return lbase;
  };

  // ------------------------------------------------------------------ channel loop
  for (int ci = 0; ci < NCH; ++ci) {
    // backward: the value channel last (its cotangent needs every other channel's final state)
    const int c = BWD ? (ci + 1 < NCH ? ci + 1 : 0) : ci;
    if (gen && !A.amp) {
      gen_tables(c);
      H2_SYNC();
    }
    // ---------------- load phase through LDS (linear mapping) unless the first round reads HBM itself
    if (!din) {
      const int tg = ftid();
      const int tsw = h2_swz<RB>(tg), tdp = abase | dep(tg);
      if constexpr (!BWD) {
        const Cplx* g = chi_of(c);
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const int a = tdp | lin_dep[q];
          t0[tsw ^ lin_sw[q]] = gen ? gen_amp(c, tg | (q << LBITS), a) : g[a];
        }
      } else {
        const Cplx* g = (LASTC && A.last) ? fin_of(c) : chi_of(c);
        const Cplx* gl = lam_of(c);
        const Cplx* g0 = fin_of(0);
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const int a = tdp | lin_dep[q];
          const Cplx x = g[a];
          Cplx y;
          if (LASTC && A.last) y = build_lam(c, tg | (q << LBITS), x, c == 0 ? x : g0[a], l0acc[LASTC ? q : 0]);
          else y = gl[a];
          t0[tsw ^ lin_sw[q]] = x;
          t1[tsw ^ lin_sw[q]] = y;
        }
      }
      H2_SYNC();
    }

    // ---------------- rounds
    for (int ri = 0; ri < sd.nr; ++ri) {
      const int rix = BWD ? sd.nr - 1 - ri : ri;
      const H2Round rd = h2_load_round(A.rounds + sd.r0 + rix);
      if (rd.kind == H2_ROUND_TABLE) {   // a table without a gate round to ride on: element-wise pass in the linear mapping
        const Cplx* tab = A.tabs + (size_t)rd.table * N;
        const int tg = ftid();
        const int tsw = h2_swz<RB>(tg), tdp = abase | dep(tg);
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const Cplx ph = tab[tdp | lin_dep[q]];
          const int li = tsw ^ lin_sw[q];
          if constexpr (!BWD) {
            t0[li] = cmul(t0[li], ph);
          } else {
            const Cplx x = t0[li], y = t1[li];
            const float tv = y.re * x.im - y.im * x.re;   // Im(conj(lam) chi), invariant under the run's gates
            if constexpr (TABC) {
              if (rd.tslot == 0) tacc0[q] += tv;
              else tacc1[q] += tv;
            }
            t0[li] = cmulc(x, ph);
            t1[li] = cmulc(y, ph);
          }
        }
        H2_SYNC();
        continue;
      }
      // gate round: per-thread and per-q parts of the mapping
      const int lbase = round_lbase(rd.rb);
      const int sl = h2_swz<RB>(lbase);
      const int alane = abase | dep(lbase);          // global index of this thread's amplitudes, register bits clear
      int roff[R], sr[R], dr[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        int o = 0, d = 0;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          o |= ((q >> j) & 1) << rd.rb[j];
          d |= ((q >> j) & 1) << sd.lb[rd.rb[j]];
        }
        roff[q] = o;
        sr[q] = h2_swz<RB>(o);
        dr[q] = d;
      }
      const bool first_r = ri == 0, last_r = ri == sd.nr - 1;
      SV<RB> v[KV];
      if (first_r && din) {
        if constexpr (!BWD) {
          const Cplx* g = chi_of(c);
#pragma unroll
          for (int q = 0; q < R; ++q) {
            const int a = alane | dr[q];
            const Cplx x = gen ? gen_amp(c, lbase | roff[q], a) : g[a];
            v[0].a[q].x = x.re;
            v[0].a[q].y = x.im;
          }
        } else {
          const Cplx* g = (LASTC && A.last) ? fin_of(c) : chi_of(c);
          const Cplx* gl = lam_of(c);
          const Cplx* g0 = fin_of(0);
#pragma unroll
          for (int q = 0; q < R; ++q) {
            const int a = alane | dr[q];
            const Cplx x = g[a];
            Cplx y;
            if (LASTC && A.last) y = build_lam(c, lbase | roff[q], x, c == 0 ? x : g0[a], l0acc[LASTC ? q : 0]);
            else y = gl[a];
            v[0].a[q].x = x.re;
            v[0].a[q].y = x.im;
            v[1].a[q].x = y.re;
            v[1].a[q].y = y.im;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const int li = sl ^ sr[q];
          const Cplx x = t0[li];
          v[0].a[q].x = x.re;
          v[0].a[q].y = x.im;
          if constexpr (BWD) {
            const Cplx y = t1[li];
            v[1].a[q].x = y.re;
            v[1].a[q].y = y.im;
          }
        }
      }
      // a diagonal table riding on this round: forward before the gates (pre) / after them (post); backward mirrored,
      // with t = Im(conj(lam) chi) accumulated per amplitude at the table's output side
      auto table_here = [&](int table, int tslot) {
        const Cplx* tab = A.tabs + (size_t)table * N;
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const Cplx ph = tab[alane | dr[q]];
          if constexpr (!BWD) {
            v[0].a[q] = qc_cmul(ph.re, ph.im, v[0].a[q]);
          } else {
            const qf2 m = v[1].a[q] * qc_swp(v[0].a[q]);   // Im(conj(lam) chi) = lo - hi
            const float tv = m.x - m.y;
            if constexpr (TABC) {
              if (tslot == 0) tacc0[q] += tv;
              else tacc1[q] += tv;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) v[k].a[q] = qc_cmul(ph.re, -ph.im, v[k].a[q]);
          }
        }
      };
      if (!BWD && rd.tab_pre >= 0) table_here(rd.tab_pre, rd.ts_pre);
      if (BWD && rd.tab_post >= 0) table_here(rd.tab_post, rd.ts_post);
#ifdef H2_ABLATE_GATES
      for (int gi = 0; gi < 0; ++gi) {
#else
      for (int gi = 0; gi < rd.ng; ++gi) {
#endif
        const H2Gate hg = h2_load_gate(A.gates + rd.g0 + (BWD ? rd.ng - 1 - gi : gi));
        float c_ = 1.f, s_ = 0.f;
        if (hg.op != QC_U4 && hg.slot >= 0) {
          const auto* tr = h2_const(A.trig + hg.gi);
          c_ = tr->c;
          s_ = tr->s;
        }
        float grad = 0.f;
        switch (hg.kind) {
          case H2_K_REG1:
          case H2_K_REG2: {
            QcGate gg;
            gg.op = hg.op;
            gg.ba = hg.kind == H2_K_REG2 ? hg.cq : hg.tq;
            gg.bb = hg.kind == H2_K_REG2 ? hg.tq : -1;
            gg.slot = hg.slot;
            if constexpr (BWD) {
              if (hg.pidx >= 0) grad = qc_gate_grad<RB>(v[1], v[0], gg);
            }
            qc_apply_gate<RB, KV, BWD>(v, gg, c_, s_, A.umat);
            break;
          }
          case H2_K_PRED: {
            const bool on = (alane >> hg.cbit) & 1;
            if (on) {
              if (hg.op == QC_CNOT) {
                h2_apply_x<RB, KV>(v, hg.tq);
              } else {   // CRX
                QcGate gg;
                gg.op = QC_RX;
                gg.ba = hg.tq;
                gg.bb = -1;
                gg.slot = hg.slot;
                if constexpr (BWD) {
                  if (hg.pidx >= 0) grad = qc_gate_grad<RB>(v[1], v[0], gg);
                }
                qc_apply_gate<RB, KV, BWD>(v, gg, c_, s_, A.umat);
              }
            }
            break;
          }
          case H2_K_PHASE: {
            const float sg = BWD ? -s_ : s_;
#pragma unroll
            for (int q = 0; q < R; ++q) {
              const int a = alane | dr[q];
              const bool on = hg.cbit < 0 || ((a >> hg.cbit) & 1);
              const bool hi = (a >> hg.tbit) & 1;
              if constexpr (BWD) {
                const qf2 m = v[1].a[q] * qc_swp(v[0].a[q]);
                const float tv = m.x - m.y;
                grad += on ? (hi ? -tv : tv) : 0.f;
              }
              const float sq = on ? (hi ? sg : -sg) : 0.f, cq = on ? c_ : 1.f;   // multiply by cq + i sq
#pragma unroll
              for (int k = 0; k < KV; ++k) v[k].a[q] = qc_cmul(cq, sq, v[k].a[q]);
            }
            break;
          }
          case H2_K_U4: {
            // the 4x4 matrix through the constant address space: (re, im) records as SGPR pairs, fetched inside the one
            // (high, low) arm that runs (32 SGPRs held across the switch would spill the interpreter's scalar state)
            const auto* um = h2_const(A.umat + (hg.slot * 2 + (BWD ? 1 : 0)) * 32);
            h2_apply_u4<RB, KV>(v, hg.tq, hg.cq, um);
            break;
          }
          default: break;
        }
        if constexpr (BWD) {
          if (hg.pidx >= 0) {
            const float tot = qc_wave_sum_to_lane63(grad);
            if (lane == 63) s_g[wave][hg.pidx] += tot;
          }
        }
      }
      if (!BWD && rd.tab_post >= 0) table_here(rd.tab_post, rd.ts_post);
      if (BWD && rd.tab_pre >= 0) table_here(rd.tab_pre, rd.ts_pre);

      if (last_r && dout) {
        // straight to HBM in this round's mapping
        if constexpr (!BWD) {
          if (LASTC && A.last) {
            Cplx fin[R];
#pragma unroll
            for (int q = 0; q < R; ++q) fin[q] = {v[0].a[q].x, v[0].a[q].y};
            int rbp[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) rbp[j] = rd.rb[j];
            expval(c, fin, lbase, rbp);
          }
          if (!(LASTC && A.last) || A.keep_final) {
            Cplx* g = (LASTC && A.last) ? fin_of(c) : chi_of(c);
#pragma unroll
            for (int q = 0; q < R; ++q) g[alane | dr[q]] = {v[0].a[q].x, v[0].a[q].y};
          }
        } else {
          Cplx* g = chi_of(c);
          Cplx* gl = lam_of(c);
#pragma unroll
          for (int q = 0; q < R; ++q) {
            if (!finl) g[alane | dr[q]] = {v[0].a[q].x, v[0].a[q].y};   // (finl: the chi half already holds it)
            gl[alane | dr[q]] = {v[1].a[q].x, v[1].a[q].y};
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const int li = sl ^ sr[q];
          t0[li] = {v[0].a[q].x, v[0].a[q].y};
          if constexpr (BWD) t1[li] = {v[1].a[q].x, v[1].a[q].y};
        }
        H2_SYNC();
      }
    }

    // ---------------- store phase through LDS (linear mapping) unless the last round wrote HBM itself
    if (!dout) {
      const int tg = ftid();
      const int tsw = h2_swz<RB>(tg), tdp = abase | dep(tg);
      if constexpr (!BWD) {
        if (LASTC && A.last) {
          Cplx fin[R];
#pragma unroll
          for (int q = 0; q < R; ++q) fin[q] = t0[tsw ^ lin_sw[q]];
          int rbp[RB];
#pragma unroll
          for (int j = 0; j < RB; ++j) rbp[j] = LBITS + j;
          expval(c, fin, tg, rbp);
          if (A.keep_final) {
            Cplx* g = fin_of(c);
#pragma unroll
            for (int q = 0; q < R; ++q) g[tdp | lin_dep[q]] = fin[q];
          }
        } else {
          Cplx* g = chi_of(c);
#pragma unroll
          for (int q = 0; q < R; ++q) g[tdp | lin_dep[q]] = t0[tsw ^ lin_sw[q]];
        }
      } else {
        if (!A.first) {
          Cplx* g = chi_of(c);
          Cplx* gl = lam_of(c);
#pragma unroll
          for (int q = 0; q < R; ++q) {
            if (!finl) g[tdp | lin_dep[q]] = t0[tsw ^ lin_sw[q]];
            gl[tdp | lin_dep[q]] = t1[tsw ^ lin_sw[q]];
          }
        } else if (A.amp) {
          // amplitude encoding: the state is linear in the (real) initial amplitudes: abar[c][w] = 2 Re lam_c[w], w < n;
          // basis state w sits in the tile whose non-local bits match it
          for (int w = tid; w < n; w += NT) {
            int nonloc = 0, l = 0;
            for (int j = 0; j < sd.ngb; ++j) nonloc |= w & (1 << sd.gb[j]);
            for (int j = 0; j < nloc; ++j) l |= ((w >> sd.lb[j]) & 1) << j;
            if (nonloc == abase) A.abar[((int64_t)c * n + w) * A.B + p] = 2.f * t1[h2_swz<RB>(l)].re;
          }
        } else {
          // un-embed lam on the local wires (RX^dagger with this point's angles), then keep the amplitudes of weight <= 3
          for (int grp = 0; grp < nloc / RB; ++grp) {
            int rbg[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < RB; ++j) rbg[j] = grp * RB + j;
            const int lb2 = round_lbase(rbg);
            const int sl2 = h2_swz<RB>(lb2);
            SV<RB> u[1];
#pragma unroll
            for (int q = 0; q < R; ++q) {
              const Cplx y = t1[sl2 ^ h2_swz<RB>(q << (grp * RB))];
              u[0].a[q].x = y.re;
              u[0].a[q].y = y.im;
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
              const float* w8 = wdp + (size_t)(n - 1 - sd.lb[grp * RB + j]) * 8;
              QcGate gg;
              gg.op = QC_RX;
              gg.ba = j;
              gg.bb = -1;
              gg.slot = 0;
              qc_apply_gate<RB, 1, true>(u, gg, h2_unif(w8[0]), h2_unif(w8[1]), A.umat);
            }
#pragma unroll
            for (int q = 0; q < R; ++q) t1[sl2 ^ h2_swz<RB>(q << (grp * RB))] = {u[0].a[q].x, u[0].a[q].y};
            H2_SYNC();
          }
          Cplx* xo = A.xi + (((size_t)c * A.pt_stride + pt) * ntau + tau) * A.nx;
          for (int j = tid; j < A.nx; j += NT) xo[j] = t1[h2_swz<RB>(A.sparse_idx[j])];
        }
      }
    }
    if (uses_lds || gen) H2_SYNC();   // the next channel overwrites the tile / the series tables
  }

  if constexpr (BWD) {
    // in-round gate gradients of this block
    H2_SYNC();
    for (int i = tid; i < sd.np; i += NT) {
      float s = 0.f;
      for (int w = 0; w < NW; ++w) s += s_g[w][i];
      A.gpart[(size_t)i * nblk + blk] = s;
    }
    // diagonal tables: Walsh-Hadamard transform of t over the local bits, coefficients of weight <= 2
    float* tf = reinterpret_cast<float*>(smem_raw);
    for (int k = 0; k < (TABC ? sd.ntab : 0); ++k) {
      H2_SYNC();
      {
        // t was accumulated in the mapping of the round the table rides on
        const H2Round rt = h2_load_round(A.rounds + sd.r0 + sd.tab_round[k]);
        if (rt.kind == H2_ROUND_TABLE) {
          const int tsw = h2_swz<RB>(ftid());
#pragma unroll
          for (int q = 0; q < R; ++q) tf[tsw ^ lin_sw[q]] = k == 0 ? tacc0[TABC ? q : 0] : tacc1[TABC ? q : 0];
        } else {
          const int sl = h2_swz<RB>(round_lbase(rt.rb));
#pragma unroll
          for (int q = 0; q < R; ++q) {
            int o = 0;
#pragma unroll
            for (int j = 0; j < RB; ++j) o |= ((q >> j) & 1) << rt.rb[j];
            tf[sl ^ h2_swz<RB>(o)] = k == 0 ? tacc0[TABC ? q : 0] : tacc1[TABC ? q : 0];
          }
        }
      }
      H2_SYNC();
      for (int grp = 0; grp < nloc / RB; ++grp) {
        int rbg[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < RB; ++j) rbg[j] = grp * RB + j;
        const int sl2 = h2_swz<RB>(round_lbase(rbg));
        float u[R];
#pragma unroll
        for (int q = 0; q < R; ++q) u[q] = tf[sl2 ^ h2_swz<RB>(q << (grp * RB))];
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
#pragma unroll
        for (int q = 0; q < R; ++q) tf[sl2 ^ h2_swz<RB>(q << (grp * RB))] = u[q];
        H2_SYNC();
      }
      for (int j = tid; j < A.nc; j += NT) A.dpart[((size_t)k * nblk + blk) * A.nc + j] = tf[h2_swz<RB>(A.wht_idx[j])];
    }
  }
}

// ---------------------------------------------------------------- small folds
// qjets[c][w][p] = sum_tau xpart[c][pt][tau][n-1-w]  (+ the 2<chi_k|Z|chi_k> sums for c = 4, 5)
template <int NCH>
__global__ void __launch_bounds__(256) k_h2_expval_fold(const float* __restrict__ xpart, int64_t pt_stride, int ntau, int n,
                                                        int64_t B, int64_t p_first, int64_t npts, float* __restrict__ qjets) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)NCH * npts * n) return;
  const int w = (int)(i % n);
  const int64_t pt = (i / n) % npts;
  const int c = (int)(i / ((int64_t)n * npts));
  float s = 0.f;
  for (int q = 0; q < ntau; ++q) s += xpart[(((size_t)c * pt_stride + pt) * ntau + q) * H2_XW + (n - 1 - w)];
  if (NCH == 6 && c >= 4)
    for (int q = 0; q < ntau; ++q) s += xpart[(((size_t)(c + 2) * pt_stride + pt) * ntau + q) * H2_XW + (n - 1 - w)];
  qjets[((int64_t)c * n + w) * B + p_first + pt] = s;
}

// theta-gradient columns of the tile rows: cleared, then every contribution writes its own slots
__global__ void k_h2_zero_rows(float* __restrict__ part, int64_t part_stride, int64_t row0, int64_t ntiles, int n_params) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntiles * n_params) return;
  part[(row0 + i / n_params) * part_stride + i % n_params] = 0.f;
}

// in-round gates: part[row0 + tile][slot_i] += sum over the tile's (t, tau) blocks of gpart[i][.]
__global__ void __launch_bounds__(256) k_h2_fold_gates(const float* __restrict__ gpart, int64_t nblk, int per_tile,
                                                       const int* __restrict__ slots, float* __restrict__ part,
                                                       int64_t part_stride, int64_t row0) {
  __shared__ float s_red[4];
  const int i = blockIdx.x;
  const int64_t tile = blockIdx.y;
  const float* src = gpart + (size_t)i * nblk + tile * per_tile;
  float t = 0.f;
  for (int b = threadIdx.x; b < per_tile; b += 256) t += src[b];
  const float w = qc_wave_sum(t);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) part[(row0 + tile) * part_stride + slots[i]] += (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// diagonal tables: per tile row, sum the blocks' Walsh-Hadamard coefficients over the 64 points (per tau), then every
// gate of the table takes its combination.  With s_b(k) = (-1)^{bit_b(k)}:
//   RZ(b):        dtheta = sum_k s_b t_k
//   CRZ(c -> b):  dtheta = sum_k [bit_c = 1] s_b t_k = (sum s_b t - sum s_c s_b t) / 2
// and sum_k s_m t_k over a tile = (+-) W[mask of the local bits of m], the sign from the tile's non-local bits.
__global__ void __launch_bounds__(256) k_h2_fold_diag(const float* __restrict__ dpart, int64_t nblk, int ntau, int nc,
                                                      H2Stage sd, const H2DiagGate* __restrict__ dg, int ng,
                                                      float* __restrict__ part, int64_t part_stride, int64_t row0) {
  extern __shared__ float s_w[];   // [ntau][nc]
  const int64_t tile = blockIdx.x;
  for (int i = threadIdx.x; i < ntau * nc; i += 256) {
    const int tau = i / nc, j = i % nc;
    float s = 0.f;
    for (int t = 0; t < 64; ++t) s += dpart[(((size_t)tile * 64 + t) * ntau + tau) * nc + j];
    s_w[i] = s;
  }
  __syncthreads();
  const int nloc = sd.nloc;
  auto posof = [&](int b) {
    for (int j = 0; j < nloc; ++j)
      if (sd.lb[j] == b) return j;
    return -1;
  };
  auto gidx = [&](int b) {
    for (int j = 0; j < sd.ngb; ++j)
      if (sd.gb[j] == b) return j;
    return 0;
  };
  // coefficient index of a local mask of weight <= 2: 0 | 1 + p | 1 + nloc + pair rank (p < q, p outer)
  auto cidx = [&](int p, int q) {
    if (p < 0) return 0;
    if (q < 0) return 1 + p;
    if (p > q) { const int x = p; p = q; q = x; }
    return 1 + nloc + p * nloc - p * (p + 1) / 2 + (q - p - 1);
  };
  for (int g = threadIdx.x; g < ng; g += 256) {
    const H2DiagGate d = dg[g];
    const int pt_ = posof(d.bt), pc_ = d.bc >= 0 ? posof(d.bc) : -1;
    float acc = 0.f;
    for (int tau = 0; tau < ntau; ++tau) {
      const float* W = s_w + (size_t)tau * nc;
      const float st = pt_ >= 0 ? 1.f : (((tau >> gidx(d.bt)) & 1) ? -1.f : 1.f);
      const float a = st * W[cidx(pt_, -1)];
      if (d.bc < 0) {
        acc += a;
      } else {
        const float sc = pc_ >= 0 ? 1.f : (((tau >> gidx(d.bc)) & 1) ? -1.f : 1.f);
        const float b = st * sc * W[pt_ >= 0 && pc_ >= 0 ? cidx(pt_, pc_) : (pt_ >= 0 ? cidx(pt_, -1) : (pc_ >= 0 ? cidx(pc_, -1) : 0))];
        acc += 0.5f * (a - b);
      }
    }
    part[(row0 + tile) * part_stride + d.slot] += acc;
  }
}

// cotangents of the angle jets from the un-embedded, sparse lam of every channel (block = one point).
// The positions of the needed amplitudes mu[k] (k of weight 1, 2, 3) inside the block's copy X[tau][rank] come from
// tables built once per plan on the host (qc_h2_create): pos1[w] = E(w), pos2[w][v] = E(w) | E(v), pos3[w][pair (u < v)]
// = E(u) ^ E(v) ^ E(w) - no per-lookup bit gathering - and the 120 pair terms of a wire are spread over the eight lanes
// of its lane group (round 2 walked them on one lane per wire with 16 of 128 lanes busy: 40 k instructions per wave).
template <int NCH>
__global__ void __launch_bounds__(128) k_h2_abar(const Cplx* __restrict__ xi, Cplx* __restrict__ xwork, int64_t pt_stride,
                                                 int ntau, int nx, H2Stage sd, int n, const float* __restrict__ wd,
                                                 const int* __restrict__ postab, int64_t B, int64_t p_first,
                                                 int64_t npts, float* __restrict__ abar, int use_lds) {
  extern __shared__ Cplx s_x[];          // [ntau][nx] when it fits, else the block works in xwork (global)
  __shared__ float s_buf[NCH][3][24];
  __shared__ float s_da[24], s_dda[24];
  const int64_t pt = blockIdx.x;
  const int64_t p = p_first + pt;
  const int tid = threadIdx.x;
  const float* wdp = wd + (size_t)pt * n * 8;
  const int npair = n * (n - 1) / 2;
  const int* pos1 = postab;                       // [n]
  const int* pos2 = postab + n;                   // [n][n]
  const int* pos3 = postab + n + n * n;           // [n][npair]
  const int* pu = pos3 + n * npair;               // [npair] u of pair
  const int* pv = pu + npair;                     // [npair] v of pair
  for (int c = 0; c < NCH; ++c) {
    const Cplx* src = xi + ((size_t)c * pt_stride + pt) * ntau * nx;
    Cplx* X = use_lds ? s_x : xwork + (size_t)pt * ntau * nx;
    for (int i = tid; i < ntau * nx; i += 128) X[i] = src[i];
    const int dsel = c == 0 ? 0 : (c <= 3 ? c - 1 : c - 3);
    if (tid < n) {
      s_da[tid] = c >= 1 ? wdp[(size_t)tid * 8 + 2 + dsel] : 0.f;
      s_dda[tid] = c >= 4 ? wdp[(size_t)tid * 8 + 5 + (c - 4)] : 0.f;
    }
    __syncthreads();
    // RX^dagger on the wires of the non-local bits: butterflies across tau
    for (int j = 0; j < sd.ngb; ++j) {
      const float* w8 = wdp + (size_t)(n - 1 - sd.gb[j]) * 8;
      const float cw = w8[0], sw = w8[1];
      for (int i = tid; i < (ntau >> 1) * nx; i += 128) {
        const int h = i / nx, e = i % nx;
        const int t0_ = ((h >> j) << (j + 1)) | (h & ((1 << j) - 1)), t1_ = t0_ | (1 << j);
        const Cplx a = X[(size_t)t0_ * nx + e], b = X[(size_t)t1_ * nx + e];
        // RX^dagger = [[c, i s], [i s, c]]
        X[(size_t)t0_ * nx + e] = {cw * a.re - sw * b.im, cw * a.im + sw * b.re};
        X[(size_t)t1_ * nx + e] = {cw * b.re - sw * a.im, cw * b.im + sw * a.re};
      }
      __syncthreads();
    }
    // wire w = lane group (8 lanes); its lanes share the sums over the other wires / pairs of wires
    for (int w = tid >> 3; w < n; w += 16) {
      const int sub = tid & 7;
      const float mu0re = X[0].re;                       // mu[0]: index 0 is rank 0 of tau 0
      const Cplx m1 = X[pos1[w]];
      float acc1 = 0.f, accb = 0.f, acca = 0.f;
      if (c >= 1)
        for (int v = sub; v < n; v += 8)
          if (v != w) acc1 = fmaf(s_da[v], X[pos2[w * n + v]].re, acc1);
      if (c >= 4) {
        for (int u = sub; u < n; u += 8)
          if (u != w) accb = fmaf(s_dda[u], X[pos2[w * n + u]].re, accb);
        for (int q = sub; q < npair; q += 8)
          acca = fmaf(2.f * s_da[pu[q]] * s_da[pv[q]], X[pos3[w * npair + q]].im, acca);
      }
      // sum over the lane group's 8 lanes (quad_perm, quad_perm, row_half_mirror: lanes 0..7 of each half row)
#define ABAR_SUM8_(x)                                                                                                         \
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));          \
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));          \
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
      ABAR_SUM8_(acc1) ABAR_SUM8_(accb) ABAR_SUM8_(acca)
#undef ABAR_SUM8_
      if (sub == 0) {
        const float ip0 = -m1.im;
        float ip1 = 0.f, ip2 = 0.f;
        if (c >= 1) ip1 = -0.5f * (s_da[w] * mu0re + acc1);
        if (c >= 4) {
          float S = 0.f;
          for (int v = 0; v < n; ++v) S = fmaf(s_da[v], s_da[v], S);
          const float a = S * m1.im + acca;
          const float b = s_dda[w] * mu0re + accb;
          ip2 = 0.25f * a - 0.5f * b;
        }
        // slots as in the register family's tail (qc_circuit_reg_kernels.h): [0] own-order term, [1], [2] lower orders
        if (c == 0) {
          s_buf[c][0][w] = ip0;
        } else if (c <= 3) {
          s_buf[c][0][w] = ip1;
          s_buf[c][1][w] = ip0;
        } else {
          s_buf[c][0][w] = ip2;
          s_buf[c][1][w] = 2.f * ip1;
          s_buf[c][2][w] = ip0;
        }
      }
    }
    __syncthreads();
  }
  if (tid < n) {
    const int w = tid;
    if constexpr (NCH == 6) {
      abar[((int64_t)0 * n + w) * B + p] = ((s_buf[0][0][w] + s_buf[1][0][w]) + (s_buf[2][0][w] + s_buf[3][0][w])) + (s_buf[4][0][w] + s_buf[5][0][w]);
      abar[((int64_t)1 * n + w) * B + p] = s_buf[1][1][w];
      abar[((int64_t)2 * n + w) * B + p] = s_buf[2][1][w] + s_buf[4][1][w];
      abar[((int64_t)3 * n + w) * B + p] = s_buf[3][1][w] + s_buf[5][1][w];
      abar[((int64_t)4 * n + w) * B + p] = s_buf[4][2][w];
      abar[((int64_t)5 * n + w) * B + p] = s_buf[5][2][w];
    } else {
      abar[(int64_t)w * B + p] = s_buf[0][0][w];
    }
  }
}

// folded RX layer: d L / d theta_w = sum over the tile's points of d L / d a_w (value-channel row w of abar)
__global__ void __launch_bounds__(64) k_h2_absorb_grad(const float* __restrict__ abar, int64_t B, int64_t p_first,
                                                       const QcGate* __restrict__ prog, float* __restrict__ part,
                                                       int64_t part_stride, int64_t row0) {
  const int w = blockIdx.x;
  const int64_t tile = blockIdx.y;
  const int64_t p = p_first + tile * 64 + threadIdx.x;
  float v = p < B ? abar[(int64_t)w * B + p] : 0.f;
  v = qc_wave_sum_to_lane63(v);
  if (threadIdx.x == 63) part[(row0 + tile) * part_stride + prog[w].slot] += v;
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct H2Ws {
  Cplx* store;
  float* wd;
  Cplx* tabs;
  Cplx* rph;
  float* xpart;
  float* gpart;
  float* dpart;
  Cplx* xi;
  Cplx* xwork;
};

}  // namespace

// ------------------------------------------------------------------ host side
struct QcH2 {
  H2Dev dev;
};

static int h2_max_np(const H2Plan& P) {
  int m = 0;
  for (const H2Stage& s : P.stages) m = s.np > m ? s.np : m;
  return m;
}
static int h2_max_ntau(const H2Plan& P) {
  int m = 1;
  for (const H2Stage& s : P.stages) m = (1 << s.ngb) > m ? (1 << s.ngb) : m;
  return m;
}

// 12-bit tiles: 8 amplitudes per thread x 512 threads by default (four waves per SIMD without spills in the
// backward kernels); QC_H2_RB=4 selects 16 amplitudes x 256 threads (fewer rounds, two waves per SIMD)
static int h2_rb12() {
  static const int rb12 = [] { const char* e = getenv("QC_H2_RB"); return (e && e[0] == '4') ? 4 : 3; }();
  return rb12;
}

// generated plans (gen/qc_static_h2_table.hip)
struct QcStaticH2Entry {
  int n_qubits, n_gates, absorb, rb;
  const int* gates;      // n_gates x 4: (op, ba, bb, slot) in the device encoding
  const int* describe;   // h2_describe() of the plan the kernels were generated from
  int n_describe;
  const H2sLaunchers* launch;
};
extern const QcStaticH2Entry qc_static_h2_table[];
extern const int qc_static_h2_count;

// the generated entry for this program, or -1.  QC_NO_STATIC=1: none; QC_H2S_RB=3|4: only entries of that geometry
static int h2s_match(const qc_program* pg, int absorb) {
  static const bool off = [] { const char* e = getenv("QC_NO_STATIC"); return e && e[0] == '1'; }();
  static const int want_rb = [] { const char* e = getenv("QC_H2S_RB"); return e ? atoi(e) : 0; }();
  if (off) return -1;
  for (int i = 0; i < qc_static_h2_count; ++i) {
    const QcStaticH2Entry& e = qc_static_h2_table[i];
    if (e.n_qubits != pg->n_qubits || e.n_gates != pg->n_gates || e.absorb != absorb) continue;
    if (want_rb && e.rb != want_rb) continue;
    bool same = true;
    for (int g = 0; g < pg->n_gates && same; ++g) {
      const QcGate& a = pg->h_gates[g];
      const int* b = e.gates + 4 * g;
      same = a.op == b[0] && a.ba == b[1] && a.bb == b[2] && a.slot == b[3];
    }
    if (same) return i;
  }
  return -1;
}

// fused RZ runs in plan order: the run-time mirror of H2sRz (qc_circuit_h2s_kernels.h); H2S_PH_DESC ints per run
static std::vector<int> h2s_enumerate_rz_runs(const H2Plan& P) {
  std::vector<int> desc;
  auto rz1 = [&](int g) { return P.gates[g].kind == H2_K_REG1 && P.gates[g].op == QC_RZ; };
  for (const H2Round& r : P.rounds) {
    if (r.kind != H2_ROUND_GATES) continue;
    for (int g = r.g0; g < r.g0 + r.ng;) {
      if (!rz1(g)) { ++g; continue; }
      int e = g;
      while (e < r.g0 + r.ng && rz1(e)) ++e;
      if (e - g >= 2) {
        std::vector<int> d(H2S_PH_DESC, 0);
        d[0] = e - g;
        for (int j = 0; j < e - g && j < 8; ++j) {
          d[1 + 2 * j] = P.gates[g + j].gi;
          d[2 + 2 * j] = P.gates[g + j].tq;
        }
        desc.insert(desc.end(), d.begin(), d.end());
      }
      g = e;
    }
  }
  return desc;
}

void* qc_h2_create(const qc_program* pg, int absorb) {
  QcH2* h = new QcH2();
  H2Dev& D = h->dev;
  const int sid = pg->amplitude ? -1 : h2s_match(pg, absorb);   // (the generated programs embed angles)
  if (sid >= 0) {
    // the kernels were generated from the plan of the SAME planner at build time: trust them only if the plan built
    // now is identical, record by record
    const QcStaticH2Entry& e = qc_static_h2_table[sid];
    D.plan = h2_make_plan(pg->h_gates, pg->n_gates, pg->n_qubits, absorb, e.rb);
    const std::vector<int> d = h2_describe(D.plan);
    if ((int)d.size() == e.n_describe && memcmp(d.data(), e.describe, sizeof(int) * d.size()) == 0) D.stat = e.launch;
  }
  if (!D.stat) D.plan = h2_make_plan(pg->h_gates, pg->n_gates, pg->n_qubits, absorb, h2_rb12());
  const H2Plan& P = D.plan;
  const int nl0 = P.stages[0].nloc;
  std::vector<int> rank((size_t)1 << nl0, -1);
  for (size_t j = 0; j < P.sparse_idx.size(); ++j) rank[P.sparse_idx[j]] = (int)j;
  D.nx = (int)P.sparse_idx.size();
  // coefficient masks in the order k_h2_fold_diag indexes them (nloc = the largest over the stages; stages share T)
  std::vector<int> wht;
  {
    const int nl = nl0;   // every stage has the same nloc (= min(n, T))
    wht.push_back(0);
    for (int p = 0; p < nl; ++p) wht.push_back(1 << p);
    for (int p = 0; p < nl; ++p)
      for (int q = p + 1; q < nl; ++q) wht.push_back((1 << p) | (1 << q));
  }
  D.nc = (int)wht.size();
  std::vector<int> pslots;
  D.pslot_off.push_back(0);
  for (const H2Stage& s : P.stages) {
    std::vector<int> sl(s.np, 0);
    for (int r = s.r0; r < s.r0 + s.nr; ++r)
      if (P.rounds[r].kind == H2_ROUND_GATES)
        for (int g = P.rounds[r].g0; g < P.rounds[r].g0 + P.rounds[r].ng; ++g)
          if (P.gates[g].pidx >= 0) sl[P.gates[g].pidx] = P.gates[g].slot;
    pslots.insert(pslots.end(), sl.begin(), sl.end());
    D.pslot_off.push_back((int)pslots.size());
  }
  bool ok = true;
  auto up = [&](const void* src, size_t bytes, void** dst) {
    if (bytes == 0) { *dst = nullptr; return; }
    ok = ok && hipMalloc(dst, bytes) == hipSuccess;
    ok = ok && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  up(P.rounds.data(), sizeof(H2Round) * P.rounds.size(), (void**)&D.d_rounds);
  up(P.gates.data(), sizeof(H2Gate) * P.gates.size(), (void**)&D.d_gates);
  up(P.dgates.data(), sizeof(H2DiagGate) * P.dgates.size(), (void**)&D.d_dgates);
  up(P.sparse_idx.data(), sizeof(int) * P.sparse_idx.size(), (void**)&D.d_sparse);
  up(rank.data(), sizeof(int) * rank.size(), (void**)&D.d_rank);
  {
    // where the amplitude of full index k sits in a point's copy X[tau][rank] (first stage's tile geometry)
    const H2Stage& s0 = P.stages[0];
    const int n = pg->n_qubits, npair = n * (n - 1) / 2;
    auto posof = [&](int64_t k) {
      int l = 0, g = 0;
      for (int j = 0; j < s0.nloc; ++j)
        if ((k >> s0.lb[j]) & 1) l |= 1 << j;
      for (int j = 0; j < s0.ngb; ++j)
        if ((k >> s0.gb[j]) & 1) g |= 1 << j;
      return g * D.nx + rank[l];
    };
    auto E = [&](int w) { return (int64_t)1 << (n - 1 - w); };
    std::vector<int> tab((size_t)n + (size_t)n * n + (size_t)n * npair + 2 * (size_t)npair, 0);
    int* pos1 = tab.data();
    int* pos2 = pos1 + n;
    int* pos3 = pos2 + n * n;
    int* pu = pos3 + n * npair;
    int* pv = pu + npair;
    int q = 0;
    for (int u = 0; u < n; ++u)
      for (int v = u + 1; v < n; ++v, ++q) {
        pu[q] = u;
        pv[q] = v;
      }
    for (int w = 0; w < n; ++w) {
      pos1[w] = posof(E(w));
      for (int v = 0; v < n; ++v) pos2[w * n + v] = v == w ? 0 : posof(E(w) | E(v));
      for (int qq = 0; qq < npair; ++qq) pos3[w * npair + qq] = posof(E(pu[qq]) ^ E(pv[qq]) ^ E(w));
    }
    up(tab.data(), sizeof(int) * tab.size(), (void**)&D.d_postab);
  }
  up(wht.data(), sizeof(int) * wht.size(), (void**)&D.d_whtidx);
  up(pslots.data(), sizeof(int) * pslots.size(), (void**)&D.d_pslots);
  if (D.stat) {
    const std::vector<int> ph = h2s_enumerate_rz_runs(P);
    D.n_ph = (int)ph.size() / H2S_PH_DESC;
    up(ph.data(), sizeof(int) * ph.size(), (void**)&D.d_phdesc);
  }
  if (!ok) {
    qc_h2_destroy(h);
    return nullptr;
  }
  return h;
}

void qc_h2_destroy(void* hp) {
  if (!hp) return;
  QcH2* h = (QcH2*)hp;
  H2Dev& D = h->dev;
  void* ptrs[] = {D.d_rounds, D.d_gates, D.d_dgates, D.d_sparse, D.d_rank, D.d_postab, D.d_whtidx, D.d_pslots, D.d_phdesc};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete h;
}

int qc_h2_describe_gates(const QcGate* gates, int n_gates, int n_qubits, int absorb, int32_t* out, int cap) {
  const std::vector<int> d = h2_describe(h2_make_plan(gates, n_gates, n_qubits, absorb, h2_rb12()));
  if (out)
    for (int i = 0; i < cap && i < (int)d.size(); ++i) out[i] = d[i];
  return (int)d.size();
}

// bytes that do not depend on the number of resident tiles, and bytes per resident 64-point tile
static size_t h2_fixed_bytes(const qc_program* pg, const H2Dev& D) {
  const size_t N = (size_t)1 << pg->n_qubits;
  return al(sizeof(Cplx) * D.plan.tables.size() * N + 256) + al(sizeof(Cplx) * ((size_t)D.n_ph << D.plan.rbits) + 256);
}
static size_t h2_tile_bytes(const qc_program* pg, const H2Dev& D, int nch, bool backward) {
  const size_t N = (size_t)1 << pg->n_qubits;
  const size_t ntau = h2_max_ntau(D.plan);
  size_t b = al(sizeof(Cplx) * (backward ? 2 : 1) * nch * 64 * N);               // slot
  b += al(sizeof(float) * 64 * pg->n_qubits * 8);                                // wd
  b += al(sizeof(float) * 8 * 64 * ntau * H2_XW);                                // xpart
  if (backward) {
    b += al(sizeof(float) * (size_t)(h2_max_np(D.plan) + 1) * 64 * ntau);        // gpart
    b += al(sizeof(float) * H2_MAXTAB * 64 * ntau * D.nc);                       // dpart
    b += al(sizeof(Cplx) * (size_t)nch * 64 * ntau * D.nx);                      // xi
    b += al(sizeof(Cplx) * 64 * ntau * D.nx);                                    // xwork
  }
  return b;
}

size_t qc_h2_bytes(const qc_program* pg, void* hp, int nch, bool backward, int64_t tiles) {
  const H2Dev& D = ((QcH2*)hp)->dev;
  return h2_fixed_bytes(pg, D) + (size_t)tiles * h2_tile_bytes(pg, D, nch, backward);
}
int64_t qc_h2_tiles_that_fit(const qc_program* pg, void* hp, int nch, bool backward, size_t ws_bytes) {
  const H2Dev& D = ((QcH2*)hp)->dev;
  const size_t f = h2_fixed_bytes(pg, D);
  if (ws_bytes <= f) return 0;
  return (int64_t)((ws_bytes - f) / h2_tile_bytes(pg, D, nch, backward));
}

static H2Ws h2_carve(const qc_program* pg, const H2Dev& D, int nch, bool backward, int64_t G, void* ws) {
  const size_t N = (size_t)1 << pg->n_qubits;
  const size_t ntau = h2_max_ntau(D.plan);
  char* p = (char*)ws;
  H2Ws w = {};
  w.tabs = (Cplx*)p; p += al(sizeof(Cplx) * D.plan.tables.size() * N + 256);
  w.rph = (Cplx*)p; p += al(sizeof(Cplx) * ((size_t)D.n_ph << D.plan.rbits) + 256);
  w.wd = (float*)p; p += (size_t)G * al(sizeof(float) * 64 * pg->n_qubits * 8);
  w.store = (Cplx*)p; p += (size_t)G * al(sizeof(Cplx) * (backward ? 2 : 1) * nch * 64 * N);
  w.xpart = (float*)p; p += (size_t)G * al(sizeof(float) * 8 * 64 * ntau * H2_XW);
  if (backward) {
    w.gpart = (float*)p; p += (size_t)G * al(sizeof(float) * (size_t)(h2_max_np(D.plan) + 1) * 64 * ntau);
    w.dpart = (float*)p; p += (size_t)G * al(sizeof(float) * H2_MAXTAB * 64 * ntau * D.nc);
    w.xi = (Cplx*)p; p += (size_t)G * al(sizeof(Cplx) * (size_t)nch * 64 * ntau * D.nx);
    w.xwork = (Cplx*)p;
  }
  return w;
}

template <int RB, int NCH, int MODE, int ROLE>
static void h2_launch_stage(const H2Args& A, int64_t npts64, hipStream_t st) {
  const int nloc = A.sd.nloc;
  const int NT = (1 << nloc) >> RB;
  const size_t sh = sizeof(Cplx) * ((size_t)1 << nloc) * (MODE == 1 ? 2 : 1);
  // the attribute is per device: set before every launch (cheap), not once per process
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_h2_stage<RB, NCH, MODE, ROLE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  const unsigned grid = (unsigned)(npts64 << A.sd.ngb);
  hipLaunchKernelGGL((k_h2_stage<RB, NCH, MODE, ROLE>), dim3(grid), dim3(NT), sh, st, A);
}
// tile bits 12 (RB = 3, 4): one instantiation per role; the small-n tiles (RB = 2, and RB = 3 at n = 9) take the
// general form
template <int RB, int NCH, int MODE>
static void h2_launch_role(const H2Args& A, int64_t npts64, hipStream_t st) {
  const int role = (A.last ? 1 : 0) | ((MODE == 1 && A.sd.ntab > 0) ? 2 : 0);
  if (A.sd.nloc < 12) return h2_launch_stage<RB, NCH, MODE, 3>(A, npts64, st);
  if constexpr (MODE == 0) {
    if (role & 1) h2_launch_stage<RB, NCH, 0, 1>(A, npts64, st);
    else h2_launch_stage<RB, NCH, 0, 0>(A, npts64, st);
  } else {
    switch (role) {
      case 0: h2_launch_stage<RB, NCH, 1, 0>(A, npts64, st); break;
      case 1: h2_launch_stage<RB, NCH, 1, 1>(A, npts64, st); break;
      case 2: h2_launch_stage<RB, NCH, 1, 2>(A, npts64, st); break;
      default: h2_launch_stage<RB, NCH, 1, 3>(A, npts64, st); break;
    }
  }
}
template <int NCH, int MODE>
static void h2_launch_stage_rb(int rb, const H2Args& A, int64_t npts64, hipStream_t st) {
  if (rb == 4) h2_launch_role<4, NCH, MODE>(A, npts64, st);
  else if (rb == 3) h2_launch_role<3, NCH, MODE>(A, npts64, st);
  else h2_launch_stage<2, NCH, MODE, 3>(A, npts64, st);
}

// One group of resident tiles [p_first, p_first + npts): forward and / or backward.
template <int NCH>
static void h2_group(const qc_program* pg, const H2Dev& D, const QcTrig* trig, const float* umat, const H2Ws& w, int64_t B,
                     int64_t p_first, int64_t npts, bool do_fwd, bool keep_final, const float* ajets, float* qjets,
                     const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, hipStream_t st) {
  const H2Plan& P = D.plan;
  const int n = pg->n_qubits;
  const size_t N = (size_t)1 << n;
  const int64_t ntiles = qc_ceil_div(npts, 64);
  const int64_t npts64 = ntiles * 64;
  const bool backward = qbar != nullptr;
  H2Args A = {};
  A.store = w.store;
  A.slot_elems = (int64_t)((backward || keep_final ? 2 : 1) * NCH * 64 * N);
  A.B = B;
  A.p_first = p_first;
  A.pt_stride = npts64;
  A.n = n;
  A.rounds = D.d_rounds;
  A.gates = D.d_gates;
  A.trig = trig;
  A.umat = umat;
  A.tabs = w.tabs;
  A.rph = w.rph;
  A.wd = w.wd;
  A.xpart = w.xpart;
  A.qbar = qbar;
  A.gpart = w.gpart;
  A.dpart = w.dpart;
  A.xi = w.xi;
  A.sparse_idx = D.d_sparse;
  A.wht_idx = D.d_whtidx;
  A.nx = D.nx;
  A.nc = D.nc;
  const int S = (int)P.stages.size();
  const bool amp = pg->amplitude != 0;
  A.amp = amp ? 1 : 0;
  A.ajets = ajets;
  A.abar = abar;
  if (do_fwd) {
    if (!amp)
      hipLaunchKernelGGL((k_h2_wiredata<NCH>), dim3(qc_ceil_div(npts * n, 256)), dim3(256), 0, st, ajets, B, p_first, npts, n, w.wd,
                         trig, P.absorb);
    for (int i = 0; i < S; ++i) {
      A.sd = P.stages[i];
      A.first = i == 0;
      A.last = i == S - 1;
      A.keep_final = keep_final ? 1 : 0;
      if (!(D.stat && D.stat->stage(NCH, 0, i, A, npts64, st))) h2_launch_stage_rb<NCH, 0>(P.rbits, A, npts64, st);
    }
    if (qjets) {
      const int ntau = 1 << P.stages[S - 1].ngb;
      hipLaunchKernelGGL((k_h2_expval_fold<NCH>), dim3(qc_ceil_div((int64_t)NCH * npts * n, 256)), dim3(256), 0, st, w.xpart,
                         npts64, ntau, n, B, p_first, npts, qjets);
    }
  }
  if (!backward) return;
  const int np_all = pg->n_params > 0 ? pg->n_params : 1;
  hipLaunchKernelGGL(k_h2_zero_rows, dim3(qc_ceil_div(ntiles * np_all, 256)), dim3(256), 0, st, part, part_stride, row0, ntiles,
                     pg->n_params);
  for (int i = S - 1; i >= 0; --i) {
    A.sd = P.stages[i];
    A.first = i == 0;
    A.last = i == S - 1;
    if (!(D.stat && D.stat->stage(NCH, 1, i, A, npts64, st))) h2_launch_stage_rb<NCH, 1>(P.rbits, A, npts64, st);
    const int ntau = 1 << A.sd.ngb;
    if (A.sd.np > 0)
      hipLaunchKernelGGL(k_h2_fold_gates, dim3(A.sd.np, (unsigned)ntiles), dim3(256), 0, st, w.gpart, npts64 * ntau, 64 * ntau,
                         D.d_pslots + D.pslot_off[i], part, part_stride, row0);
    for (int k = 0; k < A.sd.ntab; ++k) {
      const H2Table& tb = P.tables[A.sd.tab[k]];
      hipLaunchKernelGGL(k_h2_fold_diag, dim3((unsigned)ntiles), dim3(256), sizeof(float) * ntau * D.nc, st,
                         w.dpart + (size_t)k * npts64 * ntau * D.nc, npts64 * ntau, ntau, D.nc, A.sd, D.d_dgates + tb.g0, tb.ng, part,
                         part_stride, row0);
    }
  }
  if (!amp) {
    const H2Stage& s0 = P.stages[0];
    const int ntau = 1 << s0.ngb;
    const size_t xb = sizeof(Cplx) * (size_t)ntau * D.nx;
    const int use_lds = xb <= 64 * 1024 ? 1 : 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_h2_abar<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipLaunchKernelGGL((k_h2_abar<NCH>), dim3((unsigned)npts), dim3(128), use_lds ? xb : 0, st, w.xi, w.xwork, npts64, ntau, D.nx, s0,
                       n, w.wd, D.d_postab, B, p_first, npts, abar, use_lds);
    if (P.absorb)
      hipLaunchKernelGGL(k_h2_absorb_grad, dim3(n, (unsigned)ntiles), dim3(64), 0, st, abar, B, p_first, pg->d_gates, part,
                         part_stride, row0);
  }
}

// Forward and / or backward over B points with the workspace `ws`.  `resident`: the caller ran the forward pass of
// the SAME batch into the same workspace with keep = true and everything fitted -> the backward pass starts from it.
template <int NCH>
static int h2_run(const qc_program* pg, void* hp, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                  const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B, void* ws,
                  size_t ws_bytes, bool keep, bool resident, hipStream_t st) {
  const H2Dev& D = ((QcH2*)hp)->dev;
  const bool backward = qbar != nullptr;
  const bool two = backward || keep;
  const int64_t ntiles = qc_ceil_div(B, 64);
  int64_t G = qc_h2_tiles_that_fit(pg, hp, NCH, two, ws_bytes);
  if (!ws || G < 1) return QC_ERR_ARG;
  if (G > ntiles) G = ntiles;
  if (resident && G < ntiles) return QC_ERR_ARG;
  const H2Ws w = h2_carve(pg, D, NCH, two, G, ws);
  const int n = pg->n_qubits;
  const size_t N = (size_t)1 << n;
  if (!resident) {
    for (size_t k = 0; k < D.plan.tables.size(); ++k)
      hipLaunchKernelGGL(k_h2_diag_table, dim3(qc_ceil_div((int64_t)N, 256)), dim3(256), 0, st, D.d_dgates + D.plan.tables[k].g0,
                         D.plan.tables[k].ng, trig, n, w.tabs + k * N);
  }
  if (!resident && D.stat && D.n_ph > 0)
    hipLaunchKernelGGL(k_h2s_round_phases, dim3(qc_ceil_div((int64_t)D.n_ph << D.plan.rbits, 64)), dim3(64), 0, st, D.d_phdesc, D.n_ph,
                       1 << D.plan.rbits, trig, w.rph);
  for (int64_t t0 = 0; t0 < ntiles; t0 += G) {
    const int64_t p_first = t0 * 64;
    const int64_t npts = (B - p_first) < G * 64 ? (B - p_first) : G * 64;
    h2_group<NCH>(pg, D, trig, umat, w, B, p_first, npts, !resident, backward || keep, ajets, qjets, qbar, abar, part,
                  part_stride, row0 + t0, st);
  }
  return QC_OK;
}

int qc_h2_forward(const qc_program* pg, void* hp, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                  int64_t B, int nch, void* ws, size_t ws_bytes, bool keep, hipStream_t st) {
  if (nch == 6) return h2_run<6>(pg, hp, trig, umat, ajets, qjets, nullptr, nullptr, nullptr, 0, 0, B, ws, ws_bytes, keep, false, st);
  return h2_run<1>(pg, hp, trig, umat, ajets, qjets, nullptr, nullptr, nullptr, 0, 0, B, ws, ws_bytes, keep, false, st);
}
int qc_h2_backward(const qc_program* pg, void* hp, const QcTrig* trig, const float* umat, const float* ajets, const float* qbar,
                   float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B, int nch, void* ws, size_t ws_bytes,
                   bool resident, hipStream_t st) {
  if (nch == 6)
    return h2_run<6>(pg, hp, trig, umat, ajets, nullptr, qbar, abar, part, part_stride, row0, B, ws, ws_bytes, false, resident, st);
  return h2_run<1>(pg, hp, trig, umat, ajets, nullptr, qbar, abar, part, part_stride, row0, B, ws, ws_bytes, false, resident, st);
}
