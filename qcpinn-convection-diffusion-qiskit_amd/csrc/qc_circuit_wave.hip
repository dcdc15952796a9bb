// Variational-circuit kernels, "wave" family: lanes = amplitudes.  Serves every qubit count the
// register family does not (n = 1 and 6 <= n <= 8), and is the generic cross-check for n <= 5.
//
// Layout: amplitude index k of one statevector = (sub << LR) | r, where `sub` is the lane inside a
// group of G = 2^min(n,6) lanes and r indexes R = 2^LR amplitudes held in that lane's registers
// (LR = max(0, n-6)).  For n >= 6 one wave carries one collocation point; below that a wave carries
// 64/G points.  ALL derivative channels of a point live in the same lanes, so the bilinear <Z> forms
// and their cotangents are lane-local; a gate on a lane bit pairs amplitudes with __shfl_xor
// (ds_bpermute, the LDS crossbar, no LDS storage), a gate on a register bit is lane-local, and
// <Z_w> is a butterfly reduction over the group's lanes.
//
// The gate PROGRAM is run-time data (any ansatz); the opcode selects (wave-uniform branch) an
// op-specialised body: a' = c1*a + s1*partner with per-lane REAL coefficients that fold in the control
// predicate and the sign given by this amplitude's target bit: 2 mul + 2 fma per amplitude, no select;
// wires stay run-time values (lane-bit masks for __shfl_xor, register-bit switch).
//
// A block is 4 waves = one 64-point tile (one gradient partial row); each wave walks its 16 points.
#include "qc_internal.h"

#include <type_traits>

namespace {

template <int LR>
struct WV {  // this lane's share of one statevector
  float re[1 << LR];
  float im[1 << LR];
};

struct Grp {
  int lane;   // 0..63
  int sub;    // lane inside its group
  int gbase;  // first lane of the group
  int LB;     // lane bits = min(n, 6)
  int n;
};

template <int LR>
__device__ __forceinline__ bool bitval(int b, int r, int sub) {
  return b < LR ? ((r >> b) & 1) : ((sub >> (b - LR)) & 1);
}

// per-amplitude contribution to Im<lam|G|chi>: (lr,li) = lam, (xr,xi) = chi, (pr,pi) = chi's partner
__device__ __forceinline__ float grad_term(int op, bool t, bool cnd, float lr, float li, float xr, float xi,
                                           float pr, float pi) {
  if (!cnd) return 0.f;
  switch (op) {
    case QC_RX:
    case QC_CRX:
      return lr * pi - li * pr;
    case QC_RY:
      return (t ? 1.f : -1.f) * (lr * pr + li * pi);
    case QC_RZ:
    case QC_CRZ:
      return (t ? -1.f : 1.f) * (lr * xi - li * xr);
    default:
      return 0.f;
  }
}

// ---- op-specialised 2x2 updates.  Per-lane real coefficients (c1, s1) fold in the control predicate
// (identity where the control bit is clear) and the sign that depends on this amplitude's target bit,
// so every update is 2 mul + 2 fma with no select.
struct LaneCoef {
  float c1;   // multiplies the amplitude itself
  float s1;   // multiplies the partner (or, for RZ, the other component of the amplitude itself)
};

template <int OP>
__device__ __forceinline__ LaneCoef lane_coef(float c, float s, bool t, bool cnd) {
  if constexpr (OP == QC_RX || OP == QC_CRX) return {cnd ? c : 1.f, cnd ? s : 0.f};
  else if constexpr (OP == QC_RY) return {c, t ? s : -s};
  else if constexpr (OP == QC_RZ || OP == QC_CRZ) return {cnd ? c : 1.f, cnd ? (t ? -s : s) : 0.f};
  else if constexpr (OP == QC_H) return {t ? -0.70710678118654752440f : 0.70710678118654752440f, 0.70710678118654752440f};
  else return {cnd ? 0.f : 1.f, cnd ? 1.f : 0.f};   // CNOT: partner where the control is set
}

// a' for amplitude (ar, ai) with partner (pr, pi)
template <int OP>
__device__ __forceinline__ void apply_op(float& ar, float& ai, float pr, float pi, LaneCoef k) {
  const float r0 = ar, i0 = ai;
  if constexpr (OP == QC_RX || OP == QC_CRX) {          // c a - i s p
    ar = fmaf(k.s1, pi, k.c1 * r0);
    ai = fmaf(-k.s1, pr, k.c1 * i0);
  } else if constexpr (OP == QC_RY) {                    // c a -+ s p
    ar = fmaf(k.s1, pr, k.c1 * r0);
    ai = fmaf(k.s1, pi, k.c1 * i0);
  } else if constexpr (OP == QC_RZ || OP == QC_CRZ) {    // (c - i s1) a
    ar = fmaf(k.s1, i0, k.c1 * r0);
    ai = fmaf(-k.s1, r0, k.c1 * i0);
  } else {                                               // H, CNOT: c1 a + s1 p
    ar = fmaf(k.s1, pr, k.c1 * r0);
    ai = fmaf(k.s1, pi, k.c1 * i0);
  }
}

// One non-U4 gate of compile-time kind OP on K vectors.  GRAD: vectors [0,K/2) are chi, [K/2,K) are
// lam; returns this lane's partial of sum_c Im<lam_c|G|chi_c> evaluated BEFORE the (adjoint) update.
template <int LR, int K, bool ADJ, bool GRAD, int OP>
__device__ __forceinline__ float wave_gate_op(WV<LR> (&v)[K], const QcGate g, const float c, const float s_in,
                                              const Grp& G) {
  constexpr int R = 1 << LR;
  constexpr bool ctl = (OP == QC_CNOT || OP == QC_CRX || OP == QC_CRZ);
  constexpr bool needp = !(OP == QC_RZ || OP == QC_CRZ);
  const float s = ADJ ? -s_in : s_in;
  const int tb = ctl ? g.bb : g.ba;
  const int cb = ctl ? g.ba : -1;
  float grad = 0.f;

  if (tb >= LR) {  // ---- target on a lane bit
    const int mask = 1 << (tb - LR);
    const bool t = (G.sub >> (tb - LR)) & 1;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool cnd = ctl ? bitval<LR>(cb, r, G.sub) : true;
      const LaneCoef k = lane_coef<OP>(c, s, t, cnd);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        float ar = v[q].re[r], ai = v[q].im[r];
        float pr = 0.f, pi = 0.f;
        if constexpr (needp) {
          pr = __shfl_xor(ar, mask);
          pi = __shfl_xor(ai, mask);
        }
        if constexpr (GRAD) {
          if (q < K / 2) grad += grad_term(OP, t, cnd, v[q + K / 2].re[r], v[q + K / 2].im[r], ar, ai, pr, pi);
        }
        apply_op<OP>(ar, ai, pr, pi, k);
        v[q].re[r] = ar;
        v[q].im[r] = ai;
      }
    }
  } else {  // ---- target on a register bit: static pairs via a switch on the bit
    auto body = [&](auto TBC) {
      constexpr int TB = decltype(TBC)::value;
      if constexpr (TB < LR) {
#pragma unroll
        for (int h = 0; h < R / 2; ++h) {
          const int r0 = ((h >> TB) << (TB + 1)) | (h & ((1 << TB) - 1));
          const int r1 = r0 | (1 << TB);
          const bool cnd = ctl ? bitval<LR>(cb, r0, G.sub) : true;
          const LaneCoef k0 = lane_coef<OP>(c, s, false, cnd), k1 = lane_coef<OP>(c, s, true, cnd);
#pragma unroll
          for (int q = 0; q < K; ++q) {
            float ar = v[q].re[r0], ai = v[q].im[r0], br = v[q].re[r1], bi = v[q].im[r1];
            if constexpr (GRAD) {
              if (q < K / 2) {
                grad += grad_term(OP, false, cnd, v[q + K / 2].re[r0], v[q + K / 2].im[r0], ar, ai, br, bi);
                grad += grad_term(OP, true, cnd, v[q + K / 2].re[r1], v[q + K / 2].im[r1], br, bi, ar, ai);
              }
            }
            const float a0r = ar, a0i = ai;
            apply_op<OP>(ar, ai, br, bi, k0);
            apply_op<OP>(br, bi, a0r, a0i, k1);
            v[q].re[r0] = ar;
            v[q].im[r0] = ai;
            v[q].re[r1] = br;
            v[q].im[r1] = bi;
          }
        }
      }
    };
    switch (tb) {
      case 0: body(std::integral_constant<int, 0>{}); break;
      case 1: body(std::integral_constant<int, 1>{}); break;
      case 2: body(std::integral_constant<int, 2>{}); break;
      case 3: body(std::integral_constant<int, 3>{}); break;
      default: break;
    }
  }
  return grad;
}

// run-time opcode -> compile-time kind (wave-uniform branch)
template <int LR, int K, bool ADJ, bool GRAD>
__device__ __forceinline__ float wave_gate(WV<LR> (&v)[K], const QcGate g, const float c, const float s_in,
                                           const Grp& G) {
  switch (g.op) {
    case QC_RX: return wave_gate_op<LR, K, ADJ, GRAD, QC_RX>(v, g, c, s_in, G);
    case QC_RY: return wave_gate_op<LR, K, ADJ, GRAD, QC_RY>(v, g, c, s_in, G);
    case QC_RZ: return wave_gate_op<LR, K, ADJ, GRAD, QC_RZ>(v, g, c, s_in, G);
    case QC_H: return wave_gate_op<LR, K, ADJ, GRAD, QC_H>(v, g, c, s_in, G);
    case QC_CNOT: return wave_gate_op<LR, K, ADJ, GRAD, QC_CNOT>(v, g, c, s_in, G);
    case QC_CRX: return wave_gate_op<LR, K, ADJ, GRAD, QC_CRX>(v, g, c, s_in, G);
    case QC_CRZ: return wave_gate_op<LR, K, ADJ, GRAD, QC_CRZ>(v, g, c, s_in, G);
    default: return 0.f;
  }
}

// Fixed two-wire unitary on lane bits (wires [0,1] / [2,3] are always lane bits in this layout).
template <int LR, int K, bool ADJ>
__device__ __forceinline__ void wave_u4(WV<LR> (&v)[K], const QcGate g, const float* __restrict__ umat, const Grp& G) {
  constexpr int R = 1 << LR;
  const int hb = g.ba - LR, lb = g.bb - LR;
  const int mh = 1 << hb, ml = 1 << lb;
  const int row = (((G.sub >> hb) & 1) << 1) | ((G.sub >> lb) & 1);
  const float* u = umat + (g.slot * 2 + (ADJ ? 1 : 0)) * 32 + row * 8;
  float cr[4], ci[4];  // coefficient of the amplitude reached by xor-ing k into the row index
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int col = row ^ k;
    cr[k] = u[col * 2];
    ci[k] = u[col * 2 + 1];
  }
#pragma unroll
  for (int q = 0; q < K; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float ar = v[q].re[r], ai = v[q].im[r];
      const float r1 = __shfl_xor(ar, ml), i1 = __shfl_xor(ai, ml);
      const float r2 = __shfl_xor(ar, mh), i2 = __shfl_xor(ai, mh);
      const float r3 = __shfl_xor(ar, mh | ml), i3 = __shfl_xor(ai, mh | ml);
      v[q].re[r] = cr[0] * ar - ci[0] * ai + cr[1] * r1 - ci[1] * i1 + cr[2] * r2 - ci[2] * i2 + cr[3] * r3 - ci[3] * i3;
      v[q].im[r] = cr[0] * ai + ci[0] * ar + cr[1] * i1 + ci[1] * r1 + cr[2] * i2 + ci[2] * r2 + cr[3] * i3 + ci[3] * r3;
    }
}

// sum over the lanes of this lane's group (butterfly); every lane of the group gets the total
__device__ __forceinline__ float group_sum(float v, int LB) {
  for (int m = 1; m < (1 << LB); m <<= 1) v += __shfl_xor(v, m);
  return v;
}

// Per-point, per-wire embedding data, owned by lane `sub == w` of the group and broadcast on demand.
struct WireData {
  float c, s;        // cos, sin of a_w / 2
  float d[3];        // first derivatives of a_w along t, x, y
  float dd[2];       // second derivatives along x, y
};

template <int NCH>
__device__ __forceinline__ WireData load_wire(const float* __restrict__ ajets, int64_t B, int64_t p, int w, int n,
                                              bool ok) {
  WireData wd = {1.f, 0.f, {0.f, 0.f, 0.f}, {0.f, 0.f}};
  if (ok && w < n) {
    const float a = ajets[(int64_t)w * B + p];
    sincosf(0.5f * a, &wd.s, &wd.c);
    if constexpr (NCH == 6) {
#pragma unroll
      for (int k = 0; k < 3; ++k) wd.d[k] = ajets[((int64_t)(1 + k) * n + w) * B + p];
#pragma unroll
      for (int k = 0; k < 2; ++k) wd.dd[k] = ajets[((int64_t)(4 + k) * n + w) * B + p];
    }
  }
  return wd;
}

// Embedding series of this lane's amplitudes: P[0]=phi, P[1..3]=d phi (t,x,y), P[4..5]=d2 phi (x,y),
// real magnitudes; the (-i)^popcount phase is applied by the users.
template <int LR, int NCH>
__device__ __forceinline__ void embed_series(float (&P)[NCH][1 << LR], const WireData& mine, const Grp& G) {
  constexpr int R = 1 << LR;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    P[0][r] = 1.f;
#pragma unroll
    for (int c = 1; c < NCH; ++c) P[c][r] = 0.f;
  }
  for (int w = 0; w < G.n; ++w) {
    const int src = G.gbase + w;
    const float c = __shfl(mine.c, src), s = __shfl(mine.s, src);
    float d[3] = {0.f, 0.f, 0.f}, dd[2] = {0.f, 0.f};
    if constexpr (NCH == 6) {
#pragma unroll
      for (int k = 0; k < 3; ++k) d[k] = __shfl(mine.d[k], src);
#pragma unroll
      for (int k = 0; k < 2; ++k) dd[k] = __shfl(mine.dd[k], src);
    }
    const int b = G.n - 1 - w;  // index bit of wire w
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool bit = bitval<LR>(b, r, G.sub);
      const float w0 = bit ? s : c;
      const float e = bit ? c : -s;  // derivative direction of the 2-vector
      const float p0 = P[0][r];
      if constexpr (NCH == 6) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float w1 = 0.5f * d[k] * e;
          const float p1 = P[1 + k][r];
          if (k >= 1) {
            const float w2 = 0.5f * dd[k - 1] * e - 0.25f * d[k] * d[k] * w0;
            P[3 + k][r] = p0 * w2 + 2.f * p1 * w1 + P[3 + k][r] * w0;
          }
          P[1 + k][r] = p0 * w1 + p1 * w0;
        }
      }
      P[0][r] = p0 * w0;
    }
  }
}

template <int LR>
__device__ __forceinline__ int amp_index(int r, int sub) { return (sub << LR) | r; }

template <int LR, int NCH>
__device__ __forceinline__ void phase_load(WV<LR> (&v)[NCH], const float (&P)[NCH][1 << LR], const Grp& G) {
#pragma unroll
  for (int r = 0; r < (1 << LR); ++r) {
    const int ph = __popc(amp_index<LR>(r, G.sub)) & 3;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float m = P[c][r];
      v[c].re[r] = ph == 0 ? m : (ph == 2 ? -m : 0.f);
      v[c].im[r] = ph == 1 ? -m : (ph == 3 ? m : 0.f);
    }
  }
}

// amplitude encoding: the jets of the initial amplitudes are given (qc_amp.hip); amplitude k = feature k
template <int LR, int NCH>
__device__ __forceinline__ void amp_load(WV<LR> (&v)[NCH], const float* __restrict__ ujets, int64_t B, int64_t pc,
                                         bool ok, const Grp& G) {
#pragma unroll
  for (int r = 0; r < (1 << LR); ++r) {
    const int idx = amp_index<LR>(r, G.sub);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      v[c].re[r] = (ok && idx < G.n) ? ujets[((int64_t)c * G.n + idx) * B + pc] : 0.f;
      v[c].im[r] = 0.f;
    }
  }
}

template <int LR, int K, bool ADJ, bool GRAD>
__device__ __forceinline__ void run_program(WV<LR> (&v)[K], const QcGate* __restrict__ prog,
                                            const QcTrig* __restrict__ trig, const float* __restrict__ umat,
                                            int n_gates, const Grp& G, float* acc_wave) {
  for (int i = 0; i < n_gates; ++i) {
    const int g = ADJ ? n_gates - 1 - i : i;
    const QcGate gt = prog[g];
    if (gt.op == QC_U4) {
      wave_u4<LR, K, ADJ>(v, gt, umat, G);
    } else {
      const QcTrig tr = trig[g];
      const float gr = wave_gate<LR, K, ADJ, GRAD>(v, gt, tr.c, tr.s, G);
      if constexpr (GRAD) {
        if (gt.slot >= 0) {
          const float tot = qc_wave_sum_to_lane63(gr);
          if (G.lane == 63) acc_wave[gt.slot] += tot;
        }
      }
    }
  }
}

// this wave's geometry: points per pass and the group descriptor
__device__ __forceinline__ Grp make_group(int n) {
  Grp G;
  G.lane = threadIdx.x & 63;
  G.n = n;
  G.LB = n < 6 ? n : 6;
  const int gs = 1 << G.LB;
  G.sub = G.lane & (gs - 1);
  G.gbase = G.lane & ~(gs - 1);
  return G;
}

// ================================================================== forward: <Z> (NCH = 1) or <Z> jets (NCH = 6)
template <int LR, int NCH>
__global__ void __launch_bounds__(256) k_wave_fwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates, int n,
                                                  const float* __restrict__ ajets, float* __restrict__ qjets,
                                                  int64_t B, int amp) {
  constexpr int R = 1 << LR;
  const Grp G = make_group(n);
  const int wave = threadIdx.x >> 6;
  const int gs = 1 << G.LB;
  const int ppw = (64 / gs) < 16 ? (64 / gs) : 16;       // points per pass of this wave
  const int64_t first = (int64_t)blockIdx.x * 64 + wave * 16;
  for (int pass = 0; pass < 16; pass += ppw) {
    const int slot = G.lane >> G.LB;                      // which of the wave's concurrent points
    const int64_t p = first + pass + slot;
    const bool ok = slot < ppw && p < B;
    const int64_t pc = ok ? p : 0;
    WV<LR> v[NCH];
    if (amp) {
      amp_load<LR, NCH>(v, ajets, B, pc, ok, G);
    } else {
      const WireData mine = load_wire<NCH>(ajets, B, pc, G.sub, n, ok);
      float P[NCH][R];
      embed_series<LR, NCH>(P, mine, G);
      phase_load<LR, NCH>(v, P, G);
    }
    run_program<LR, NCH, false, false>(v, prog, trig, umat, n_gates, G, nullptr);
    // per-amplitude weights of the bilinear forms
    float t[NCH][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float r0 = v[0].re[r], i0 = v[0].im[r];
      t[0][r] = r0 * r0 + i0 * i0;
      if constexpr (NCH == 6) {
#pragma unroll
        for (int c = 1; c < 6; ++c) t[c][r] = 2.f * (r0 * v[c].re[r] + i0 * v[c].im[r]);
        t[4][r] += 2.f * (v[2].re[r] * v[2].re[r] + v[2].im[r] * v[2].im[r]);
        t[5][r] += 2.f * (v[3].re[r] * v[3].re[r] + v[3].im[r] * v[3].im[r]);
      }
    }
    for (int w = 0; w < n; ++w) {
      const int b = n - 1 - w;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) acc += bitval<LR>(b, r, G.sub) ? -t[c][r] : t[c][r];
        acc = group_sum(acc, G.LB);
        if (ok && G.sub == 0) qjets[((int64_t)c * n + w) * B + p] = acc;
      }
    }
  }
}

// ================================================================== backward
template <int LR, int NCH>
__global__ void __launch_bounds__(256) k_wave_bwd(const QcGate* __restrict__ prog, const QcTrig* __restrict__ trig,
                                                  const float* __restrict__ umat, int n_gates, int n_params, int n,
                                                  const float* __restrict__ ajets, const float* __restrict__ qbar,
                                                  float* __restrict__ abar, float* __restrict__ part,
                                                  int64_t part_stride, int64_t row0, int64_t B, int amp) {
  constexpr int R = 1 << LR;
  extern __shared__ float s_acc[];  // [4 waves][n_params]
  const Grp G = make_group(n);
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * n_params; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  const int gs = 1 << G.LB;
  const int ppw = (64 / gs) < 16 ? (64 / gs) : 16;
  const int64_t first = (int64_t)blockIdx.x * 64 + wave * 16;
  for (int pass = 0; pass < 16; pass += ppw) {
    const int slot = G.lane >> G.LB;
    const int64_t p = first + pass + slot;
    const bool ok = slot < ppw && p < B;
    const int64_t pc = ok ? p : 0;
    float P[NCH][R];
    WV<LR> v[2 * NCH];  // [0,NCH) chi, [NCH,2NCH) lam
    {
      WV<LR> f[NCH];
      if (amp) {
        amp_load<LR, NCH>(f, ajets, B, pc, ok, G);
      } else {
        const WireData mine = load_wire<NCH>(ajets, B, pc, G.sub, n, ok);
        embed_series<LR, NCH>(P, mine, G);
        phase_load<LR, NCH>(f, P, G);
      }
      run_program<LR, NCH, false, false>(f, prog, trig, umat, n_gates, G, nullptr);
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[c] = f[c];
    }
    // ---- cotangents of the final states: D_c[k] = sum_w qbar[c][w] * sign_w(k)
    float D[NCH][R];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r) D[c][r] = 0.f;
    for (int w = 0; w < n; ++w) {
      const int b = n - 1 - w;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float qb = ok ? qbar[((int64_t)c * n + w) * B + pc] : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) D[c][r] += bitval<LR>(b, r, G.sub) ? -qb : qb;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float r0 = v[0].re[r], i0 = v[0].im[r];
      float lr = D[0][r] * r0, li = D[0][r] * i0;
      if constexpr (NCH == 6) {
#pragma unroll
        for (int c = 1; c < 6; ++c) {
          lr += D[c][r] * v[c].re[r];
          li += D[c][r] * v[c].im[r];
          v[NCH + c].re[r] = D[c][r] * r0;
          v[NCH + c].im[r] = D[c][r] * i0;
        }
        v[NCH + 2].re[r] += 2.f * D[4][r] * v[2].re[r];
        v[NCH + 2].im[r] += 2.f * D[4][r] * v[2].im[r];
        v[NCH + 3].re[r] += 2.f * D[5][r] * v[3].re[r];
        v[NCH + 3].im[r] += 2.f * D[5][r] * v[3].im[r];
      }
      v[NCH].re[r] = lr;
      v[NCH].im[r] = li;
    }
    run_program<LR, 2 * NCH, true, true>(v, prog, trig, umat, n_gates, G, s_acc + wave * n_params);

    if (amp) {   // d L / d(initial amplitude k of channel c) = 2 Re Lambda_c[k], k < n
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int idx = amp_index<LR>(r, G.sub);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
          if (ok && idx < n) abar[((int64_t)c * n + idx) * B + p] = 2.f * v[NCH + c].re[r];
      }
      continue;
    }
    // ---- cotangents of the angle jets: T(Lam, phi)[w] = Im <Lam| X_w |phi>, phi = (-i)^pop * P
    for (int w = 0; w < n; ++w) {
      const int b = n - 1 - w;
      float out[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) out[c] = 0.f;
      auto accum = [&](auto TBC) {
        // partner magnitudes of the series across bit b (register or lane)
        constexpr int TB = decltype(TBC)::value;  // >= 0: register bit TB; -1: lane bit
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float Pp[NCH];
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            if constexpr (TB >= 0) Pp[c] = P[c][r ^ (1 << (TB < LR ? TB : 0))];
            else Pp[c] = __shfl_xor(P[c][r], 1 << (b - LR));
          }
          const int kp = amp_index<LR>(r, G.sub) ^ (1 << b);
          const int ph = __popc(kp) & 3;
          // Im(conj(L) * (-i)^ph * m): ph0 -> -Li m, ph1 -> -Lr m, ph2 -> +Li m, ph3 -> +Lr m
          auto T = [&](int lc, float m) {
            const float Lr = v[NCH + lc].re[r], Li = v[NCH + lc].im[r];
            return (ph == 0 ? -Li : ph == 1 ? -Lr : ph == 2 ? Li : Lr) * m;
          };
          float a0 = T(0, Pp[0]);
          if constexpr (NCH == 6) {
            a0 += T(1, Pp[1]) + T(2, Pp[2]) + T(3, Pp[3]) + T(4, Pp[4]) + T(5, Pp[5]);
            out[1] += T(1, Pp[0]);
            out[2] += T(2, Pp[0]) + 2.f * T(4, Pp[2]);
            out[3] += T(3, Pp[0]) + 2.f * T(5, Pp[3]);
            out[4] += T(4, Pp[0]);
            out[5] += T(5, Pp[0]);
          }
          out[0] += a0;
        }
      };
      if (b >= LR) {
        accum(std::integral_constant<int, -1>{});
      } else {
        switch (b) {
          case 0: accum(std::integral_constant<int, 0>{}); break;
          case 1: accum(std::integral_constant<int, 1>{}); break;
          case 2: accum(std::integral_constant<int, 2>{}); break;
          case 3: accum(std::integral_constant<int, 3>{}); break;
          default: break;
        }
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float tot = group_sum(out[c], G.LB);
        if (ok && G.sub == 0) abar[((int64_t)c * n + w) * B + p] = tot;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_params; i += 256)
    part[(row0 + blockIdx.x) * part_stride + i] =
        (s_acc[i] + s_acc[n_params + i]) + (s_acc[2 * n_params + i] + s_acc[3 * n_params + i]);
}

}  // namespace

// ------------------------------------------------------------------ launchers
#define QC_WAVE_DISPATCH(n, CALL)                       \
  {                                                     \
    const int lr_ = (n) > 6 ? (n)-6 : 0;                \
    switch (lr_) {                                      \
      case 0: { CALL(0) } break;                        \
      case 1: { CALL(1) } break;                        \
      case 2: { CALL(2) } break;                        \
      default: return QC_ERR_UNSUPPORTED;               \
    }                                                   \
  }

int qc_wave_value_fwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* angles,
                      float* expval, int64_t B, hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
#define CALL(LRR) \
  hipLaunchKernelGGL((k_wave_fwd<LRR, 1>), dim3(grid), dim3(256), 0, st, pg->d_gates, trig, umat, pg->n_gates, pg->n_qubits, angles, expval, B, pg->amplitude);
  QC_WAVE_DISPATCH(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}

int qc_wave_jets_fwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                     int64_t B, hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
#define CALL(LRR) \
  hipLaunchKernelGGL((k_wave_fwd<LRR, 6>), dim3(grid), dim3(256), 0, st, pg->d_gates, trig, umat, pg->n_gates, pg->n_qubits, ajets, qjets, B, pg->amplitude);
  QC_WAVE_DISPATCH(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}

int qc_wave_value_bwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* angles,
                      const float* cot, float* d_angles, float* part, int64_t part_stride, int64_t row0, int64_t B,
                      hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
  const size_t sh = (size_t)4 * pg->n_params * sizeof(float);
#define CALL(LRR)                                                                                               \
  hipLaunchKernelGGL((k_wave_bwd<LRR, 1>), dim3(grid), dim3(256), sh, st, pg->d_gates, trig, umat, pg->n_gates, \
                     pg->n_params, pg->n_qubits, angles, cot, d_angles, part, part_stride, row0, B, pg->amplitude);
  QC_WAVE_DISPATCH(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}

int qc_wave_jets_bwd(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                     const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B,
                     hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
  const size_t sh = (size_t)4 * pg->n_params * sizeof(float);
#define CALL(LRR)                                                                                               \
  hipLaunchKernelGGL((k_wave_bwd<LRR, 6>), dim3(grid), dim3(256), sh, st, pg->d_gates, trig, umat, pg->n_gates, \
                     pg->n_params, pg->n_qubits, ajets, qbar, abar, part, part_stride, row0, B, pg->amplitude);
  QC_WAVE_DISPATCH(pg->n_qubits, CALL)
#undef CALL
  return QC_OK;
}
