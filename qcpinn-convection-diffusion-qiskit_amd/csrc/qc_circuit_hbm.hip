// Variational-circuit kernels, "hbm" family: 9 <= n <= 20 qubits, statevectors resident in HBM.
//
// At n = 16 (BASELINE config 5, cross_mesh) one statevector is 512 KiB — beyond registers, lanes and
// LDS — so the state of every (point, channel) lives in a caller-provided HBM workspace and each
// gate is one pass of index-paired amplitude updates over it (one launch per gate; coalesced: the
// pair stride is a power of two and consecutive lanes take consecutive pair indices).  This is the
// straightforward, correct-first form of the path; the bandwidth-optimal form (fused diagonal
// layers, LDS-tiled qubit blocking) replaces the per-gate passes, not the interface.
//
// Workspace layout for a tile of T <= 64 points, V = NCH or 2*NCH vectors per point, N = 2^n:
//   chi  [NCH][T][N] float2     final / un-computed states
//   lam  [NCH][T][N] float2     cotangent states (backward only)
//   ser  [NCH][T][N] float      embedding series magnitudes (backward only)
//   wd   [T][n][8]   float      per-point per-wire embedding data
//   gblk [blocks]    float      per-block gradient partials of the current gate
// One 64-point tile at a time, so every tile owns one gradient partial row like the other families.
#include "qc_internal.h"
#include "qc_hbm_plan.h"

#include <type_traits>

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

struct Cplx {
  float re, im;
};

// ---------------------------------------------------------------- per-point, per-wire embedding data
template <int NCH>
__global__ void k_hbm_wiredata(const float* __restrict__ ajets, int64_t B, int64_t p0, int T, int n,
                               float* __restrict__ wd, const QcTrig* __restrict__ trig, int absorb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * n) return;
  const int t = i / n, w = i % n;
  const int64_t p = p0 + t;
  float* o = wd + (size_t)i * 8;
  // absorb: gate w is RX(theta_w) on wire w right after the embedding RX(a_w): one rotation by a_w + theta_w
  const float a = ajets[(int64_t)w * B + p] + (absorb ? trig[w].th : 0.f);
  float s, c;
  sincosf(0.5f * a, &s, &c);
  o[0] = c;
  o[1] = s;
#pragma unroll
  for (int k = 0; k < 5; ++k) o[2 + k] = 0.f;
  if constexpr (NCH == 6) {
    for (int k = 0; k < 3; ++k) o[2 + k] = ajets[((int64_t)(1 + k) * n + w) * B + p];
    for (int k = 0; k < 2; ++k) o[5 + k] = ajets[((int64_t)(4 + k) * n + w) * B + p];
  }
}

// ---------------------------------------------------------------- embedded product state + its jets
// thread = (point t, amplitude k): series recursion over the wires (qc_gates.h::qc_embed_series).
template <int NCH, bool KEEP_SERIES>
__global__ void __launch_bounds__(256) k_hbm_init(const float* __restrict__ wd, int T, int n, Cplx* __restrict__ chi,
                                                  float* __restrict__ ser) {
  const int64_t N = (int64_t)1 << n;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)T * N) return;
  const int t = (int)(gid >> n);
  const int64_t k = gid & (N - 1);
  float P[NCH];
  P[0] = 1.f;
#pragma unroll
  for (int c = 1; c < NCH; ++c) P[c] = 0.f;
  const float* w8 = wd + (size_t)t * n * 8;
  for (int w = 0; w < n; ++w) {
    const float c = w8[w * 8], s = w8[w * 8 + 1];
    const bool bit = (k >> (n - 1 - w)) & 1;
    const float w0 = bit ? s : c, e = bit ? c : -s;
    const float p0 = P[0];
    if constexpr (NCH == 6) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float da = w8[w * 8 + 2 + d];
        const float w1 = 0.5f * da * e;
        const float p1 = P[1 + d];
        if (d >= 1) {
          const float w2 = 0.5f * w8[w * 8 + 4 + d] * e - 0.25f * da * da * w0;
          P[3 + d] = p0 * w2 + 2.f * p1 * w1 + P[3 + d] * w0;
        }
        P[1 + d] = p0 * w1 + p1 * w0;
      }
    }
    P[0] = p0 * w0;
  }
  const int ph = __popcll((unsigned long long)k) & 3;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float m = P[c];
    Cplx v;
    v.re = ph == 0 ? m : (ph == 2 ? -m : 0.f);
    v.im = ph == 1 ? -m : (ph == 3 ? m : 0.f);
    if (chi != nullptr) chi[((size_t)c * T + t) * N + k] = v;   // (nullptr: only the series is wanted)
    if constexpr (KEEP_SERIES) ser[((size_t)c * T + t) * N + k] = m;
  }
}

// amplitude encoding: chi[c][t][k] = (k < n ? ujets[c][k][p0+t] : 0), real
template <int NCH>
__global__ void __launch_bounds__(256) k_hbm_init_amp(const float* __restrict__ ujets, int64_t B, int64_t p0, int T, int n,
                                                      Cplx* __restrict__ chi) {
  const int64_t N = (int64_t)1 << n;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)T * N) return;
  const int t = (int)(gid >> n);
  const int64_t k = gid & (N - 1);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float m = k < n ? ujets[((int64_t)c * n + k) * B + p0 + t] : 0.f;
    chi[((size_t)c * T + t) * N + k] = {m, 0.f};
  }
}

// amplitude encoding: abar[c][w][p] = 2 Re lam[c][t][w]
template <int NCH>
__global__ void k_hbm_abar_amp(const Cplx* __restrict__ lam, int T, int n, int64_t B, int64_t p0, float* __restrict__ abar) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NCH * T * n) return;
  const int w = i % n, t = (i / n) % T, c = i / (n * T);
  abar[((int64_t)c * n + w) * B + p0 + t] = 2.f * lam[((size_t)c * T + t) * ((size_t)1 << n) + w].re;
}

// ---------------------------------------------------------------- one gate over S statevectors
__device__ __forceinline__ void coef_of(int op, float c, float s, bool t, float& ar, float& ai, float& br, float& bi) {
  ar = 1.f; ai = 0.f; br = 0.f; bi = 0.f;
  switch (op) {
    case QC_RX: case QC_CRX: ar = c; bi = -s; break;
    case QC_RY: ar = c; br = t ? s : -s; break;
    case QC_RZ: case QC_CRZ: ar = c; ai = t ? s : -s; break;
    case QC_H: ar = t ? -0.70710678118654752440f : 0.70710678118654752440f; br = 0.70710678118654752440f; break;
    case QC_CNOT: ar = 0.f; br = 1.f; break;
    default: break;
  }
}

// thread = (statevector sidx, pair j).  GRAD: vectors [0,S/2) are chi, [S/2,S) lam of the same (channel,
// point); every block writes its partial of sum Im<lam|G|chi> (taken before the adjoint update) to gblk.
template <bool ADJ, bool GRAD>
__global__ void __launch_bounds__(256) k_hbm_gate(Cplx* __restrict__ st, int64_t S, int n, QcGate g,
                                                  const QcTrig* __restrict__ trig, int gi, float* __restrict__ gblk) {
  __shared__ float s_red[4];
  const QcTrig tr = trig[gi];
  const float c = tr.c;
  const float s = ADJ ? -tr.s : tr.s;
  const int64_t N = (int64_t)1 << n, H = N >> 1;
  const bool ctl = (g.op == QC_CNOT || g.op == QC_CRX || g.op == QC_CRZ);
  const int tb = ctl ? g.bb : g.ba, cb = ctl ? g.ba : -1;
  const int64_t nvec = GRAD ? S / 2 : S;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float grad = 0.f;
  if (gid < nvec * H) {
    const int64_t v = gid / H, j = gid % H;
    const int64_t i0 = ((j >> tb) << (tb + 1)) | (j & (((int64_t)1 << tb) - 1));
    const int64_t i1 = i0 | ((int64_t)1 << tb);
    const bool cnd = cb < 0 ? true : ((i0 >> cb) & 1);
    if (cnd) {
      float a0r, a0i, b0r, b0i, a1r, a1i, b1r, b1i;
      coef_of(g.op, c, s, false, a0r, a0i, b0r, b0i);
      coef_of(g.op, c, s, true, a1r, a1i, b1r, b1i);
      Cplx* base = st + v * N;
      const Cplx x0 = base[i0], x1 = base[i1];
      if constexpr (GRAD) {
        Cplx* lb = st + (v + nvec) * N;
        const Cplx l0 = lb[i0], l1 = lb[i1];
        switch (g.op) {
          case QC_RX: case QC_CRX:
            grad = (l0.re * x1.im - l0.im * x1.re) + (l1.re * x0.im - l1.im * x0.re);
            break;
          case QC_RY:
            grad = -(l0.re * x1.re + l0.im * x1.im) + (l1.re * x0.re + l1.im * x0.im);
            break;
          case QC_RZ: case QC_CRZ:
            grad = (l0.re * x0.im - l0.im * x0.re) - (l1.re * x1.im - l1.im * x1.re);
            break;
          default: break;
        }
        Cplx m0, m1;
        m0.re = a0r * l0.re - a0i * l0.im + b0r * l1.re - b0i * l1.im;
        m0.im = a0r * l0.im + a0i * l0.re + b0r * l1.im + b0i * l1.re;
        m1.re = a1r * l1.re - a1i * l1.im + b1r * l0.re - b1i * l0.im;
        m1.im = a1r * l1.im + a1i * l1.re + b1r * l0.im + b1i * l0.re;
        lb[i0] = m0;
        lb[i1] = m1;
      }
      Cplx y0, y1;
      y0.re = a0r * x0.re - a0i * x0.im + b0r * x1.re - b0i * x1.im;
      y0.im = a0r * x0.im + a0i * x0.re + b0r * x1.im + b0i * x1.re;
      y1.re = a1r * x1.re - a1i * x1.im + b1r * x0.re - b1i * x0.im;
      y1.im = a1r * x1.im + a1i * x1.re + b1r * x0.im + b1i * x0.re;
      base[i0] = y0;
      base[i1] = y1;
    }
  }
  if constexpr (GRAD) {
    const float w = qc_wave_sum(grad);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) gblk[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
}

// acc[slot] += sum of the per-block partials, fixed order (one thread: the count is small)
__global__ void k_hbm_fold(const float* __restrict__ gblk, int nblk, float* __restrict__ acc) {
  __shared__ float s_red[4];
  float t = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) t += gblk[i];
  const float w = qc_wave_sum(t);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) acc[0] += (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// fixed two-wire unitary: thread = (statevector, group of 4 amplitudes)
__global__ void __launch_bounds__(256) k_hbm_u4(Cplx* __restrict__ st, int64_t S, int n, int hb, int lb,
                                                const float* __restrict__ u /*32 floats*/) {
  const int64_t N = (int64_t)1 << n, Q = N >> 2;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= S * Q) return;
  const int64_t v = gid / Q, j = gid % Q;
  const int lo = hb < lb ? hb : lb, hi = hb < lb ? lb : hb;
  int64_t b = ((j >> lo) << (lo + 1)) | (j & (((int64_t)1 << lo) - 1));
  b = ((b >> hi) << (hi + 1)) | (b & (((int64_t)1 << hi) - 1));
  const int64_t idx[4] = {b, b | ((int64_t)1 << lb), b | ((int64_t)1 << hb), b | ((int64_t)1 << hb) | ((int64_t)1 << lb)};
  Cplx* base = st + v * N;
  Cplx x[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = base[idx[q]];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float yr = 0.f, yi = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ur = u[(r * 4 + q) * 2], ui = u[(r * 4 + q) * 2 + 1];
      yr += ur * x[q].re - ui * x[q].im;
      yi += ur * x[q].im + ui * x[q].re;
    }
    base[idx[r]] = {yr, yi};
  }
}

// ---------------------------------------------------------------- <Z_w> (jets): block = one (channel, point)
template <int NCH>
__global__ void __launch_bounds__(256) k_hbm_expval(const Cplx* __restrict__ chi, int T, int n, int64_t B, int64_t p0,
                                                    float* __restrict__ qjets) {
  __shared__ float s_red[4][20];
  const int c = blockIdx.x / T, t = blockIdx.x % T;
  const int64_t N = (int64_t)1 << n;
  const Cplx* x0 = chi + ((size_t)0 * T + t) * N;
  const Cplx* xc = chi + ((size_t)c * T + t) * N;
  const Cplx* xk = (NCH == 6 && c >= 4) ? chi + ((size_t)(c - 2) * T + t) * N : nullptr;
  float acc[20];
#pragma unroll
  for (int w = 0; w < 20; ++w) acc[w] = 0.f;
  for (int64_t k = threadIdx.x; k < N; k += 256) {
    const Cplx a = x0[k];
    float wgt;
    if (c == 0) {
      wgt = a.re * a.re + a.im * a.im;
    } else {
      const Cplx b = xc[k];
      wgt = 2.f * (a.re * b.re + a.im * b.im);
      if (xk != nullptr) {
        const Cplx d = xk[k];
        wgt += 2.f * (d.re * d.re + d.im * d.im);
      }
    }
#pragma unroll
    for (int w = 0; w < 20; ++w)
      if (w < n) acc[w] += ((k >> (n - 1 - w)) & 1) ? -wgt : wgt;
  }
  for (int w = 0; w < n; ++w) {
    const float v = qc_wave_sum(acc[w]);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < n) {
    const int w = threadIdx.x;
    qjets[((int64_t)c * n + w) * B + p0 + t] = (s_red[0][w] + s_red[1][w]) + (s_red[2][w] + s_red[3][w]);
  }
}

// ---- the same sums with the amplitude range of one (channel, point) vector split over QC_XCH blocks (the one-block
// form above walks 2^n amplitudes with one wave per SIMD: 0.24 ms per tile at n = 16).  Index bits of an amplitude
// k = chunk * APB + j * 256 + tid: bits [0,8) = thread, [8, 8 + log2(APB/256)) = loop index j, the rest = chunk: only
// the j bits need a per-amplitude conditional add, thread and chunk bits sign whole partial sums.
constexpr int QC_XCH = 8;
template <int NCH>
__global__ void __launch_bounds__(256) k_hbm_expval_chunk(const Cplx* __restrict__ chi, int T, int n,
                                                          float* __restrict__ xpart, int chunks) {
  __shared__ float s_red[4][24];
  const int c = blockIdx.x / T, t = blockIdx.x % T, chunk = blockIdx.y;
  const int64_t N = (int64_t)1 << n;
  const int64_t apb = N / chunks;                       // amplitudes per block (a power of two >= 256)
  const int jbits = 31 - __clz((int)(apb >> 8));         // log2(apb / 256)
  const Cplx* x0 = chi + ((size_t)0 * T + t) * N + chunk * apb;
  const Cplx* xc = chi + ((size_t)c * T + t) * N + chunk * apb;
  const Cplx* xk = (NCH == 6 && c >= 4) ? chi + ((size_t)(c - 2) * T + t) * N + chunk * apb : nullptr;
  float tot = 0.f, sj[8];
#pragma unroll
  for (int b = 0; b < 8; ++b) sj[b] = 0.f;
  for (int j = 0; j < (int)(apb >> 8); ++j) {
    const int64_t k = (int64_t)j * 256 + threadIdx.x;
    const Cplx a = x0[k];
    float wgt;
    if (c == 0) {
      wgt = a.re * a.re + a.im * a.im;
    } else {
      const Cplx b = xc[k];
      wgt = 2.f * (a.re * b.re + a.im * b.im);
      if (xk != nullptr) {
        const Cplx d = xk[k];
        wgt += 2.f * (d.re * d.re + d.im * d.im);
      }
    }
    tot += wgt;
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (b < jbits && ((j >> b) & 1)) sj[b] += wgt;
  }
  // per index bit: signed sum = (sum over bit clear) - (sum over bit set)
  float v[24];
  for (int b = 0; b < n; ++b) {
    float mine;
    if (b < 8) mine = ((threadIdx.x >> b) & 1) ? -tot : tot;                 // thread bit
    else if (b < 8 + jbits) mine = tot - 2.f * sj[b - 8];                    // loop bit
    else mine = ((chunk >> (b - 8 - jbits)) & 1) ? -tot : tot;               // chunk bit
    v[b] = qc_wave_sum(mine);
  }
  for (int b = 0; b < n; ++b)
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][b] = v[b];
  __syncthreads();
  if (threadIdx.x < n) {
    const int b = threadIdx.x;
    xpart[(((size_t)c * T + t) * chunks + chunk) * 24 + b] = (s_red[0][b] + s_red[1][b]) + (s_red[2][b] + s_red[3][b]);
  }
}
// qjets[c][w][p0 + t] = sum of the chunks' partials of index bit n-1-w, in order
__global__ void __launch_bounds__(256) k_hbm_expval_fold(const float* __restrict__ xpart, int nvec, int T, int n, int chunks,
                                                         int64_t B, int64_t p0, float* __restrict__ qjets) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec * n) return;
  const int vec = i / n, w = i % n, c = vec / T, t = vec % T;
  float sum = 0.f;
  for (int q = 0; q < chunks; ++q) sum += xpart[((size_t)vec * chunks + q) * 24 + (n - 1 - w)];
  qjets[((int64_t)c * n + w) * B + p0 + t] = sum;
}

// ---------------------------------------------------------------- cotangents of the final states
template <int NCH>
__global__ void __launch_bounds__(256) k_hbm_lambda(const Cplx* __restrict__ chi, Cplx* __restrict__ lam, int T, int n,
                                                    int64_t B, int64_t p0, const float* __restrict__ qbar) {
  const int64_t N = (int64_t)1 << n;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (int64_t)T * N) return;
  const int t = (int)(gid >> n);
  const int64_t k = gid & (N - 1);
  float D[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) D[c] = 0.f;
  for (int w = 0; w < n; ++w) {
    const bool bit = (k >> (n - 1 - w)) & 1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const float qb = qbar[((int64_t)c * n + w) * B + p0 + t];
      D[c] += bit ? -qb : qb;
    }
  }
  Cplx x[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) x[c] = chi[((size_t)c * T + t) * N + k];
  Cplx l0 = {D[0] * x[0].re, D[0] * x[0].im};
  if constexpr (NCH == 6) {
#pragma unroll
    for (int c = 1; c < 6; ++c) {
      l0.re += D[c] * x[c].re;
      l0.im += D[c] * x[c].im;
      Cplx lc = {D[c] * x[0].re, D[c] * x[0].im};
      if (c == 2 || c == 3) {
        lc.re += 2.f * D[c + 2] * x[c].re;
        lc.im += 2.f * D[c + 2] * x[c].im;
      }
      lam[((size_t)c * T + t) * N + k] = lc;
    }
  }
  lam[((size_t)0 * T + t) * N + k] = l0;
}

// ---------------------------------------------------------------- cotangents of the angle jets
// block = (wire w, point t): T(Lam_c, phi_b)[w] = Im <Lam_c| X_w |phi_b> summed over amplitudes.
template <int NCH>
__global__ void __launch_bounds__(256) k_hbm_abar(const Cplx* __restrict__ lam, const float* __restrict__ ser, int T, int n,
                                                  int64_t B, int64_t p0, float* __restrict__ abar) {
  __shared__ float s_red[4][NCH];
  const int w = blockIdx.x / T, t = blockIdx.x % T;
  const int64_t N = (int64_t)1 << n;
  const int b = n - 1 - w;
  float out[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) out[c] = 0.f;
  for (int64_t k = threadIdx.x; k < N; k += 256) {
    const int64_t kp = k ^ ((int64_t)1 << b);
    const int ph = __popcll((unsigned long long)kp) & 3;
    float Pp[NCH];
    Cplx L[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      Pp[c] = ser[((size_t)c * T + t) * N + kp];
      L[c] = lam[((size_t)c * T + t) * N + k];
    }
    auto Tm = [&](int lc, float m) {
      return (ph == 0 ? -L[lc].im : ph == 1 ? -L[lc].re : ph == 2 ? L[lc].im : L[lc].re) * m;
    };
    float a0 = Tm(0, Pp[0]);
    if constexpr (NCH == 6) {
      a0 += Tm(1, Pp[1]) + Tm(2, Pp[2]) + Tm(3, Pp[3]) + Tm(4, Pp[4]) + Tm(5, Pp[5]);
      out[1] += Tm(1, Pp[0]);
      out[2] += Tm(2, Pp[0]) + 2.f * Tm(4, Pp[2]);
      out[3] += Tm(3, Pp[0]) + 2.f * Tm(5, Pp[3]);
      out[4] += Tm(4, Pp[0]);
      out[5] += Tm(5, Pp[0]);
    }
    out[0] += a0;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float v = qc_wave_sum(out[c]);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < NCH) {
    const int c = threadIdx.x;
    abar[((int64_t)c * n + w) * B + p0 + t] = (s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c]);
  }
}

__global__ void k_hbm_zero(float* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
// folded RX layer: d L / d theta_w = sum over the tile's points of d L / d a_w (value-channel row w of abar)
__global__ void __launch_bounds__(64) k_hbm_absorb_grad(const float* __restrict__ abar, int64_t B, int64_t p0, int T,
                                                        const QcGate* __restrict__ prog, float* __restrict__ acc) {
  const int w = blockIdx.x;
  float v = (int)threadIdx.x < T ? abar[(int64_t)w * B + p0 + threadIdx.x] : 0.f;
  v = qc_wave_sum_to_lane63(v);
  if (threadIdx.x == 63) acc[prog[w].slot] += v;
}
__global__ void k_hbm_store_row(const float* __restrict__ acc, int n_params, float* __restrict__ row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_params) row[i] = acc[i];
}


// ================================================================== staged execution (qc_hbm_plan.h)
__device__ __forceinline__ int64_t dep_local(int l, const QcStage& sd, int l0) {
  int64_t a = l & ((1 << l0) - 1);
  for (int j = l0; j < sd.nloc; ++j)
    if ((l >> j) & 1) a |= (int64_t)1 << sd.lb[j];
  return a;
}
__device__ __forceinline__ int64_t dep_global(int tau, const QcStage& sd) {
  int64_t a = 0;
  for (int j = 0; j < sd.ngb; ++j)
    if ((tau >> j) & 1) a |= (int64_t)1 << sd.gb[j];
  return a;
}
__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cplx cmulc(Cplx a, Cplx b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }  // a * conj(b)

// block-wide sum (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
  const float w = qc_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = w;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// One L stage on one tile of one statevector (GRAD: of the chi/lam pair v, v + nvec).
// grid = nvec * 2^ngb blocks.  Forward: gates in order, then the fused diagonal table at the store.
// ADJ: the table's conjugate at the load, then the adjoint gates in reverse order; GRAD additionally
// writes, per parametric gate, this block's partial of Im<lam|G|chi> to gpart[gate][block].
template <bool ADJ, bool GRAD>
__global__ void __launch_bounds__(256) k_hbm_stage(Cplx* __restrict__ st, int64_t nvec, int n, QcStage sd,
                                                   const QcStageGate* __restrict__ gl, const QcTrig* __restrict__ trig,
                                                   const float* __restrict__ umat, const Cplx* __restrict__ tab,
                                                   float* __restrict__ gpart) {
  extern __shared__ Cplx tile[];        // [GRAD ? 2 : 1][2^nloc]
  __shared__ float s_red[4];
  const int64_t N = (int64_t)1 << n;
  const int TS = 1 << sd.nloc;
  const int l0 = sd.nloc < QC_HBM_L0 ? sd.nloc : QC_HBM_L0;
  const int tiles = 1 << sd.ngb;
  const int64_t v = blockIdx.x / tiles;
  const int tau = blockIdx.x % tiles;
  const int64_t base = dep_global(tau, sd);
  Cplx* t0 = tile;
  Cplx* t1 = tile + TS;
  Cplx* g0 = st + v * N;
  Cplx* g1 = st + (v + nvec) * N;
  for (int l = threadIdx.x; l < TS; l += 256) {
    const int64_t a = base | dep_local(l, sd, l0);
    Cplx x = g0[a];
    if (ADJ && tab != nullptr) x = cmulc(x, tab[a]);
    t0[l] = x;
    if constexpr (GRAD) {
      Cplx y = g1[a];
      if (ADJ && tab != nullptr) y = cmulc(y, tab[a]);
      t1[l] = y;
    }
  }
  __syncthreads();
  int pidx = 0;  // running index of parametric gates in this stage (execution order)
  for (int i = 0; i < sd.ng; ++i) {
    const QcStageGate g = gl[sd.g0 + (ADJ ? sd.ng - 1 - i : i)];
    if (g.op == QC_U4) {
      const float* u = umat + (g.slot * 2 + (ADJ ? 1 : 0)) * 32;
      const int lo = g.jt < g.jc ? g.jt : g.jc, hi = g.jt < g.jc ? g.jc : g.jt;
      for (int q = threadIdx.x; q < (TS >> 2); q += 256) {
        int b = ((q >> lo) << (lo + 1)) | (q & ((1 << lo) - 1));
        b = ((b >> hi) << (hi + 1)) | (b & ((1 << hi) - 1));
        const int idx[4] = {b, b | (1 << g.jc), b | (1 << g.jt), b | (1 << g.jt) | (1 << g.jc)};
#pragma unroll
        for (int w = 0; w < (GRAD ? 2 : 1); ++w) {
          Cplx* t = w == 0 ? t0 : t1;
          Cplx x[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = t[idx[k]];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float yr = 0.f, yi = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float ur = u[(r * 4 + k) * 2], ui = u[(r * 4 + k) * 2 + 1];
              yr += ur * x[k].re - ui * x[k].im;
              yi += ur * x[k].im + ui * x[k].re;
            }
            t[idx[r]] = {yr, yi};
          }
        }
      }
      __syncthreads();
      continue;
    }
    const QcTrig tr = trig[g.gi];
    const float c = tr.c, s = ADJ ? -tr.s : tr.s;
    float grad = 0.f;
    // op-specialised pair update (wave-uniform switch outside the pair loop): 4 mul + 4 fma per pair and
    // vector for the rotations instead of the generic 2x2 complex form (16), gradient term inline.
    // (Batching several gates per LDS round trip - a thread owning the 2^m amplitudes spanned by m target bits -
    // was measured slower: adjacent gates target adjacent low bits, so lanes stride 2^m amplitudes and collide
    // in the LDS banks; it needs a padded/swizzled tile, for which the 64 KiB of the sweep's two tiles leave no room.)
    auto pairs = [&](auto OPC) {
      constexpr int OP = decltype(OPC)::value;
      constexpr bool ctl = (OP == QC_CNOT || OP == QC_CRX || OP == QC_CRZ);
      auto upd = [&](Cplx& a, Cplx& b) {   // (a, b) = amplitudes with target bit 0 / 1
        const Cplx x0 = a, x1 = b;
        if constexpr (OP == QC_RX || OP == QC_CRX) {
          a = {fmaf(s, x1.im, c * x0.re), fmaf(-s, x1.re, c * x0.im)};
          b = {fmaf(s, x0.im, c * x1.re), fmaf(-s, x0.re, c * x1.im)};
        } else if constexpr (OP == QC_RY) {
          a = {fmaf(-s, x1.re, c * x0.re), fmaf(-s, x1.im, c * x0.im)};
          b = {fmaf(s, x0.re, c * x1.re), fmaf(s, x0.im, c * x1.im)};
        } else if constexpr (OP == QC_RZ || OP == QC_CRZ) {
          a = {fmaf(s, x0.im, c * x0.re), fmaf(-s, x0.re, c * x0.im)};
          b = {fmaf(-s, x1.im, c * x1.re), fmaf(s, x1.re, c * x1.im)};
        } else if constexpr (OP == QC_H) {
          const float h = 0.70710678118654752440f;
          a = {h * (x0.re + x1.re), h * (x0.im + x1.im)};
          b = {h * (x0.re - x1.re), h * (x0.im - x1.im)};
        } else {   // CNOT
          a = x1;
          b = x0;
        }
      };
#pragma unroll 2
      for (int p = threadIdx.x; p < (TS >> 1); p += 256) {
        const int i0 = ((p >> g.jt) << (g.jt + 1)) | (p & ((1 << g.jt) - 1));
        const int i1 = i0 | (1 << g.jt);
        if constexpr (ctl) {
          if (!((i0 >> g.jc) & 1)) continue;
        }
        Cplx x0 = t0[i0], x1 = t0[i1];
        if constexpr (GRAD) {
          Cplx q0 = t1[i0], q1 = t1[i1];
          if constexpr (OP == QC_RX || OP == QC_CRX)
            grad += (q0.re * x1.im - q0.im * x1.re) + (q1.re * x0.im - q1.im * x0.re);
          else if constexpr (OP == QC_RY)
            grad += -(q0.re * x1.re + q0.im * x1.im) + (q1.re * x0.re + q1.im * x0.im);
          else if constexpr (OP == QC_RZ || OP == QC_CRZ)
            grad += (q0.re * x0.im - q0.im * x0.re) - (q1.re * x1.im - q1.im * x1.re);
          upd(q0, q1);
          t1[i0] = q0;
          t1[i1] = q1;
        }
        upd(x0, x1);
        t0[i0] = x0;
        t0[i1] = x1;
      }
    };
    switch (g.op) {
      case QC_RX: pairs(std::integral_constant<int, QC_RX>{}); break;
      case QC_RY: pairs(std::integral_constant<int, QC_RY>{}); break;
      case QC_RZ: pairs(std::integral_constant<int, QC_RZ>{}); break;
      case QC_H: pairs(std::integral_constant<int, QC_H>{}); break;
      case QC_CNOT: pairs(std::integral_constant<int, QC_CNOT>{}); break;
      case QC_CRX: pairs(std::integral_constant<int, QC_CRX>{}); break;
      case QC_CRZ: pairs(std::integral_constant<int, QC_CRZ>{}); break;
      default: break;
    }
    if constexpr (GRAD) {
      if (g.slot >= 0) {
        const float tot = block_sum_256(grad, s_red);
        if (threadIdx.x == 0) gpart[(size_t)pidx * gridDim.x + blockIdx.x] = tot;
        ++pidx;
      }
    }
    __syncthreads();
  }
  for (int l = threadIdx.x; l < TS; l += 256) {
    const int64_t a = base | dep_local(l, sd, l0);
    Cplx x = t0[l];
    if (!ADJ && tab != nullptr) x = cmul(x, tab[a]);
    g0[a] = x;
    if constexpr (GRAD) {
      Cplx y = t1[l];
      if (!ADJ && tab != nullptr) y = cmul(y, tab[a]);
      g1[a] = y;
    }
  }
}

// acc[slot_i] += sum_b gpart[i][b] for the parametric gates of one stage, in execution order
__global__ void __launch_bounds__(256) k_hbm_fold_stage(const float* __restrict__ gpart, int nblk,
                                                        const int* __restrict__ slots, float* __restrict__ acc) {
  __shared__ float s_red[4];
  const int i = blockIdx.x;
  float t = 0.f;
  for (int b = threadIdx.x; b < nblk; b += 256) t += gpart[(size_t)i * nblk + b];
  const float tot = block_sum_256(t, s_red);
  if (threadIdx.x == 0) acc[slots[i]] += tot;
}

// phase table of one diagonal run: tab[k] = prod_g phase_g(k), accumulated in double
__global__ void __launch_bounds__(256) k_hbm_diag_table(const QcDiagGate* __restrict__ dg, int ng,
                                                        const QcTrig* __restrict__ trig, int n, Cplx* __restrict__ tab) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= ((int64_t)1 << n)) return;
  double zr = 1.0, zi = 0.0;
  for (int i = 0; i < ng; ++i) {
    const QcDiagGate g = dg[i];
    if (g.bc >= 0 && !((k >> g.bc) & 1)) continue;
    const QcTrig tr = trig[g.gi];
    const double c = tr.c, s = ((k >> g.bt) & 1) ? (double)tr.s : -(double)tr.s;   // bit 0: c - i s, bit 1: c + i s
    const double nr = zr * c - zi * s, ni = zr * s + zi * c;
    zr = nr;
    zi = ni;
  }
  tab[k] = {(float)zr, (float)zi};
}

template <bool ADJ>
__global__ void __launch_bounds__(256) k_hbm_diag_apply(Cplx* __restrict__ st, int64_t S, int n, const Cplx* __restrict__ tab) {
  const int64_t N = (int64_t)1 << n;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= S * N) return;
  const int64_t k = gid & (N - 1);
  st[gid] = ADJ ? cmulc(st[gid], tab[k]) : cmul(st[gid], tab[k]);
}

// W partials: wp[grp][k] = sum over this group's (chi, lam) pairs of Im(conj(lam_k) chi_k)
__global__ void __launch_bounds__(256) k_hbm_diag_w(const Cplx* __restrict__ st, int64_t nvec, int n, int groups,
                                                    float* __restrict__ wp) {
  const int64_t N = (int64_t)1 << n;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int grp = blockIdx.y;
  if (k >= N) return;
  float acc = 0.f;
  for (int64_t v = grp; v < nvec; v += groups) {
    const Cplx x = st[v * N + k], l = st[(v + nvec) * N + k];
    acc += l.re * x.im - l.im * x.re;
  }
  wp[(size_t)grp * N + k] = acc;
}

// W[k] = sum_grp wp[grp][k], written over group 0 (the other group rows become scratch for the gate partials)
__global__ void __launch_bounds__(256) k_hbm_diag_collapse(float* __restrict__ wp, int n, int groups) {
  const int64_t N = (int64_t)1 << n;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= N) return;
  float w = 0.f;
  for (int q = 0; q < groups; ++q) w += wp[(size_t)q * N + k];
  wp[k] = w;
}

// block (gate, chunk): gp[gate][chunk] = sum over the chunk's k of g(k) * W[k],  g = +-1 (target bit), 0 if the
// control bit is clear.  (One block per gate walking all 2^n entries was latency-bound: 0.77 ms per diagonal run
// at n = 16; 16 chunks per gate fill the chip.)
constexpr int DIAG_CHUNKS = 16;
__global__ void __launch_bounds__(256) k_hbm_diag_grad(const QcDiagGate* __restrict__ dg, const float* __restrict__ W,
                                                       int n, float* __restrict__ gp) {
  __shared__ float s_red[4];
  const QcDiagGate g = dg[blockIdx.x];
  const int64_t N = (int64_t)1 << n;
  const int64_t per = (N + DIAG_CHUNKS - 1) / DIAG_CHUNKS;
  const int64_t k0 = (int64_t)blockIdx.y * per, k1 = (k0 + per) < N ? (k0 + per) : N;
  float t = 0.f;
  for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
    if (g.bc >= 0 && !((k >> g.bc) & 1)) continue;
    const float w = W[k];
    t += ((k >> g.bt) & 1) ? -w : w;
  }
  const float tot = block_sum_256(t, s_red);
  if (threadIdx.x == 0) gp[(size_t)blockIdx.x * DIAG_CHUNKS + blockIdx.y] = tot;
}

// acc[slot_i] += gp[i][0] + ... + gp[i][DIAG_CHUNKS-1], in order
__global__ void __launch_bounds__(256) k_hbm_diag_fold(const QcDiagGate* __restrict__ dg, int ng, const float* __restrict__ gp,
                                                       float* __restrict__ acc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ng) return;
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < DIAG_CHUNKS; ++c) t += gp[(size_t)i * DIAG_CHUNKS + c];
  acc[dg[i].slot] += t;
}

struct Ws {
  Cplx* chi;
  Cplx* lam;
  float* ser;
  float* wd;
  float* gblk;
  float* acc;
  Cplx* tabs;    // [n_tables][N]
  float* wpart;  // [W_GROUPS][N]
  float* gpart;  // [max_param_lgates][nvec * tiles]
  float* xpart;  // [nch * 64][QC_XCH][24] partial <Z> sums of k_hbm_expval_chunk
};

constexpr int W_GROUPS = 16;

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

inline bool no_absorb() {
  static const bool f = [] { const char* e = getenv("QC_NO_ABSORB"); return e && e[0] == '1'; }();
  return f;
}

inline bool use_simple() {
  static const bool f = [] { const char* e = getenv("QC_HBM_SIMPLE"); return e && e[0] == '1'; }();
  return f;   // test hook: one pass per gate (the staged path's cross-check)
}

}  // namespace

// ------------------------------------------------------------------ plan (host)
struct QcHbmPlanPriv {
  QcHbmPlan pub;
  std::vector<QcStage> stages;
  std::vector<QcStageGate> lgates;
  std::vector<QcDiagGate> dgates;
  std::vector<int> slots_rev;      // per L stage: parameter slots of its parametric gates, reverse order
  std::vector<int> slots_off;      // offset of each stage's list in slots_rev (size n_stages + 1)
  int* d_slots_rev = nullptr;
};

static bool is_diag_op(int op) { return op == QC_RZ || op == QC_CRZ; }

QcHbmPlan* qc_hbm_plan_create(const qc_program* pg) {
  const int n = pg->n_qubits;
  const int T = n < QC_HBM_T ? n : QC_HBM_T;
  const int L0 = T < QC_HBM_L0 ? T : QC_HBM_L0;
  QcHbmPlanPriv* P = new QcHbmPlanPriv();
  struct Tmp { int kind; std::vector<int> gates; std::vector<int> extra; };
  std::vector<Tmp> tmp;
  const int absorb = (pg->lead_rx && !pg->amplitude && !no_absorb()) ? 1 : 0;
  P->pub.absorb = absorb;
  for (int g = absorb ? n : 0; g < pg->n_gates; ++g) {
    const QcGate& gt = pg->h_gates[g];
    if (is_diag_op(gt.op)) {
      if (tmp.empty() || tmp.back().kind != 1) tmp.push_back({1, {}, {}});
      tmp.back().gates.push_back(g);
      continue;
    }
    std::vector<int> bits = {gt.ba};
    if (gt.bb >= 0) bits.push_back(gt.bb);
    std::vector<int> need;
    const bool open_l = !tmp.empty() && tmp.back().kind == 0;
    for (int b : bits)
      if (b >= L0 && (!open_l || std::find(tmp.back().extra.begin(), tmp.back().extra.end(), b) == tmp.back().extra.end()) &&
          std::find(need.begin(), need.end(), b) == need.end())
        need.push_back(b);
    if (open_l && (int)(tmp.back().extra.size() + need.size()) <= T - L0) {
      for (int b : need) tmp.back().extra.push_back(b);
      tmp.back().gates.push_back(g);
    } else {
      std::vector<int> ex;
      for (int b : bits)
        if (b >= L0 && std::find(ex.begin(), ex.end(), b) == ex.end()) ex.push_back(b);
      tmp.push_back({0, {g}, ex});
    }
  }
  int n_tables = 0, max_pl = 0;
  P->slots_off.push_back(0);
  for (size_t i = 0; i < tmp.size(); ++i) {
    QcStage sd;
    memset(&sd, 0, sizeof(sd));
    sd.post_diag = -1;
    sd.table = -1;
    if (tmp[i].kind == 1) {
      sd.kind = 1;
      sd.g0 = (int)P->dgates.size();
      sd.ng = (int)tmp[i].gates.size();
      sd.table = n_tables++;
      for (int g : tmp[i].gates) {
        const QcGate& gt = pg->h_gates[g];
        const bool ctl = gt.op == QC_CRZ;
        P->dgates.push_back({gt.op, ctl ? gt.bb : gt.ba, ctl ? gt.ba : -1, g, gt.slot});
      }
      if (!P->stages.empty() && P->stages.back().kind == 0) {
        P->stages.back().post_diag = (int)P->stages.size();
        sd.fused = 1;
      }
    } else {
      sd.kind = 0;
      std::vector<int> loc;
      for (int b = 0; b < L0; ++b) loc.push_back(b);
      for (int b : tmp[i].extra) loc.push_back(b);
      for (int b = L0; b < n && (int)loc.size() < T; ++b)
        if (std::find(loc.begin(), loc.end(), b) == loc.end()) loc.push_back(b);
      std::sort(loc.begin(), loc.end());
      sd.nloc = (int)loc.size();
      for (int j = 0; j < sd.nloc; ++j) sd.lb[j] = loc[j];
      sd.ngb = 0;
      for (int b = 0; b < n; ++b)
        if (std::find(loc.begin(), loc.end(), b) == loc.end()) sd.gb[sd.ngb++] = b;
      auto local = [&](int b) { return (int)(std::find(loc.begin(), loc.end(), b) - loc.begin()); };
      sd.g0 = (int)P->lgates.size();
      sd.ng = (int)tmp[i].gates.size();
      std::vector<int> pslots;
      for (int g : tmp[i].gates) {
        const QcGate& gt = pg->h_gates[g];
        QcStageGate lg;
        lg.op = gt.op;
        lg.gi = g;
        lg.slot = gt.slot;
        if (gt.op == QC_U4) {
          lg.jt = local(gt.ba);
          lg.jc = local(gt.bb);
        } else if (gt.op == QC_CNOT || gt.op == QC_CRX) {
          lg.jt = local(gt.bb);
          lg.jc = local(gt.ba);
        } else {
          lg.jt = local(gt.ba);
          lg.jc = -1;
        }
        if (gt.op != QC_U4 && gt.slot >= 0) pslots.push_back(gt.slot);
        P->lgates.push_back(lg);
      }
      max_pl = std::max(max_pl, (int)pslots.size());
      for (auto it = pslots.rbegin(); it != pslots.rend(); ++it) P->slots_rev.push_back(*it);
    }
    P->stages.push_back(sd);
    P->slots_off.push_back((int)P->slots_rev.size());
  }
  QcHbmPlan& q = P->pub;
  q.n_stages = (int)P->stages.size();
  q.n_tables = n_tables;
  q.n_lgates = (int)P->lgates.size();
  q.n_dgates = (int)P->dgates.size();
  q.stages = P->stages.data();
  q.h_lgates = P->lgates.data();
  q.h_dgates = P->dgates.data();
  q.d_lgates = nullptr;
  q.d_dgates = nullptr;
  q.max_param_lgates = max_pl;
  bool ok = true;
  if (q.n_lgates) {
    ok = ok && hipMalloc((void**)&q.d_lgates, sizeof(QcStageGate) * q.n_lgates) == hipSuccess;
    ok = ok && hipMemcpy(q.d_lgates, q.h_lgates, sizeof(QcStageGate) * q.n_lgates, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok && q.n_dgates) {
    ok = ok && hipMalloc((void**)&q.d_dgates, sizeof(QcDiagGate) * q.n_dgates) == hipSuccess;
    ok = ok && hipMemcpy(q.d_dgates, q.h_dgates, sizeof(QcDiagGate) * q.n_dgates, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (ok && !P->slots_rev.empty()) {
    ok = ok && hipMalloc((void**)&P->d_slots_rev, sizeof(int) * P->slots_rev.size()) == hipSuccess;
    ok = ok && hipMemcpy(P->d_slots_rev, P->slots_rev.data(), sizeof(int) * P->slots_rev.size(), hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!ok) {
    qc_hbm_plan_destroy(&P->pub);
    return nullptr;
  }
  return &P->pub;
}

void qc_hbm_plan_destroy(QcHbmPlan* plan) {
  if (!plan) return;
  QcHbmPlanPriv* P = reinterpret_cast<QcHbmPlanPriv*>(plan);   // pub is the first member
  if (plan->d_lgates) (void)hipFree(plan->d_lgates);
  if (plan->d_dgates) (void)hipFree(plan->d_dgates);
  if (P->d_slots_rev) (void)hipFree(P->d_slots_rev);
  delete P;
}

// ------------------------------------------------------------------ workspace
// bytes of workspace for one call (tile of up to 64 points at a time)
size_t qc_hbm_workspace_bytes(const qc_program* pg, int nch, bool backward) {
  const size_t N = (size_t)1 << pg->n_qubits, T = 64;
  const QcHbmPlan* plan = (const QcHbmPlan*)pg->hbm_plan;
  size_t b = align_up(sizeof(Cplx) * nch * T * N);                 // chi
  if (backward) {
    b += align_up(sizeof(Cplx) * nch * T * N);                     // lam (placed directly behind the used part of chi)
    b += align_up(sizeof(float) * nch * T * N);                    // series
    b += align_up(sizeof(float) * (size_t)qc_ceil_div((int64_t)nch * T * (N / 2), 256));  // gblk
    b += align_up(sizeof(float) * (pg->n_params > 0 ? pg->n_params : 1));                  // acc
    b += align_up(sizeof(float) * W_GROUPS * N);                   // wpart
    const size_t tiles = N >> (pg->n_qubits < QC_HBM_T ? pg->n_qubits : QC_HBM_T);
    b += align_up(sizeof(float) * (size_t)(plan ? plan->max_param_lgates : 0) * nch * T * tiles + 256);  // gpart
  }
  b += align_up(sizeof(Cplx) * (size_t)(plan ? plan->n_tables : 0) * N + 256);          // diagonal tables
  b += align_up(sizeof(float) * T * pg->n_qubits * 8);
  b += align_up(sizeof(float) * (size_t)nch * T * QC_XCH * 24);                        // expval chunk partials
  return b;
}

// one tile's [chi | lam] slot of the kept-state store (6 channels x 64 points x 2^n amplitudes, twice), in complex64
size_t qc_hbm_keep_slot_elems(const qc_program* pg) { return (size_t)2 * 6 * 64 * ((size_t)1 << pg->n_qubits); }
// bytes of the store for B residual points, or 0 when it would exceed the budget (QC_HBM_KEEP_GB, default 96 GiB)
size_t qc_hbm_keep_bytes(const qc_program* pg, int64_t B) {
  static const double cap_gb = [] { const char* e = getenv("QC_HBM_KEEP_GB"); return e ? atof(e) : 96.0; }();
  if (B <= 0) return 0;
  const double bytes = (double)qc_ceil_div(B, 64) * (double)qc_hbm_keep_slot_elems(pg) * sizeof(Cplx);
  return bytes <= cap_gb * 1073741824.0 ? (size_t)bytes : 0;
}

static Ws carve(const qc_program* pg, int nch, bool backward, void* ws) {
  const size_t N = (size_t)1 << pg->n_qubits, T = 64;
  const QcHbmPlan* plan = (const QcHbmPlan*)pg->hbm_plan;
  char* p = (char*)ws;
  Ws w = {};
  w.chi = (Cplx*)p; p += align_up(sizeof(Cplx) * nch * T * N);
  if (backward) {
    w.lam = (Cplx*)p; p += align_up(sizeof(Cplx) * nch * T * N);
    w.ser = (float*)p; p += align_up(sizeof(float) * nch * T * N);
    w.gblk = (float*)p; p += align_up(sizeof(float) * (size_t)qc_ceil_div((int64_t)nch * T * (N / 2), 256));
    w.acc = (float*)p; p += align_up(sizeof(float) * (pg->n_params > 0 ? pg->n_params : 1));
    w.wpart = (float*)p; p += align_up(sizeof(float) * W_GROUPS * N);
    const size_t tiles = N >> (pg->n_qubits < QC_HBM_T ? pg->n_qubits : QC_HBM_T);
    w.gpart = (float*)p; p += align_up(sizeof(float) * (size_t)(plan ? plan->max_param_lgates : 0) * nch * T * tiles + 256);
  }
  w.tabs = (Cplx*)p; p += align_up(sizeof(Cplx) * (size_t)(plan ? plan->n_tables : 0) * N + 256);
  w.wd = (float*)p; p += align_up(sizeof(float) * T * pg->n_qubits * 8);
  w.xpart = (float*)p;
  return w;
}

// ------------------------------------------------------------------ staged forward / backward of one tile
// dynamic LDS of a stage: 2^12 complex64 per tile, x2 with the cotangent tile = 64 KiB (the default limit)
static void staged_forward(const qc_program* pg, const QcTrig* trig, const float* umat, const Ws& w, int64_t S,
                           hipStream_t st) {
  const QcHbmPlan* plan = (const QcHbmPlan*)pg->hbm_plan;
  const int n = pg->n_qubits;
  const int64_t N = (int64_t)1 << n;
  for (int i = 0; i < plan->n_stages; ++i) {
    const QcStage& sd = plan->stages[i];
    if (sd.kind == 0) {
      const Cplx* tab = sd.post_diag >= 0 ? w.tabs + (size_t)plan->stages[sd.post_diag].table * N : nullptr;
      const size_t sh = sizeof(Cplx) << sd.nloc;
      hipLaunchKernelGGL((k_hbm_stage<false, false>), dim3((unsigned)(S << sd.ngb)), dim3(256), sh, st, w.chi, S, n, sd,
                         plan->d_lgates, trig, umat, tab, (float*)nullptr);
    } else if (!sd.fused) {
      hipLaunchKernelGGL((k_hbm_diag_apply<false>), dim3(qc_ceil_div(S * N, 256)), dim3(256), 0, st, w.chi, S, n,
                         w.tabs + (size_t)sd.table * N);
    }
  }
}

static void build_tables(const qc_program* pg, const QcTrig* trig, const Ws& w, hipStream_t st) {
  const QcHbmPlan* plan = (const QcHbmPlan*)pg->hbm_plan;
  const int n = pg->n_qubits;
  const int64_t N = (int64_t)1 << n;
  for (int i = 0; i < plan->n_stages; ++i) {
    const QcStage& sd = plan->stages[i];
    if (sd.kind == 1)
      hipLaunchKernelGGL(k_hbm_diag_table, dim3(qc_ceil_div(N, 256)), dim3(256), 0, st, plan->d_dgates + sd.g0, sd.ng, trig, n,
                         w.tabs + (size_t)sd.table * N);
  }
}

static void staged_backward(const qc_program* pg, const QcTrig* trig, const float* umat, const Ws& w, int64_t S,
                            hipStream_t st) {
  const QcHbmPlanPriv* P = reinterpret_cast<const QcHbmPlanPriv*>(pg->hbm_plan);
  const QcHbmPlan* plan = &P->pub;
  const int n = pg->n_qubits;
  const int64_t N = (int64_t)1 << n;
  auto diag_grads = [&](const QcStage& d) {   // memory holds chi, lam at the OUTPUT of the diagonal run d
    hipLaunchKernelGGL(k_hbm_diag_w, dim3(qc_ceil_div(N, 256), W_GROUPS), dim3(256), 0, st, w.chi, S, n, W_GROUPS, w.wpart);
    // (the gate partials live in group rows 1.. of wpart, free after the collapse: needs ng * 16 <= 15 * 2^n)
    hipLaunchKernelGGL(k_hbm_diag_collapse, dim3(qc_ceil_div(N, 256)), dim3(256), 0, st, w.wpart, n, W_GROUPS);
    hipLaunchKernelGGL(k_hbm_diag_grad, dim3(d.ng, DIAG_CHUNKS), dim3(256), 0, st, plan->d_dgates + d.g0, w.wpart, n, w.wpart + N);
    hipLaunchKernelGGL(k_hbm_diag_fold, dim3(qc_ceil_div(d.ng, 256)), dim3(256), 0, st, plan->d_dgates + d.g0, d.ng, w.wpart + N, w.acc);
  };
  for (int i = plan->n_stages - 1; i >= 0; --i) {
    const QcStage& sd = plan->stages[i];
    if (sd.kind == 0) {
      const Cplx* tab = nullptr;
      if (sd.post_diag >= 0) {
        const QcStage& d = plan->stages[sd.post_diag];
        diag_grads(d);
        tab = w.tabs + (size_t)d.table * N;
      }
      const size_t sh = 2 * (sizeof(Cplx) << sd.nloc);
      const unsigned nblk = (unsigned)(S << sd.ngb);
      hipLaunchKernelGGL((k_hbm_stage<true, true>), dim3(nblk), dim3(256), sh, st, w.chi, S, n, sd, plan->d_lgates, trig, umat,
                         tab, w.gpart);
      const int np = P->slots_off[i + 1] - P->slots_off[i];
      if (np > 0)
        hipLaunchKernelGGL(k_hbm_fold_stage, dim3(np), dim3(256), 0, st, w.gpart, (int)nblk, P->d_slots_rev + P->slots_off[i],
                           w.acc);
    } else if (!sd.fused) {
      diag_grads(sd);
      hipLaunchKernelGGL((k_hbm_diag_apply<true>), dim3(qc_ceil_div(2 * S * N, 256)), dim3(256), 0, st, w.chi, 2 * S, n,
                         w.tabs + (size_t)sd.table * N);
    }
  }
}

template <int NCH>
static int hbm_run(const qc_program* pg, const QcTrig* trig_dev, const float* umat,
                   const float* ajets, float* qjets, const float* qbar, float* abar, float* part, int64_t part_stride,
                   int64_t row0, int64_t B, void* ws, size_t ws_bytes, hipStream_t st, Cplx* store = nullptr) {
  // store != nullptr (fused step, enough HBM): every 64-point tile owns a [chi | lam] slot there.  The forward
  // call evolves each tile in its slot and leaves the final states; the backward call finds them and skips
  // the forward recompute (only the embedding series is rebuilt).  288 GB of HBM holds ~700 such tiles at n = 16.
  const bool backward = qbar != nullptr;
  if (!ws || ws_bytes < qc_hbm_workspace_bytes(pg, NCH, backward)) return QC_ERR_ARG;
  const int n = pg->n_qubits;
  const int64_t N = (int64_t)1 << n;
  const bool kept = backward && store != nullptr;
  Ws w = carve(pg, NCH, backward, ws);
  const bool staged = pg->hbm_plan != nullptr && !use_simple();
  const int absorb = (staged && ((const QcHbmPlan*)pg->hbm_plan)->absorb) ? 1 : 0;   // the per-gate path runs every gate
  if (staged) build_tables(pg, trig_dev, w, st);
  for (int64_t p0 = 0; p0 < B; p0 += 64) {
    const int T = (int)((B - p0) < 64 ? (B - p0) : 64);
    const int TA = T;                                   // layout stride = points of this tile
    if (store != nullptr) w.chi = store + (size_t)(p0 / 64) * qc_hbm_keep_slot_elems(pg);
    if (backward) w.lam = w.chi + (size_t)NCH * T * N;  // lam directly behind chi: GRAD pairs v with v + nvec
    const int64_t amps = (int64_t)TA * N;
    if (pg->amplitude) {
      if (!kept)
        hipLaunchKernelGGL((k_hbm_init_amp<NCH>), dim3(qc_ceil_div(amps, 256)), dim3(256), 0, st, ajets, B, p0, TA, n, w.chi);
    } else {
      hipLaunchKernelGGL((k_hbm_wiredata<NCH>), dim3(qc_ceil_div((int64_t)T * n, 256)), dim3(256), 0, st, ajets, B, p0, T, n, w.wd,
                         trig_dev, absorb);
      if (backward)
        hipLaunchKernelGGL((k_hbm_init<NCH, true>), dim3(qc_ceil_div(amps, 256)), dim3(256), 0, st, w.wd, TA, n,
                           kept ? (Cplx*)nullptr : w.chi, w.ser);
      else
        hipLaunchKernelGGL((k_hbm_init<NCH, false>), dim3(qc_ceil_div(amps, 256)), dim3(256), 0, st, w.wd, TA, n, w.chi, w.ser);
    }
    const int64_t S = (int64_t)NCH * TA;
    if (staged && !kept) staged_forward(pg, trig_dev, umat, w, S, st);
    for (int g = 0; g < pg->n_gates && !staged && !kept; ++g) {
      const QcGate gt = pg->h_gates[g];
      if (gt.op == QC_U4)
        hipLaunchKernelGGL(k_hbm_u4, dim3(qc_ceil_div(S * (N / 4), 256)), dim3(256), 0, st, w.chi, S, n, gt.ba, gt.bb,
                           umat + (gt.slot * 2) * 32);
      else
        hipLaunchKernelGGL((k_hbm_gate<false, false>), dim3(qc_ceil_div(S * (N / 2), 256)), dim3(256), 0, st, w.chi, S, n, gt,
                           trig_dev, g, (float*)nullptr);
    }
    if (!backward) {
      if (N >= 256 * QC_XCH && n <= 19) {   // (the chunk kernel keeps 8 loop-bit accumulators: 2^19 / 8 / 256 = 2^8 iterations)
        hipLaunchKernelGGL((k_hbm_expval_chunk<NCH>), dim3(NCH * T, QC_XCH), dim3(256), 0, st, w.chi, TA, n, w.xpart, QC_XCH);
        hipLaunchKernelGGL(k_hbm_expval_fold, dim3(qc_ceil_div(NCH * T * n, 256)), dim3(256), 0, st, w.xpart, NCH * T, T, n,
                           QC_XCH, B, p0, qjets);
      } else {
        hipLaunchKernelGGL((k_hbm_expval<NCH>), dim3(NCH * T), dim3(256), 0, st, w.chi, TA, n, B, p0, qjets);
      }
      continue;
    }
    hipLaunchKernelGGL((k_hbm_lambda<NCH>), dim3(qc_ceil_div(amps, 256)), dim3(256), 0, st, w.chi, w.lam, TA, n, B, p0, qbar);
    hipLaunchKernelGGL(k_hbm_zero, dim3(qc_ceil_div(pg->n_params > 0 ? pg->n_params : 1, 256)), dim3(256), 0, st, w.acc,
                       pg->n_params > 0 ? pg->n_params : 1);
    const int nblk = qc_ceil_div(S * (N / 2), 256);
    if (staged) staged_backward(pg, trig_dev, umat, w, S, st);
    for (int g = pg->n_gates - 1; g >= 0 && !staged; --g) {
      const QcGate gt = pg->h_gates[g];
      if (gt.op == QC_U4) {
        hipLaunchKernelGGL(k_hbm_u4, dim3(qc_ceil_div(2 * S * (N / 4), 256)), dim3(256), 0, st, w.chi, 2 * S, n, gt.ba, gt.bb,
                           umat + (gt.slot * 2 + 1) * 32);
      } else if (gt.slot >= 0) {
        hipLaunchKernelGGL((k_hbm_gate<true, true>), dim3(nblk), dim3(256), 0, st, w.chi, 2 * S, n, gt, trig_dev, g, w.gblk);
        hipLaunchKernelGGL(k_hbm_fold, dim3(1), dim3(256), 0, st, w.gblk, nblk, w.acc + gt.slot);
      } else {
        hipLaunchKernelGGL((k_hbm_gate<true, false>), dim3(qc_ceil_div(2 * S * (N / 2), 256)), dim3(256), 0, st, w.chi, 2 * S,
                           n, gt, trig_dev, g, (float*)nullptr);
      }
    }
    if (pg->amplitude)
      hipLaunchKernelGGL((k_hbm_abar_amp<NCH>), dim3(qc_ceil_div((int64_t)NCH * T * n, 256)), dim3(256), 0, st, w.lam, TA, n, B, p0, abar);
    else
    {
      hipLaunchKernelGGL((k_hbm_abar<NCH>), dim3(n * T), dim3(256), 0, st, w.lam, w.ser, TA, n, B, p0, abar);
      if (absorb) hipLaunchKernelGGL(k_hbm_absorb_grad, dim3(n), dim3(64), 0, st, abar, B, p0, T, pg->d_gates, w.acc);
    }
    hipLaunchKernelGGL(k_hbm_store_row, dim3(qc_ceil_div(pg->n_params > 0 ? pg->n_params : 1, 256)), dim3(256), 0, st,
                       w.acc, pg->n_params, part + (row0 + p0 / 64) * part_stride);
  }
  return QC_OK;
}

// fused step with the final states kept per tile: `store` = ceil(B/64) slots of qc_hbm_keep_slot_elems() complex64
int qc_hbm_forward_keep(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                        int64_t B, void* ws, size_t ws_bytes, void* store, hipStream_t st) {
  return hbm_run<6>(pg, trig, umat, ajets, qjets, nullptr, nullptr, nullptr, 0, 0, B, ws, ws_bytes, st, (Cplx*)store);
}
int qc_hbm_backward_kept(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                         const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B,
                         void* ws, size_t ws_bytes, void* store, hipStream_t st) {
  return hbm_run<6>(pg, trig, umat, ajets, nullptr, qbar, abar, part, part_stride, row0, B, ws, ws_bytes, st, (Cplx*)store);
}

int qc_hbm_forward(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                   int64_t B, int nch, void* ws, size_t ws_bytes, hipStream_t st) {
  if (nch == 6) return hbm_run<6>(pg, trig, umat, ajets, qjets, nullptr, nullptr, nullptr, 0, 0, B, ws, ws_bytes, st);
  return hbm_run<1>(pg, trig, umat, ajets, qjets, nullptr, nullptr, nullptr, 0, 0, B, ws, ws_bytes, st);
}

int qc_hbm_backward(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                    const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B, int nch,
                    void* ws, size_t ws_bytes, hipStream_t st) {
  if (nch == 6) return hbm_run<6>(pg, trig, umat, ajets, nullptr, qbar, abar, part, part_stride, row0, B, ws, ws_bytes, st);
  return hbm_run<1>(pg, trig, umat, ajets, nullptr, qbar, abar, part, part_stride, row0, B, ws, ws_bytes, st);
}
