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

namespace {

struct Cplx {
  float re, im;
};

// ---------------------------------------------------------------- per-point, per-wire embedding data
template <int NCH>
__global__ void k_hbm_wiredata(const float* __restrict__ ajets, int64_t B, int64_t p0, int T, int n,
                               float* __restrict__ wd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * n) return;
  const int t = i / n, w = i % n;
  const int64_t p = p0 + t;
  float* o = wd + (size_t)i * 8;
  const float a = ajets[(int64_t)w * B + p];
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
    chi[((size_t)c * T + t) * N + k] = v;
    if constexpr (KEEP_SERIES) ser[((size_t)c * T + t) * N + k] = m;
  }
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
__global__ void k_hbm_store_row(const float* __restrict__ acc, int n_params, float* __restrict__ row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_params) row[i] = acc[i];
}

struct Ws {
  Cplx* chi;
  Cplx* lam;
  float* ser;
  float* wd;
  float* gblk;
  float* acc;
};

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

// bytes of workspace for one call (tile of up to 64 points at a time)
size_t qc_hbm_workspace_bytes(const qc_program* pg, int nch, bool backward) {
  const size_t N = (size_t)1 << pg->n_qubits, T = 64;
  size_t b = align_up(sizeof(Cplx) * nch * T * N);                 // chi
  if (backward) {
    b += align_up(sizeof(Cplx) * nch * T * N);                     // lam (placed directly behind the used part of chi)
    b += align_up(sizeof(float) * nch * T * N);                    // series
    b += align_up(sizeof(float) * (size_t)qc_ceil_div((int64_t)nch * T * (N / 2), 256));  // gblk
    b += align_up(sizeof(float) * (pg->n_params > 0 ? pg->n_params : 1));                  // acc
  }
  b += align_up(sizeof(float) * T * pg->n_qubits * 8);
  return b;
}

static Ws carve(const qc_program* pg, int nch, bool backward, void* ws) {
  const size_t N = (size_t)1 << pg->n_qubits, T = 64;
  char* p = (char*)ws;
  Ws w = {};
  w.chi = (Cplx*)p; p += align_up(sizeof(Cplx) * nch * T * N);
  if (backward) {
    w.lam = (Cplx*)p; p += align_up(sizeof(Cplx) * nch * T * N);
    w.ser = (float*)p; p += align_up(sizeof(float) * nch * T * N);
    w.gblk = (float*)p; p += align_up(sizeof(float) * (size_t)qc_ceil_div((int64_t)nch * T * (N / 2), 256));
    w.acc = (float*)p; p += align_up(sizeof(float) * (pg->n_params > 0 ? pg->n_params : 1));
  }
  w.wd = (float*)p;
  return w;
}

template <int NCH>
static int hbm_run(const qc_program* pg, const QcTrig* trig_dev, const float* umat,
                   const float* ajets, float* qjets, const float* qbar, float* abar, float* part, int64_t part_stride,
                   int64_t row0, int64_t B, void* ws, size_t ws_bytes, hipStream_t st) {
  const bool backward = qbar != nullptr;
  if (!ws || ws_bytes < qc_hbm_workspace_bytes(pg, NCH, backward)) return QC_ERR_ARG;
  const int n = pg->n_qubits;
  const int64_t N = (int64_t)1 << n;
  Ws w = carve(pg, NCH, backward, ws);
  for (int64_t p0 = 0; p0 < B; p0 += 64) {
    const int T = (int)((B - p0) < 64 ? (B - p0) : 64);
    const int TA = T;                                   // layout stride = points of this tile
    if (backward) w.lam = w.chi + (size_t)NCH * T * N;  // lam directly behind chi: GRAD pairs v with v + nvec
    hipLaunchKernelGGL((k_hbm_wiredata<NCH>), dim3(qc_ceil_div((int64_t)T * n, 256)), dim3(256), 0, st, ajets, B, p0, T, n, w.wd);
    const int64_t amps = (int64_t)TA * N;
    if (backward)
      hipLaunchKernelGGL((k_hbm_init<NCH, true>), dim3(qc_ceil_div(amps, 256)), dim3(256), 0, st, w.wd, TA, n, w.chi, w.ser);
    else
      hipLaunchKernelGGL((k_hbm_init<NCH, false>), dim3(qc_ceil_div(amps, 256)), dim3(256), 0, st, w.wd, TA, n, w.chi, w.ser);
    const int64_t S = (int64_t)NCH * TA;
    for (int g = 0; g < pg->n_gates; ++g) {
      const QcGate gt = pg->h_gates[g];
      if (gt.op == QC_U4)
        hipLaunchKernelGGL(k_hbm_u4, dim3(qc_ceil_div(S * (N / 4), 256)), dim3(256), 0, st, w.chi, S, n, gt.ba, gt.bb,
                           umat + (gt.slot * 2) * 32);
      else
        hipLaunchKernelGGL((k_hbm_gate<false, false>), dim3(qc_ceil_div(S * (N / 2), 256)), dim3(256), 0, st, w.chi, S, n, gt,
                           trig_dev, g, (float*)nullptr);
    }
    if (!backward) {
      hipLaunchKernelGGL((k_hbm_expval<NCH>), dim3(NCH * T), dim3(256), 0, st, w.chi, TA, n, B, p0, qjets);
      continue;
    }
    hipLaunchKernelGGL((k_hbm_lambda<NCH>), dim3(qc_ceil_div(amps, 256)), dim3(256), 0, st, w.chi, w.lam, TA, n, B, p0, qbar);
    hipLaunchKernelGGL(k_hbm_zero, dim3(qc_ceil_div(pg->n_params > 0 ? pg->n_params : 1, 256)), dim3(256), 0, st, w.acc,
                       pg->n_params > 0 ? pg->n_params : 1);
    const int nblk = qc_ceil_div(S * (N / 2), 256);
    for (int g = pg->n_gates - 1; g >= 0; --g) {
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
    hipLaunchKernelGGL((k_hbm_abar<NCH>), dim3(n * T), dim3(256), 0, st, w.lam, w.ser, TA, n, B, p0, abar);
    hipLaunchKernelGGL(k_hbm_store_row, dim3(qc_ceil_div(pg->n_params > 0 ? pg->n_params : 1, 256)), dim3(256), 0, st,
                       w.acc, pg->n_params, part + (row0 + p0 / 64) * part_stride);
  }
  return QC_OK;
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
