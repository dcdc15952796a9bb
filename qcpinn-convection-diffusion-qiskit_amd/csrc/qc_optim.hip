// Gradient-row reduction, gradient clipping, Adam and the plateau scheduler, all on device so a
// training step never synchronises with the host.
//
// Mirrors the optimiser block of the reference step (trainer/diffusion_train.py:81-90):
//   clip_grad_norm_(max_norm=1) -> Adam(lr) -> ReduceLROnPlateau(min, factor 0.9, patience 1000)
// as configured in nn/DVPDESolver.py:59-64 (torch defaults otherwise: betas (0.9, 0.999),
// eps 1e-8, threshold 1e-4 relative, cooldown 0, min_lr 0, scheduler eps 1e-8).
#include "qc_internal.h"

namespace {

// Fixed-order column sums of the partial-row matrix: out[c] = sum_r part[r][c].
// grid.x = ceil(ncols/64); 16 waves per block, wave w takes rows w, w+16, ...
__global__ void __launch_bounds__(1024) k_reduce_rows(const float* __restrict__ part, int64_t rows,
                                                      int64_t stride, int ncols, float* __restrict__ out) {
  __shared__ float s[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < ncols) {
    int64_t r = wave;
    for (; r + 48 < rows; r += 64) {
      a0 += part[r * stride + col];
      a1 += part[(r + 16) * stride + col];
      a2 += part[(r + 32) * stride + col];
      a3 += part[(r + 48) * stride + col];
    }
    for (; r < rows; r += 16) a0 += part[r * stride + col];
  }
  s[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave == 0 && col < ncols) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += s[w][lane];
    out[col] = t;
  }
}

// beta^step by repeated squaring in double (<= 31 dependent multiplies; the generic pow() costs ~1 us of a one-block
// kernel's latency chain); agrees with pow() to a few ulp of double, far below the float it is converted to
__device__ __forceinline__ double qc_ipow(double b, int e) {
  double r = 1.0;
  while (e > 0) {
    if (e & 1) r *= b;
    b *= b;
    e >>= 1;
  }
  return r;
}

__device__ __forceinline__ float block_sum_1024(float v, float* s_red) {
  v = qc_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}

// One block.  flat = [grad[NP] | L_r, L_bc, L_ic].  Updates prm/m/v in place, advances the
// scheduler, appends the loss to hist[step], and rebuilds the gate trig table from the NEW theta.
struct QcAdamArgs {
  float* flat;
  int NP;
  float *prm, *m, *v;
  QcOptState* st;
  QcOptHyper hp;
  float* hist;
  int hist_cap;
  const QcGate* prog;
  int n_gates, theta_off, n_qubits;
  QcTrig* trig;
  QcDiagRuns runs;
};

__device__ __forceinline__ void adam_block(const QcAdamArgs& a, float* s_red) {
  float* __restrict__ flat = a.flat;
  const int NP = a.NP;
  float *__restrict__ prm = a.prm, *__restrict__ m = a.m, *__restrict__ v = a.v;
  QcOptState* __restrict__ st = a.st;
  const QcOptHyper& hp = a.hp;
  float* __restrict__ hist = a.hist;
  const int hist_cap = a.hist_cap, n_gates = a.n_gates, theta_off = a.theta_off;
  const QcGate* __restrict__ prog = a.prog;
  QcTrig* __restrict__ trig = a.trig;
  float ss = 0.f;
  for (int i = threadIdx.x; i < NP; i += blockDim.x) ss += flat[i] * flat[i];
  const float norm = sqrtf(block_sum_1024(ss, s_red));
  float coef = hp.max_norm / (norm + 1e-6f);           // torch.nn.utils.clip_grad_norm_
  coef = coef > 1.f ? 1.f : coef;

  const int step = st->step + 1;
  const float lr = st->lr;
  const float b1 = (float)hp.beta1, b2 = (float)hp.beta2;
  const double bc1 = 1.0 - qc_ipow(hp.beta1, step);
  const float bc2s = (float)sqrt(1.0 - qc_ipow(hp.beta2, step));
  const float step_size = (float)((double)lr / bc1);
  for (int i = threadIdx.x; i < NP; i += blockDim.x) {
    const float g = flat[i] * coef;
    flat[i] = g;                                       // leave the clipped gradient visible
    const float mi = m[i] + (1.f - b1) * (g - m[i]);       // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + hp.eps;
    prm[i] -= step_size * (mi / denom);
  }
  __syncthreads();
  if (prog != nullptr) {
    for (int g = threadIdx.x; g < n_gates; g += blockDim.x) {
      const QcGate gt = prog[g];
      QcTrig tr = {1.f, 0.f, 0.f, 0.f};
      if (gt.op != QC_U4 && gt.slot >= 0) {
        tr.th = prm[theta_off + gt.slot];
        sincosf(0.5f * tr.th, &tr.s, &tr.c);
      }
      trig[g] = tr;
    }
    __syncthreads();
    qc_fill_diag_tables(prog, n_gates, a.n_qubits, trig, threadIdx.x, a.runs);
  }
  if (threadIdx.x == 0) {
    const float lr_ = flat[NP], lb = flat[NP + 1], li = flat[NP + 2];
    const float loss = hp.w_res * lr_ + hp.w_bc * lb + hp.w_ic * li;
    // ReduceLROnPlateau.step(loss), mode "min", threshold_mode "rel", cooldown 0
    float best = st->best;
    int bad = st->num_bad;
    float new_lr = lr;
    if (loss < best * (1.f - hp.sched_threshold)) {
      best = loss;
      bad = 0;
    } else {
      bad += 1;
    }
    if (bad > hp.sched_patience) {
      const float cand = fmaxf(lr * hp.sched_factor, hp.sched_min_lr);
      if (lr - cand > hp.sched_eps) new_lr = cand;
      bad = 0;
    }
    st->lr = new_lr;
    st->best = best;
    st->num_bad = bad;
    st->step = step;
    st->last_loss = loss;
    st->last_norm = norm;
    st->loss_parts[0] = lr_;
    st->loss_parts[1] = lb;
    st->loss_parts[2] = li;
    const int hi = step - 1 - st->hist_base;
    if (hist != nullptr && hi >= 0 && hi < hist_cap) hist[hi] = loss;
  }
}

__global__ void __launch_bounds__(1024) k_adam(QcAdamArgs a) {
  __shared__ float s_red[16];
  adam_block(a, s_red);
}

// First level of the step's own row reduction, in place: block (cb, rs) folds rows rs, rs + RS,
// rs + 2 RS, ... of column block cb into row rs (only that block ever touches row rs of those columns).
// The RS = gridDim.y surviving rows are added, in order, by k_adam_fold (same call) or k_reduce_rows.
// 1 700 rows become ~7 dependent loads per wave on 384 blocks instead of ~27 on 12.
constexpr int QC_RED_WAVES = 8;
__global__ void __launch_bounds__(64 * QC_RED_WAVES) k_fold_rows(float* __restrict__ part, int64_t rows, int64_t stride,
                                                                 int ncols) {
  __shared__ float s[QC_RED_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const int RS = gridDim.y, rs = blockIdx.y;
  // rows r, r + hop, r + 2 hop, ... of this wave: the first EIGHT are requested together (config 2 has ~7 per wave: one
  // memory round trip instead of four dependent ones), then summed in the same fixed order as a two-chain loop
  float a0 = 0.f, a1 = 0.f;
  if (col < ncols) {
    const int64_t r0 = rs + (int64_t)RS * wave;
    const int64_t hop = (int64_t)RS * QC_RED_WAVES;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t r = r0 + k * hop;
      v[k] = r < rows ? part[r * stride + col] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      a0 += v[k];
      a1 += v[k + 1];
    }
    for (int64_t r = r0 + 8 * hop; r < rows; r += 2 * hop) {
      a0 += part[r * stride + col];
      if (r + hop < rows) a1 += part[(r + hop) * stride + col];
    }
  }
  s[wave][lane] = a0 + a1;
  __syncthreads();
  if (wave == 0 && col < ncols) {
    float t = s[0][lane];
#pragma unroll
    for (int w = 1; w < QC_RED_WAVES; ++w) t += s[w][lane];
    part[(int64_t)rs * stride + col] = t;
  }
}

// The optimiser update with everything it reads fetched in ONE round trip (state record, gradient or the
// RS folded rows of it, moments, parameters), for NP + 3 <= 3 * 1024: a single-block kernel is a chain of
// memory latencies, so the loads are issued together up front and the new theta reaches the trig table
// through LDS instead of a store -> load of global memory.  Arithmetic identical to adam_block.
// FOLD: flat[c] = part[0][c] + ... + part[RS-1][c] (second reduction level, fixed order) is formed here.
constexpr int QC_ADAM_K = 3;
template <bool FOLD>
__global__ void __launch_bounds__(1024) k_adam_fast(QcAdamArgs a, const float* __restrict__ part, int64_t stride, int RS) {
  __shared__ float s_red[16];
  __shared__ float s_loss[3];
  extern __shared__ float s_theta[];   // [n_theta], then (cos, sin)[n_gates][2]
  const int NP = a.NP, tid = threadIdx.x;
  const QcOptHyper& hp = a.hp;
  const int step = a.st->step + 1;
  const float lr = a.st->lr;
  float best = a.st->best;
  int bad = a.st->num_bad;
  const int hist_base = a.st->hist_base;
  float g[QC_ADAM_K], mi[QC_ADAM_K], vi[QC_ADAM_K], pi[QC_ADAM_K];
#pragma unroll
  for (int k = 0; k < QC_ADAM_K; ++k) {
    const int i = tid + k * 1024;
    g[k] = mi[k] = vi[k] = pi[k] = 0.f;
    if (i < NP) {
      mi[k] = a.m[i];
      vi[k] = a.v[i];
      pi[k] = a.prm[i];
    }
    if (i < NP + 3) {
      if constexpr (FOLD) {
        // all (<= 32) folded rows requested together: one memory round trip, then the same left-to-right sum
        float t = 0.f;
        float x[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = j < RS ? part[(int64_t)j * stride + i] : 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) t += x[j];       // (rows >= RS add +0.f: the sum is unchanged)
        for (int q = 32; q < RS; ++q) t += part[(int64_t)q * stride + i];
        g[k] = t;
      } else {
        g[k] = a.flat[i];
      }
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < QC_ADAM_K; ++k) {
    const int i = tid + k * 1024;
    if (i < NP) ss += g[k] * g[k];
    else if (i < NP + 3) s_loss[i - NP] = g[k];
  }
  const float norm = sqrtf(block_sum_1024(ss, s_red));   // (its barriers also publish s_loss)
  float coef = hp.max_norm / (norm + 1e-6f);
  coef = coef > 1.f ? 1.f : coef;
  const float b1 = (float)hp.beta1, b2 = (float)hp.beta2;
  const double bc1 = 1.0 - qc_ipow(hp.beta1, step);
  const float bc2s = (float)sqrt(1.0 - qc_ipow(hp.beta2, step));
  const float step_size = (float)((double)lr / bc1);
  const int n_theta = NP - a.theta_off;
  float* s_cs = s_theta + (a.prog != nullptr ? n_theta : 0);
#pragma unroll
  for (int k = 0; k < QC_ADAM_K; ++k) {
    const int i = tid + k * 1024;
    if (i < NP) {
      const float gc = g[k] * coef;
      const float mn = mi[k] + (1.f - b1) * (gc - mi[k]);
      const float vn = b2 * vi[k] + (1.f - b2) * gc * gc;
      const float denom = sqrtf(vn) / bc2s + hp.eps;
      const float pn = pi[k] - step_size * (mn / denom);
      a.flat[i] = gc;
      a.m[i] = mn;
      a.v[i] = vn;
      a.prm[i] = pn;
      if (a.prog != nullptr && i >= a.theta_off) s_theta[i - a.theta_off] = pn;
    } else if (FOLD && i < NP + 3) {
      a.flat[i] = g[k];
    }
  }
  __syncthreads();
  if (a.prog != nullptr)
    for (int gi = tid; gi < a.n_gates; gi += 1024) {
      const QcGate gt = a.prog[gi];
      QcTrig tr = {1.f, 0.f, 0.f, 0.f};
      if (gt.op != QC_U4 && gt.slot >= 0 && gt.slot < n_theta) {
        tr.th = s_theta[gt.slot];
        sincosf(0.5f * tr.th, &tr.s, &tr.c);
      }
      a.trig[gi] = tr;
      if (a.runs.n > 0) {             // (cos, sin) of every gate also to LDS, for the diagonal-run tables below
        s_cs[2 * gi] = tr.c;
        s_cs[2 * gi + 1] = tr.s;
      }
    }
  if (a.prog != nullptr && a.runs.n > 0) {
    __syncthreads();
    qc_fill_diag_tables(a.prog, a.n_gates, a.n_qubits, a.trig, tid, a.runs, s_cs);
  }
  if (tid == 0) {
    const float lr_ = s_loss[0], lb = s_loss[1], li = s_loss[2];
    const float loss = hp.w_res * lr_ + hp.w_bc * lb + hp.w_ic * li;
    float new_lr = lr;
    if (loss < best * (1.f - hp.sched_threshold)) {
      best = loss;
      bad = 0;
    } else {
      bad += 1;
    }
    if (bad > hp.sched_patience) {
      const float cand = fmaxf(lr * hp.sched_factor, hp.sched_min_lr);
      if (lr - cand > hp.sched_eps) new_lr = cand;
      bad = 0;
    }
    QcOptState o;
    o.lr = new_lr;
    o.best = best;
    o.num_bad = bad;
    o.step = step;
    o.last_loss = loss;
    o.last_norm = norm;
    o.loss_parts[0] = lr_;
    o.loss_parts[1] = lb;
    o.loss_parts[2] = li;
    o.hist_base = hist_base;
#pragma unroll
    for (int k = 0; k < 6; ++k) o.pad[k] = 0;
    *a.st = o;
    const int hi = step - 1 - hist_base;
    if (a.hist != nullptr && hi >= 0 && hi < a.hist_cap) a.hist[hi] = loss;
  }
}

// one block: per-gate (cos, sin, theta) entries, then the phase tables of the fused diagonal runs
__global__ void __launch_bounds__(256) k_prep_trig(const QcGate* __restrict__ prog, int n_gates, int n_qubits,
                                                   const float* __restrict__ theta, QcTrig* __restrict__ trig,
                                                   QcDiagRuns runs) {
  for (int g = threadIdx.x; g < n_gates; g += 256) {
    const QcGate gt = prog[g];
    QcTrig tr = {1.f, 0.f, 0.f, 0.f};
    if (gt.op != QC_U4 && gt.slot >= 0) {
      tr.th = theta[gt.slot];
      sincosf(0.5f * tr.th, &tr.s, &tr.c);
    }
    trig[g] = tr;
  }
  __syncthreads();
  qc_fill_diag_tables(prog, n_gates, n_qubits, trig, threadIdx.x, runs);
}

}  // namespace

int qc_opt_reduce_rows(const float* part, int64_t rows, int64_t stride, int ncols, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_rows, dim3(qc_ceil_div(ncols, 64)), dim3(1024), 0, st, part, rows, stride, ncols, out);
  return QC_OK;
}

static QcAdamArgs adam_args(float* flat, int NP, float* prm, float* m, float* v, QcOptState* state, QcOptHyper hp,
                            float* hist, int hist_cap, const qc_program* pg, int theta_off, QcTrig* trig) {
  QcAdamArgs a;
  a.flat = flat; a.NP = NP; a.prm = prm; a.m = m; a.v = v; a.st = state; a.hp = hp; a.hist = hist; a.hist_cap = hist_cap;
  a.prog = pg ? pg->d_gates : nullptr; a.n_gates = pg ? pg->n_gates : 0; a.theta_off = theta_off; a.trig = trig;
  a.n_qubits = pg ? pg->n_qubits : 0;
  a.runs = qc_diag_runs_of(pg);
  return a;
}

static bool adam_fast_ok(int NP) { return NP + 3 <= QC_ADAM_K * 1024; }

int qc_opt_adam(float* flat, int NP, float* prm, float* m, float* v, QcOptState* state, QcOptHyper hp,
                float* hist, int hist_cap, const qc_program* pg, int theta_off, QcTrig* trig, hipStream_t st) {
  const QcAdamArgs a = adam_args(flat, NP, prm, m, v, state, hp, hist, hist_cap, pg, theta_off, trig);
  if (adam_fast_ok(NP))
    hipLaunchKernelGGL((k_adam_fast<false>), dim3(1), dim3(1024), sizeof(float) * (pg ? NP - theta_off + 2 * pg->n_gates : 0), st, a,
                       (const float*)nullptr, (int64_t)0, 0);
  else
    hipLaunchKernelGGL(k_adam, dim3(1), dim3(1024), 0, st, a);
  return QC_OK;
}

// In-place first level over the step's own partial-row matrix; returns the number of surviving rows.
int qc_opt_fold_rows(float* part, int64_t rows, int64_t stride, int ncols, hipStream_t st) {
  const int RS = (int)(rows < 32 ? rows : 32);
  hipLaunchKernelGGL(k_fold_rows, dim3(qc_ceil_div(ncols, 64), RS), dim3(64 * QC_RED_WAVES), 0, st, part, rows, stride, ncols);
  return RS;
}

// Second level + optimiser update (flat = [grad | 3 loss parts] is written as well).
int qc_opt_adam_fold(const float* part, int64_t stride, int RS, float* flat, int NP, float* prm, float* m, float* v,
                     QcOptState* state, QcOptHyper hp, float* hist, int hist_cap, const qc_program* pg, int theta_off,
                     QcTrig* trig, hipStream_t st) {
  if (adam_fast_ok(NP)) {
    hipLaunchKernelGGL((k_adam_fast<true>), dim3(1), dim3(1024), sizeof(float) * (pg ? NP - theta_off + 2 * pg->n_gates : 0), st,
                       adam_args(flat, NP, prm, m, v, state, hp, hist, hist_cap, pg, theta_off, trig), part, stride, RS);
    return QC_OK;
  }
  qc_opt_reduce_rows(part, RS, stride, NP + 3, flat, st);
  return qc_opt_adam(flat, NP, prm, m, v, state, hp, hist, hist_cap, pg, theta_off, trig, st);
}

int qc_opt_prep_trig(const qc_program* pg, const float* theta, QcTrig* trig, hipStream_t st) {
  hipLaunchKernelGGL(k_prep_trig, dim3(1), dim3(256), 0, st, pg->d_gates, pg->n_gates, pg->n_qubits, theta, trig,
                     qc_diag_runs_of(pg));
  return QC_OK;
}
