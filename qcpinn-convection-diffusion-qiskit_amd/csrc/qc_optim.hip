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
__global__ void __launch_bounds__(1024) k_adam(float* __restrict__ flat, int NP, float* __restrict__ prm,
                                               float* __restrict__ m, float* __restrict__ v,
                                               QcOptState* __restrict__ st, QcOptHyper hp,
                                               float* __restrict__ hist, int hist_cap,
                                               const QcGate* __restrict__ prog, int n_gates, int theta_off,
                                               QcTrig* __restrict__ trig) {
  __shared__ float s_red[16];
  float ss = 0.f;
  for (int i = threadIdx.x; i < NP; i += blockDim.x) ss += flat[i] * flat[i];
  const float norm = sqrtf(block_sum_1024(ss, s_red));
  float coef = hp.max_norm / (norm + 1e-6f);           // torch.nn.utils.clip_grad_norm_
  coef = coef > 1.f ? 1.f : coef;

  const int step = st->step + 1;
  const float lr = st->lr;
  const float b1 = (float)hp.beta1, b2 = (float)hp.beta2;
  const double bc1 = 1.0 - pow(hp.beta1, (double)step);
  const float bc2s = (float)sqrt(1.0 - pow(hp.beta2, (double)step));
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
  if (prog != nullptr)
    for (int g = threadIdx.x; g < n_gates; g += blockDim.x) {
      const QcGate gt = prog[g];
      QcTrig tr = {1.f, 0.f, 0.f, 0.f};
      if (gt.op != QC_U4 && gt.slot >= 0) {
        tr.th = prm[theta_off + gt.slot];
        sincosf(0.5f * tr.th, &tr.s, &tr.c);
      }
      trig[g] = tr;
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
    if (hist != nullptr && step - 1 < hist_cap) hist[step - 1] = loss;
  }
}

__global__ void k_prep_trig(const QcGate* __restrict__ prog, int n_gates, const float* __restrict__ theta,
                            QcTrig* __restrict__ trig) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_gates) return;
  const QcGate gt = prog[g];
  QcTrig tr = {1.f, 0.f, 0.f, 0.f};
  if (gt.op != QC_U4 && gt.slot >= 0) {
    tr.th = theta[gt.slot];
    sincosf(0.5f * tr.th, &tr.s, &tr.c);
  }
  trig[g] = tr;
}

}  // namespace

int qc_opt_reduce_rows(const float* part, int64_t rows, int64_t stride, int ncols, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_rows, dim3(qc_ceil_div(ncols, 64)), dim3(1024), 0, st, part, rows, stride, ncols, out);
  return QC_OK;
}

int qc_opt_adam(float* flat, int NP, float* prm, float* m, float* v, QcOptState* state, QcOptHyper hp,
                float* hist, int hist_cap, const qc_program* pg, int theta_off, QcTrig* trig, hipStream_t st) {
  hipLaunchKernelGGL(k_adam, dim3(1), dim3(1024), 0, st, flat, NP, prm, m, v, state, hp, hist, hist_cap,
                     pg ? pg->d_gates : nullptr, pg ? pg->n_gates : 0, theta_off, trig);
  return QC_OK;
}

int qc_opt_prep_trig(const qc_program* pg, const float* theta, QcTrig* trig, hipStream_t st) {
  hipLaunchKernelGGL(k_prep_trig, dim3(qc_ceil_div(pg->n_gates, 256)), dim3(256), 0, st, pg->d_gates,
                     pg->n_gates, theta, trig);
  return QC_OK;
}
