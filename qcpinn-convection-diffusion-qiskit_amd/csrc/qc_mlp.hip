// Classical pre/post networks of DVPDESolver with forward-mode derivative channels, the PDE
// residual, the weighted MSE loss and their reverse pass.
//
// Replaces, for the fused path, what the reference runs as torch modules + 5 autograd.grad calls
// + loss.backward():
//   pre  : Linear(3,H) -> Tanh -> Linear(H,n)      nn/DVPDESolver.py:37-43
//   post : Linear(n,H) -> Tanh -> Linear(H,1)      nn/DVPDESolver.py:45-51
//   residual = u_t/s_t + v_x u_x/s_x + v_y u_y/s_y - D (u_xx/s_x^2 + u_yy/s_y^2)    nn/pde.py:53-72
//            = c_t u_t + c_x u_x + c_y u_y - (d_xx u_xx + d_yy u_yy) with the coefficients of QcPde
//   targets u(t,x,y), r(t,x,y) (incl. the -400 constant)           data/diffusion_dataset.py:20-38
//   loss parts = MSE(residual, r), MSE(u_bc, u), MSE(u_ic, u)      trainer/diffusion_train.py:44-47
//
// Parameters live in ONE flat fp32 buffer in torch's model.parameters() order:
//   W1[H][3] b1[H] W2[n][H] b2[n] | W3[H][n] b3[H] W4[H] b4 | theta[L*P]
// Gradients leave every kernel as one "partial row" per block ([rows][stride] buffer, same column
// order, 3 extra columns for the loss sums); qc_optim.hip reduces rows in a fixed order, so the
// result is bit-reproducible (no float atomics).
//
// NCH = 6 : residual points, channels {value, d/dt, d/dx, d/dy, d2/dx2, d2/dy2};
// NCH = 1 : boundary / initial-condition points, value channel only.
#include "qc_internal.h"
#include "qc_philox.h"

namespace {

// tanh(x) = 1 - 2/(exp(2x)+1) on the hardware exp/rcp units (abs. error ~1e-7; saturates cleanly
// to +-1 for |x| large, where exp overflows to inf or underflows to 0).
// (__frcp_rn is a correctly rounded division on this target: ten instructions with v_div_scale / v_div_fmas / v_div_fixup;
// the hardware reciprocal is 1 ulp, far inside the 1e-7 of the exp approximation.)
__device__ __forceinline__ float qc_tanh(float x) {
  const float e = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// register pairs for the contraction blocks of the six-channel kernels: two derivative channels per 64-bit register,
// v_pk_fma_f32 with a broadcast scalar weight (two multiply-adds per instruction; csrc/qc_gates.h has the issue rates)
typedef float mf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mf2 m_dup(const float a) { return (mf2){a, a}; }
__device__ __forceinline__ mf2 m_fma(const mf2 a, const mf2 b, const mf2 c) { return __builtin_elementwise_fma(a, b, c); }

// analytic solution and forcing term, data/diffusion_dataset.py:20-38
__device__ __forceinline__ float analytic_u(float t, float x, float y) {
  const float dx = x - 0.5f, dy = y - 0.5f;
  return expf(-100.f * (dx * dx + dy * dy)) * expf(-t);
}
__device__ __forceinline__ float analytic_r(float t, float x, float y, float D, float vx, float vy) {
  const float u = analytic_u(t, x, y);
  const float dx = x - 0.5f, dy = y - 0.5f;
  const float ut = -u, ux = -200.f * dx * u, uy = -200.f * dy * u;
  const float uxx = (40000.f * dx * dx - 400.f) * u;  // the reference's constant (:31-34)
  const float uyy = (40000.f * dy * dy - 400.f) * u;
  return ut + vx * ux + vy * uy - D * (uxx + uyy);
}

// second workload (train_hybrid_qpinn.py:116-131): u = sin(pi x) sin(pi y) exp(-2 pi^2 D t); its PDE
// u_t = D (u_xx + u_yy) has no forcing term, and u vanishes on the four boundary faces
__device__ __forceinline__ float analytic_u_diffusion(float t, float x, float y, float D) {
  const float pi = 3.14159265358979323846f;
  return sinf(pi * x) * sinf(pi * y) * expf(-2.f * pi * pi * D * t);
}

// ================================================================== pre network, forward jets
// Block = 4 waves on ONE 64-point tile: lane = collocation point, wave w takes a quarter of the
// hidden units (wave-uniform weights -> scalar loads); the four partial angle jets meet in LDS.
// Four waves per tile (instead of one) keep >= 4 waves per SIMD in flight at B = 65 536.
constexpr int QC_MS = 4;

template <int N, int NCH>
__device__ __forceinline__ void k_pre_fwd_body(const int64_t bid, const float* __restrict__ X, const float* __restrict__ prm,
                                                 QcLayout L, float* __restrict__ ajets, int64_t B,
                                                 const QcDraw* __restrict__ draw = nullptr, float* __restrict__ Xout = nullptr) {
  __shared__ float s_part[QC_MS][NCH * N][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t p = (int64_t)bid * 64 + lane;
  const int64_t pc = p < B ? p : B - 1;
  float t, x, y;
  if (draw != nullptr && draw->enabled) {
    // the step's sampler folded into its first stage: this tile draws its own points (same Philox counters as
    // k_sample) and wave 0 leaves them in X for the later stages
    if (NCH == 6) qc_draw_point(0, draw->off_res + pc, 0, draw->seed, draw->step, t, x, y);
    else if (pc < draw->n_ic) qc_draw_point(1, draw->off_ic + pc, 0, draw->seed, draw->step, t, x, y);
    else qc_draw_point(2, draw->off_bc + (pc - draw->n_ic), draw->face_pts, draw->seed, draw->step, t, x, y);
    if (wave == 0 && p < B) {
      Xout[p * 3 + 0] = t;
      Xout[p * 3 + 1] = x;
      Xout[p * 3 + 2] = y;
    }
  } else {
    t = X[pc * 3 + 0];
    x = X[pc * 3 + 1];
    y = X[pc * 3 + 2];
  }
  // six channels: the accumulators of channels (0,1), (2,3), (4,5) share a register pair (packed multiply-adds)
  constexpr int NP2 = NCH == 6 ? 3 : 1;
  mf2 acc2[NP2][N];
#pragma unroll
  for (int cp = 0; cp < NP2; ++cp)
#pragma unroll
    for (int i = 0; i < N; ++i) acc2[cp][i] = (mf2){0.f, 0.f};
  const float* W1 = prm + L.oW1;
  const float* b1 = prm + L.ob1;
  const float* W2 = prm + L.oW2;
  const int hq = (L.H + QC_MS - 1) / QC_MS;
  const int m0 = wave * hq, m1 = (m0 + hq) < L.H ? (m0 + hq) : L.H;
  for (int m = m0; m < m1; ++m) {
    const float w0 = W1[3 * m], w1 = W1[3 * m + 1], w2 = W1[3 * m + 2];
    const float h = fmaf(w0, t, fmaf(w1, x, fmaf(w2, y, b1[m])));
    const float z = qc_tanh(h);
    if constexpr (NCH == 6) {
      const float d1 = 1.f - z * z, d2 = -2.f * z * d1;
      const mf2 zc2[3] = {(mf2){z, d1 * w0}, (mf2){d1 * w1, d1 * w2}, (mf2){d2 * w1 * w1, d2 * w2 * w2}};
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const mf2 wi = m_dup(W2[i * L.H + m]);
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) acc2[cp][i] = m_fma(wi, zc2[cp], acc2[cp][i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) acc2[0][i].x = fmaf(W2[i * L.H + m], z, acc2[0][i].x);
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) s_part[wave][c * N + i][lane] = (c & 1) ? acc2[c >> 1][i].y : acc2[c >> 1][i].x;
  __syncthreads();
  if (p < B) {
    for (int f = wave; f < NCH * N; f += QC_MS) {
      float v = (s_part[0][f][lane] + s_part[1][f][lane]) + (s_part[2][f][lane] + s_part[3][f][lane]);
      if (f < N) v += prm[L.ob2 + f];
      ajets[(int64_t)f * B + p] = v;
    }
  }
}

template <int N, int NCH>
__global__ void __launch_bounds__(256) k_pre_fwd(const float* __restrict__ X, const float* __restrict__ prm,
                                                 QcLayout L, float* __restrict__ ajets, int64_t B) {
  k_pre_fwd_body<N, NCH>(blockIdx.x, X, prm, L, ajets, B);
}

// ================================================================== pre network, reverse pass
// lane = hidden unit (each lane owns one row of W1 / column of W2, so weight gradients need no
// cross-lane reduction).  The block's 64-point tile is staged in LDS and split over PS groups of HB
// threads (16 points each, read as LDS broadcasts); the groups' accumulators meet in LDS.
template <int N, int NCH>
__device__ __forceinline__ void k_pre_bwd_body(const int64_t bid, const float* __restrict__ X, const float* __restrict__ prm, QcLayout L,
                          const float* __restrict__ abar, float* __restrict__ part, int64_t part_stride,
                          int64_t row0, int64_t B, int HB, int PS) {
  // six channels: the cotangents of channels (0,1), (2,3), (4,5) of one wire sit side by side in LDS (one 64-bit
  // broadcast read) and ride through the two contraction blocks as register pairs (packed multiply-adds)
  constexpr int NP2 = NCH == 6 ? 3 : 1;
  __shared__ float sX[3][64];
  __shared__ mf2 sA2[NP2 * N][64];   // [cp * N + i][point] = (abar of channel 2 cp, channel 2 cp + 1) (NCH = 1: .x only)
  extern __shared__ float s_acc[];  // [PS][4 + N][HB]
  const int64_t base = (int64_t)bid * 64;
  const int cnt = (int)((B - base) < 64 ? (B - base) : 64);
  for (int i = threadIdx.x; i < 64 * 3; i += blockDim.x) {
    const int pp = i / 3, k = i % 3;
    sX[k][pp] = pp < cnt ? X[(base + pp) * 3 + k] : 0.f;
  }
  for (int i = threadIdx.x; i < NCH * N * 64; i += blockDim.x) {
    const int f = i >> 6, pp = i & 63;      // f = c * N + i
    const int c = f / N, w = f % N;
    const float v = pp < cnt ? abar[(int64_t)f * B + base + pp] : 0.f;
    if (c & 1) sA2[(c >> 1) * N + w][pp].y = v;
    else sA2[(c >> 1) * N + w][pp].x = v;
  }
  __syncthreads();

  const int grp = threadIdx.x / HB, m = threadIdx.x % HB;   // threads past HB * PS (block rounded up to waves) idle
  const int per = (64 + PS - 1) / PS;
  const int p0 = grp * per, p1 = (p0 + per) < cnt ? (p0 + per) : cnt;
  float gW1[3] = {0.f, 0.f, 0.f}, gb1 = 0.f;
  mf2 gW2p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) gW2p[i] = (mf2){0.f, 0.f};
  if (grp < PS && m < L.H) {
    const float w0 = prm[L.oW1 + 3 * m], w1 = prm[L.oW1 + 3 * m + 1], w2 = prm[L.oW1 + 3 * m + 2];
    const float bb = prm[L.ob1 + m];
    float w2c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w2c[i] = prm[L.oW2 + i * L.H + m];
    for (int pp = p0; pp < p1; ++pp) {
      const float t = sX[0][pp], x = sX[1][pp], y = sX[2][pp];
      const float h = fmaf(w0, t, fmaf(w1, x, fmaf(w2, y, bb)));
      const float z = qc_tanh(h);
      const float d1 = 1.f - z * z;
      if constexpr (NCH == 6) {
        mf2 a2[3][N], zb2[3];
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {
          zb2[cp] = (mf2){0.f, 0.f};
#pragma unroll
          for (int i = 0; i < N; ++i) {
            a2[cp][i] = sA2[cp * N + i][pp];
            zb2[cp] = m_fma(m_dup(w2c[i]), a2[cp][i], zb2[cp]);
          }
        }
        const float d2 = -2.f * z * d1;
        const float d3 = -2.f * (d1 * d1 + z * d2);
        const mf2 zc2[3] = {(mf2){z, d1 * w0}, (mf2){d1 * w1, d1 * w2}, (mf2){d2 * w1 * w1, d2 * w2 * w2}};
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int cp = 0; cp < 3; ++cp) gW2p[i] = m_fma(a2[cp][i], zc2[cp], gW2p[i]);
        const float zb[6] = {zb2[0].x, zb2[0].y, zb2[1].x, zb2[1].y, zb2[2].x, zb2[2].y};
        const float hb = zb[0] * d1 + (zb[1] * w0 + zb[2] * w1 + zb[3] * w2) * d2 +
                         (zb[4] * w1 * w1 + zb[5] * w2 * w2) * d3;
        gW1[0] += hb * t + zb[1] * d1;
        gW1[1] += hb * x + zb[2] * d1 + 2.f * zb[4] * d2 * w1;
        gW1[2] += hb * y + zb[3] * d1 + 2.f * zb[5] * d2 * w2;
        gb1 += hb;
      } else {
        float zb0 = 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const float a = sA2[i][pp].x;
          zb0 = fmaf(w2c[i], a, zb0);
          gW2p[i].x = fmaf(a, z, gW2p[i].x);
        }
        const float hb = zb0 * d1;
        gW1[0] += hb * t;
        gW1[1] += hb * x;
        gW1[2] += hb * y;
        gb1 += hb;
      }
    }
  }
  if (grp < PS) {
    float* mine = s_acc + (size_t)grp * (4 + N) * HB;
    mine[0 * HB + m] = gW1[0];
    mine[1 * HB + m] = gW1[1];
    mine[2 * HB + m] = gW1[2];
    mine[3 * HB + m] = gb1;
#pragma unroll
    for (int i = 0; i < N; ++i) mine[(4 + i) * HB + m] = gW2p[i].x + gW2p[i].y;
  }
  __syncthreads();
  float* row = part + (row0 + bid) * part_stride;
  if (grp == 0 && m < L.H) {
    float tot[4 + N];
#pragma unroll
    for (int k = 0; k < 4 + N; ++k) {
      float sum = 0.f;
      for (int g2 = 0; g2 < PS; ++g2) sum += s_acc[((size_t)g2 * (4 + N) + k) * HB + m];
      tot[k] = sum;
    }
    row[L.oW1 + 3 * m] = tot[0];
    row[L.oW1 + 3 * m + 1] = tot[1];
    row[L.oW1 + 3 * m + 2] = tot[2];
    row[L.ob1 + m] = tot[3];
#pragma unroll
    for (int i = 0; i < N; ++i) row[L.oW2 + i * L.H + m] = tot[4 + i];
  }
  if (threadIdx.x < N) {  // b2 only feeds the value channel
    float sum = 0.f;
    for (int pp = 0; pp < cnt; ++pp) sum += sA2[threadIdx.x][pp].x;
    row[L.ob2 + threadIdx.x] = sum;
  }
}

template <int N, int NCH>
__global__ void k_pre_bwd(const float* __restrict__ X, const float* __restrict__ prm, QcLayout L,
                          const float* __restrict__ abar, float* __restrict__ part, int64_t part_stride,
                          int64_t row0, int64_t B, int HB, int PS) {
  k_pre_bwd_body<N, NCH>(blockIdx.x, X, prm, L, abar, part, part_stride, row0, B, HB, PS);
}

// ================================================================== post network + PDE + loss
// Point kernel: lane = collocation point (scalar weights), 4 tiles per block.
// MODE 0: forward only  -> u, residual
// MODE 1: reverse only: cotangents (ubar, rbar) come from memory (autograd path) -> qbar
// MODE 2: forward + analytic targets + squared error (loss sums into the tile's partial row)
//         + reverse -> qbar; the per-point cotangents (ubar, rbar-scale) are left in ub_out[2][B]
// MODE 3: reverse only, GENERAL cotangents: in_ubar = [6][B], one per derivative channel of u (operators that are not
//         linear in the channels, e.g. Navier-Stokes' u u_x) -> qbar
// MODE 4: forward only, all six channels of u -> out_u[6][B] (value, d/dt, d/dx, d/dy, d2/dx2, d2/dy2)
// The weight gradients of W3/b3/W4/b4 are NOT formed here (they would need ~300 cross-lane
// reductions per wave): k_post_wg below forms them with lane = hidden unit.
template <int N, int NCH>
__device__ __forceinline__ void post_cotangents(float (&gb)[NCH], float& gw4, const float (&g)[NCH],
                                                const float (&ub)[NCH], const float z, const float w4) {
  const float d1 = 1.f - z * z;
  gw4 = ub[0] * z;
  gb[0] = ub[0] * d1;
  if constexpr (NCH == 6) {
    const float d2 = -2.f * z * d1;
    const float d3 = -2.f * (d1 * d1 + z * d2);
    const float gx2 = g[2] * g[2], gy2 = g[3] * g[3];
    const float lin = ub[1] * g[1] + ub[2] * g[2] + ub[3] * g[3];
    gw4 += d1 * lin + ub[4] * (d2 * gx2 + d1 * g[4]) + ub[5] * (d2 * gy2 + d1 * g[5]);
    gb[0] += d2 * lin + ub[4] * (d3 * gx2 + d2 * g[4]) + ub[5] * (d3 * gy2 + d2 * g[5]);
    gb[1] = ub[1] * d1;
    gb[2] = ub[2] * d1 + 2.f * ub[4] * d2 * g[2];
    gb[3] = ub[3] * d1 + 2.f * ub[5] * d2 * g[3];
    gb[4] = ub[4] * d1;
    gb[5] = ub[5] * d1;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) gb[c] *= w4;
}

template <int NCH>
__device__ __forceinline__ void expand_ub(float (&ub)[NCH], float ub0, float gsc, const QcPde& pde) {
  ub[0] = ub0;
  if constexpr (NCH == 6) {
    ub[1] = gsc * pde.c_t;
    ub[2] = gsc * pde.c_x;
    ub[3] = gsc * pde.c_y;
    ub[4] = -pde.d_xx * gsc;
    ub[5] = -pde.d_yy * gsc;
  }
}

template <int N, int NCH, int MODE>
__device__ __forceinline__ void k_post_body(const int64_t bid, const float* __restrict__ X, const float* __restrict__ prm, QcLayout L,
                                              QcPde pde, const float* __restrict__ qjets,
                                              float* __restrict__ out_u, float* __restrict__ out_res,
                                              const float* __restrict__ in_ubar, const float* __restrict__ in_rbar,
                                              float* __restrict__ qbar, float* __restrict__ part,
                                              int64_t part_stride, int64_t row0, int64_t B) {
  // block = 4 waves on one 64-point tile; wave w owns a quarter of the hidden units
  __shared__ float s_buf[QC_MS][NCH * N][64];   // partial u jets first (NCH rows), partial qbar later
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = bid;
  const int64_t p = tile * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  float q[NCH][N];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) q[c][i] = qjets[((int64_t)c * N + i) * B + pc];
  const float* W3 = prm + L.oW3;
  const float* b3 = prm + L.ob3;
  const float* W4 = prm + L.oW4;
  const int hq = (L.H + QC_MS - 1) / QC_MS;
  const int m0 = wave * hq, m1 = (m0 + hq) < L.H ? (m0 + hq) : L.H;

  float ub0 = 0.f, gsc = 0.f;  // cotangent of u, and of the residual
  // MODE 2: the reverse pass is linear in its single non-zero cotangent (the residual's for residual points,
  // u's for value points), so it is accumulated for a UNIT cotangent inside the forward loop and scaled
  // afterwards - the pre-activations and tanh are formed once per hidden unit, not twice.
  float qbu[MODE == 2 ? NCH : 1][N];
  if constexpr (MODE == 2) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) qbu[c][i] = 0.f;
  }
  if constexpr (MODE == 0 || MODE == 2 || MODE == 4) {
    float u[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) u[c] = 0.f;
    float ub_unit[NCH];
    expand_ub<NCH>(ub_unit, NCH == 6 ? 0.f : 1.f, NCH == 6 ? 1.f : 0.f, pde);
    for (int m = m0; m < m1; ++m) {
      float g[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float sum = (c == 0) ? b3[m] : 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) sum = fmaf(W3[m * N + i], q[c][i], sum);
        g[c] = sum;
      }
      const float z = qc_tanh(g[0]);
      const float w4 = W4[m];
      u[0] = fmaf(w4, z, u[0]);
      if constexpr (NCH == 6) {
        const float d1 = 1.f - z * z, d2 = -2.f * z * d1;
        u[1] = fmaf(w4, d1 * g[1], u[1]);
        u[2] = fmaf(w4, d1 * g[2], u[2]);
        u[3] = fmaf(w4, d1 * g[3], u[3]);
        u[4] = fmaf(w4, d2 * g[2] * g[2] + d1 * g[4], u[4]);
        u[5] = fmaf(w4, d2 * g[3] * g[3] + d1 * g[5], u[5]);
      }
      if constexpr (MODE == 2) {
        float gb[NCH], gw4;
        post_cotangents<N, NCH>(gb, gw4, g, ub_unit, z, w4);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const float w3 = W3[m * N + i];
#pragma unroll
          for (int c = 0; c < NCH; ++c) qbu[c][i] = fmaf(w3, gb[c], qbu[c][i]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) s_buf[wave][c][lane] = u[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      u[c] = (s_buf[0][c][lane] + s_buf[1][c][lane]) + (s_buf[2][c][lane] + s_buf[3][c][lane]);
    __syncthreads();   // s_buf is reused for the qbar partials below
    u[0] += prm[L.ob4];
    if constexpr (MODE == 4) {
      if (live && wave == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) out_u[(int64_t)c * B + p] = u[c];
      }
      return;
    }
    float res = 0.f;
    if constexpr (NCH == 6) res = pde.c_t * u[1] + pde.c_x * u[2] + pde.c_y * u[3] - (pde.d_xx * u[4] + pde.d_yy * u[5]);
    if constexpr (MODE == 0) {
      if (live && wave == 0) {
        if (out_u) out_u[p] = u[0];
        if constexpr (NCH == 6)
          if (out_res) out_res[p] = res;
      }
      return;
    } else {
      const float t = X[pc * 3 + 0], x = X[pc * 3 + 1], y = X[pc * 3 + 2];
      float* row = part + (row0 + tile) * part_stride;
      if constexpr (NCH == 6) {
        const float target = pde.problem == QC_PB_PURE_DIFFUSION ? 0.f : analytic_r(t, x, y, pde.D, pde.vx, pde.vy);
        const float e = live ? res - target : 0.f;
        gsc = pde.w_res * e;
        if (wave == 0) {
          const float ls = qc_wave_sum_to_lane63(e * e * pde.inv_n_res);
          if (lane == 63) {
            row[L.NP + 0] = ls;
            row[L.NP + 1] = 0.f;
            row[L.NP + 2] = 0.f;
          }
        }
      } else {
        const bool seg_a = p < pde.n_seg_a;
        const float target = pde.problem == QC_PB_PURE_DIFFUSION
                                 ? (seg_a ? analytic_u_diffusion(t, x, y, pde.D) : 0.f)
                                 : analytic_u(t, x, y);
        const float e = live ? u[0] - target : 0.f;
        ub0 = (seg_a ? pde.w_val_a : pde.w_val_b) * e;
        if (wave == 0) {
          const float la = qc_wave_sum_to_lane63(seg_a ? e * e * pde.inv_n_a : 0.f);
          const float lb = qc_wave_sum_to_lane63(seg_a ? 0.f : e * e * pde.inv_n_b);
          if (lane == 63) {
            row[L.NP + 0] = 0.f;
            row[L.NP + 1] = lb;  // column order: residual, BC, IC; segment a = IC, b = BC
            row[L.NP + 2] = la;
          }
        }
      }
      if (live && wave == 0) {  // hand the per-point cotangents to k_post_wg
        out_u[p] = ub0;
        if constexpr (NCH == 6) out_res[p] = gsc;
      }
    }
  }
  if constexpr (MODE == 1) {
    ub0 = (live && in_ubar) ? in_ubar[pc] : 0.f;
    if constexpr (NCH == 6) gsc = (live && in_rbar) ? in_rbar[pc] : 0.f;
  }
  if constexpr (MODE == 2) {
    const float scale = NCH == 6 ? gsc : ub0;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) s_buf[wave][c * N + i][lane] = scale * qbu[c][i];
    __syncthreads();
    if (live) {
      for (int f = wave; f < NCH * N; f += QC_MS)
        qbar[(int64_t)f * B + p] = (s_buf[0][f][lane] + s_buf[1][f][lane]) + (s_buf[2][f][lane] + s_buf[3][f][lane]);
    }
  }
  if constexpr (MODE == 1 || MODE == 3) {
    float ub[NCH];
    if constexpr (MODE == 3) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) ub[c] = live ? in_ubar[(int64_t)c * B + pc] : 0.f;
    } else {
      expand_ub<NCH>(ub, ub0, gsc, pde);
    }
    float qb[NCH][N];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) qb[c][i] = 0.f;
    for (int m = m0; m < m1; ++m) {
      float g[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float sum = (c == 0) ? b3[m] : 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) sum = fmaf(W3[m * N + i], q[c][i], sum);
        g[c] = sum;
      }
      const float z = qc_tanh(g[0]);
      float gb[NCH], gw4;
      post_cotangents<N, NCH>(gb, gw4, g, ub, z, W4[m]);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float w3 = W3[m * N + i];
#pragma unroll
        for (int c = 0; c < NCH; ++c) qb[c][i] = fmaf(w3, gb[c], qb[c][i]);
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) s_buf[wave][c * N + i][lane] = qb[c][i];
    __syncthreads();
    if (live) {
      for (int f = wave; f < NCH * N; f += QC_MS)
        qbar[(int64_t)f * B + p] = (s_buf[0][f][lane] + s_buf[1][f][lane]) + (s_buf[2][f][lane] + s_buf[3][f][lane]);
    }
  }
}

template <int N, int NCH, int MODE>
__global__ void __launch_bounds__(256) k_post(const float* __restrict__ X, const float* __restrict__ prm, QcLayout L,
                                              QcPde pde, const float* __restrict__ qjets,
                                              float* __restrict__ out_u, float* __restrict__ out_res,
                                              const float* __restrict__ in_ubar, const float* __restrict__ in_rbar,
                                              float* __restrict__ qbar, float* __restrict__ part,
                                              int64_t part_stride, int64_t row0, int64_t B) {
  k_post_body<N, NCH, MODE>(blockIdx.x, X, prm, L, pde, qjets, out_u, out_res, in_ubar, in_rbar, qbar, part, part_stride, row0, B);
}

// Weight gradients of the post network: lane = hidden unit m (owns row m of W3, b3[m], W4[m]); the
// block walks its 64-point tile, reading the tile's <Z> jets and per-point cotangents from LDS as
// broadcasts.  No cross-lane reduction; one partial row per tile.
template <int N, int NCH>
__device__ __forceinline__ void k_post_wg_body(const int64_t bid, const float* __restrict__ prm, QcLayout L, QcPde pde, const float* __restrict__ qjets,
                          const float* __restrict__ ubar, const float* __restrict__ rbar,
                          float* __restrict__ part, int64_t part_stride, int64_t row0, int64_t B, int HB, int PS,
                          int gen = 0) {
  __shared__ float sQ[NCH * N][64];
  __shared__ float sU[NCH == 6 ? 6 : 2][64];   // rows 0, 1: (ubar, rbar); gen: one row per derivative channel of u
  extern __shared__ float s_acc[];  // [PS][N + 2][HB]
  const int64_t base = (int64_t)bid * 64;
  const int cnt = (int)((B - base) < 64 ? (B - base) : 64);
  for (int i = threadIdx.x; i < NCH * N * 64; i += blockDim.x) {
    const int f = i >> 6, pp = i & 63;
    sQ[f][pp] = pp < cnt ? qjets[(int64_t)f * B + base + pp] : 0.f;
  }
  if (NCH == 6 && gen) {
    for (int i = threadIdx.x; i < 6 * 64; i += blockDim.x) {
      const int k = i >> 6, pp = i & 63;
      sU[k][pp] = pp < cnt ? ubar[(int64_t)k * B + base + pp] : 0.f;
    }
  } else {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) {
      const int k = i >> 6, pp = i & 63;
      const float* src = k == 0 ? ubar : rbar;
      sU[k][pp] = (pp < cnt && src != nullptr) ? src[base + pp] : 0.f;
    }
  }
  __syncthreads();
  const int grp = threadIdx.x / HB, m = threadIdx.x % HB;   // threads past HB * PS (block rounded up to waves) idle
  const int per = (64 + PS - 1) / PS;
  const int p0 = grp * per, p1 = (p0 + per) < cnt ? (p0 + per) : cnt;
  float gW3[N], gb3 = 0.f, gW4 = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) gW3[i] = 0.f;
  if (grp < PS && m < L.H) {
    float w3[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w3[i] = prm[L.oW3 + m * N + i];
    const float b3m = prm[L.ob3 + m], w4 = prm[L.oW4 + m];
    for (int pp = p0; pp < p1; ++pp) {
      float g[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float sum = (c == 0) ? b3m : 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) sum = fmaf(w3[i], sQ[c * N + i][pp], sum);
        g[c] = sum;
      }
      const float z = qc_tanh(g[0]);
      float ub[NCH];
      if (NCH == 6 && gen) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) ub[c] = sU[c < (NCH == 6 ? 6 : 2) ? c : 0][pp];
      } else {
        expand_ub<NCH>(ub, sU[0][pp], sU[1][pp], pde);
      }
      float gb[NCH], gw4;
      post_cotangents<N, NCH>(gb, gw4, g, ub, z, w4);
      gW4 += gw4;
      gb3 += gb[0];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        float sum = gW3[i];
#pragma unroll
        for (int c = 0; c < NCH; ++c) sum = fmaf(gb[c], sQ[c * N + i][pp], sum);
        gW3[i] = sum;
      }
    }
  }
  if (grp < PS) {
    float* mine = s_acc + (size_t)grp * (N + 2) * HB;
#pragma unroll
    for (int i = 0; i < N; ++i) mine[i * HB + m] = gW3[i];
    mine[N * HB + m] = gb3;
    mine[(N + 1) * HB + m] = gW4;
  }
  __syncthreads();
  float* row = part + (row0 + bid) * part_stride;
  if (grp == 0 && m < L.H) {
    float tot[N + 2];
#pragma unroll
    for (int k = 0; k < N + 2; ++k) {
      float sum = 0.f;
      for (int g2 = 0; g2 < PS; ++g2) sum += s_acc[((size_t)g2 * (N + 2) + k) * HB + m];
      tot[k] = sum;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) row[L.oW3 + m * N + i] = tot[i];
    row[L.ob3 + m] = tot[N];
    row[L.oW4 + m] = tot[N + 1];
  }
  if (threadIdx.x == 0) {
    float sum = 0.f;
    for (int pp = 0; pp < cnt; ++pp) sum += sU[0][pp];
    row[L.ob4] = sum;
  }
}

template <int N, int NCH>
__global__ void k_post_wg(const float* __restrict__ prm, QcLayout L, QcPde pde, const float* __restrict__ qjets,
                          const float* __restrict__ ubar, const float* __restrict__ rbar,
                          float* __restrict__ part, int64_t part_stride, int64_t row0, int64_t B, int HB, int PS, int gen) {
  k_post_wg_body<N, NCH>(blockIdx.x, prm, L, pde, qjets, ubar, rbar, part, part_stride, row0, B, HB, PS, gen);
}

// ================================================================== value tiles, one wave per tile
// In the merged launches a value tile (64 boundary / initial points, value channel only) is too little work to
// split four ways with two LDS round trips: here each of a block's 4 waves owns one whole tile (all hidden units,
// scalar weights, no LDS, no barrier).  Same arithmetic as the NCH = 1 instances of the kernels above; the four
// per-quarter partial sums are still formed and added as (p0 + p1) + (p2 + p3) (same association as the m-split form).
template <int N>
__device__ __forceinline__ void k_pre_fwd_value4(const int64_t bid, const float* __restrict__ X, const float* __restrict__ prm,
                                                 QcLayout L, float* __restrict__ ajets, int64_t B,
                                                 const QcDraw* __restrict__ draw, float* __restrict__ Xout) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = bid * 4 + wave;
  if (tile * 64 >= B) return;
  const int64_t p = tile * 64 + lane;
  const int64_t pc = p < B ? p : B - 1;
  float t, x, y;
  if (draw != nullptr && draw->enabled) {
    if (pc < draw->n_ic) qc_draw_point(1, draw->off_ic + pc, 0, draw->seed, draw->step, t, x, y);
    else qc_draw_point(2, draw->off_bc + (pc - draw->n_ic), draw->face_pts, draw->seed, draw->step, t, x, y);
    if (p < B) {
      Xout[p * 3 + 0] = t;
      Xout[p * 3 + 1] = x;
      Xout[p * 3 + 2] = y;
    }
  } else {
    t = X[pc * 3 + 0];
    x = X[pc * 3 + 1];
    y = X[pc * 3 + 2];
  }
  const float* W1 = prm + L.oW1;
  const float* b1 = prm + L.ob1;
  const float* W2 = prm + L.oW2;
  const int hq = (L.H + QC_MS - 1) / QC_MS;
  float part[QC_MS][N];
#pragma unroll
  for (int k = 0; k < QC_MS; ++k) {
#pragma unroll
    for (int i = 0; i < N; ++i) part[k][i] = 0.f;
    const int m0 = k * hq, m1 = (m0 + hq) < L.H ? (m0 + hq) : L.H;
    for (int m = m0; m < m1; ++m) {
      const float h = fmaf(W1[3 * m], t, fmaf(W1[3 * m + 1], x, fmaf(W1[3 * m + 2], y, b1[m])));
      const float z = qc_tanh(h);
#pragma unroll
      for (int i = 0; i < N; ++i) part[k][i] = fmaf(W2[i * L.H + m], z, part[k][i]);
    }
  }
  if (p < B) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      ajets[(int64_t)i * B + p] = ((part[0][i] + part[1][i]) + (part[2][i] + part[3][i])) + prm[L.ob2 + i];
  }
}

// fused (mode 2) post stage of a value tile: u, squared error against the analytic target, cotangent of <Z>
template <int N>
__device__ __forceinline__ void k_post_value4(const int64_t bid, const float* __restrict__ X, const float* __restrict__ prm,
                                              QcLayout L, QcPde pde, const float* __restrict__ qjets,
                                              float* __restrict__ out_u, float* __restrict__ qbar, float* __restrict__ part,
                                              int64_t part_stride, int64_t row0, int64_t B) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = bid * 4 + wave;
  if (tile * 64 >= B) return;
  const int64_t p = tile * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  float q[N];
#pragma unroll
  for (int i = 0; i < N; ++i) q[i] = qjets[(int64_t)i * B + pc];
  const float* W3 = prm + L.oW3;
  const float* b3 = prm + L.ob3;
  const float* W4 = prm + L.oW4;
  const int hq = (L.H + QC_MS - 1) / QC_MS;
  float up[QC_MS], qbu[QC_MS][N];
#pragma unroll
  for (int k = 0; k < QC_MS; ++k) {
    up[k] = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) qbu[k][i] = 0.f;
    const int m0 = k * hq, m1 = (m0 + hq) < L.H ? (m0 + hq) : L.H;
    for (int m = m0; m < m1; ++m) {
      float g = b3[m];
#pragma unroll
      for (int i = 0; i < N; ++i) g = fmaf(W3[m * N + i], q[i], g);
      const float z = qc_tanh(g);
      const float w4 = W4[m];
      up[k] = fmaf(w4, z, up[k]);
      const float gb = (1.f - z * z) * w4;          // post_cotangents<N, 1> for a unit cotangent of u
#pragma unroll
      for (int i = 0; i < N; ++i) qbu[k][i] = fmaf(W3[m * N + i], gb, qbu[k][i]);
    }
  }
  const float u = ((up[0] + up[1]) + (up[2] + up[3])) + prm[L.ob4];
  const float t = X[pc * 3 + 0], x = X[pc * 3 + 1], y = X[pc * 3 + 2];
  const bool seg_a = p < pde.n_seg_a;
  const float target = pde.problem == QC_PB_PURE_DIFFUSION ? (seg_a ? analytic_u_diffusion(t, x, y, pde.D) : 0.f)
                                                           : analytic_u(t, x, y);
  const float e = live ? u - target : 0.f;
  const float ub0 = (seg_a ? pde.w_val_a : pde.w_val_b) * e;
  const float la = qc_wave_sum_to_lane63(seg_a ? e * e * pde.inv_n_a : 0.f);
  const float lb = qc_wave_sum_to_lane63(seg_a ? 0.f : e * e * pde.inv_n_b);
  if (lane == 63) {
    float* row = part + (row0 + tile) * part_stride;
    row[L.NP + 0] = 0.f;
    row[L.NP + 1] = lb;  // column order: residual, BC, IC; segment a = IC, b = BC
    row[L.NP + 2] = la;
  }
  if (live) {
    out_u[p] = ub0;
#pragma unroll
    for (int i = 0; i < N; ++i)
      qbar[(int64_t)i * B + p] = (ub0 * qbu[0][i] + ub0 * qbu[1][i]) + (ub0 * qbu[2][i] + ub0 * qbu[3][i]);
  }
}

// ================================================================== post stage of the training step in ONE kernel
// (MODE 2 of k_post + k_post_wg).  lane = collocation point throughout, weights are scalar operands:
//   phase A  u jets of the tile (pre-activations, tanh once per (point, hidden unit); the tanh values of a residual tile
//            are parked in LDS), residual, analytic target, squared error -> the point's cotangent (gsc or ub0);
//   phase B  with the ACTUAL cotangent: cotangents of the pre-activations gb_c, the <Z> jet cotangents
//            qbar_c[i] += W3[m][i] gb_c, and the weight gradients in the same lanes: the per-point products
//            sum_c gb_c q_c[i] (W3), gb_0 (b3), gw4 (W4) are summed over the wave's 64 points by DPP reductions and
//            stored straight into the tile's partial row - one row entry has exactly one producing wave.
// The lane = hidden-unit kernel (k_post_wg) recomputed pre-activations, tanh and cotangents per (hidden unit, point)
// from LDS copies of the jets: ~130 wave-instructions per (point, 64 hidden lanes) for what costs 24 multiply-adds and
// a few reductions here.  WPT = waves per tile: 4 (residual tiles: hidden units split four ways, partial sums meet in
// LDS) or 1 (value tiles: one wave owns the tile, four tiles per block).
template <int N, int NCH, int WPT>
__device__ __forceinline__ void k_post_fused_body(const int64_t bid, const float* __restrict__ X, const float* __restrict__ prm,
                                                   QcLayout L, QcPde pde, const float* __restrict__ qjets,
                                                   float* __restrict__ out_u, float* __restrict__ out_res,
                                                   float* __restrict__ qbar, float* __restrict__ part, int64_t part_stride,
                                                   int64_t row0, int64_t B, float* __restrict__ s_z) {
  __shared__ float s_buf[WPT == 4 ? QC_MS : 1][WPT == 4 ? NCH * N : 1][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = WPT == 4 ? bid : bid * 4 + wave;
  if (WPT == 1 && tile * 64 >= B) return;
  const int64_t p = tile * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  float q[NCH][N];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) q[c][i] = qjets[((int64_t)c * N + i) * B + pc];
  const float* W3 = prm + L.oW3;
  const float* b3 = prm + L.ob3;
  const float* W4 = prm + L.oW4;
  const int hq = (L.H + QC_MS - 1) / QC_MS;
  const int m0 = WPT == 4 ? wave * hq : 0, m1 = WPT == 4 ? ((m0 + hq) < L.H ? (m0 + hq) : L.H) : L.H;
  // ---------------- phase A
  float u[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) u[c] = 0.f;
  if constexpr (WPT == 4) {
    for (int m = m0; m < m1; ++m) {
      float g[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float sum = (c == 0) ? b3[m] : 0.f;
#pragma unroll
        for (int i = 0; i < N; ++i) sum = fmaf(W3[m * N + i], q[c][i], sum);
        g[c] = sum;
      }
      const float z = qc_tanh(g[0]);
      s_z[m * 64 + lane] = z;
      const float w4 = W4[m];
      u[0] = fmaf(w4, z, u[0]);
      if constexpr (NCH == 6) {
        const float d1 = 1.f - z * z, d2 = -2.f * z * d1;
        u[1] = fmaf(w4, d1 * g[1], u[1]);
        u[2] = fmaf(w4, d1 * g[2], u[2]);
        u[3] = fmaf(w4, d1 * g[3], u[3]);
        u[4] = fmaf(w4, d2 * g[2] * g[2] + d1 * g[4], u[4]);
        u[5] = fmaf(w4, d2 * g[3] * g[3] + d1 * g[5], u[5]);
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) s_buf[wave][c][lane] = u[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      u[c] = (s_buf[0][c][lane] + s_buf[1][c][lane]) + (s_buf[2][c][lane] + s_buf[3][c][lane]);
    __syncthreads();   // s_buf is reused for the qbar partials below
  } else {
    // one wave, all hidden units; the four quarter sums are formed and added as in the four-wave form
    float up[QC_MS];
#pragma unroll
    for (int k = 0; k < QC_MS; ++k) {
      up[k] = 0.f;
      const int k0 = k * hq, k1 = (k0 + hq) < L.H ? (k0 + hq) : L.H;
      for (int m = k0; m < k1; ++m) {
        float g = b3[m];
#pragma unroll
        for (int i = 0; i < N; ++i) g = fmaf(W3[m * N + i], q[0][i], g);
        up[k] = fmaf(W4[m], qc_tanh(g), up[k]);
      }
    }
    u[0] = (up[0] + up[1]) + (up[2] + up[3]);
  }
  u[0] += prm[L.ob4];
  // ---------------- residual / error / loss sums / per-point cotangent
  const float t = X[pc * 3 + 0], x = X[pc * 3 + 1], y = X[pc * 3 + 2];
  float* row = part + (row0 + tile) * part_stride;
  float ub0 = 0.f, gsc = 0.f;
  if constexpr (NCH == 6) {
    const float res = pde.c_t * u[1] + pde.c_x * u[2] + pde.c_y * u[3] - (pde.d_xx * u[4] + pde.d_yy * u[5]);
    const float target = pde.problem == QC_PB_PURE_DIFFUSION ? 0.f : analytic_r(t, x, y, pde.D, pde.vx, pde.vy);
    const float e = live ? res - target : 0.f;
    gsc = pde.w_res * e;
    if (wave == 0 || WPT == 1) {
      const float ls = qc_wave_sum_to_lane63(e * e * pde.inv_n_res);
      if (lane == 63) {
        row[L.NP + 0] = ls;
        row[L.NP + 1] = 0.f;
        row[L.NP + 2] = 0.f;
      }
    }
  } else {
    const bool seg_a = p < pde.n_seg_a;
    const float target = pde.problem == QC_PB_PURE_DIFFUSION ? (seg_a ? analytic_u_diffusion(t, x, y, pde.D) : 0.f)
                                                             : analytic_u(t, x, y);
    const float e = live ? u[0] - target : 0.f;
    ub0 = (seg_a ? pde.w_val_a : pde.w_val_b) * e;
    if (wave == 0 || WPT == 1) {
      const float la = qc_wave_sum_to_lane63(seg_a ? e * e * pde.inv_n_a : 0.f);
      const float lb = qc_wave_sum_to_lane63(seg_a ? 0.f : e * e * pde.inv_n_b);
      if (lane == 63) {
        row[L.NP + 0] = 0.f;
        row[L.NP + 1] = lb;  // column order: residual, BC, IC; segment a = IC, b = BC
        row[L.NP + 2] = la;
      }
    }
  }
  if (live && (wave == 0 || WPT == 1)) {   // the per-point cotangents (MODE 2 contract of qc_post)
    out_u[p] = ub0;
    if constexpr (NCH == 6) out_res[p] = gsc;
  }
  // d loss / d b4 = sum of the points' cotangents of u (zero for residual tiles: the loss sees u only through the residual)
  if (wave == 0 || WPT == 1) {
    const float sb4 = qc_wave_sum_to_lane63(ub0);
    if (lane == 63) row[L.ob4] = sb4;
  }
  // ---------------- phase B
  float ub[NCH];
  expand_ub<NCH>(ub, ub0, gsc, pde);
  float qb[NCH][N];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) qb[c][i] = 0.f;
  for (int m = m0; m < m1; ++m) {
    float g[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (WPT == 4 && c == 0) {   // only tanh(g_0) is needed, and that is parked
        g[0] = 0.f;
        continue;
      }
      float sum = (c == 0) ? b3[m] : 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) sum = fmaf(W3[m * N + i], q[c][i], sum);
      g[c] = sum;
    }
    const float z = WPT == 4 ? s_z[m * 64 + lane] : qc_tanh(g[0]);
    float gb[NCH], gw4;
    post_cotangents<N, NCH>(gb, gw4, g, ub, z, W4[m]);
    float wg[N + 2];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float w3 = W3[m * N + i];
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        qb[c][i] = fmaf(w3, gb[c], qb[c][i]);
        sum = fmaf(gb[c], q[c][i], sum);
      }
      wg[i] = sum;
    }
    wg[N] = gb[0];
    wg[N + 1] = gw4;
    qc_wave_sum_multi_to_lane63<N + 2>(wg);
    if (lane == 63) {
#pragma unroll
      for (int i = 0; i < N; ++i) row[L.oW3 + m * N + i] = wg[i];
      row[L.ob3 + m] = wg[N];
      row[L.oW4 + m] = wg[N + 1];
    }
  }
  if constexpr (WPT == 4) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) s_buf[wave][c * N + i][lane] = qb[c][i];
    __syncthreads();
    if (live) {
      for (int f = wave; f < NCH * N; f += QC_MS)
        qbar[(int64_t)f * B + p] = (s_buf[0][f][lane] + s_buf[1][f][lane]) + (s_buf[2][f][lane] + s_buf[3][f][lane]);
    }
  } else {
    if (live) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < N; ++i) qbar[((int64_t)c * N + i) * B + p] = qb[c][i];
    }
  }
}

// Residual tiles (six channels, four waves per tile) of the fused post stage with the three contraction blocks on
// register PAIRS: channels (0,1), (2,3), (4,5) of the <Z> jets ride in the two halves of one 64-bit register, so
//   g_c = sum_i W3[m][i] q_c[i],   qbar_c[i] += W3[m][i] gb_c,   sum_c gb_c q_c[i]
// are v_pk_fma_f32 with a broadcast scalar weight: two multiply-adds per instruction at the issue cost of one and a
// bit (csrc/qc_gates.h, tools/ubench/valu_issue.hip).  Same arithmetic as k_post_fused_body<N, 6, 4>, other order of the
// sums over channel pairs in the weight-gradient products.

template <int N>
__device__ __forceinline__ void k_post_fused6_body(const int64_t bid, const float* __restrict__ X, const float* __restrict__ prm,
                                                    QcLayout L, QcPde pde, const float* __restrict__ qjets,
                                                    float* __restrict__ out_u, float* __restrict__ out_res,
                                                    float* __restrict__ qbar, float* __restrict__ part, int64_t part_stride,
                                                    int64_t row0, int64_t B, float* __restrict__ s_z) {
  constexpr int NCH = 6;
  __shared__ float s_buf[QC_MS][NCH * N][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = bid;
  const int64_t p = tile * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  mf2 q2[3][N];
#pragma unroll
  for (int cp = 0; cp < 3; ++cp)
#pragma unroll
    for (int i = 0; i < N; ++i)
      q2[cp][i] = (mf2){qjets[((int64_t)(2 * cp) * N + i) * B + pc], qjets[((int64_t)(2 * cp + 1) * N + i) * B + pc]};
  const float* W3 = prm + L.oW3;
  const float* b3 = prm + L.ob3;
  const float* W4 = prm + L.oW4;
  const int hq = (L.H + QC_MS - 1) / QC_MS;
  const int m0 = wave * hq, m1 = (m0 + hq) < L.H ? (m0 + hq) : L.H;
  auto preact = [&](int m, mf2 (&g2)[3]) {
    g2[0] = (mf2){b3[m], 0.f};
    g2[1] = g2[2] = (mf2){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const mf2 w = m_dup(W3[m * N + i]);
#pragma unroll
      for (int cp = 0; cp < 3; ++cp) g2[cp] = m_fma(w, q2[cp][i], g2[cp]);
    }
  };
  // ---------------- phase A
  float u[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) u[c] = 0.f;
  for (int m = m0; m < m1; ++m) {
    mf2 g2[3];
    preact(m, g2);
    const float z = qc_tanh(g2[0].x);
    s_z[m * 64 + lane] = z;
    const float w4 = W4[m];
    const float d1 = 1.f - z * z, d2 = -2.f * z * d1;
    u[0] = fmaf(w4, z, u[0]);
    u[1] = fmaf(w4, d1 * g2[0].y, u[1]);
    u[2] = fmaf(w4, d1 * g2[1].x, u[2]);
    u[3] = fmaf(w4, d1 * g2[1].y, u[3]);
    u[4] = fmaf(w4, d2 * g2[1].x * g2[1].x + d1 * g2[2].x, u[4]);
    u[5] = fmaf(w4, d2 * g2[1].y * g2[1].y + d1 * g2[2].y, u[5]);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) s_buf[wave][c][lane] = u[c];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCH; ++c)
    u[c] = (s_buf[0][c][lane] + s_buf[1][c][lane]) + (s_buf[2][c][lane] + s_buf[3][c][lane]);
  __syncthreads();   // s_buf is reused for the qbar partials below
  u[0] += prm[L.ob4];
  // ---------------- residual / error / loss sums / per-point cotangent
  const float t = X[pc * 3 + 0], x = X[pc * 3 + 1], y = X[pc * 3 + 2];
  float* row = part + (row0 + tile) * part_stride;
  const float res = pde.c_t * u[1] + pde.c_x * u[2] + pde.c_y * u[3] - (pde.d_xx * u[4] + pde.d_yy * u[5]);
  const float target = pde.problem == QC_PB_PURE_DIFFUSION ? 0.f : analytic_r(t, x, y, pde.D, pde.vx, pde.vy);
  const float e = live ? res - target : 0.f;
  const float gsc = pde.w_res * e;
  if (wave == 0) {
    const float ls = qc_wave_sum_to_lane63(e * e * pde.inv_n_res);
    if (lane == 63) {
      row[L.NP + 0] = ls;
      row[L.NP + 1] = 0.f;
      row[L.NP + 2] = 0.f;
      row[L.ob4] = 0.f;   // the loss sees u only through the residual: no cotangent on u itself
    }
    if (live) {           // the per-point cotangents (MODE 2 contract of qc_post)
      out_u[p] = 0.f;
      out_res[p] = gsc;
    }
  }
  // ---------------- phase B
  float ub[NCH];
  expand_ub<NCH>(ub, 0.f, gsc, pde);
  mf2 qb2[3][N];
#pragma unroll
  for (int cp = 0; cp < 3; ++cp)
#pragma unroll
    for (int i = 0; i < N; ++i) qb2[cp][i] = (mf2){0.f, 0.f};
  for (int m = m0; m < m1; ++m) {
    mf2 g2[3];
    preact(m, g2);
    const float g[NCH] = {0.f, g2[0].y, g2[1].x, g2[1].y, g2[2].x, g2[2].y};
    const float z = s_z[m * 64 + lane];
    float gb[NCH], gw4;
    post_cotangents<N, NCH>(gb, gw4, g, ub, z, W4[m]);
    const mf2 gb2[3] = {(mf2){gb[0], gb[1]}, (mf2){gb[2], gb[3]}, (mf2){gb[4], gb[5]}};
    float wg[N + 2];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const mf2 w3 = m_dup(W3[m * N + i]);
      mf2 acc = gb2[0] * q2[0][i];
#pragma unroll
      for (int cp = 0; cp < 3; ++cp) {
        qb2[cp][i] = m_fma(w3, gb2[cp], qb2[cp][i]);
        if (cp > 0) acc = m_fma(gb2[cp], q2[cp][i], acc);
      }
      wg[i] = acc.x + acc.y;
    }
    wg[N] = gb[0];
    wg[N + 1] = gw4;
    qc_wave_sum_multi_to_lane63<N + 2>(wg);
    if (lane == 63) {
#pragma unroll
      for (int i = 0; i < N; ++i) row[L.oW3 + m * N + i] = wg[i];
      row[L.ob3 + m] = wg[N];
      row[L.oW4 + m] = wg[N + 1];
    }
  }
#pragma unroll
  for (int cp = 0; cp < 3; ++cp)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      s_buf[wave][(2 * cp) * N + i][lane] = qb2[cp][i].x;
      s_buf[wave][(2 * cp + 1) * N + i][lane] = qb2[cp][i].y;
    }
  __syncthreads();
  if (live) {
    for (int f = wave; f < NCH * N; f += QC_MS)
      qbar[(int64_t)f * B + p] = (s_buf[0][f][lane] + s_buf[1][f][lane]) + (s_buf[2][f][lane] + s_buf[3][f][lane]);
  }
}

// hidden widths the fused post kernel parks tanh values for (LDS: H x 64 floats per residual tile)
constexpr int QC_POST_FUSED_MAXH = 128;

template <int N, int NCH>
__global__ void __launch_bounds__(256) k_post_fused(const float* __restrict__ X, const float* __restrict__ prm, QcLayout L,
                                                    QcPde pde, const float* __restrict__ qjets, float* __restrict__ out_u,
                                                    float* __restrict__ out_res, float* __restrict__ qbar,
                                                    float* __restrict__ part, int64_t part_stride, int64_t row0, int64_t B) {
  extern __shared__ float s_dyn[];
  if constexpr (NCH == 6) k_post_fused6_body<N>(blockIdx.x, X, prm, L, pde, qjets, out_u, out_res, qbar, part, part_stride, row0, B, s_dyn);
  else k_post_fused_body<N, 1, 1>(blockIdx.x, X, prm, L, pde, qjets, out_u, out_res, qbar, part, part_stride, row0, B, s_dyn);
}

// ================================================================== K outputs behind one shared network (Navier-Stokes)
// A K-output post network Linear(n, H) -> Tanh -> Linear(H, K) (reference nn/pde.py:2-27 differentiates (u, v, p) of ONE
// model) shares pre network, circuit and the hidden layer; only the last layer has a row per output.  The six
// derivative channels of all K outputs come from one pass: f_c(m) (the hidden unit's channel values) is formed once,
// u_k,c = sum_m W4[k][m] f_c(m).  w4k = [K][H + 1] rows (W4[k][0..H-1], b4[k]); the W4 / b4 slots of the flat vector
// are not read.  MODE 4: forward, out = [K][6][B].  MODE 3: reverse, ubar = [K][6][B] -> qbar [6][n][B], the tile's row
// of the shared parameters (W3, b3) in `part` and of the last layer in `partk` ([rows][K * (H + 1)]).  lane = point;
// weight gradients by wave reductions as in k_post_fused_body.
constexpr int QC_KMAX = 4;

template <int N, int MODE>
__global__ void __launch_bounds__(256) k_post_multi(const float* __restrict__ prm, QcLayout L, int K,
                                                    const float* __restrict__ w4k, const float* __restrict__ qjets,
                                                    float* __restrict__ out_u, const float* __restrict__ ubar,
                                                    float* __restrict__ qbar, float* __restrict__ part, int64_t part_stride,
                                                    float* __restrict__ partk, int64_t partk_stride, int64_t row0, int64_t B) {
  constexpr int NCH = 6;
  __shared__ float s_buf[QC_MS][NCH * (N > QC_KMAX ? N : QC_KMAX)][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t tile = blockIdx.x;
  const int64_t p = tile * 64 + lane;
  const bool live = p < B;
  const int64_t pc = live ? p : B - 1;
  float q[NCH][N];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < N; ++i) q[c][i] = qjets[((int64_t)c * N + i) * B + pc];
  const float* W3 = prm + L.oW3;
  const float* b3 = prm + L.ob3;
  const int H1 = L.H + 1;
  const int hq = (L.H + QC_MS - 1) / QC_MS;
  const int m0 = wave * hq, m1 = (m0 + hq) < L.H ? (m0 + hq) : L.H;
  auto hidden = [&](int m, float (&g)[NCH], float& z, float (&f)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float sum = (c == 0) ? b3[m] : 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) sum = fmaf(W3[m * N + i], q[c][i], sum);
      g[c] = sum;
    }
    z = qc_tanh(g[0]);
    const float d1 = 1.f - z * z, d2 = -2.f * z * d1;
    f[0] = z;
    f[1] = d1 * g[1];
    f[2] = d1 * g[2];
    f[3] = d1 * g[3];
    f[4] = d2 * g[2] * g[2] + d1 * g[4];
    f[5] = d2 * g[3] * g[3] + d1 * g[5];
  };
  if constexpr (MODE == 4) {
    float u[QC_KMAX][NCH];
#pragma unroll
    for (int k = 0; k < QC_KMAX; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) u[k][c] = 0.f;
    for (int m = m0; m < m1; ++m) {
      float g[NCH], z, f[NCH];
      hidden(m, g, z, f);
#pragma unroll
      for (int k = 0; k < QC_KMAX; ++k)
        if (k < K) {
          const float w4 = w4k[k * H1 + m];
#pragma unroll
          for (int c = 0; c < NCH; ++c) u[k][c] = fmaf(w4, f[c], u[k][c]);
        }
    }
#pragma unroll
    for (int k = 0; k < QC_KMAX; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) s_buf[wave][k * NCH + c][lane] = u[k][c];
    __syncthreads();
    if (live) {
      for (int f = wave; f < K * NCH; f += QC_MS) {
        float v = (s_buf[0][f][lane] + s_buf[1][f][lane]) + (s_buf[2][f][lane] + s_buf[3][f][lane]);
        if (f % NCH == 0) v += w4k[(f / NCH) * H1 + L.H];
        out_u[(int64_t)f * B + p] = v;
      }
    }
  } else {
    float ub[QC_KMAX][NCH];
#pragma unroll
    for (int k = 0; k < QC_KMAX; ++k)
#pragma unroll
      for (int c = 0; c < NCH; ++c) ub[k][c] = (live && k < K) ? ubar[((int64_t)k * NCH + c) * B + pc] : 0.f;
    float* row = part + (row0 + tile) * part_stride;
    float* rowk = partk + (row0 + tile) * partk_stride;
    if (wave == 0) {   // d / d b4[k] = sum of the points' cotangents of u_k
#pragma unroll
      for (int k = 0; k < QC_KMAX; ++k)
        if (k < K) {
          const float sb = qc_wave_sum_to_lane63(ub[k][0]);
          if (lane == 63) rowk[k * H1 + L.H] = sb;
        }
    }
    float qb[NCH][N];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) qb[c][i] = 0.f;
    for (int m = m0; m < m1; ++m) {
      float g[NCH], z, f[NCH];
      hidden(m, g, z, f);
      float ubw[NCH], gk[QC_KMAX];
#pragma unroll
      for (int c = 0; c < NCH; ++c) ubw[c] = 0.f;
#pragma unroll
      for (int k = 0; k < QC_KMAX; ++k) {
        gk[k] = 0.f;
        if (k < K) {
          const float w4 = w4k[k * H1 + m];
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            ubw[c] = fmaf(w4, ub[k][c], ubw[c]);
            gk[k] = fmaf(ub[k][c], f[c], gk[k]);
          }
        }
      }
      float gb[NCH], gw4;
      post_cotangents<N, NCH>(gb, gw4, g, ubw, z, 1.f);
      float wg[N + 1];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float w3 = W3[m * N + i];
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          qb[c][i] = fmaf(w3, gb[c], qb[c][i]);
          sum = fmaf(gb[c], q[c][i], sum);
        }
        wg[i] = sum;
      }
      wg[N] = gb[0];
      qc_wave_sum_multi_to_lane63<N + 1>(wg);
#pragma unroll
      for (int k = 0; k < QC_KMAX; ++k)
        if (k < K) gk[k] = qc_wave_sum_to_lane63(gk[k]);
      if (lane == 63) {
#pragma unroll
        for (int i = 0; i < N; ++i) row[L.oW3 + m * N + i] = wg[i];
        row[L.ob3 + m] = wg[N];
#pragma unroll
        for (int k = 0; k < QC_KMAX; ++k)
          if (k < K) rowk[k * H1 + m] = gk[k];
      }
    }
    if (threadIdx.x == 0) {   // the single-output slots of the flat layout are not parameters here
      for (int m = 0; m < L.H; ++m) row[L.oW4 + m] = 0.f;
      row[L.ob4] = 0.f;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int i = 0; i < N; ++i) s_buf[wave][c * N + i][lane] = qb[c][i];
    __syncthreads();
    if (live) {
      for (int f = wave; f < NCH * N; f += QC_MS)
        qbar[(int64_t)f * B + p] = (s_buf[0][f][lane] + s_buf[1][f][lane]) + (s_buf[2][f][lane] + s_buf[3][f][lane]);
    }
  }
}

// ================================================================== residual + value tiles in ONE launch
// The fused step's two pipelines (65 536 residual points with 6 channels, 2 x 21 845 boundary / initial points with
// the value channel) are independent until the row reduction.  Launching each stage once over the blocks of BOTH
// (the lighter value tiles first, so the launch ends on full-occupancy residual tiles; block-uniform branch) removes the side stream, its two
// cross-queue event waits (~7 us of idle queue each) and 6 of the step's 15 launches.
template <int N>
__global__ void __launch_bounds__(256) k_pre_fwd_both(float* __restrict__ Xr, float* __restrict__ Xv,
                                                      const float* __restrict__ prm, QcLayout L, float* __restrict__ ajr,
                                                      float* __restrict__ ajv, int64_t Br, int64_t Bv, int n_val, QcDraw draw) {
  if ((int)blockIdx.x >= n_val) k_pre_fwd_body<N, 6>(blockIdx.x - n_val, Xr, prm, L, ajr, Br, &draw, Xr);
  else k_pre_fwd_value4<N>(blockIdx.x, Xv, prm, L, ajv, Bv, &draw, Xv);
}

template <int N>
__global__ void k_pre_bwd_both(const float* __restrict__ Xr, const float* __restrict__ Xv, const float* __restrict__ prm,
                               QcLayout L, const float* __restrict__ abr, const float* __restrict__ abv,
                               float* __restrict__ part, int64_t part_stride, int64_t row0_r, int64_t row0_v, int64_t Br,
                               int64_t Bv, int HB, int PS, int n_val) {
  if ((int)blockIdx.x >= n_val) k_pre_bwd_body<N, 6>(blockIdx.x - n_val, Xr, prm, L, abr, part, part_stride, row0_r, Br, HB, PS);
  else k_pre_bwd_body<N, 1>(blockIdx.x, Xv, prm, L, abv, part, part_stride, row0_v, Bv, HB, PS);
}

struct QcPostSeg {   // one pipeline's arguments of the fused (mode 2) post kernels
  const float* X;
  const float* qjets;
  float* ub;       // per-point cotangent of u (scratch, B floats)
  float* rb;       // per-point cotangent of the residual (scratch, B floats; residual pipeline only)
  float* qbar;
  int64_t row0, B;
};

template <int N>
__global__ void __launch_bounds__(256) k_post_both(const float* __restrict__ prm, QcLayout L, QcPde pde, QcPostSeg r,
                                                   QcPostSeg v, float* __restrict__ part, int64_t part_stride,
                                                   int n_val) {
  if ((int)blockIdx.x >= n_val)
    k_post_body<N, 6, 2>(blockIdx.x - n_val, r.X, prm, L, pde, r.qjets, r.ub, r.rb, nullptr, nullptr, r.qbar, part, part_stride,
                         r.row0, r.B);
  else
    k_post_value4<N>(blockIdx.x, v.X, prm, L, pde, v.qjets, v.ub, v.qbar, part, part_stride, v.row0, v.B);
}

template <int N>
__global__ void __launch_bounds__(256) k_post_fused_both(const float* __restrict__ prm, QcLayout L, QcPde pde, QcPostSeg r,
                                                         QcPostSeg v, float* __restrict__ part, int64_t part_stride,
                                                         int n_val) {
  extern __shared__ float s_dyn[];
  if ((int)blockIdx.x >= n_val)
    k_post_fused6_body<N>(blockIdx.x - n_val, r.X, prm, L, pde, r.qjets, r.ub, r.rb, r.qbar, part, part_stride, r.row0, r.B, s_dyn);
  else
    k_post_fused_body<N, 1, 1>(blockIdx.x, v.X, prm, L, pde, v.qjets, v.ub, nullptr, v.qbar, part, part_stride, v.row0, v.B, s_dyn);
}

template <int N>
__global__ void k_post_wg_both(const float* __restrict__ prm, QcLayout L, QcPde pde, QcPostSeg r, QcPostSeg v,
                               float* __restrict__ part, int64_t part_stride, int HB, int PS, int n_val) {
  if ((int)blockIdx.x >= n_val)
    k_post_wg_body<N, 6>(blockIdx.x - n_val, prm, L, pde, r.qjets, r.ub, r.rb, part, part_stride, r.row0, r.B, HB, PS);
  else
    k_post_wg_body<N, 1>(blockIdx.x, prm, L, pde, v.qjets, v.ub, nullptr, part, part_stride, v.row0, v.B, HB, PS);
}

}  // namespace

// ------------------------------------------------------------------ launchers
#define QC_MLP_DISPATCH(n, CALL)                                                                   \
  switch (n) {                                                                                     \
    case 1: { CALL(1) } break;  case 2: { CALL(2) } break;  case 3: { CALL(3) } break;             \
    case 4: { CALL(4) } break;  case 5: { CALL(5) } break;  case 6: { CALL(6) } break;             \
    case 7: { CALL(7) } break;  case 8: { CALL(8) } break;  case 9: { CALL(9) } break;             \
    case 10: { CALL(10) } break; case 11: { CALL(11) } break; case 12: { CALL(12) } break;         \
    case 13: { CALL(13) } break; case 14: { CALL(14) } break; case 15: { CALL(15) } break;         \
    case 16: { CALL(16) } break;                                                                   \
    default: return QC_ERR_UNSUPPORTED;                                                            \
  }

// lane = hidden unit kernels: HB lanes per group (one per hidden unit), PS groups share the tile's 64
// points.  Narrow hidden layers pack groups back to back (HB = H: 5 groups of 50 fill 250 of 256 lanes
// instead of 4 x 64 with 14 idle lanes each); the block is rounded up to whole waves.  Target block size 256
// (QC_MLP_THREADS): the 1 708 tiles of BASELINE config 2 are then resident in one round of 8 blocks per CU (512: 1.7
// rounds of 4; measured 15.3 against 16.4 us for the stage; 128 / 192 / 320 / 384 / 768 / 1024: 17.3 .. 24.4).
static inline void hidden_geometry(int H, int* HB, int* PS, int* threads) {
  static const int target = [] { const char* e = getenv("QC_MLP_THREADS"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 1024 ? v : 256; }();
  if (H <= target) {
    *HB = H;
    int ps = target / H;
    *PS = ps > 64 ? 64 : ps;
  } else {
    *HB = 64 * qc_ceil_div(H, 64);
    int ps = 1024 / *HB;
    *PS = ps >= 4 ? 4 : (ps >= 2 ? 2 : 1);
  }
  *threads = 64 * qc_ceil_div(*HB * *PS, 64);
}

static inline bool post_fused_ok(const QcLayout& L) {
  static const bool split = [] { const char* e = getenv("QC_POST_SPLIT"); return e && e[0] == '1'; }();
  return !split && L.H <= QC_POST_FUSED_MAXH;
}

int qc_mlp_pre_fwd(const float* X, const float* prm, QcLayout L, float* ajets, int64_t B, int nch,
                   hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
#define CALL(NN)                                                                                   \
  if (nch == 6) hipLaunchKernelGGL((k_pre_fwd<NN, 6>), dim3(grid), dim3(256), 0, st, X, prm, L, ajets, B); \
  else hipLaunchKernelGGL((k_pre_fwd<NN, 1>), dim3(grid), dim3(256), 0, st, X, prm, L, ajets, B);
  QC_MLP_DISPATCH(L.n, CALL)
#undef CALL
  return QC_OK;
}

int qc_mlp_pre_bwd(const float* X, const float* prm, QcLayout L, const float* abar, float* part,
                   int64_t part_stride, int64_t row0, int64_t B, int nch, hipStream_t st) {
  const int grid = qc_ceil_div(B, 64);
  if (L.H > 1024 || L.n > 64) return QC_ERR_UNSUPPORTED;
  int HB, PS, threads;
  hidden_geometry(L.H, &HB, &PS, &threads);
  const size_t sh = (size_t)PS * (4 + L.n) * HB * sizeof(float);
#define CALL(NN)                                                                                               \
  if (nch == 6) hipLaunchKernelGGL((k_pre_bwd<NN, 6>), dim3(grid), dim3(threads), sh, st, X, prm, L, abar, part, \
                                   part_stride, row0, B, HB, PS);                                              \
  else hipLaunchKernelGGL((k_pre_bwd<NN, 1>), dim3(grid), dim3(threads), sh, st, X, prm, L, abar, part,         \
                          part_stride, row0, B, HB, PS);
  QC_MLP_DISPATCH(L.n, CALL)
#undef CALL
  return QC_OK;
}

int qc_mlp_post(int mode, const float* X, const float* prm, QcLayout L, QcPde pde, const float* qjets,
                float* out_u, float* out_res, const float* in_ubar, const float* in_rbar, float* qbar,
                float* part, int64_t part_stride, int64_t row0, int64_t B, int nch, hipStream_t st) {
  const int tiles = qc_ceil_div(B, 64);
  if (L.H > 1024) return QC_ERR_UNSUPPORTED;
  int HB, PS, threads;
  hidden_geometry(L.H, &HB, &PS, &threads);
  const size_t sh = (size_t)PS * (L.n + 2) * HB * sizeof(float);
  // cotangent sources of the weight-gradient kernel: given (mode 1) or produced by the point kernel (mode 2)
  const float* ub_src = (mode == 1 || mode == 3) ? in_ubar : out_u;
  const float* rb_src = mode == 1 ? in_rbar : out_res;
  const int gen = mode == 3 ? 1 : 0;
  const bool fused = post_fused_ok(L);
#define LAUNCH(NN, CC, MM)                                                                              \
  hipLaunchKernelGGL((k_post<NN, CC, MM>), dim3(tiles), dim3(256), 0, st, X, prm, L, pde, qjets, out_u,  \
                     out_res, in_ubar, in_rbar, qbar, part, part_stride, row0, B)
#define LAUNCH_WG(NN, CC)                                                                               \
  hipLaunchKernelGGL((k_post_wg<NN, CC>), dim3(tiles), dim3(threads), sh, st, prm, L, pde, qjets,        \
                     ub_src, (CC == 6 ? rb_src : nullptr), part, part_stride, row0, B, HB, PS, gen)
  /* the step's form (mode 2): one kernel, lane = point in both phases (k_post_fused_body); QC_POST_SPLIT=1 keeps the pair */ \
#define LAUNCH_FUSED(NN, CC)                                                                            \
  hipLaunchKernelGGL((k_post_fused<NN, CC>), dim3(CC == 6 ? tiles : qc_ceil_div(tiles, 4)), dim3(256),   \
                     CC == 6 ? (size_t)L.H * 64 * sizeof(float) : 0, st, X, prm, L, pde, qjets, out_u,   \
                     out_res, qbar, part, part_stride, row0, B)
#define CALL(NN)                                                         \
  if (nch == 6) {                                                        \
    if (mode == 0) LAUNCH(NN, 6, 0);                                     \
    else if (mode == 1) { LAUNCH(NN, 6, 1); LAUNCH_WG(NN, 6); }          \
    else if (mode == 3) { LAUNCH(NN, 6, 3); LAUNCH_WG(NN, 6); }          \
    else if (mode == 4) LAUNCH(NN, 6, 4);                                \
    else if (fused) LAUNCH_FUSED(NN, 6);                                 \
    else { LAUNCH(NN, 6, 2); LAUNCH_WG(NN, 6); }                         \
  } else {                                                               \
    if (mode == 0) LAUNCH(NN, 1, 0);                                     \
    else if (mode == 1) { LAUNCH(NN, 1, 1); LAUNCH_WG(NN, 1); }          \
    else if (fused) LAUNCH_FUSED(NN, 1);                                 \
    else { LAUNCH(NN, 1, 2); LAUNCH_WG(NN, 1); }                         \
  }
  QC_MLP_DISPATCH(L.n, CALL)
#undef CALL
#undef LAUNCH
#undef LAUNCH_WG
#undef LAUNCH_FUSED
  return QC_OK;
}

// ------------------------------------------------------------------ merged residual + value launches (fused step)
// draw_*: when `draw` != 0 the launch first draws its own points (qc_sample_collocation_faces semantics) into Xr / Xv
int qc_mlp_pre_fwd_both(float* Xr, float* Xv, const float* prm, QcLayout L, float* ajr, float* ajv, int64_t Br, int64_t Bv,
                        int draw, int64_t n_ic, int64_t off_res, int64_t off_ic, int64_t off_bc, int64_t face_pts,
                        uint64_t seed, uint64_t step, hipStream_t st) {
  const int nr = qc_ceil_div(Br, 64), nv = qc_ceil_div(qc_ceil_div(Bv, 64), 4);   // value tiles: 4 per block, one per wave
  QcDraw dr;
  dr.enabled = draw;
  dr.n_ic = n_ic; dr.off_res = off_res; dr.off_ic = off_ic; dr.off_bc = off_bc; dr.face_pts = face_pts;
  dr.seed = seed; dr.step = step;
#define CALL(NN) \
  hipLaunchKernelGGL((k_pre_fwd_both<NN>), dim3(nr + nv), dim3(256), 0, st, Xr, Xv, prm, L, ajr, ajv, Br, Bv, nv, dr);
  QC_MLP_DISPATCH(L.n, CALL)
#undef CALL
  return QC_OK;
}

int qc_mlp_pre_bwd_both(const float* Xr, const float* Xv, const float* prm, QcLayout L, const float* abr, const float* abv,
                        float* part, int64_t part_stride, int64_t row0_r, int64_t row0_v, int64_t Br, int64_t Bv,
                        hipStream_t st) {
  if (L.H > 1024 || L.n > 64) return QC_ERR_UNSUPPORTED;
  const int nr = qc_ceil_div(Br, 64), nv = qc_ceil_div(Bv, 64);
  int HB, PS, threads;
  hidden_geometry(L.H, &HB, &PS, &threads);
  const size_t sh = (size_t)PS * (4 + L.n) * HB * sizeof(float);
#define CALL(NN)                                                                                                   \
  hipLaunchKernelGGL((k_pre_bwd_both<NN>), dim3(nr + nv), dim3(threads), sh, st, Xr, Xv, prm, L, abr, abv, part,    \
                     part_stride, row0_r, row0_v, Br, Bv, HB, PS, nv);
  QC_MLP_DISPATCH(L.n, CALL)
#undef CALL
  return QC_OK;
}

// mode-2 post stage of both pipelines: point kernel, then weight-gradient kernel
int qc_mlp_post_both(const float* prm, QcLayout L, QcPde pde, const float* Xr, const float* qjr, float* ubr, float* rbr,
                     float* qbr, int64_t row0_r, int64_t Br, const float* Xv, const float* qjv, float* ubv, float* qbv,
                     int64_t row0_v, int64_t Bv, float* part, int64_t part_stride, hipStream_t st) {
  if (L.H > 1024) return QC_ERR_UNSUPPORTED;
  const int nr = qc_ceil_div(Br, 64), nv = qc_ceil_div(Bv, 64);
  int HB, PS, threads;
  hidden_geometry(L.H, &HB, &PS, &threads);
  const size_t sh = (size_t)PS * (L.n + 2) * HB * sizeof(float);
  const int nv4 = qc_ceil_div(nv, 4);   // point kernel: 4 value tiles per block, one per wave
  const QcPostSeg r = {Xr, qjr, ubr, rbr, qbr, row0_r, Br}, v = {Xv, qjv, ubv, nullptr, qbv, row0_v, Bv};
  const bool fused = post_fused_ok(L);
#define CALL(NN)                                                                                                        \
  if (fused) {                                                                                                          \
    hipLaunchKernelGGL((k_post_fused_both<NN>), dim3(nr + nv4), dim3(256), (size_t)L.H * 64 * sizeof(float), st, prm, L, \
                       pde, r, v, part, part_stride, nv4);                                                              \
  } else {                                                                                                              \
    hipLaunchKernelGGL((k_post_both<NN>), dim3(nr + nv4), dim3(256), 0, st, prm, L, pde, r, v, part, part_stride, nv4); \
    hipLaunchKernelGGL((k_post_wg_both<NN>), dim3(nr + nv), dim3(threads), sh, st, prm, L, pde, r, v, part, part_stride, \
                       HB, PS, nv);                                                                                     \
  }
  QC_MLP_DISPATCH(L.n, CALL)
#undef CALL
  return QC_OK;
}

// K-output post stage (k_post_multi): mode 4 forward / mode 3 reverse, six channels
int qc_mlp_post_multi(int mode, const float* prm, QcLayout L, int K, const float* w4k, const float* qjets, float* out_u,
                      const float* ubar, float* qbar, float* part, int64_t part_stride, float* partk, int64_t partk_stride,
                      int64_t row0, int64_t B, hipStream_t st) {
  if (K < 1 || K > QC_KMAX || L.H > 1024) return QC_ERR_UNSUPPORTED;
  const int tiles = qc_ceil_div(B, 64);
#define CALL(NN)                                                                                                     \
  if (mode == 4) hipLaunchKernelGGL((k_post_multi<NN, 4>), dim3(tiles), dim3(256), 0, st, prm, L, K, w4k, qjets, out_u, \
                                    ubar, qbar, part, part_stride, partk, partk_stride, row0, B);                      \
  else hipLaunchKernelGGL((k_post_multi<NN, 3>), dim3(tiles), dim3(256), 0, st, prm, L, K, w4k, qjets, out_u, ubar,     \
                          qbar, part, part_stride, partk, partk_stride, row0, B);
  QC_MLP_DISPATCH(L.n, CALL)
#undef CALL
  return QC_OK;
}
