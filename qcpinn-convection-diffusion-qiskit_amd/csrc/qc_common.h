// Shared definitions for the gfx950 QCPINN kernels (device + host side of the C-ABI library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qc_types.h"

// Channel numbering of the derivative ("jet") channels carried through the network:
// 0 value, 1 d/dt, 2 d/dx, 3 d/dy, 4 d2/dx2, 5 d2/dy2.
#define QC_NCH 6

#define QC_WAVE 64

static inline int qc_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ wave-level reductions
// ---- fused diagonal runs (register family, n <= 5; compile-time programs of the wave family: qc_wave_sched.h)
// A run of >= 2 consecutive RZ / CRZ gates is one element-wise multiply by D[k] = prod_g phase_g(k), phase =
// c -+ i s by the target bit (CRZ: only where the control bit is set).  The tables live behind the n_gates per-gate
// entries of the trig buffer ({c, s} = {Re D, Im D}), run r at [n_gates + r * 2^n, ...), and are rebuilt with it.
__host__ __device__ inline bool qc_is_diag_op(int op) { return op == 2 /* QC_RZ */ || op == 6 /* QC_CRZ */; }

struct QcDiagRuns {   // kernel-argument copy of qc_program's run list
  int n;
  int g0[QC_MAX_DIAG_RUNS], g1[QC_MAX_DIAG_RUNS];
  const int* list;    // null: run r = gates g0[r] .. g1[r]-1; else gates list[g0[r]] .. list[g1[r]-1] (qc_wave_sched.h)
};

// fills pg->n_diag_runs / diag_g0 / diag_g1 (host); more than QC_MAX_DIAG_RUNS runs: none is fused
inline void qc_find_diag_runs(qc_program* pg) {
  pg->n_diag_runs = 0;
  pg->d_diag_list = nullptr;
  if (pg->n_qubits > 5) return;
  for (int g = 0; g < pg->n_gates;) {
    if (!qc_is_diag_op(pg->h_gates[g].op)) { ++g; continue; }
    int e = g;
    while (e < pg->n_gates && qc_is_diag_op(pg->h_gates[e].op)) ++e;
    if (e - g >= 2) {
      if (pg->n_diag_runs == QC_MAX_DIAG_RUNS) { pg->n_diag_runs = 0; return; }
      pg->diag_g0[pg->n_diag_runs] = g;
      pg->diag_g1[pg->n_diag_runs] = e;
      ++pg->n_diag_runs;
    }
    g = e;
  }
}
inline QcDiagRuns qc_diag_runs_of(const qc_program* pg) {
  QcDiagRuns r;
  r.n = pg ? pg->n_diag_runs : 0;
  r.list = pg ? pg->d_diag_list : nullptr;
  for (int i = 0; i < QC_MAX_DIAG_RUNS; ++i) {
    r.g0[i] = (pg && i < r.n) ? pg->diag_g0[i] : 0;
    r.g1[i] = (pg && i < r.n) ? pg->diag_g1[i] : 0;
  }
  return r;
}

// one thread per amplitude k (tid < 2^n) after the per-gate entries of `trig` are complete and block-visible.
// (`cs`: optional LDS copy [n_gates][2] of the (cos, sin) pairs; else read back from `trig`)
__device__ inline void qc_fill_diag_tables(const QcGate* __restrict__ prog, int n_gates, int n_qubits,
                                           QcTrig* __restrict__ trig, int tid, const QcDiagRuns& runs,
                                           const float* cs = nullptr) {
  if (runs.n == 0 || n_qubits > 8 || tid >= (1 << n_qubits)) return;
  const int k = tid;
  for (int r = 0; r < runs.n; ++r) {
    float dr = 1.f, di = 0.f;
    for (int hh = runs.g0[r]; hh < runs.g1[r]; ++hh) {
      const int h = runs.list ? runs.list[hh] : hh;
      const QcGate gt = prog[h];
      const bool ctl = gt.op == 6;
      const int tb = ctl ? gt.bb : gt.ba;
      const float c = cs ? cs[2 * h] : trig[h].c, s0 = cs ? cs[2 * h + 1] : trig[h].s;
      if (ctl && !((k >> gt.ba) & 1)) continue;
      const float s = ((k >> tb) & 1) ? s0 : -s0;
      const float nr = dr * c - di * s, ni = dr * s + di * c;
      dr = nr;
      di = ni;
    }
    QcTrig t = {dr, di, 0.f, 0.f};
    trig[n_gates + r * (1 << n_qubits) + k] = t;
  }
}

// Sum over the 64 lanes of a wave with DPP row operations (no LDS traffic).  The total is
// valid in lane 63 (and only there).
__device__ __forceinline__ float qc_wave_sum_to_lane63(float v) {
  // quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror, row_bcast15, row_bcast31
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
  return v;
}

// The same sum for NV independent values at once (valid in lane 63 of each): the four in-row steps interleaved over
// the values, the two row_bcast steps as ONE v_add_f32_dpp each with a row mask (the compiler expands the builtin form
// of those two steps into v_mov 0 / v_mov_dpp / v_add: ten instructions per value instead of six).
template <int NV>
__device__ __forceinline__ void qc_wave_sum_multi_to_lane63(float (&v)[NV]) {
#define QC_DPP_STEP_(CTRL)                                                                                       \
  _Pragma("unroll") for (int k = 0; k < NV; ++k)                                                                 \
    v[k] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), CTRL, 0xF, 0xF, true));
  QC_DPP_STEP_(0xB1) QC_DPP_STEP_(0x4E) QC_DPP_STEP_(0x141) QC_DPP_STEP_(0x140)
#undef QC_DPP_STEP_
  // every value's last in-row step is complete before the hand-written DPP reads (the assembler does not pad the
  // VALU-write -> DPP-read hazard inside asm: two wait states, provided by the s_nop and by the other values' steps)
#pragma unroll
  for (int k = 0; k < NV; ++k) asm volatile("" : "+v"(v[k]));
  asm volatile("s_nop 1");
#pragma unroll
  for (int k = 0; k < NV; ++k) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[k]));
  if constexpr (NV < 3) asm volatile("s_nop 1");
#pragma unroll
  for (int k = 0; k < NV; ++k) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v[k]));
}

// Sum over the wave, result broadcast to every lane (uniform value).
__device__ __forceinline__ float qc_wave_sum(float v) {
  v = qc_wave_sum_to_lane63(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
