// Shared definitions for the gfx950 QCPINN kernels (device + host side of the C-ABI library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- error codes returned by every extern "C" entry point (0 = ok, negative = failure)
#define QC_OK 0
#define QC_ERR_ARG (-1)          // null pointer / bad size / unsupported combination
#define QC_ERR_UNSUPPORTED (-2)  // shape outside what the kernels are built for
#define QC_ERR_HIP (-3)          // a HIP runtime call failed (see qc_last_hip_error)
#define QC_ERR_ALLOC (-4)

// ---- gate opcodes: must match circuits.py
enum QcOp : int { QC_RX = 0, QC_RY = 1, QC_RZ = 2, QC_H = 3, QC_CNOT = 4, QC_CRX = 5, QC_CRZ = 6, QC_U4 = 7 };

// One gate of the device-resident program.  `ba`/`bb` are BIT positions of the amplitude
// index (bit = n-1-wire: wire 0 is the most significant bit), not wires.
struct QcGate {
  int op;
  int ba;    // target bit (1q), control bit (controlled), high bit of the 4x4 index (U4)
  int bb;    // target bit (controlled), low bit of the 4x4 index (U4), -1 otherwise
  int slot;  // flat parameter index, U4 slot, or -1
};

// Per-gate trig table entry, rebuilt on device every time the parameters change.
struct QcTrig {
  float c;   // cos(theta/2)
  float s;   // sin(theta/2)
  float th;  // theta itself (used when a leading RX layer is folded into the embedding angles)
  float pad;
};

#define QC_MAX_DIAG_RUNS 8

struct qc_program {
  int n_qubits;
  int n_gates;
  int n_params;
  int n_u4;
  int static_id;    // index into the compile-time specialised programs, or -1
  QcGate* d_gates;  // device
  QcGate* h_gates;  // host copy
  void* hbm_plan;   // QcHbmPlan* for n >= 9 (round-1 staged execution: amplitude encoding, cross-check hook), else null
  void* h2;         // QcH2* for n >= 9: the round-structured plan of qc_hbm2_plan.h (angle encoding), else null
  int amplitude;    // 1: amplitude encoding (initial state given directly), 0: RX angle embedding
  int lead_rx;      // 1: gates 0..n-1 are RX on wires 0..n-1 (cascade, cross_mesh): RX(p_w) RX(a_w) = RX(a_w + p_w)
  int n_diag_runs;  // n <= 5: runs of >= 2 consecutive diagonal gates (RZ / CRZ); each owns a 2^n phase table behind
                    // the per-gate entries of the trig buffer (see qc_fill_diag_tables)
  int diag_g0[QC_MAX_DIAG_RUNS], diag_g1[QC_MAX_DIAG_RUNS];   // gate ranges [g0, g1) of those runs
};

// Channel numbering of the derivative ("jet") channels carried through the network:
// 0 value, 1 d/dt, 2 d/dx, 3 d/dy, 4 d2/dx2, 5 d2/dy2.
#define QC_NCH 6

#define QC_WAVE 64

static inline int qc_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------ wave-level reductions
// ---- fused diagonal runs (register family, n <= 5)
// A run of >= 2 consecutive RZ / CRZ gates is one element-wise multiply by D[k] = prod_g phase_g(k), phase =
// c -+ i s by the target bit (CRZ: only where the control bit is set).  The tables live behind the n_gates per-gate
// entries of the trig buffer ({c, s} = {Re D, Im D}), run r at [n_gates + r * 2^n, ...), and are rebuilt with it.
__host__ __device__ inline bool qc_is_diag_op(int op) { return op == 2 /* QC_RZ */ || op == 6 /* QC_CRZ */; }

struct QcDiagRuns {   // kernel-argument copy of qc_program's run list
  int n;
  int g0[QC_MAX_DIAG_RUNS], g1[QC_MAX_DIAG_RUNS];
};

// fills pg->n_diag_runs / diag_g0 / diag_g1 (host); more than QC_MAX_DIAG_RUNS runs: none is fused
inline void qc_find_diag_runs(qc_program* pg) {
  pg->n_diag_runs = 0;
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
  if (n_qubits > 5 || tid >= (1 << n_qubits)) return;
  const int k = tid;
  for (int r = 0; r < runs.n; ++r) {
    float dr = 1.f, di = 0.f;
    for (int h = runs.g0[r]; h < runs.g1[r]; ++h) {
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

// Sum over the wave, result broadcast to every lane (uniform value).
__device__ __forceinline__ float qc_wave_sum(float v) {
  v = qc_wave_sum_to_lane63(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
