// Plain-data definitions shared by the device code, the host side of the C-ABI library and the build-time plan tool
// (tools that never see HIP headers include this file alone).
#pragma once
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
  void* h2;         // QcH2* for n >= 9: the round-structured plan of qc_hbm2_plan.h, else null
  int amplitude;    // 1: amplitude encoding (initial state given directly), 0: RX angle embedding
  int lead_rx;      // 1: gates 0..n-1 are RX on wires 0..n-1 (cascade, cross_mesh): RX(p_w) RX(a_w) = RX(a_w + p_w)
  int n_diag_runs;  // n <= 5: runs of >= 2 consecutive diagonal gates (RZ / CRZ); n = 6..8 with a compile-time program:
                    // the RZ runs of qc_wave_sched.h.  Each owns a 2^n phase table behind the per-gate entries of the
                    // trig buffer (see qc_fill_diag_tables)
  int diag_g0[QC_MAX_DIAG_RUNS], diag_g1[QC_MAX_DIAG_RUNS];   // gate ranges [g0, g1) of those runs, or (d_diag_list set)
  int* d_diag_list;                                           // ranges of this device list of gate indices
};
