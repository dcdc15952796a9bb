// Execution plan of the HBM-resident circuit family (9 <= n <= 20 qubits): the gate program is cut,
// once, on the host, into STAGES so that a statevector crosses HBM a handful of times per circuit
// instead of once per gate:
//   * L stage: a run of gates whose bits all lie in a "local" set of T <= 12 index bits.  A block loads
//     one tile (2^T amplitudes: the local bits vary, the others are fixed) into LDS, applies the whole run
//     there, stores it back.  The local set always contains the low 7 bits, so tiles move as 1 KiB runs.
//   * D stage: a run of diagonal gates (RZ, CRZ).  All of them commute, so the run is ONE table of 2^n
//     unit phases (built per call from the trig table), applied as an element-wise multiply fused into
//     the store of the preceding L stage.  Their parameter gradients come from one real vector
//     W[k] = sum_states Im(conj(lam_k) chi_k) at the output of the run and per-gate signed sums of W.
#pragma once
#include "qc_common.h"

constexpr int QC_HBM_T = 12;    // local bits per tile (32 KiB of complex64 per statevector tile)
constexpr int QC_HBM_L0 = 7;    // low bits that are always local (1 KiB contiguous runs)

struct QcStageGate {   // gate in tile-local bit numbering
  int op;
  int jt;    // local index of the target bit (U4: high bit of the 4x4 index)
  int jc;    // local index of the control bit (U4: low bit), -1 otherwise
  int gi;    // index into the trig table (= position in the program)
  int slot;  // parameter slot / U4 slot / -1
};

struct QcStage {
  int kind;             // 0 = L (LDS tile), 1 = D (diagonal table)
  int g0, ng;           // range in the stage-gate array (L) or in the diagonal-gate array (D)
  int nloc;             // L: number of local bits (<= QC_HBM_T)
  int lb[QC_HBM_T];     // L: global bit position of local bit j (ascending)
  int ngb;              // L: number of non-local bits
  int gb[24];           // L: their positions (ascending)
  int post_diag;        // L: index of the D stage fused into this stage's store, or -1
  int table;            // D: index of its phase table
  int fused;            // D: 1 if applied by the preceding L stage
};

struct QcDiagGate {     // diagonal gate in GLOBAL bit numbering
  int op;               // QC_RZ or QC_CRZ
  int bt, bc;           // target / control bit (-1)
  int gi, slot;
};

struct QcHbmPlan {
  int n_stages, n_tables;
  int n_lgates, n_dgates;
  QcStage* stages;        // host
  QcStageGate* h_lgates;  // host
  QcStageGate* d_lgates;  // device
  QcDiagGate* h_dgates;
  QcDiagGate* d_dgates;
  int max_param_lgates;   // largest number of parametric gates in one L stage
  int absorb;             // 1: gates 0..n-1 (leading RX layer) are folded into the embedding angles and not staged
};

QcHbmPlan* qc_hbm_plan_create(const qc_program* pg);
void qc_hbm_plan_destroy(QcHbmPlan* plan);
