// Diagonal-run schedule of a gate program for the compile-time programs of the lanes-as-amplitudes family (n = 6..8).
//
// RZ and CRZ gates commute with every gate that does not act on their wires and with everything diagonal on them (RZ,
// CRZ, the control of CNOT / CRX).  The schedule moves each of them as far as that allows and collects the ones that meet into runs;
// a run is ONE element-wise multiply by a 2^n phase table (built with the trig table, qc_fill_diag_tables) instead
// of one 2x2 update per gate, and its gradients come from one pass over t[k] = Im(conj(lam_k) chi_k).  The reference's
// layered ansatz (nn/DVQuantumLayer.py:184-212: RZ RX per wire | CNOT ring | RX RZ per wire, per layer) turns into
// [run | RX x n | ring | RX x n | run of 2n | RX x n | ring | RX x n | run]: 3 table multiplies for 4n RZ gates; the
// cross-mesh ansatz (:348-371: RX, RZ per wire | CRZ between every ordered pair | RX, RZ per wire) into
// [RX x n | run of n + n(n-1) | RX x n | run of n].
//
// One constexpr function serves both sides: the generated kernels evaluate it at compile time on their gate list,
// the host evaluates it at program creation on the same list to know which gates feed which table.  Plain C++.
#pragma once
#include "qc_types.h"

#define QC_WS_MAX_ITEMS 256

struct QcWaveSched {
  int n_items;
  int item[QC_WS_MAX_ITEMS];        // >= 0: gate index (its own trig entry); < 0: -(run + 1)
  int n_runs;
  int run_off[QC_MAX_DIAG_RUNS + 1];
  int entry[QC_WS_MAX_ITEMS];       // gate indices of the runs' RZ / CRZ gates, run r at [run_off[r], run_off[r + 1])
  bool ok;                          // false: program too long for the fixed arrays (no run is formed)
};

// does gate g act non-diagonally on bit b?
template <class G>
constexpr bool qc_ws_blocks(const G& g, int b) {
  switch (g.op) {
    case QC_RX: case QC_RY: case QC_H: return g.ba == b;
    case QC_CNOT: case QC_CRX: return g.bb == b;          // the control (ba) sees a diagonal action
    case QC_U4: return g.ba == b || g.bb == b;
    default: return false;                                // RZ, CRZ
  }
}

// every gate a gate, in program order
constexpr QcWaveSched qc_wave_plain_schedule(int n_gates) {
  QcWaveSched s{};
  s.ok = n_gates <= QC_WS_MAX_ITEMS;
  for (int g = 0; g < n_gates && g < QC_WS_MAX_ITEMS; ++g) s.item[g] = g;
  s.n_items = n_gates < QC_WS_MAX_ITEMS ? n_gates : QC_WS_MAX_ITEMS;
  return s;
}

template <class G>
constexpr QcWaveSched qc_wave_schedule(const G* gates, int n_gates, int n_qubits) {
  QcWaveSched s{};
  s.ok = n_gates <= QC_WS_MAX_ITEMS;
  // pass 1: a run id (or -1) per gate, and the position (item order) of each run
  // order[]: emitted sequence of gate indices / run markers; runs as -(id + 1), ids in order of creation
  int order[QC_WS_MAX_ITEMS + QC_MAX_DIAG_RUNS * 2 + 8] = {};
  int n_order = 0;
  int run_of[QC_WS_MAX_ITEMS] = {};
  int run_size[QC_WS_MAX_ITEMS] = {};
  int n_ids = 0;
  int last_run = -1;              // the most recently emitted run; `blocked` = bits acted on non-diagonally since
  unsigned blocked = 0;
  int fl[QC_WS_MAX_ITEMS] = {};   // floating diagonal gates (not yet emitted), any number per bit, in program order
  int n_fl = 0;
  unsigned fl_bits = 0;
  if (!s.ok) return qc_wave_plain_schedule(n_gates);
  for (int g = 0; g < n_gates; ++g) run_of[g] = -1;
  auto flush = [&]() {
    if (n_fl == 0) return;
    const int id = n_ids++;
    for (int i = 0; i < n_fl; ++i) run_of[fl[i]] = id;
    run_size[id] = n_fl;
    order[n_order++] = -(id + 1);
    last_run = id;
    blocked = 0;
    n_fl = 0;
    fl_bits = 0;
  };
  for (int g = 0; g < n_gates; ++g) {
    const G& gt = gates[g];
    if ((gt.op == QC_RZ || gt.op == QC_CRZ) && gt.slot >= 0) {
      const unsigned m = (1u << gt.ba) | (gt.op == QC_CRZ ? (1u << gt.bb) : 0u);
      if (last_run >= 0 && !(blocked & m)) {
        run_of[g] = last_run;                              // moves back to the emitted run
        ++run_size[last_run];
      } else {
        fl[n_fl++] = g;                                    // floats forward
        fl_bits |= m;
      }
      continue;
    }
    bool conflict = false;
    for (int b = 0; b < n_qubits; ++b)
      if (((fl_bits >> b) & 1u) && qc_ws_blocks(gt, b)) conflict = true;
    if (conflict) flush();
    order[n_order++] = g;
    for (int b = 0; b < n_qubits; ++b)
      if (qc_ws_blocks(gt, b)) blocked |= 1u << b;
  }
  flush();
  // pass 2: runs of one gate stay plain gates; at most QC_MAX_DIAG_RUNS tables
  int new_id[QC_WS_MAX_ITEMS] = {};
  int kept = 0;
  for (int id = 0; id < n_ids; ++id) new_id[id] = (run_size[id] >= 2 && kept < QC_MAX_DIAG_RUNS) ? kept++ : -1;
  s.n_runs = kept;
  int e = 0;
  for (int id = 0; id < n_ids; ++id) {
    if (new_id[id] < 0) continue;
    s.run_off[new_id[id]] = e;
    for (int g = 0; g < n_gates; ++g)
      if (run_of[g] == id) s.entry[e++] = g;
  }
  for (int r = kept; r <= QC_MAX_DIAG_RUNS; ++r) s.run_off[r] = e;
  for (int i = 0; i < n_order; ++i) {
    if (order[i] >= 0) {
      s.item[s.n_items++] = order[i];
    } else {
      const int id = -order[i] - 1;
      if (new_id[id] >= 0) {
        s.item[s.n_items++] = -(new_id[id] + 1);
      } else {
        for (int g = 0; g < n_gates; ++g)
          if (run_of[g] == id) s.item[s.n_items++] = g;
      }
    }
  }
  return s;
}
