// extern "C" entry points of libqcpinn_hip.so (declared in include/qcpinn_hip.h).
// Argument checking happens here, once, on the host: the kernels assume validated shapes.
#include "qc_internal.h"
#include "qc_wave_sched.h"
#include "../../include/qcpinn_hip.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>

static_assert(sizeof(QcPde) == sizeof(qc_pde), "qc_pde layout");
static_assert(QC_PB_CONVECTION_DIFFUSION == QC_PROBLEM_CONVECTION_DIFFUSION && QC_PB_PURE_DIFFUSION == QC_PROBLEM_PURE_DIFFUSION,
              "problem ids");
static_assert(sizeof(QcOptHyper) == sizeof(qc_opt_hyper), "qc_opt_hyper layout");

static thread_local int g_last_hip = 0;

static inline int hip_fail(hipError_t e) {
  g_last_hip = (int)e;
  return QC_ERR_HIP;
}
static inline int after_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? QC_OK : hip_fail(e);
}

static QcLayout make_layout(int H, int n, int n_theta) { return qc_layout(H, n, n_theta); }
static QcPde to_pde(const qc_pde* p) {
  QcPde q;
  memcpy(&q, p, sizeof(q));
  return q;
}

// Which kernel family serves n qubits: registers (one lane per statevector) up to 5,
// lanes-as-amplitudes above.
static inline bool force_wave() {
  static const bool f = [] { const char* e = getenv("QC_FORCE_WAVE"); return e && e[0] == '1'; }();
  return f;   // test hook: route n <= 5 through the wave family too (cross-checks the two families)
}
static inline bool use_reg(int n) { return n >= 2 && n <= 5 && !force_wave(); }
static inline bool use_wave(int n) { return n >= 1 && n <= 8; }
static inline bool use_hbm(int n) { return n >= 9 && n <= 20; }
// n >= 9: the round-structured plan (qc_circuit_hbm2.hip; compile-time stage programs where one is registered).  The
// round-1 kernels (one LDS round trip or one pass per gate) are gone since round 3; the plan interpreter
// (QC_NO_STATIC=1) is the cross-check of the generated programs.
static inline bool use_h2(const qc_program* p) { return p->h2 != nullptr; }
// Budget of the resident per-tile stores: QC_HBM_KEEP_GB (default 96), never more than 85 % of the memory that is free
// on the current device when a workspace is sized (a smaller or partly occupied GPU gets fewer resident tiles, not an
// allocation failure).
static inline double hbm_budget_bytes() {
  static const double cap_gb = [] { const char* e = getenv("QC_HBM_KEEP_GB"); return e ? atof(e) : 96.0; }();
  double b = cap_gb * 1073741824.0;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
    const double avail = 0.85 * (double)free_b;
    if (avail < b) b = avail;
  }
  return b;
}

// A side stream per device so the (small) boundary/initial-value pipeline of a step can overlap the
// residual pipeline; created once, on first use, never inside a graph capture of the caller.
struct QcSide {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false, tried = false;
};
static QcSide* side_stream() {
  static QcSide tab[64];
  static std::mutex mu;
  static const bool off = [] { const char* e = getenv("QC_NO_OVERLAP"); return e && e[0] == '1'; }();
  if (off) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  QcSide& q = tab[dev];
  if (!q.tried) {
    q.tried = true;
    q.ok = hipStreamCreateWithFlags(&q.s, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&q.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&q.join, hipEventDisableTiming) == hipSuccess;
  }
  return q.ok ? &q : nullptr;
}

extern "C" {

int qc_version(void) { return QC_ABI_VERSION; }

const char* qc_error_string(int code) {
  switch (code) {
    case QC_OK: return "ok";
    case QC_ERR_ARG: return "invalid argument";
    case QC_ERR_UNSUPPORTED: return "unsupported shape (qubit count / hidden width outside the built kernels)";
    case QC_ERR_HIP: return "HIP runtime error (see qc_last_hip_error)";
    case QC_ERR_ALLOC: return "allocation failed";
    default: return "unknown error";
  }
}

int qc_last_hip_error(void) { return g_last_hip; }

// gate rows (opcode, wire_a, wire_b, slot) -> bit-indexed gates; returns false on an invalid row
static bool parse_rows(const int32_t* rows, int n_gates, int n_qubits, int n_params, QcGate* h, int* n_u4_out) {
  int n_u4 = 0;
  for (int g = 0; g < n_gates; ++g) {
    const int op = rows[4 * g], a = rows[4 * g + 1], b = rows[4 * g + 2], slot = rows[4 * g + 3];
    bool ok = op >= QC_RX && op <= QC_U4 && a >= 0 && a < n_qubits;
    const bool two = (op == QC_CNOT || op == QC_CRX || op == QC_CRZ || op == QC_U4);
    if (two) ok = ok && b >= 0 && b < n_qubits && b != a;
    const bool par = (op == QC_RX || op == QC_RY || op == QC_RZ || op == QC_CRX || op == QC_CRZ);
    if (par) ok = ok && slot >= 0 && slot < n_params;
    if (op == QC_U4) {  // the kernels hard-wire slot 0 = wires [0,1], slot 1 = wires [2,3]
      ok = ok && n_qubits >= 4 && ((slot == 0 && a == 0 && b == 1) || (slot == 1 && a == 2 && b == 3));
      ++n_u4;
    }
    if (!ok) return false;
    h[g].op = op;
    h[g].ba = n_qubits - 1 - a;
    h[g].bb = two ? n_qubits - 1 - b : -1;
    h[g].slot = (par || op == QC_U4) ? slot : -1;
  }
  *n_u4_out = n_u4;
  return true;
}
// RX(p_w) right after the embedding RX(a_w), wire by wire, distinct slots
static int detect_lead_rx(const QcGate* h, int n_gates, int n_qubits) {
  if (n_gates < n_qubits) return 0;
  for (int g = 0; g < n_qubits; ++g)
    if (h[g].op != QC_RX || h[g].ba != n_qubits - 1 - g || h[g].slot < 0) return 0;
  for (int g = 0; g < n_qubits; ++g)
    for (int k = 0; k < g; ++k)
      if (h[k].slot == h[g].slot) return 0;
  return 1;
}
static bool absorb_enabled() {
  static const bool no_absorb = [] { const char* e = getenv("QC_NO_ABSORB"); return e && e[0] == '1'; }();
  return !no_absorb;
}

int qc_program_create(const int32_t* rows, int n_gates, int n_qubits, int n_params, qc_program** out) {
  if (!rows || !out || n_gates <= 0 || n_qubits < 1 || n_qubits > 24 || n_params < 0) return QC_ERR_ARG;
  QcGate* h = (QcGate*)malloc(sizeof(QcGate) * n_gates);
  if (!h) return QC_ERR_ALLOC;
  int n_u4 = 0;
  if (!parse_rows(rows, n_gates, n_qubits, n_params, h, &n_u4)) {
    free(h);
    return QC_ERR_ARG;
  }
  qc_program* p = (qc_program*)malloc(sizeof(qc_program));
  if (!p) {
    free(h);
    return QC_ERR_ALLOC;
  }
  p->n_qubits = n_qubits; p->n_gates = n_gates; p->n_params = n_params; p->n_u4 = n_u4;
  p->h_gates = h; p->d_gates = nullptr;
  p->static_id = (n_qubits >= 2 && n_qubits <= 5) ? qc_reg_match_static(p)
                 : ((n_qubits >= 6 && n_qubits <= 8) ? qc_wave_match_static(p) : -1);
  p->h2 = nullptr;
  p->amplitude = 0;
  qc_find_diag_runs(p);
  p->lead_rx = detect_lead_rx(h, n_gates, n_qubits);
  hipError_t e = hipMalloc((void**)&p->d_gates, sizeof(QcGate) * n_gates);
  if (e == hipSuccess) e = hipMemcpy(p->d_gates, h, sizeof(QcGate) * n_gates, hipMemcpyHostToDevice);
#ifndef QC_WAVE_NO_RUNS
  if (e == hipSuccess && n_qubits >= 6 && n_qubits <= 8 && p->static_id >= 0) {
    // the compile-time program of this gate list multiplies by one phase table per RZ run: the same schedule, evaluated
    // here, names the gates behind each table for the kernels that (re)build the trig buffer
    const QcWaveSched ws = qc_wave_schedule(h, n_gates, n_qubits);
    if (ws.n_runs > 0) {
      const int ne = ws.run_off[ws.n_runs];
      e = hipMalloc((void**)&p->d_diag_list, sizeof(int) * ne);
      if (e == hipSuccess) e = hipMemcpy(p->d_diag_list, ws.entry, sizeof(int) * ne, hipMemcpyHostToDevice);
      p->n_diag_runs = ws.n_runs;
      for (int r = 0; r < ws.n_runs; ++r) {
        p->diag_g0[r] = ws.run_off[r];
        p->diag_g1[r] = ws.run_off[r + 1];
      }
    }
  }
#endif
  if (e != hipSuccess) {
    if (p->d_gates) (void)hipFree(p->d_gates);
    if (p->d_diag_list) (void)hipFree(p->d_diag_list);
    free(h);
    free(p);
    return hip_fail(e);
  }
  if (n_qubits >= 9 && n_qubits <= 20) {
    p->h2 = qc_h2_create(p, (p->lead_rx && absorb_enabled()) ? 1 : 0);
    if (!p->h2) {
      (void)hipFree(p->d_gates);
      if (p->d_diag_list) (void)hipFree(p->d_diag_list);
      free(h);
      free(p);
      return QC_ERR_ALLOC;
    }
  }
  *out = p;
  return QC_OK;
}

int qc_program_destroy(qc_program* p) {
  if (!p) return QC_ERR_ARG;
  if (p->h2) qc_h2_destroy(p->h2);
  if (p->d_gates) (void)hipFree(p->d_gates);
  if (p->d_diag_list) (void)hipFree(p->d_diag_list);
  free(p->h_gates);
  free(p);
  return QC_OK;
}

int qc_program_set_encoding(qc_program* p, int amplitude) {
  if (!p || (amplitude != 0 && amplitude != 1)) return QC_ERR_ARG;
  if (amplitude && ((int64_t)1 << p->n_qubits) < p->n_qubits) return QC_ERR_ARG;
  if (p->amplitude != amplitude && p->h2) {   // the staged plan folds the leading RX layer only for angle encoding, and
    p->amplitude = amplitude;                 // the generated programs embed angles
    void* h2 = qc_h2_create(p, (p->lead_rx && !amplitude && absorb_enabled()) ? 1 : 0);
    if (!h2) return QC_ERR_ALLOC;
    qc_h2_destroy(p->h2);
    p->h2 = h2;
  }
  p->amplitude = amplitude;
  return QC_OK;
}

int qc_amp_forward(const float* ajets, float* ujets, int n, int64_t B, int nch, void* stream) {
  if (!ajets || !ujets || n < 1 || n > 24 || B <= 0 || (nch != 1 && nch != 6)) return QC_ERR_ARG;
  qc_amp_fwd_launch(ajets, ujets, n, B, nch, (hipStream_t)stream);
  return after_launch();
}

int qc_amp_backward(const float* ajets, const float* ubar, float* abar, int n, int64_t B, int nch, void* stream) {
  if (!ajets || !ubar || !abar || n < 1 || n > 24 || B <= 0 || (nch != 1 && nch != 6)) return QC_ERR_ARG;
  qc_amp_bwd_launch(ajets, ubar, abar, n, B, nch, (hipStream_t)stream);
  return after_launch();
}

size_t qc_trig_bytes(const qc_program* p) {
  return p ? sizeof(QcTrig) * ((size_t)p->n_gates + (size_t)p->n_diag_runs * ((size_t)1 << p->n_qubits)) : 0;
}

int qc_prepare_gates(const qc_program* p, const float* theta, void* trig, void* stream) {
  if (!p || !trig || (p->n_params > 0 && !theta)) return QC_ERR_ARG;
  qc_opt_prep_trig(p, theta, (QcTrig*)trig, (hipStream_t)stream);
  return after_launch();
}

static int check_circuit(const qc_program* p, const void* trig, const float* umat, int64_t B) {
  if (!p || !trig || B <= 0) return QC_ERR_ARG;
  if (p->n_u4 > 0 && !umat) return QC_ERR_ARG;
  if (!use_reg(p->n_qubits) && !use_wave(p->n_qubits) && !use_hbm(p->n_qubits)) return QC_ERR_UNSUPPORTED;
  return QC_OK;
}

size_t qc_circuit_workspace_bytes(const qc_program* p, int nch, int backward) {
  if (!p || !use_hbm(p->n_qubits) || (nch != 1 && nch != 6)) return 0;
  return qc_h2_bytes(p, p->h2, nch, backward != 0, 1);
}

size_t qc_circuit_workspace_bytes_batch(const qc_program* p, int nch, int backward, int64_t B) {
  if (!p || B <= 0 || !use_hbm(p->n_qubits) || (nch != 1 && nch != 6)) return 0;
  const int64_t tiles = qc_ceil_div(B, 64);
  const size_t all = qc_h2_bytes(p, p->h2, nch, backward != 0, tiles);
  if ((double)all <= hbm_budget_bytes()) return all;
  int64_t fit = qc_h2_tiles_that_fit(p, p->h2, nch, backward != 0, (size_t)hbm_budget_bytes());
  return qc_h2_bytes(p, p->h2, nch, backward != 0, fit < 1 ? 1 : fit);
}

static size_t round256(size_t v) { return (v + 255) & ~(size_t)255; }
// h2, fused step: residual tiles (6 channels) and value tiles (1 channel) each keep their own resident slots
static size_t h2_res_bytes(const qc_program* p, int64_t B_res) {
  return B_res > 0 ? ((qc_h2_bytes(p, p->h2, 6, true, qc_ceil_div(B_res, 64)) + 255) & ~(size_t)255) : 0;
}
static size_t h2_val_bytes(const qc_program* p, int64_t B_val) {
  return B_val > 0 ? ((qc_h2_bytes(p, p->h2, 1, true, qc_ceil_div(B_val, 64)) + 255) & ~(size_t)255) : 0;
}
static size_t h2_min_bytes(const qc_program* p) { return (qc_h2_bytes(p, p->h2, 6, true, 1) + 255) & ~(size_t)255; }
static size_t step_circuit_bytes(const qc_program* p, int64_t B_res, int64_t B_val = 0) {
  if (use_hbm(p->n_qubits) && use_h2(p)) {
    const size_t all = h2_res_bytes(p, B_res) + h2_val_bytes(p, B_val);
    const double budget = hbm_budget_bytes();
    if ((double)all <= budget) return all > h2_min_bytes(p) ? all : h2_min_bytes(p);
    // not everything fits: as many six-channel tiles as the budget holds, shared by both pipelines (one launch
    // sequence per group of resident tiles, the adjoint pass recomputes its group's forward pass)
    const int64_t fit = qc_h2_tiles_that_fit(p, p->h2, 6, true, (size_t)budget);
    const size_t some = (qc_h2_bytes(p, p->h2, 6, true, fit < 1 ? 1 : fit) + 255) & ~(size_t)255;
    return some > h2_min_bytes(p) ? some : h2_min_bytes(p);
  }
  if (use_reg(p->n_qubits)) return qc_reg_chi_store_bytes(p, B_res);   // optional: enables the no-recompute adjoint
  if (use_wave(p->n_qubits)) {   // same, compile-time programs at n = 6..8: residual store, then the value pipeline's
    const size_t rb = round256(qc_wave_chi_store_bytes(p, B_res));
    return rb > 0 ? rb + qc_wave_val_store_bytes(p, B_val) : 0;
  }
  return 0;
}

int qc_hbm_plan_describe(const int32_t* rows, int n_gates, int n_qubits, int n_params, int32_t* out, int cap) {
  if (!rows || n_gates <= 0 || n_qubits < 9 || n_qubits > 20 || n_params < 0 || cap < 0) return 0;
  QcGate* h = (QcGate*)malloc(sizeof(QcGate) * n_gates);
  if (!h) return 0;
  int n_u4 = 0, len = 0;
  if (parse_rows(rows, n_gates, n_qubits, n_params, h, &n_u4))
    len = qc_h2_describe_gates(h, n_gates, n_qubits, (detect_lead_rx(h, n_gates, n_qubits) && absorb_enabled()) ? 1 : 0, out, cap);
  free(h);
  return len;
}

int qc_wave_sched_describe(const int32_t* rows, int n_gates, int n_qubits, int n_params, int32_t* out, int cap) {
  if (!rows || n_gates <= 0 || n_gates > QC_WS_MAX_ITEMS || n_qubits < 1 || n_qubits > 8 || n_params < 0 || cap < 0) return 0;
  QcGate* h = (QcGate*)malloc(sizeof(QcGate) * n_gates);
  if (!h) return 0;
  int n_u4 = 0, len = 0;
  if (parse_rows(rows, n_gates, n_qubits, n_params, h, &n_u4)) {
    const QcWaveSched ws = qc_wave_schedule(h, n_gates, n_qubits);
    auto put = [&](int v) {
      if (out && len < cap) out[len] = v;
      ++len;
    };
    put(ws.n_items);
    put(ws.n_runs);
    for (int i = 0; i < ws.n_items; ++i) put(ws.item[i]);
    for (int r = 0; r < ws.n_runs; ++r) {
      put(ws.run_off[r + 1] - ws.run_off[r]);
      for (int e = ws.run_off[r]; e < ws.run_off[r + 1]; ++e) put(ws.entry[e]);
    }
  }
  free(h);
  return len;
}

size_t qc_step_workspace_bytes(const qc_program* p, int64_t B_res, int64_t B_val) {
  if (!p || B_res < 0 || B_val < 0) return 0;
  size_t b = round256(step_circuit_bytes(p, B_res, B_val));
  if (p->amplitude)   // initial-amplitude jets and their cotangents, both pipelines
    b += 2 * round256(sizeof(float) * 6 * p->n_qubits * (size_t)B_res) + 2 * round256(sizeof(float) * p->n_qubits * (size_t)B_val);
  return b;
}

int qc_forward_expval(const qc_program* p, const void* trig, const float* umat, const float* angles,
                      float* expval, int64_t B, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_circuit(p, trig, umat, B);
  if (rc) return rc;
  if (!angles || !expval) return QC_ERR_ARG;
  if (use_hbm(p->n_qubits)) {
    rc = qc_h2_forward(p, p->h2, (const QcTrig*)trig, umat, angles, expval, B, 1, ws, ws_bytes, false, (hipStream_t)stream);
    return rc ? rc : after_launch();
  }
  rc = use_reg(p->n_qubits)
           ? qc_reg_value_fwd(p, (const QcTrig*)trig, umat, angles, expval, B, (hipStream_t)stream)
           : qc_wave_value_fwd(p, (const QcTrig*)trig, umat, angles, expval, B, nullptr, (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_backward_expval(const qc_program* p, const void* trig, const float* umat, const float* angles,
                       const float* cot, float* d_angles, float* part, int64_t part_stride, int64_t row0,
                       int64_t B, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_circuit(p, trig, umat, B);
  if (rc) return rc;
  if (!angles || !cot || !d_angles || !part || part_stride < p->n_params || row0 < 0) return QC_ERR_ARG;
  if (use_hbm(p->n_qubits)) {
    rc = qc_h2_backward(p, p->h2, (const QcTrig*)trig, umat, angles, cot, d_angles, part, part_stride, row0, B, 1, ws,
                                    ws_bytes, false, (hipStream_t)stream);
    return rc ? rc : after_launch();
  }
  rc = use_reg(p->n_qubits)
           ? qc_reg_value_bwd(p, (const QcTrig*)trig, umat, angles, cot, d_angles, part, part_stride, row0, B,
                              (hipStream_t)stream)
           : qc_wave_value_bwd(p, (const QcTrig*)trig, umat, angles, cot, d_angles, part, part_stride, row0, B, nullptr,
                               (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_forward_jets(const qc_program* p, const void* trig, const float* umat, const float* ajets, float* qjets,
                    int64_t B, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_circuit(p, trig, umat, B);
  if (rc) return rc;
  if (!ajets || !qjets) return QC_ERR_ARG;
  if (use_hbm(p->n_qubits)) {
    rc = qc_h2_forward(p, p->h2, (const QcTrig*)trig, umat, ajets, qjets, B, 6, ws, ws_bytes, false, (hipStream_t)stream);
    return rc ? rc : after_launch();
  }
  rc = use_reg(p->n_qubits)
           ? qc_reg_jets_fwd(p, (const QcTrig*)trig, umat, ajets, qjets, B, nullptr, (hipStream_t)stream)
           : qc_wave_jets_fwd(p, (const QcTrig*)trig, umat, ajets, qjets, B, nullptr, (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_backward_jets(const qc_program* p, const void* trig, const float* umat, const float* ajets,
                     const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B,
                     void* ws, size_t ws_bytes, void* stream) {
  int rc = check_circuit(p, trig, umat, B);
  if (rc) return rc;
  if (!ajets || !qbar || !abar || !part || part_stride < p->n_params || row0 < 0) return QC_ERR_ARG;
  if (use_hbm(p->n_qubits)) {
    rc = qc_h2_backward(p, p->h2, (const QcTrig*)trig, umat, ajets, qbar, abar, part, part_stride, row0, B, 6, ws,
                                    ws_bytes, false, (hipStream_t)stream);
    return rc ? rc : after_launch();
  }
  rc = use_reg(p->n_qubits)
           ? qc_reg_jets_bwd(p, (const QcTrig*)trig, umat, ajets, qbar, abar, part, part_stride, row0, B, nullptr,
                             (hipStream_t)stream)
           : qc_wave_jets_bwd(p, (const QcTrig*)trig, umat, ajets, qbar, abar, part, part_stride, row0, B, nullptr,
                              (hipStream_t)stream);
  return rc ? rc : after_launch();
}

// Register-family variants that hand the forward pass's final states to the adjoint pass through
// chi_dev [6][2*2^n][B] instead of recomputing them (12 instead of 18 circuit-equivalents per point).
int qc_forward_jets_keep(const qc_program* p, const void* trig, const float* umat, const float* ajets, float* qjets,
                         int64_t B, float* chi, void* stream) {
  int rc = check_circuit(p, trig, umat, B);
  if (rc) return rc;
  if (!ajets || !qjets || !chi) return QC_ERR_ARG;
  if (!use_reg(p->n_qubits)) return QC_ERR_UNSUPPORTED;
  rc = qc_reg_jets_fwd(p, (const QcTrig*)trig, umat, ajets, qjets, B, chi, (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_backward_jets_kept(const qc_program* p, const void* trig, const float* umat, const float* ajets,
                          const float* qbar, float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B,
                          const float* chi, void* stream) {
  int rc = check_circuit(p, trig, umat, B);
  if (rc) return rc;
  if (!ajets || !qbar || !abar || !part || !chi || part_stride < p->n_params || row0 < 0) return QC_ERR_ARG;
  if (!use_reg(p->n_qubits)) return QC_ERR_UNSUPPORTED;
  rc = qc_reg_jets_bwd(p, (const QcTrig*)trig, umat, ajets, qbar, abar, part, part_stride, row0, B, chi,
                       (hipStream_t)stream);
  return rc ? rc : after_launch();
}

static int check_mlp(int H, int n, int n_theta, int64_t B, int nch) {
  if (H < 1 || H > 1024 || n < 1 || n > 16 || n_theta < 0 || B <= 0 || (nch != 1 && nch != 6)) return QC_ERR_ARG;
  return QC_OK;
}

int qc_pre_forward(const float* X, const float* prm, int H, int n, int n_theta, float* ajets, int64_t B, int nch,
                   void* stream) {
  int rc = check_mlp(H, n, n_theta, B, nch);
  if (rc) return rc;
  if (!X || !prm || !ajets) return QC_ERR_ARG;
  rc = qc_mlp_pre_fwd(X, prm, make_layout(H, n, n_theta), ajets, B, nch, (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_pre_backward(const float* X, const float* prm, int H, int n, int n_theta, const float* abar, float* part,
                    int64_t part_stride, int64_t row0, int64_t B, int nch, void* stream) {
  int rc = check_mlp(H, n, n_theta, B, nch);
  if (rc) return rc;
  const QcLayout L = make_layout(H, n, n_theta);
  if (!X || !prm || !abar || !part || part_stride < L.NP || row0 < 0) return QC_ERR_ARG;
  rc = qc_mlp_pre_bwd(X, prm, L, abar, part, part_stride, row0, B, nch, (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_post(int mode, const float* X, const float* prm, int H, int n, int n_theta, const qc_pde* pde,
            const float* qjets, float* out_u, float* out_res, const float* in_ubar, const float* in_rbar,
            float* qbar, float* part, int64_t part_stride, int64_t row0, int64_t B, int nch, void* stream) {
  int rc = check_mlp(H, n, n_theta, B, nch);
  if (rc) return rc;
  const QcLayout L = make_layout(H, n, n_theta);
  if (mode < 0 || mode > 4 || !X || !prm || !pde || !qjets) return QC_ERR_ARG;
  if (mode >= 3 && nch != 6) return QC_ERR_ARG;          // general jets: six channels only
  if (mode >= 1 && mode <= 3 && (!qbar || !part || row0 < 0 || part_stride < L.NP + (mode == 2 ? 3 : 0))) return QC_ERR_ARG;
  if (mode == 2 && (!out_u || (nch == 6 && !out_res))) return QC_ERR_ARG;  // per-point cotangent scratch
  if ((mode == 3 && !in_ubar) || (mode == 4 && !out_u)) return QC_ERR_ARG;
  rc = qc_mlp_post(mode, X, prm, L, to_pde(pde), qjets, out_u, out_res, in_ubar, in_rbar, qbar, part, part_stride,
                   row0, B, nch, (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_post_multi(int mode, const float* prm, int H, int n, int n_theta, int K, const float* w4k, const float* qjets,
                  float* out_u, const float* in_ubar, float* qbar, float* part, int64_t part_stride, float* partk,
                  int64_t partk_stride, int64_t row0, int64_t B, void* stream) {
  int rc = check_mlp(H, n, n_theta, B, 6);
  if (rc) return rc;
  const QcLayout L = make_layout(H, n, n_theta);
  if ((mode != 3 && mode != 4) || K < 1 || K > 4 || !prm || !w4k || !qjets) return QC_ERR_ARG;
  if (mode == 4 && !out_u) return QC_ERR_ARG;
  if (mode == 3 && (!in_ubar || !qbar || !part || !partk || row0 < 0 || part_stride < L.NP || partk_stride < (int64_t)K * (H + 1)))
    return QC_ERR_ARG;
  rc = qc_mlp_post_multi(mode, prm, L, K, w4k, qjets, out_u, in_ubar, qbar, part, part_stride, partk, partk_stride, row0, B,
                         (hipStream_t)stream);
  return rc ? rc : after_launch();
}

int qc_reduce_rows(const float* part, int64_t rows, int64_t stride, int ncols, float* out, void* stream) {
  if (!part || !out || rows <= 0 || ncols <= 0 || stride < ncols) return QC_ERR_ARG;
  qc_opt_reduce_rows(part, rows, stride, ncols, out, (hipStream_t)stream);
  return after_launch();
}

int qc_adam_step(float* flat, int NP, float* prm, float* m, float* v, void* state, const qc_opt_hyper* hp,
                 float* hist, int hist_cap, const qc_program* prog, int theta_off, void* trig, void* stream) {
  if (!flat || NP <= 0 || !prm || !m || !v || !state || !hp) return QC_ERR_ARG;
  if (prog && (!trig || theta_off < 0 || theta_off + prog->n_params > NP)) return QC_ERR_ARG;
  QcOptHyper h;
  memcpy(&h, hp, sizeof(h));
  qc_opt_adam(flat, NP, prm, m, v, (QcOptState*)state, h, hist, hist_cap, prog, theta_off, (QcTrig*)trig,
              (hipStream_t)stream);
  return after_launch();
}

int qc_sample_collocation_faces(float* X_res, int64_t n_res, int64_t off_res, float* X_val, int64_t n_ic, int64_t off_ic,
                                int64_t n_bc, int64_t off_bc, int64_t bc_face_points, uint64_t seed, uint64_t step,
                                void* stream) {
  if (n_res < 0 || n_ic < 0 || n_bc < 0 || off_res < 0 || off_ic < 0 || off_bc < 0 || bc_face_points < 0) return QC_ERR_ARG;
  if ((n_res > 0 && !X_res) || (n_ic + n_bc > 0 && !X_val)) return QC_ERR_ARG;
  qc_sample_launch(X_res, n_res, off_res, X_val, n_ic, off_ic, n_bc, off_bc, bc_face_points, seed, step, (hipStream_t)stream);
  return after_launch();
}

int qc_sample_collocation(float* X_res, int64_t n_res, int64_t off_res, float* X_val, int64_t n_ic, int64_t off_ic,
                          int64_t n_bc, int64_t off_bc, uint64_t seed, uint64_t step, void* stream) {
  return qc_sample_collocation_faces(X_res, n_res, off_res, X_val, n_ic, off_ic, n_bc, off_bc, 0, seed, step, stream);
}

// ---- merged residual + value stages of the fused step (register family, angle encoding)
static bool merged_ok(const qc_step_desc* d) {
  static const bool no_merge = [] { const char* e = getenv("QC_NO_MERGE"); return e && e[0] == '1'; }();
  return !no_merge && use_reg(d->n) && !d->prog->amplitude && d->B_res > 0 && d->B_val > 0 && d->circ_ws_dev &&
         d->circ_ws_bytes >= qc_reg_chi_store_bytes(d->prog, d->B_res) && d->X_res_dev && d->ajets_res_dev &&
         d->qjets_res_dev && d->qbar_res_dev && d->abar_res_dev && d->X_val_dev && d->ajets_val_dev && d->qjets_val_dev &&
         d->qbar_val_dev && d->abar_val_dev;
}

static int merged_stage(const qc_step_desc* d, int stage, hipStream_t st, bool draw = false) {
  const QcLayout L = make_layout(d->H, d->n, d->n_theta);
  const int64_t rows_res = qc_ceil_div(d->B_res, 64);
  const QcTrig* trig = (const QcTrig*)d->trig_dev;
  const float* prm = d->params_dev;
  float* chi_store = (float*)d->circ_ws_dev;
  QcPde pde;
  memcpy(&pde, &d->pde, sizeof(pde));
  switch (stage) {
    case QC_STAGE_PRE_FWD:
      return qc_mlp_pre_fwd_both((float*)d->X_res_dev, (float*)d->X_val_dev, prm, L, d->ajets_res_dev, d->ajets_val_dev,
                                 d->B_res, d->B_val, draw ? 1 : 0, d->n_ic, d->sample_off_res, d->sample_off_ic,
                                 d->sample_off_bc, d->sample_bc_face_points, d->sample_seed, d->sample_step, st);
    case QC_STAGE_CIRCUIT_FWD:
      return qc_reg_circ_fwd_both(d->prog, trig, d->umat_dev, d->ajets_res_dev, d->qjets_res_dev, d->B_res, chi_store,
                                  d->ajets_val_dev, d->qjets_val_dev, d->B_val, st);
    case QC_STAGE_POST:
      // abar_* are written only by the adjoint sweep: their heads serve as per-point cotangent scratch here
      return qc_mlp_post_both(prm, L, pde, (const float*)d->X_res_dev, d->qjets_res_dev, d->abar_res_dev,
                              d->abar_res_dev + d->B_res, d->qbar_res_dev, 0, d->B_res, (const float*)d->X_val_dev,
                              d->qjets_val_dev, d->abar_val_dev, d->qbar_val_dev, rows_res, d->B_val, d->part_dev,
                              d->part_stride, st);
    case QC_STAGE_CIRCUIT_BWD:
      return qc_reg_circ_bwd_both(d->prog, trig, d->umat_dev, d->ajets_res_dev, d->qbar_res_dev, d->abar_res_dev, 0, d->B_res,
                                  chi_store, d->ajets_val_dev, d->qbar_val_dev, d->abar_val_dev, rows_res, d->B_val,
                                  d->part_dev + L.oTh, d->part_stride, st);
    case QC_STAGE_PRE_BWD:
      return qc_mlp_pre_bwd_both((const float*)d->X_res_dev, (const float*)d->X_val_dev, prm, L, d->abar_res_dev,
                                 d->abar_val_dev, d->part_dev, d->part_stride, 0, rows_res, d->B_res, d->B_val, st);
    default: return QC_ERR_ARG;
  }
}

int qc_fused_step_stage(const qc_step_desc* d, int stage, void* stream) {
  if (!d || !d->prog || !d->trig_dev || !d->params_dev || !d->part_dev) return QC_ERR_ARG;
  if (d->prog->n_qubits != d->n || d->prog->n_params != d->n_theta) return QC_ERR_ARG;
  if (stage < 0 || stage >= QC_STAGE_COUNT) return QC_ERR_ARG;
  if (!merged_ok(d)) return QC_ERR_UNSUPPORTED;
  int rc = check_mlp(d->H, d->n, d->n_theta, d->B_res, 6);
  if (rc) return rc;
  if ((rc = merged_stage(d, stage, (hipStream_t)stream))) return rc;
  return after_launch();
}

int qc_fused_pinn_residual_step(const qc_step_desc* d, int phases, void* stream) {
  if (!d || !d->prog || !d->trig_dev || !d->params_dev || !d->part_dev || !d->flat_dev) return QC_ERR_ARG;
  const int n = d->n, H = d->H;
  if (d->prog->n_qubits != n || d->prog->n_params != d->n_theta) return QC_ERR_ARG;
  int rc = check_mlp(H, n, d->n_theta, d->B_res > 0 ? d->B_res : 1, 6);
  if (rc) return rc;
  const QcLayout L = make_layout(H, n, d->n_theta);
  const int64_t rows_res = d->B_res > 0 ? qc_ceil_div(d->B_res, 64) : 0;
  const int64_t rows_val = d->B_val > 0 ? qc_ceil_div(d->B_val, 64) : 0;
  const int64_t rows = rows_res + rows_val;
  if (rows <= 0 || rows > d->part_rows_cap || d->part_stride < L.NP + 3) return QC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const QcTrig* trig = (const QcTrig*)d->trig_dev;
  // amplitude encoding: the circuit kernels run on the initial-amplitude jets u(a) and return cotangents
  // w.r.t. them; both live behind the circuit scratch in the step workspace
  const bool amp = d->prog->amplitude != 0;
  float *u_res = nullptr, *ub_res = nullptr, *u_val = nullptr, *ub_val = nullptr;
  void* cws = d->circ_ws_dev;
  size_t cws_bytes = d->circ_ws_bytes;
  if (amp) {
    if (!d->circ_ws_dev || d->circ_ws_bytes < qc_step_workspace_bytes(d->prog, d->B_res, d->B_val)) return QC_ERR_ARG;
    char* base = (char*)d->circ_ws_dev + round256(step_circuit_bytes(d->prog, d->B_res));
    const size_t rb = round256(sizeof(float) * 6 * n * (size_t)d->B_res), vb = round256(sizeof(float) * n * (size_t)d->B_val);
    u_res = (float*)base; ub_res = (float*)(base + rb);
    u_val = (float*)(base + 2 * rb); ub_val = (float*)(base + 2 * rb + vb);
    cws_bytes = round256(step_circuit_bytes(d->prog, d->B_res));
    if (cws_bytes == 0) cws = nullptr;
  }

  // (in the merged form below the first stage draws the points itself)
  const bool draw_in_stage = (phases & QC_PHASE_SAMPLE) && (phases & QC_PHASE_GRADS) && merged_ok(d);
  if ((phases & QC_PHASE_SAMPLE) && !draw_in_stage) {
    if ((rc = qc_sample_collocation_faces((float*)d->X_res_dev, d->B_res, d->sample_off_res, (float*)d->X_val_dev, d->n_ic,
                                          d->sample_off_ic, d->B_val - d->n_ic, d->sample_off_bc,
                                          d->sample_bc_face_points, d->sample_seed, d->sample_step, st))) return rc;
  }
  // register family, angle encoding, both pipelines present, final-state store available: every stage is ONE launch
  // over the value tiles and the residual tiles together (no side stream, 9 launches per step); QC_NO_MERGE=1 keeps
  // the two-stream form below
  const bool merged = (phases & QC_PHASE_GRADS) && merged_ok(d);
  if (merged) {
    for (int stage = 0; stage < QC_STAGE_COUNT; ++stage)
      if ((rc = merged_stage(d, stage, st, draw_in_stage && stage == QC_STAGE_PRE_FWD))) return rc;
    if ((rc = after_launch())) return rc;
  }
  if ((phases & QC_PHASE_GRADS) && !merged) {
    // the two pipelines are independent until the row reduction: fork the value pipeline onto a side
    // stream (not for n >= 9, where both would share the HBM statevector workspace)
    QcSide* side = (d->B_res > 0 && d->B_val > 0 && !use_hbm(n)) ? side_stream() : nullptr;
    // n >= 9 (round-structured plan): residual and value tiles keep their own resident slots when the caller's
    // workspace holds them all; otherwise both pipelines share the workspace and the adjoint pass recomputes
    const bool h2 = use_hbm(n) && use_h2(d->prog);
    bool h2_resident = false;
    void *h2_res_ws = cws, *h2_val_ws = cws;
    size_t h2_res_b = cws_bytes, h2_val_b = cws_bytes;
    if (h2) {
      if (!cws || cws_bytes < h2_min_bytes(d->prog)) return QC_ERR_ARG;
      const size_t rb = h2_res_bytes(d->prog, d->B_res), vb = h2_val_bytes(d->prog, d->B_val);
      h2_resident = cws_bytes >= rb + vb;
      if (h2_resident) {
        h2_res_b = rb;
        h2_val_ws = (char*)cws + rb;
        h2_val_b = vb;
      }
    }
    hipStream_t sv = st;
    if (side) {
      if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess)
        side = nullptr;
      else
        sv = side->s;
    }
    // lanes-as-amplitudes family, compile-time program, angle encoding: the value pipeline's final states are kept as
    // well when the workspace holds both stores (behind the residual pipeline's)
    float* wave_val_store = nullptr;
    if (!use_reg(n) && use_wave(n) && !amp && cws && qc_wave_val_store_bytes(d->prog, d->B_val) > 0) {
      const size_t rb = round256(qc_wave_chi_store_bytes(d->prog, d->B_res));
      if (rb > 0 && cws_bytes >= rb + qc_wave_val_store_bytes(d->prog, d->B_val)) wave_val_store = (float*)((char*)cws + rb);
    }
    if (d->B_val > 0) {
      if (!d->X_val_dev || !d->ajets_val_dev || !d->qjets_val_dev || !d->qbar_val_dev || !d->abar_val_dev)
        return QC_ERR_ARG;
      if ((rc = qc_pre_forward(d->X_val_dev, d->params_dev, H, n, d->n_theta, d->ajets_val_dev, d->B_val, 1, sv))) return rc;
      if (amp && (rc = qc_amp_forward(d->ajets_val_dev, u_val, n, d->B_val, 1, sv))) return rc;
      const float* cin_val = amp ? u_val : d->ajets_val_dev;
      float* cout_val = amp ? ub_val : d->abar_val_dev;
      if (h2) {
        if ((rc = qc_h2_forward(d->prog, d->prog->h2, trig, d->umat_dev, cin_val, d->qjets_val_dev, d->B_val, 1, h2_val_ws, h2_val_b,
                                h2_resident, sv))) return rc;
      } else if (wave_val_store) {
        if ((rc = qc_wave_value_fwd(d->prog, trig, d->umat_dev, cin_val, d->qjets_val_dev, d->B_val, wave_val_store, sv))) return rc;
        if ((rc = after_launch())) return rc;
      } else if ((rc = qc_forward_expval(d->prog, trig, d->umat_dev, cin_val, d->qjets_val_dev, d->B_val, cws, cws_bytes, sv))) return rc;
      if ((rc = qc_post(2, d->X_val_dev, d->params_dev, H, n, d->n_theta, &d->pde, d->qjets_val_dev,
                        d->abar_val_dev, nullptr, nullptr, nullptr, d->qbar_val_dev, d->part_dev, d->part_stride,
                        rows_res, d->B_val, 1, sv))) return rc;
      if (h2) {
        if ((rc = qc_h2_backward(d->prog, d->prog->h2, trig, d->umat_dev, cin_val, d->qbar_val_dev, cout_val, d->part_dev + L.oTh,
                                 d->part_stride, rows_res, d->B_val, 1, h2_val_ws, h2_val_b, h2_resident, sv))) return rc;
      } else if (wave_val_store) {
        if ((rc = qc_wave_value_bwd(d->prog, trig, d->umat_dev, cin_val, d->qbar_val_dev, cout_val, d->part_dev + L.oTh,
                                    d->part_stride, rows_res, d->B_val, wave_val_store, sv))) return rc;
        if ((rc = after_launch())) return rc;
      } else if ((rc = qc_backward_expval(d->prog, trig, d->umat_dev, cin_val, d->qbar_val_dev, cout_val,
                                          d->part_dev + L.oTh, d->part_stride, rows_res, d->B_val, cws, cws_bytes, sv))) return rc;
      if (amp && (rc = qc_amp_backward(d->ajets_val_dev, ub_val, d->abar_val_dev, n, d->B_val, 1, sv))) return rc;
      if ((rc = qc_pre_backward(d->X_val_dev, d->params_dev, H, n, d->n_theta, d->abar_val_dev, d->part_dev,
                                d->part_stride, rows_res, d->B_val, 1, sv))) return rc;
    }
    if (d->B_res > 0) {
      if (!d->X_res_dev || !d->ajets_res_dev || !d->qjets_res_dev || !d->qbar_res_dev || !d->abar_res_dev)
        return QC_ERR_ARG;
      if ((rc = qc_pre_forward(d->X_res_dev, d->params_dev, H, n, d->n_theta, d->ajets_res_dev, d->B_res, 6, st))) return rc;
      // register family: keep the final states of the forward pass for the adjoint kernel of this step
      float* chi_store = (use_reg(n) && cws && cws_bytes >= qc_reg_chi_store_bytes(d->prog, d->B_res)) ? (float*)cws : nullptr;
      // lanes-as-amplitudes family, compile-time program: final states of the forward kernel kept for the adjoint kernel
      float* wave_store = nullptr;
      if (!use_reg(n) && use_wave(n) && cws && qc_wave_chi_store_bytes(d->prog, d->B_res) > 0 &&
          cws_bytes >= qc_wave_chi_store_bytes(d->prog, d->B_res))
        wave_store = (float*)cws;
      if (amp && (rc = qc_amp_forward(d->ajets_res_dev, u_res, n, d->B_res, 6, st))) return rc;
      const float* cin_res = amp ? u_res : d->ajets_res_dev;
      float* cout_res = amp ? ub_res : d->abar_res_dev;
      if (use_reg(n)) {
        if ((rc = qc_reg_jets_fwd(d->prog, trig, d->umat_dev, cin_res, d->qjets_res_dev, d->B_res, chi_store, st)))
          return rc;
        if ((rc = after_launch())) return rc;
      } else if (wave_store) {
        if ((rc = qc_wave_jets_fwd(d->prog, trig, d->umat_dev, cin_res, d->qjets_res_dev, d->B_res, wave_store, st))) return rc;
        if ((rc = after_launch())) return rc;
      } else if (h2) {
        if ((rc = qc_h2_forward(d->prog, d->prog->h2, trig, d->umat_dev, cin_res, d->qjets_res_dev, d->B_res, 6, h2_res_ws, h2_res_b,
                                h2_resident, st))) return rc;
        if ((rc = after_launch())) return rc;
      } else if ((rc = qc_forward_jets(d->prog, trig, d->umat_dev, cin_res, d->qjets_res_dev, d->B_res, cws, cws_bytes, st)))
        return rc;
      // abar_res is written only by the adjoint sweep below: its head serves as cotangent scratch here
      if ((rc = qc_post(2, d->X_res_dev, d->params_dev, H, n, d->n_theta, &d->pde, d->qjets_res_dev,
                        d->abar_res_dev, d->abar_res_dev + d->B_res, nullptr, nullptr, d->qbar_res_dev, d->part_dev,
                        d->part_stride, 0, d->B_res, 6, st))) return rc;
      if (use_reg(n)) {
        if ((rc = qc_reg_jets_bwd(d->prog, trig, d->umat_dev, cin_res, d->qbar_res_dev, cout_res,
                                  d->part_dev + L.oTh, d->part_stride, 0, d->B_res, chi_store, st))) return rc;
        if ((rc = after_launch())) return rc;
      } else if (wave_store) {
        if ((rc = qc_wave_jets_bwd(d->prog, trig, d->umat_dev, cin_res, d->qbar_res_dev, cout_res, d->part_dev + L.oTh,
                                   d->part_stride, 0, d->B_res, wave_store, st))) return rc;
        if ((rc = after_launch())) return rc;
      } else if (h2) {
        if ((rc = qc_h2_backward(d->prog, d->prog->h2, trig, d->umat_dev, cin_res, d->qbar_res_dev, cout_res, d->part_dev + L.oTh,
                                 d->part_stride, 0, d->B_res, 6, h2_res_ws, h2_res_b, h2_resident, st))) return rc;
        if ((rc = after_launch())) return rc;
      } else if ((rc = qc_backward_jets(d->prog, trig, d->umat_dev, cin_res, d->qbar_res_dev, cout_res,
                                        d->part_dev + L.oTh, d->part_stride, 0, d->B_res, cws, cws_bytes, st))) return rc;
      if (amp && (rc = qc_amp_backward(d->ajets_res_dev, ub_res, d->abar_res_dev, n, d->B_res, 6, st))) return rc;
      if ((rc = qc_pre_backward(d->X_res_dev, d->params_dev, H, n, d->n_theta, d->abar_res_dev, d->part_dev,
                                d->part_stride, 0, d->B_res, 6, st))) return rc;
    }
    if (side) {
      hipError_t e = hipEventRecord(side->join, sv);
      if (e == hipSuccess) e = hipStreamWaitEvent(st, side->join, 0);
      if (e != hipSuccess) return hip_fail(e);
    }
  }
  if (phases & QC_PHASE_GRADS) {
    // the step owns its partial-row matrix, so the reduction folds it in place (two levels, fixed order);
    // with the update phase in the same call the second level rides in the optimiser launch
    const int RS = qc_opt_fold_rows(d->part_dev, rows, d->part_stride, L.NP + 3, st);
    if ((phases & QC_PHASE_UPDATE) && !d->comm) {
      if (!d->m_dev || !d->v_dev || !d->opt_state_dev) return QC_ERR_ARG;
      QcOptHyper h;
      memcpy(&h, &d->hyper, sizeof(h));
      qc_opt_adam_fold(d->part_dev, d->part_stride, RS, d->flat_dev, L.NP, d->params_dev, d->m_dev, d->v_dev,
                       (QcOptState*)d->opt_state_dev, h, d->hist_dev, d->hist_cap, d->prog, L.oTh, (QcTrig*)d->trig_dev, st);
      return after_launch();
    }
    if ((rc = qc_reduce_rows(d->part_dev, RS, d->part_stride, L.NP + 3, d->flat_dev, st))) return rc;
    // data parallelism inside the library: sum the flat [gradient | 3 loss sums] vector over the ranks (RCCL, same stream)
    if (d->comm && (phases & QC_PHASE_UPDATE) && (rc = qc_comm_allreduce(d->flat_dev, L.NP + 3, d->comm, st))) return rc;
  }
  if (phases & QC_PHASE_UPDATE) {
    if (!d->m_dev || !d->v_dev || !d->opt_state_dev) return QC_ERR_ARG;
    if ((rc = qc_adam_step(d->flat_dev, L.NP, d->params_dev, d->m_dev, d->v_dev, d->opt_state_dev, &d->hyper,
                           d->hist_dev, d->hist_cap, d->prog, L.oTh, d->trig_dev, st))) return rc;
  }
  return QC_OK;
}

}  // extern "C"
