// Internal structs and launcher prototypes shared by the translation units of libqcpinn_hip.so.
#pragma once
#include "qc_common.h"

struct QcLayout {  // column offsets of the flat parameter / gradient vector
  int H, n;
  int oW1, ob1, oW2, ob2, oW3, ob3, oW4, ob4, oTh, NP;
};

inline QcLayout qc_layout(int H, int n, int n_theta) {
  QcLayout L;
  L.H = H;
  L.n = n;
  L.oW1 = 0;
  L.ob1 = 3 * H;
  L.oW2 = L.ob1 + H;
  L.ob2 = L.oW2 + n * H;
  L.oW3 = L.ob2 + n;
  L.ob3 = L.oW3 + H * n;
  L.oW4 = L.ob3 + H;
  L.ob4 = L.oW4 + H;
  L.oTh = L.ob4 + 1;
  L.NP = L.oTh + n_theta;
  return L;
}

constexpr int QC_PB_CONVECTION_DIFFUSION = 0, QC_PB_PURE_DIFFUSION = 1;   // == QC_PROBLEM_* of the public header

struct QcPde {  // == qc_pde of the public header
  float D, vx, vy;                 // physical constants: analytic targets of mode 2
  float c_t, c_x, c_y, d_xx, d_yy; // operator coefficients (sigma scalings folded in)
  float w_res;
  float inv_n_res;
  float w_val_a, w_val_b;
  float inv_n_a, inv_n_b;
  int problem;
  int64_t n_seg_a;
};

// device-resident optimiser state (64-byte record; host reads it back on demand)
struct QcOptState {
  float lr;
  float best;
  int num_bad;
  int step;  // number of Adam steps taken
  float last_loss;
  float last_norm;
  float loss_parts[3];
  int hist_base;  // steps taken before this history buffer started: the loss of step s goes to hist[s - 1 - hist_base]
  int pad[6];
};

struct QcOptHyper {  // == qc_opt_hyper of the public header
  double beta1, beta2;  // kept in double: torch forms the bias corrections in Python floats
  float eps, max_norm;
  float sched_factor, sched_threshold, sched_min_lr, sched_eps;
  int sched_patience;
  float w_res, w_bc, w_ic;  // loss = w_res*L_r + w_bc*L_bc + w_ic*L_ic   (2, 4, 2)
};

// ---- launchers, one group per .hip file
int qc_reg_match_static(const qc_program* pg);
int qc_reg_value_fwd(const qc_program*, const QcTrig*, const float* umat, const float* angles, float* expval,
                     int64_t B, hipStream_t);
int qc_reg_value_bwd(const qc_program*, const QcTrig*, const float* umat, const float* angles, const float* cot,
                     float* d_angles, float* part, int64_t part_stride, int64_t row0, int64_t B, hipStream_t);
int qc_reg_jets_fwd(const qc_program*, const QcTrig*, const float* umat, const float* ajets, float* qjets,
                    int64_t B, float* chi_store, hipStream_t);
int qc_reg_jets_bwd(const qc_program*, const QcTrig*, const float* umat, const float* ajets, const float* qbar,
                    float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B, const float* chi_store,
                    hipStream_t);
int qc_reg_circ_fwd_both(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                         int64_t Br, float* chi_store, const float* angles, float* expval, int64_t Bv, hipStream_t st);
int qc_reg_circ_bwd_both(const qc_program* pg, const QcTrig* trig, const float* umat, const float* ajets,
                         const float* qbar, float* abar, int64_t row0_r, int64_t Br, const float* chi_store,
                         const float* angles, const float* cot, float* d_angles, int64_t row0_v, int64_t Bv, float* part,
                         int64_t part_stride, hipStream_t st);
int qc_mlp_pre_fwd_both(float* Xr, float* Xv, const float* prm, QcLayout L, float* ajr, float* ajv, int64_t Br, int64_t Bv,
                        int draw, int64_t n_ic, int64_t off_res, int64_t off_ic, int64_t off_bc, int64_t face_pts,
                        uint64_t seed, uint64_t step, hipStream_t st);
int qc_mlp_pre_bwd_both(const float* Xr, const float* Xv, const float* prm, QcLayout L, const float* abr, const float* abv,
                        float* part, int64_t part_stride, int64_t row0_r, int64_t row0_v, int64_t Br, int64_t Bv,
                        hipStream_t st);
int qc_mlp_post_both(const float* prm, QcLayout L, QcPde pde, const float* Xr, const float* qjr, float* ubr, float* rbr,
                     float* qbr, int64_t row0_r, int64_t Br, const float* Xv, const float* qjv, float* ubv, float* qbv,
                     int64_t row0_v, int64_t Bv, float* part, int64_t part_stride, hipStream_t st);
size_t qc_reg_chi_store_bytes(const qc_program* pg, int64_t B);
int qc_wave_match_static(const qc_program*);
int qc_wave_value_fwd(const qc_program*, const QcTrig*, const float* umat, const float* angles, float* expval,
                      int64_t B, float* value_store, hipStream_t);
int qc_wave_value_bwd(const qc_program*, const QcTrig*, const float* umat, const float* angles, const float* cot,
                      float* d_angles, float* part, int64_t part_stride, int64_t row0, int64_t B, const float* value_store,
                      hipStream_t);
size_t qc_wave_val_store_bytes(const qc_program* pg, int64_t B);
int qc_wave_jets_fwd(const qc_program*, const QcTrig*, const float* umat, const float* ajets, float* qjets,
                     int64_t B, float* chi_store, hipStream_t);
int qc_wave_jets_bwd(const qc_program*, const QcTrig*, const float* umat, const float* ajets, const float* qbar,
                     float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B, const float* chi_store,
                     hipStream_t);
size_t qc_wave_chi_store_bytes(const qc_program* pg, int64_t B);
int qc_mlp_pre_fwd(const float* X, const float* prm, QcLayout L, float* ajets, int64_t B, int nch, hipStream_t);
int qc_mlp_pre_bwd(const float* X, const float* prm, QcLayout L, const float* abar, float* part,
                   int64_t part_stride, int64_t row0, int64_t B, int nch, hipStream_t);
int qc_mlp_post(int mode, const float* X, const float* prm, QcLayout L, QcPde pde, const float* qjets,
                float* out_u, float* out_res, const float* in_ubar, const float* in_rbar, float* qbar,
                float* part, int64_t part_stride, int64_t row0, int64_t B, int nch, hipStream_t);
int qc_mlp_post_multi(int mode, const float* prm, QcLayout L, int K, const float* w4k, const float* qjets, float* out_u,
                      const float* ubar, float* qbar, float* part, int64_t part_stride, float* partk, int64_t partk_stride,
                      int64_t row0, int64_t B, hipStream_t);
int qc_opt_reduce_rows(const float* part, int64_t rows, int64_t stride, int ncols, float* out, hipStream_t);
int qc_opt_fold_rows(float* part, int64_t rows, int64_t stride, int ncols, hipStream_t);
int qc_opt_adam_fold(const float* part, int64_t stride, int RS, float* flat, int NP, float* prm, float* m, float* v,
                     QcOptState* state, QcOptHyper hp, float* hist, int hist_cap, const qc_program* pg, int theta_off,
                     QcTrig* trig, hipStream_t);
int qc_opt_adam(float* flat, int NP, float* prm, float* m, float* v, QcOptState* state, QcOptHyper hp,
                float* hist, int hist_cap, const qc_program* pg, int theta_off, QcTrig* trig, hipStream_t);
int qc_opt_prep_trig(const qc_program* pg, const float* theta, QcTrig* trig, hipStream_t);
int qc_sample_launch(float* X_res, int64_t n_res, int64_t off_res, float* X_val, int64_t n_ic, int64_t off_ic,
                     int64_t n_bc, int64_t off_bc, int64_t bc_face_points, uint64_t seed, uint64_t step, hipStream_t);
// HBM family, n >= 9 (qc_circuit_hbm2.hip, qc_circuit_h2s_kernels.h): all tiles of a batch resident when the workspace allows
void* qc_h2_create(const qc_program* pg, int absorb);
void qc_h2_destroy(void* h2);
int qc_h2_describe_gates(const QcGate* gates, int n_gates, int n_qubits, int absorb, int32_t* out, int cap);   // host only
size_t qc_h2_bytes(const qc_program* pg, void* h2, int nch, bool backward, int64_t tiles);
int64_t qc_h2_tiles_that_fit(const qc_program* pg, void* h2, int nch, bool backward, size_t ws_bytes);
int qc_h2_forward(const qc_program* pg, void* h2, const QcTrig* trig, const float* umat, const float* ajets, float* qjets,
                  int64_t B, int nch, void* ws, size_t ws_bytes, bool keep, hipStream_t st);
int qc_h2_backward(const qc_program* pg, void* h2, const QcTrig* trig, const float* umat, const float* ajets, const float* qbar,
                   float* abar, float* part, int64_t part_stride, int64_t row0, int64_t B, int nch, void* ws, size_t ws_bytes,
                   bool resident, hipStream_t st);
int qc_comm_allreduce(float* buf, int64_t count, void* comm, hipStream_t st);   // qc_comm.hip: RCCL sum, fp32, in place
int qc_amp_fwd_launch(const float* a, float* u, int n, int64_t B, int nch, hipStream_t);
int qc_amp_bwd_launch(const float* a, const float* ub, float* ab, int n, int64_t B, int nch, hipStream_t);
