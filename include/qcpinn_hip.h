/* libqcpinn_hip.so — C ABI of the MI355X (gfx950) QCPINN hot path.
 *
 * The reference (masapasa/qcpinn-convection-diffusion-qiskit) has no FFI: its hot path is Python
 * calling PennyLane + torch autograd.  These entry points are what a binding for that path binds
 * instead; each one names the reference code whose arithmetic it replaces.  Plain pointers and
 * sizes only — no torch types.  Every pointer marked "dev" is a device (HBM) pointer owned by the
 * caller; the library never allocates or frees caller memory, never synchronises the stream, and
 * only enqueues work on `stream` (a hipStream_t passed as void*; NULL = default stream).
 * Every function returns 0 on success or a negative QC_ERR_* code; nothing throws.
 *
 * Layouts (all fp32):
 *   [f][B]      "batch-minor": feature f of point p at f*B + p (coalesced over the batch).
 *   jets        6 derivative channels {value, d/dt, d/dx, d/dy, d2/dx2, d2/dy2} x n wires: [6][n][B].
 *   flat params W1[H][3] b1[H] W2[n][H] b2[n] W3[H][n] b3[H] W4[H] b4 theta[n_theta]
 *               (= torch's model.parameters() order of the reference DVPDESolver).
 *   partial rows [rows][stride] gradients, one row per 64-point tile, columns in flat-param order
 *               followed by 3 loss columns (residual, BC, IC); reduced in fixed order (no atomics).
 */
#ifndef QCPINN_HIP_H
#define QCPINN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QC_ABI_VERSION 4

typedef struct qc_program qc_program; /* device-resident gate program (opaque) */

int qc_version(void);
const char* qc_error_string(int code);
int qc_last_hip_error(void); /* hipError_t of the last failing runtime call, 0 if none */

/* ---- gate program: the lowered ansatz of DVQuantumLayer._quantum_circuit
 * (reference nn/DVQuantumLayer.py:176-214 and the builders :246-371).
 * gate_rows: host int32 [n_gates][4] = (opcode, wire_a, wire_b, slot); opcodes
 * RX=0 RY=1 RZ=2 H=3 CNOT=4 CRX=5 CRZ=6 U4=7 (fixed two-wire unitary, slot 0 on wires [0,1],
 * slot 1 on wires [2,3]).  The embedding RX(x_i) is implicit. */
int qc_program_create(const int32_t* gate_rows, int n_gates, int n_qubits, int n_params, qc_program** out);
int qc_program_destroy(qc_program* prog);
size_t qc_trig_bytes(const qc_program* prog); /* size of the per-gate cos/sin workspace */
/* 0 (default): AngleEmbedding, RX(x_i) on wire i (nn/DVQuantumLayer.py:182).  1: AmplitudeEmbedding(normalize=True,
 * pad_with=0) (:177-180); the circuit entry points then take the jets of the INITIAL AMPLITUDES (see
 * qc_amp_forward) where they take angle jets otherwise, and return cotangents w.r.t. them. */
int qc_program_set_encoding(qc_program* prog, int amplitude);
/* angle-jet layout [nch][n][B] -> jets of u = a/|a| (same layout), and the pull-back of their cotangents */
int qc_amp_forward(const float* ajets_dev, float* ujets_dev, int n, int64_t B, int nch, void* stream);
int qc_amp_backward(const float* ajets_dev, const float* ubar_dev, float* abar_dev, int n, int64_t B, int nch,
                    void* stream);

/* cos/sin(theta/2) per gate; call after theta changes (the fused step does it itself). */
int qc_prepare_gates(const qc_program* prog, const float* theta_dev, void* trig_dev, void* stream);

/* Scratch the circuit entry points need for this program (0 for n <= 8: registers / lanes only;
 * for 9 <= n <= 20 the statevectors of one 64-point tile live in HBM).  nch = 1 or 6. */
size_t qc_circuit_workspace_bytes(const qc_program* prog, int nch, int backward);
/* The same for a batch of B points: 9 <= n <= 20 with angle encoding processes every 64-point tile that fits the
 * workspace in ONE launch per stage, so more scratch than the one-tile minimum above buys fewer, larger launches;
 * this returns the bytes that keep the whole batch resident (capped by the QC_HBM_KEEP_GB budget, default 96). */
size_t qc_circuit_workspace_bytes_batch(const qc_program* prog, int nch, int backward, int64_t B);
/* Execution plan of the HBM-resident family (9 <= n <= 20, angle encoding) for the gate rows of qc_program_create, as
 * a flat int32 record: stages (local bit sets), rounds (register bit sets), gates and diagonal tables -- the schedule
 * the kernels run for DVQuantumLayer._quantum_circuit (nn/DVQuantumLayer.py:176-214), which is NOT program order (gates
 * on disjoint wires and diagonal gates are commuted).  Host-only (no GPU needed).  Returns the number of int32 values
 * of the record (written up to `cap`), 0 on invalid input.  Layout: csrc/qc_hbm2_plan.h::h2_describe;
 * tests/test_hbm_plan.py re-executes the record on the CPU against the gate-by-gate program. */
int qc_hbm_plan_describe(const int32_t* gate_rows, int n_gates, int n_qubits, int n_params, int32_t* out, int cap);

/* The execution order of the compile-time programs of the n = 6..8 family (csrc/qc_wave_sched.h): RZ / CRZ gates of the
 * reference circuit (nn/DVQuantumLayer.py:246-262, :348-371) that commute into each other are collected into runs, each
 * applied as one phase-table multiply.  Host-only; the same function the kernels evaluate at compile time.  Record:
 * {n_items, n_runs, items[n_items] (gate index, or -(run + 1)), then per run {count, gate indices}}.  Returns the
 * number of int32 values (written up to `cap`), 0 on invalid input; tests/test_wave_sched.py re-executes it on the CPU. */
int qc_wave_sched_describe(const int32_t* gate_rows, int n_gates, int n_qubits, int n_params, int32_t* out, int cap);

/* ---- DVQuantumLayer.forward, simulator batch branch (nn/DVQuantumLayer.py:151-154):
 * angles [n][B] -> <Z_w> [n][B].  umat_dev: [2 slots][fwd, adjoint][4x4 complex] floats or NULL.
 * ws_dev / ws_bytes: caller-owned scratch of at least qc_circuit_workspace_bytes() (may be NULL/0 if that is 0). */
int qc_forward_expval(const qc_program* prog, const void* trig_dev, const float* umat_dev,
                      const float* angles_dev, float* expval_dev, int64_t B, void* ws_dev, size_t ws_bytes,
                      void* stream);
/* its vector-Jacobian product (what loss.backward() asks of PennyLane's backprop):
 * cot [n][B] -> d_angles [n][B] and d_theta as partial rows at columns [0, n_params) of `part`. */
int qc_backward_expval(const qc_program* prog, const void* trig_dev, const float* umat_dev,
                       const float* angles_dev, const float* cot_dev, float* d_angles_dev,
                       float* part_dev, int64_t part_stride, int64_t row0, int64_t B, void* ws_dev, size_t ws_bytes,
                       void* stream);

/* ---- the same layer carrying the derivative channels that nn/pde.py:59-70 obtains with five
 * torch.autograd.grad(create_graph=True) calls through the simulator. */
int qc_forward_jets(const qc_program* prog, const void* trig_dev, const float* umat_dev,
                    const float* ajets_dev, float* qjets_dev, int64_t B, void* ws_dev, size_t ws_bytes, void* stream);
int qc_backward_jets(const qc_program* prog, const void* trig_dev, const float* umat_dev,
                     const float* ajets_dev, const float* qbar_dev, float* abar_dev, float* part_dev,
                     int64_t part_stride, int64_t row0, int64_t B, void* ws_dev, size_t ws_bytes, void* stream);

/* Register-family (2 <= n <= 5) pair that passes the forward pass's final statevectors to the adjoint
 * pass through chi_dev [6][2*2^n][B] floats instead of recomputing them; both calls must see the same
 * angle jets and parameters (the fused step uses them back to back). */
int qc_forward_jets_keep(const qc_program* prog, const void* trig_dev, const float* umat_dev, const float* ajets_dev,
                         float* qjets_dev, int64_t B, float* chi_dev, void* stream);
int qc_backward_jets_kept(const qc_program* prog, const void* trig_dev, const float* umat_dev, const float* ajets_dev,
                          const float* qbar_dev, float* abar_dev, float* part_dev, int64_t part_stride, int64_t row0,
                          int64_t B, const float* chi_dev, void* stream);

/* ---- classical pre/post networks of DVPDESolver.forward (nn/DVPDESolver.py:37-51,81-110) with
 * the same channels.  nch = 6 (residual points) or 1 (boundary/initial points). */
int qc_pre_forward(const float* X_dev /*[B][3]*/, const float* params_dev, int H, int n, int n_theta,
                   float* ajets_dev, int64_t B, int nch, void* stream);
int qc_pre_backward(const float* X_dev, const float* params_dev, int H, int n, int n_theta,
                    const float* abar_dev, float* part_dev, int64_t part_stride, int64_t row0, int64_t B,
                    int nch, void* stream);

typedef struct qc_pde {
  float D, vx, vy;        /* nn/pde.py:53-55 defaults 0.01, 1, 1: the constants of the analytic targets (mode 2) */
  /* operator coefficients, residual = c_t u_t + c_x u_x + c_y u_y - (d_xx u_xx + d_yy u_yy): with the sigma scalings of
   * nn/pde.py:60-70 folded in, c_t = 1/sigma_t, c_x = v_x/sigma_x, c_y = v_y/sigma_y, d_xx = D/sigma_x^2, d_yy = D/sigma_y^2 */
  float c_t, c_x, c_y, d_xx, d_yy;
  float w_res;            /* d loss / d residual scale: 2*weight/N (weight 2, trainer/diffusion_train.py:47) */
  float inv_n_res;        /* 1/N for the logged MSE */
  float w_val_a, w_val_b; /* same for the value segments: a = IC (weight 2), b = BC (weight 4) */
  float inv_n_a, inv_n_b;
  int problem;            /* analytic targets of mode 2: QC_PROBLEM_CONVECTION_DIFFUSION (0) or QC_PROBLEM_PURE_DIFFUSION (1) */
  int64_t n_seg_a;        /* leading value points that are IC points */
} qc_pde;

/* 0: trainer/diffusion_train.py + data/diffusion_dataset.py:20-38 (Gaussian u on IC and BC1 points, forcing
 *    term r incl. its -400 constant on residual points);
 * 1: the reference's second workload train_hybrid_qpinn.py:116-131,159-203: u = sin(pi x) sin(pi y) exp(-2 pi^2 D t)
 *    on IC points, 0 on the four boundary faces, residual target 0 (pure diffusion: set vx = vy = 0). */
#define QC_PROBLEM_CONVECTION_DIFFUSION 0
#define QC_PROBLEM_PURE_DIFFUSION 1

/* mode 0: qjets -> u [B], residual [B] (nn/pde.py:71);
 * mode 1: cotangents (ubar, rbar) [B] -> qbar jets + weight-gradient partial rows;
 * mode 2: forward + analytic targets (data/diffusion_dataset.py:20-38) + squared error + reverse;
 *         out_u_dev / out_res_dev are then B-float scratch buffers (per-point cotangents);
 * mode 4 (nch = 6): qjets -> all six derivative channels of u, out_u_dev = [6][B] (value, d/dt, d/dx, d/dy, d2/dx2,
 *         d2/dy2): operators that are not linear in the channels (Navier-Stokes, nn/pde.py:2-27) combine them outside;
 * mode 3 (nch = 6): its reverse: in_ubar_dev = [6][B] cotangents of those channels -> qbar jets + partial rows. */
int qc_post(int mode, const float* X_dev, const float* params_dev, int H, int n, int n_theta,
            const qc_pde* pde, const float* qjets_dev, float* out_u_dev, float* out_res_dev,
            const float* in_ubar_dev, const float* in_rbar_dev, float* qbar_dev, float* part_dev,
            int64_t part_stride, int64_t row0, int64_t B, int nch, void* stream);

/* K outputs behind one shared network: a post network Linear(n, H) -> Tanh -> Linear(H, K), 1 <= K <= 4 (the (u, v, p)
 * model nn/pde.py:2-27 differentiates for Navier-Stokes).  params_dev: the flat vector of the single-output layout
 * (its W4 / b4 slots are not read); w4k_dev = [K][H + 1] rows (W4[k][0..H-1], b4[k]).
 * mode 4: out_u_dev = [K][6][B], the six derivative channels of every output, from ONE evaluation of the hidden layer;
 * mode 3: its reverse, in_ubar_dev = [K][6][B] -> qbar jets [6][n][B], the tile rows of the shared parameters in
 * part_dev (W3, b3 columns; the W4 / b4 columns are zeroed) and of the last layer in partk_dev ([rows][K * (H + 1)]). */
int qc_post_multi(int mode, const float* params_dev, int H, int n, int n_theta, int K, const float* w4k_dev,
                  const float* qjets_dev, float* out_u_dev, const float* in_ubar_dev, float* qbar_dev, float* part_dev,
                  int64_t part_stride, float* partk_dev, int64_t partk_stride, int64_t row0, int64_t B, void* stream);

/* ---- optimiser block of the step (trainer/diffusion_train.py:81-90) */
int qc_reduce_rows(const float* part_dev, int64_t rows, int64_t stride, int ncols, float* out_dev, void* stream);

typedef struct qc_opt_hyper {
  double beta1, beta2;
  float eps, max_norm;
  float sched_factor, sched_threshold, sched_min_lr, sched_eps;
  int sched_patience;
  float w_res, w_bc, w_ic;
} qc_opt_hyper;

/* opt_state_dev: 64-byte record {float lr, best; int num_bad, step; float last_loss, last_norm, loss_parts[3];
 * int hist_base; pad}.  flat_dev = [grad[NP] | L_r, L_bc, L_ic].  Clips, applies Adam, steps the plateau
 * scheduler, writes the loss of (1-based) step s to hist_dev[s - 1 - hist_base] when that index lies in
 * [0, hist_cap) (hist_base = steps taken before this history buffer started, e.g. by an earlier train() call on
 * the same optimiser state) and refreshes the trig table from the new theta. */
int qc_adam_step(float* flat_dev, int NP, float* params_dev, float* m_dev, float* v_dev, void* opt_state_dev,
                 const qc_opt_hyper* hp, float* hist_dev, int hist_cap, const qc_program* prog, int theta_off,
                 void* trig_dev, void* stream);

/* ---- the three uniform batches of a step (trainer/diffusion_train.py:9-20,34-36: IC t=0 face, BC1 x=0
 * face, residual in [0,1]^3; data/diffusion_dataset.py:12-19) drawn on device with Philox4x32-10 keyed by
 * (seed, step, batch) and indexed by the GLOBAL point index off_* + i: ranks of a data-parallel run fill
 * disjoint shards of the batch a single GPU would draw.  X_val holds the IC points first, then BC. */
int qc_sample_collocation(float* X_res_dev, int64_t n_res, int64_t off_res, float* X_val_dev, int64_t n_ic,
                          int64_t off_ic, int64_t n_bc, int64_t off_bc, uint64_t seed, uint64_t step, void* stream);
/* Same, with the boundary batch spread over the four faces x=0, x=1, y=0, y=1 in that order
 * (train_hybrid_qpinn.py:166-176,689-697): GLOBAL boundary point g lies on face g / bc_face_points.
 * bc_face_points = 0 is the single x=0 face of qc_sample_collocation. */
int qc_sample_collocation_faces(float* X_res_dev, int64_t n_res, int64_t off_res, float* X_val_dev, int64_t n_ic,
                                int64_t off_ic, int64_t n_bc, int64_t off_bc, int64_t bc_face_points, uint64_t seed,
                                uint64_t step, void* stream);

/* ---- one whole training step (trainer/diffusion_train.py:30-49,81-90) on resident batches:
 * residual batch through the 6-channel pipeline, IC+BC batch through the value pipeline,
 * row reduction, then (phase 2) clip + Adam + scheduler.  With `phases` = 1 it stops after the
 * reduction so the caller can all-reduce `flat_dev` across ranks, then calls again with 2. */
typedef struct qc_step_desc {
  const qc_program* prog;
  void* trig_dev;
  const float* umat_dev;
  int H, n, n_theta;
  float* params_dev; float* m_dev; float* v_dev; void* opt_state_dev;
  float* hist_dev; int hist_cap;
  const float* X_res_dev; int64_t B_res;
  const float* X_val_dev; int64_t B_val;   /* IC points first, then BC points */
  float* ajets_res_dev; float* qjets_res_dev; float* qbar_res_dev; float* abar_res_dev; /* 6*n*B_res each */
  float* ajets_val_dev; float* qjets_val_dev; float* qbar_val_dev; float* abar_val_dev; /* n*B_val each */
  float* part_dev; int64_t part_stride; int64_t part_rows_cap;   /* scratch: rewritten and folded in place every step */
  float* flat_dev;                                                                       /* NP+3 */
  qc_pde pde;
  qc_opt_hyper hyper;
  /* on-device sampler (QC_PHASE_SAMPLE): global index of this rank's first point in each batch */
  int64_t n_ic;                 /* leading IC points of the value batch (== pde.n_seg_a) */
  int64_t sample_off_res, sample_off_ic, sample_off_bc;
  uint64_t sample_seed, sample_step;
  int64_t sample_bc_face_points; /* 0: BC batch on the x=0 face; > 0: four faces, see qc_sample_collocation_faces */
  void* circ_ws_dev; size_t circ_ws_bytes;   /* qc_step_workspace_bytes(prog, B_res, B_val); NULL/0 allowed for angle encoding at n <= 8 */
  /* data parallelism inside the library: a communicator of qc_comm_create (or NULL).  With it a call with
   * QC_PHASE_GRADS | QC_PHASE_UPDATE all-reduces flat_dev across the ranks between the two phases, on `stream`. */
  void* comm;
} qc_step_desc;

/* Scratch for one fused step on B_res residual points: the HBM statevector tile for n >= 9; for the
 * register family (2 <= n <= 5) an optional [6][2*2^n][B_res] store of the forward pass's final states
 * that lets the adjoint kernel skip recomputing them (pass less and it recomputes); with amplitude encoding
 * additionally the initial-amplitude jets and their cotangents of both pipelines (required). */
size_t qc_step_workspace_bytes(const qc_program* prog, int64_t B_res, int64_t B_val);

/* One stage of the step above when it runs in its merged form (2 <= n <= 5, angle encoding, both batches non-empty,
 * workspace present): each stage is a single launch over the value tiles and the residual tiles together.  For
 * per-kernel timing; QC_ERR_UNSUPPORTED when the step would take the two-stream form instead. */
#define QC_STAGE_PRE_FWD 0
#define QC_STAGE_CIRCUIT_FWD 1
#define QC_STAGE_POST 2          /* point kernel + weight-gradient kernel */
#define QC_STAGE_CIRCUIT_BWD 3
#define QC_STAGE_PRE_BWD 4
#define QC_STAGE_COUNT 5
int qc_fused_step_stage(const qc_step_desc* desc, int stage, void* stream);

/* ---- data-parallel collective (collocation batches shard over the GPUs of a node, ONE all-reduce of the
 * flat [gradient | L_r, L_bc, L_ic] vector per step; the reference is single-process and has no counterpart).  RCCL over
 * xGMI, loaded on first use (QC_ERR_UNSUPPORTED when librccl is not available).  Call with the rank's device current.
 * id: 128 bytes (ncclUniqueId), produced on one rank and distributed to the others by the caller's own means. */
int qc_comm_unique_id(void* id_out_128_bytes);
int qc_comm_create(const void* id_128_bytes, int world_size, int rank, void** comm_out);
int qc_comm_destroy(void* comm);
/* in-place sum over the ranks of `count` floats (fp32) on `stream` */
int qc_allreduce_grads(float* buf_dev, int64_t count, void* comm, void* stream);

#define QC_PHASE_GRADS 1
#define QC_PHASE_UPDATE 2
#define QC_PHASE_SAMPLE 4 /* fill X_res / X_val first (see qc_sample_collocation) */
int qc_fused_pinn_residual_step(const qc_step_desc* desc, int phases, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QCPINN_HIP_H */
