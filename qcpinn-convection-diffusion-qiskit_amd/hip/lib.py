"""ctypes binding of ``libqcpinn_hip.so`` (C ABI in ``include/qcpinn_hip.h``).

The library is the product: there is no CPU or torch fallback behind these calls.  Loading
fails loudly when the shared object is missing, and every call raises ``QcError`` on a
non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# (QC_LIB: a diagnostic build of the same library, e.g. one compiled with -DH2_ABLATE_GATES / -DH2_ABLATE_SYNC, csrc/qc_h2_shared.h)
LIB_PATH = os.environ.get("QC_LIB") or os.path.join(os.path.dirname(_HERE), "libqcpinn_hip.so")

QC_PHASE_GRADS = 1
QC_PHASE_UPDATE = 2
QC_PHASE_SAMPLE = 4
QC_STAGE_PRE_FWD, QC_STAGE_CIRCUIT_FWD, QC_STAGE_POST, QC_STAGE_CIRCUIT_BWD, QC_STAGE_PRE_BWD = range(5)

# every symbol include/qcpinn_hip.h declares
EXPORTS = (
    "qc_version", "qc_error_string", "qc_last_hip_error", "qc_program_create", "qc_program_destroy",
    "qc_trig_bytes", "qc_program_set_encoding", "qc_amp_forward", "qc_amp_backward", "qc_prepare_gates",
    "qc_circuit_workspace_bytes", "qc_circuit_workspace_bytes_batch", "qc_hbm_plan_describe", "qc_wave_sched_describe", "qc_forward_expval", "qc_backward_expval", "qc_forward_jets",
    "qc_backward_jets", "qc_forward_jets_keep", "qc_backward_jets_kept", "qc_pre_forward", "qc_pre_backward", "qc_post", "qc_reduce_rows", "qc_adam_step",
    "qc_sample_collocation", "qc_sample_collocation_faces", "qc_step_workspace_bytes", "qc_fused_pinn_residual_step",
    "qc_fused_step_stage", "qc_post_multi", "qc_comm_unique_id", "qc_comm_create", "qc_comm_destroy", "qc_allreduce_grads",
)


QC_PROBLEM_CONVECTION_DIFFUSION, QC_PROBLEM_PURE_DIFFUSION = 0, 1      # qc_pde.problem


class QcError(RuntimeError):
    pass


class QcPde(C.Structure):
    _fields_ = [("D", C.c_float), ("vx", C.c_float), ("vy", C.c_float),
                ("c_t", C.c_float), ("c_x", C.c_float), ("c_y", C.c_float), ("d_xx", C.c_float), ("d_yy", C.c_float),
                ("w_res", C.c_float),
                ("inv_n_res", C.c_float), ("w_val_a", C.c_float), ("w_val_b", C.c_float),
                ("inv_n_a", C.c_float), ("inv_n_b", C.c_float), ("problem", C.c_int), ("n_seg_a", C.c_int64)]


class QcOptHyper(C.Structure):
    _fields_ = [("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_float), ("max_norm", C.c_float),
                ("sched_factor", C.c_float), ("sched_threshold", C.c_float), ("sched_min_lr", C.c_float),
                ("sched_eps", C.c_float), ("sched_patience", C.c_int), ("w_res", C.c_float),
                ("w_bc", C.c_float), ("w_ic", C.c_float)]


class QcStepDesc(C.Structure):
    _fields_ = [
        ("prog", C.c_void_p), ("trig_dev", C.c_void_p), ("umat_dev", C.c_void_p),
        ("H", C.c_int), ("n", C.c_int), ("n_theta", C.c_int),
        ("params_dev", C.c_void_p), ("m_dev", C.c_void_p), ("v_dev", C.c_void_p), ("opt_state_dev", C.c_void_p),
        ("hist_dev", C.c_void_p), ("hist_cap", C.c_int),
        ("X_res_dev", C.c_void_p), ("B_res", C.c_int64),
        ("X_val_dev", C.c_void_p), ("B_val", C.c_int64),
        ("ajets_res_dev", C.c_void_p), ("qjets_res_dev", C.c_void_p), ("qbar_res_dev", C.c_void_p),
        ("abar_res_dev", C.c_void_p),
        ("ajets_val_dev", C.c_void_p), ("qjets_val_dev", C.c_void_p), ("qbar_val_dev", C.c_void_p),
        ("abar_val_dev", C.c_void_p),
        ("part_dev", C.c_void_p), ("part_stride", C.c_int64), ("part_rows_cap", C.c_int64),
        ("flat_dev", C.c_void_p),
        ("pde", QcPde), ("hyper", QcOptHyper),
        ("n_ic", C.c_int64), ("sample_off_res", C.c_int64), ("sample_off_ic", C.c_int64),
        ("sample_off_bc", C.c_int64), ("sample_seed", C.c_uint64), ("sample_step", C.c_uint64),
        ("sample_bc_face_points", C.c_int64),
        ("circ_ws_dev", C.c_void_p), ("circ_ws_bytes", C.c_size_t),
        ("comm", C.c_void_p),
    ]


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """dlopen the in-tree library (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QcError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, fp = C.c_void_p, C.c_int, C.c_int64, C.c_void_p
    lib.qc_version.restype = i32
    lib.qc_error_string.restype = C.c_char_p
    lib.qc_error_string.argtypes = [i32]
    lib.qc_last_hip_error.restype = i32
    lib.qc_program_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    lib.qc_program_destroy.argtypes = [vp]
    lib.qc_trig_bytes.restype = C.c_size_t
    lib.qc_trig_bytes.argtypes = [vp]
    lib.qc_prepare_gates.argtypes = [vp, fp, vp, vp]
    lib.qc_circuit_workspace_bytes.restype = C.c_size_t
    lib.qc_circuit_workspace_bytes.argtypes = [vp, i32, i32]
    lib.qc_circuit_workspace_bytes_batch.restype = C.c_size_t
    lib.qc_circuit_workspace_bytes_batch.argtypes = [vp, i32, i32, i64]
    lib.qc_hbm_plan_describe.argtypes = [vp, i32, i32, i32, vp, i32]
    lib.qc_wave_sched_describe.argtypes = [vp, i32, i32, i32, vp, i32]
    lib.qc_forward_expval.argtypes = [vp, vp, fp, fp, fp, i64, vp, C.c_size_t, vp]
    lib.qc_backward_expval.argtypes = [vp, vp, fp, fp, fp, fp, fp, i64, i64, i64, vp, C.c_size_t, vp]
    lib.qc_forward_jets.argtypes = [vp, vp, fp, fp, fp, i64, vp, C.c_size_t, vp]
    lib.qc_backward_jets.argtypes = [vp, vp, fp, fp, fp, fp, fp, i64, i64, i64, vp, C.c_size_t, vp]
    lib.qc_forward_jets_keep.argtypes = [vp, vp, fp, fp, fp, i64, fp, vp]
    lib.qc_backward_jets_kept.argtypes = [vp, vp, fp, fp, fp, fp, fp, i64, i64, i64, fp, vp]
    lib.qc_pre_forward.argtypes = [fp, fp, i32, i32, i32, fp, i64, i32, vp]
    lib.qc_pre_backward.argtypes = [fp, fp, i32, i32, i32, fp, fp, i64, i64, i64, i32, vp]
    lib.qc_post.argtypes = [i32, fp, fp, i32, i32, i32, C.POINTER(QcPde), fp, fp, fp, fp, fp, fp, fp, i64, i64,
                            i64, i32, vp]
    lib.qc_post_multi.argtypes = [i32, fp, i32, i32, i32, i32, fp, fp, fp, fp, fp, fp, i64, fp, i64, i64, i64, vp]
    lib.qc_reduce_rows.argtypes = [fp, i64, i64, i32, fp, vp]
    lib.qc_adam_step.argtypes = [fp, i32, fp, fp, fp, vp, C.POINTER(QcOptHyper), fp, i32, vp, i32, vp, vp]
    lib.qc_sample_collocation.argtypes = [fp, i64, i64, fp, i64, i64, i64, i64, C.c_uint64, C.c_uint64, vp]
    lib.qc_sample_collocation_faces.argtypes = [fp, i64, i64, fp, i64, i64, i64, i64, i64, C.c_uint64, C.c_uint64, vp]
    lib.qc_fused_step_stage.argtypes = [C.POINTER(QcStepDesc), i32, vp]
    lib.qc_step_workspace_bytes.restype = C.c_size_t
    lib.qc_step_workspace_bytes.argtypes = [vp, i64, i64]
    lib.qc_program_set_encoding.argtypes = [vp, i32]
    lib.qc_amp_forward.argtypes = [fp, fp, i32, i64, i32, vp]
    lib.qc_amp_backward.argtypes = [fp, fp, fp, i32, i64, i32, vp]
    lib.qc_fused_pinn_residual_step.argtypes = [C.POINTER(QcStepDesc), i32, vp]
    lib.qc_comm_unique_id.argtypes = [vp]
    lib.qc_comm_create.argtypes = [vp, i32, i32, C.POINTER(vp)]
    lib.qc_comm_destroy.argtypes = [vp]
    lib.qc_allreduce_grads.argtypes = [fp, i64, vp, vp]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("qc_error_string", "qc_trig_bytes", "qc_circuit_workspace_bytes", "qc_circuit_workspace_bytes_batch",
                        "qc_step_workspace_bytes"):
            fn.restype = i32
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        lib = load()
        msg = lib.qc_error_string(rc).decode()
        raise QcError(f"{what or 'libqcpinn_hip'} failed: {msg} (code {rc}, hipError {lib.qc_last_hip_error()})")
