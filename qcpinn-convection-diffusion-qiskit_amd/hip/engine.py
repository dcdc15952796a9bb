"""Host-side driver of the HIP kernels: owns device workspaces, hands raw pointers and the
current HIP stream to ``libqcpinn_hip.so``.  torch is plumbing here (device memory, streams,
autograd glue); all arithmetic of the hot path runs in the library.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from ..circuits import GateProgram

NCH = 6  # derivative channels: value, d/dt, d/dx, d/dy, d2/dx2, d2/dy2


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, device, what: str) -> torch.Tensor:
    if t.device != device or t.dtype != torch.float32:
        raise L.QcError(f"{what}: expected a float32 tensor on {device}, got {t.dtype} on {t.device}")
    return t.contiguous()


def param_layout(H: int, n: int, n_theta: int) -> Dict[str, Tuple[int, Tuple[int, ...]]]:
    """Offsets/shapes of the flat parameter vector = the reference's ``model.parameters()`` order
    (nn/DVPDESolver.py:28-57: preprocessor, postprocessor, quantum_layer)."""
    out, off = {}, 0
    for name, shape in (("preprocessor.0.weight", (H, 3)), ("preprocessor.0.bias", (H,)),
                        ("preprocessor.2.weight", (n, H)), ("preprocessor.2.bias", (n,)),
                        ("postprocessor.0.weight", (H, n)), ("postprocessor.0.bias", (H,)),
                        ("postprocessor.2.weight", (1, H)), ("postprocessor.2.bias", (1,)),
                        ("quantum_layer.params", (n_theta,))):
        out[name] = (off, shape)
        off += int(np.prod(shape))
    out["__total__"] = (off, ())
    return out


class Circuit:
    """A device-resident gate program + its per-gate trig table and fixed-unitary table."""

    def __init__(self, program: GateProgram, haar: Optional[np.ndarray], device: torch.device,
                 amplitude: bool = False):
        if device.type != "cuda":
            raise L.QcError("the HIP kernels need a GPU device (torch device type 'cuda' on ROCm); "
                            "there is no CPU fallback")
        self.lib = L.load()
        self.program = program
        self.device = device
        self.n = program.n_qubits
        self.n_params = program.n_params
        rows = np.ascontiguousarray(program.rows())
        self.handle = C.c_void_p()
        with torch.cuda.device(device):
            L.check(self.lib.qc_program_create(rows.ctypes.data_as(C.c_void_p), program.n_gates, self.n,
                                               self.n_params, C.byref(self.handle)), "qc_program_create")
        self.amplitude = bool(amplitude)
        if self.amplitude:
            L.check(self.lib.qc_program_set_encoding(self.handle, 1), "qc_program_set_encoding")
        self.trig = torch.zeros(int(self.lib.qc_trig_bytes(self.handle)) // 4, dtype=torch.float32, device=device)
        self.umat = None
        if program.use_haar:
            if haar is None:
                raise L.QcError("program uses the fixed two-wire unitaries but none were given")
            u = np.asarray(haar, dtype=np.complex128)                       # (2,4,4)
            both = np.stack([u, np.conj(np.transpose(u, (0, 2, 1)))], axis=1)  # [slot][fwd|adj][4][4]
            packed = np.stack([both.real, both.imag], axis=-1).astype(np.float32)
            self.umat = torch.from_numpy(np.ascontiguousarray(packed)).to(device)

    def __del__(self):
        try:
            if self.handle:
                self.lib.qc_program_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    # -- scratch for the HBM-resident family (n >= 9); empty for the register / lane families
    def workspace(self, nch: int, backward: bool, B: int = 64):
        # the whole batch resident when it fits (fewer, larger launches), never less than the one-tile minimum
        need = max(int(self.lib.qc_circuit_workspace_bytes(self.handle, nch, 1 if backward else 0)),
                   int(self.lib.qc_circuit_workspace_bytes_batch(self.handle, nch, 1 if backward else 0, B)))
        if need == 0:
            return None, 0
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < need:
            self._ws = ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws.data_ptr(), ws.numel()

    # -- parameters changed: refresh cos/sin(theta/2)
    def prepare(self, theta: torch.Tensor) -> None:
        theta = _need(theta.reshape(-1), self.device, "theta")
        L.check(self.lib.qc_prepare_gates(self.handle, theta.data_ptr(), self.trig.data_ptr(),
                                          _stream(self.device)), "qc_prepare_gates")

    # -- amplitude encoding: features -> jets of the normalised initial amplitudes, and back
    def _amp_fwd(self, a: torch.Tensor, nch: int) -> torch.Tensor:
        u = torch.empty_like(a)
        L.check(self.lib.qc_amp_forward(a.data_ptr(), u.data_ptr(), self.n, a.shape[-1], nch, _stream(self.device)),
                "qc_amp_forward")
        return u

    def _amp_bwd(self, a: torch.Tensor, ubar: torch.Tensor, nch: int) -> torch.Tensor:
        ab = torch.empty_like(a)
        L.check(self.lib.qc_amp_backward(a.data_ptr(), ubar.data_ptr(), ab.data_ptr(), self.n, a.shape[-1], nch,
                                         _stream(self.device)), "qc_amp_backward")
        return ab

    def forward_expval(self, angles_nB: torch.Tensor) -> torch.Tensor:
        a = _need(angles_nB, self.device, "angles")
        if self.amplitude:
            a = self._amp_fwd(a, 1)
        B = a.shape[1]
        out = torch.empty_like(a)
        wp, wb = self.workspace(1, False, B)
        L.check(self.lib.qc_forward_expval(self.handle, self.trig.data_ptr(), _ptr(self.umat), a.data_ptr(),
                                           out.data_ptr(), B, wp, wb, _stream(self.device)), "qc_forward_expval")
        return out

    def backward_expval(self, angles_nB: torch.Tensor, cot_nB: torch.Tensor):
        a = a_in = _need(angles_nB, self.device, "angles")
        if self.amplitude:
            a = self._amp_fwd(a_in, 1)
        g = _need(cot_nB, self.device, "cotangent")
        B = a.shape[1]
        rows = (B + 63) // 64
        P = max(self.n_params, 1)
        part = torch.empty(rows, P, dtype=torch.float32, device=self.device)
        d_angles = torch.empty_like(a)
        st = _stream(self.device)
        wp, wb = self.workspace(1, True, B)
        L.check(self.lib.qc_backward_expval(self.handle, self.trig.data_ptr(), _ptr(self.umat), a.data_ptr(),
                                            g.data_ptr(), d_angles.data_ptr(), part.data_ptr(), P, 0, B, wp, wb, st),
                "qc_backward_expval")
        d_theta = torch.empty(P, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_reduce_rows(part.data_ptr(), rows, P, P, d_theta.data_ptr(), st), "qc_reduce_rows")
        if self.amplitude:
            d_angles = self._amp_bwd(a_in, d_angles, 1)
        return d_angles, d_theta[: self.n_params]

    def forward_jets(self, ajets: torch.Tensor) -> torch.Tensor:
        a = _need(ajets, self.device, "angle jets")            # (6, n, B)
        if self.amplitude:
            a = self._amp_fwd(a, NCH)
        B = a.shape[2]
        out = torch.empty_like(a)
        wp, wb = self.workspace(NCH, False, B)
        L.check(self.lib.qc_forward_jets(self.handle, self.trig.data_ptr(), _ptr(self.umat), a.data_ptr(),
                                         out.data_ptr(), B, wp, wb, _stream(self.device)), "qc_forward_jets")
        return out

    def backward_jets(self, ajets: torch.Tensor, qbar: torch.Tensor):
        a = a_in = _need(ajets, self.device, "angle jets")
        if self.amplitude:
            a = self._amp_fwd(a_in, NCH)
        g = _need(qbar, self.device, "cotangent jets")
        B = a.shape[2]
        rows = (B + 63) // 64
        P = max(self.n_params, 1)
        part = torch.empty(rows, P, dtype=torch.float32, device=self.device)
        abar = torch.empty_like(a)
        st = _stream(self.device)
        wp, wb = self.workspace(NCH, True, B)
        L.check(self.lib.qc_backward_jets(self.handle, self.trig.data_ptr(), _ptr(self.umat), a.data_ptr(),
                                          g.data_ptr(), abar.data_ptr(), part.data_ptr(), P, 0, B, wp, wb, st),
                "qc_backward_jets")
        d_theta = torch.empty(P, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_reduce_rows(part.data_ptr(), rows, P, P, d_theta.data_ptr(), st), "qc_reduce_rows")
        if self.amplitude:
            abar = self._amp_bwd(a_in, abar, NCH)
        return abar, d_theta[: self.n_params]


class SolverEngine:
    """Kernel pipelines of DVPDESolver on one GPU: value forward, residual forward, their reverse
    passes (autograd path), and the fused training step."""

    def __init__(self, circuit: Circuit, hidden: int, flat_params: torch.Tensor,
                 D: float = 0.01, vx: float = 1.0, vy: float = 1.0):
        self.lib = circuit.lib
        self.circuit = circuit
        self.device = circuit.device
        self.n = circuit.n
        self.H = int(hidden)
        self.n_theta = circuit.n_params
        self.layout = param_layout(self.H, self.n, self.n_theta)
        self.NP = self.layout["__total__"][0]
        self.theta_off = self.layout["quantum_layer.params"][0]
        if flat_params.numel() != self.NP:
            raise L.QcError(f"flat parameter vector has {flat_params.numel()} entries, layout needs {self.NP}")
        self.flat = _need(flat_params, self.device, "flat params")
        self.D, self.vx, self.vy = float(D), float(vx), float(vy)
        self.sigma = (1.0, 1.0, 1.0)                           # sigma_t, sigma_x, sigma_y of nn/pde.py:53-70
        self.coeffs = None      # explicit (c_t, c_x, c_y, d_xx, d_yy) of another linear operator on the same channels
        self.problem = L.QC_PROBLEM_CONVECTION_DIFFUSION      # analytic targets of the fused loss (qcpinn_hip.h)
        self._fused: Dict[Tuple[int, int, int], "FusedStep"] = {}

    # ------------------------------------------------------------------ helpers
    def _pde(self, n_res=1, n_ic=1, n_bc=1, n_seg_a=0) -> L.QcPde:
        # loss = 2*MSE_res + 4*MSE_bc + 2*MSE_ic (trainer/diffusion_train.py:47); d/d(err) = 2*w/N * err
        # u_k/sigma_k and u_kk/sigma_k^2 (nn/pde.py:60-70) are scalings of channels the kernels already carry
        st, sx, sy = self.sigma
        co = self.coeffs if self.coeffs is not None else (1.0 / st, self.vx / sx, self.vy / sy, self.D / (sx * sx),
                                                          self.D / (sy * sy))
        return L.QcPde(self.D, self.vx, self.vy, co[0], co[1], co[2], co[3], co[4], 4.0 / n_res, 1.0 / n_res, 4.0 / n_ic, 8.0 / n_bc,
                       1.0 / n_ic, 1.0 / n_bc, self.problem, n_seg_a)

    def refresh_gates(self) -> None:
        self.circuit.prepare(self.flat[self.theta_off: self.theta_off + self.n_theta])

    def _X(self, X: torch.Tensor) -> torch.Tensor:
        X = _need(X, self.device, "collocation points")
        if X.dim() != 2 or X.shape[1] != 3:
            raise L.QcError(f"collocation points must have shape (B, 3), got {tuple(X.shape)}")
        return X

    # ------------------------------------------------------------------ forward passes
    def forward(self, X: torch.Tensor, nch: int, refresh: bool = True, ujets: bool = False):
        """Returns (u, residual_or_None, ajets, qjets).  nch = 1: value only; 6: with residual.  ``ujets`` (nch = 6):
        the first element is the [6, B] tensor of u's derivative channels instead."""
        X = self._X(X)
        B = X.shape[0]
        st = _stream(self.device)
        if refresh:
            self.refresh_gates()
        ajets = torch.empty(nch, self.n, B, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_pre_forward(X.data_ptr(), self.flat.data_ptr(), self.H, self.n, self.n_theta,
                                        ajets.data_ptr(), B, nch, st), "qc_pre_forward")
        if nch == 1:
            qjets = self.circuit.forward_expval(ajets[0]).unsqueeze(0)
        else:
            qjets = self.circuit.forward_jets(ajets)
        pde = self._pde()
        if ujets:      # all six derivative channels of u (qc_post mode 4) instead of (u, residual)
            uj = torch.empty(NCH, B, dtype=torch.float32, device=self.device)
            L.check(self.lib.qc_post(4, X.data_ptr(), self.flat.data_ptr(), self.H, self.n, self.n_theta,
                                     C.byref(pde), qjets.data_ptr(), uj.data_ptr(), None, None, None, None, None,
                                     0, 0, B, nch, st), "qc_post(forward, six channels)")
            return uj, None, ajets, qjets
        u = torch.empty(B, 1, dtype=torch.float32, device=self.device)
        res = torch.empty(B, 1, dtype=torch.float32, device=self.device) if nch == NCH else None
        L.check(self.lib.qc_post(0, X.data_ptr(), self.flat.data_ptr(), self.H, self.n, self.n_theta,
                                 C.byref(pde), qjets.data_ptr(), u.data_ptr(), _ptr(res), None, None, None, None,
                                 0, 0, B, nch, st), "qc_post(forward)")
        return u, res, ajets, qjets

    # ------------------------------------------------------------------ reverse pass (autograd path)
    def backward(self, X, ajets, qjets, ubar, rbar, nch: int, ujets: bool = False) -> torch.Tensor:
        """Vector-Jacobian product of (u, residual) w.r.t. the flat parameters: returns d_flat (NP).  ``ujets``:
        ``ubar`` is the [6, B] cotangent of u's six derivative channels (qc_post mode 3), ``rbar`` unused."""
        X = self._X(X)
        B = X.shape[0]
        st = _stream(self.device)
        rows = (B + 63) // 64
        part = torch.empty(rows, self.NP, dtype=torch.float32, device=self.device)
        qbar = torch.empty_like(qjets)
        pde = self._pde()
        ub = None if ubar is None else _need(ubar.reshape(-1), self.device, "ubar")
        rb = None if (rbar is None or ujets) else _need(rbar.reshape(-1), self.device, "rbar")
        if ujets and (ub is None or ub.numel() != NCH * B or nch != NCH):
            raise L.QcError("the six-channel cotangent must be a [6, B] tensor")
        L.check(self.lib.qc_post(3 if ujets else 1, X.data_ptr(), self.flat.data_ptr(), self.H, self.n, self.n_theta,
                                 C.byref(pde), qjets.data_ptr(), None, None, _ptr(ub), _ptr(rb), qbar.data_ptr(),
                                 part.data_ptr(), self.NP, 0, B, nch, st), "qc_post(backward)")
        abar = torch.empty_like(ajets)
        th = part.data_ptr() + 4 * self.theta_off
        c = self.circuit
        wp, wb = c.workspace(nch, True, B)
        cin = c._amp_fwd(ajets, nch) if c.amplitude else ajets
        if nch == 1:
            L.check(self.lib.qc_backward_expval(c.handle, c.trig.data_ptr(), _ptr(c.umat), cin.data_ptr(),
                                                qbar.data_ptr(), abar.data_ptr(), th, self.NP, 0, B, wp, wb, st),
                    "qc_backward_expval")
        else:
            L.check(self.lib.qc_backward_jets(c.handle, c.trig.data_ptr(), _ptr(c.umat), cin.data_ptr(),
                                              qbar.data_ptr(), abar.data_ptr(), th, self.NP, 0, B, wp, wb, st),
                    "qc_backward_jets")
        if c.amplitude:
            abar = c._amp_bwd(ajets, abar, nch)
        L.check(self.lib.qc_pre_backward(X.data_ptr(), self.flat.data_ptr(), self.H, self.n, self.n_theta,
                                         abar.data_ptr(), part.data_ptr(), self.NP, 0, B, nch, st),
                "qc_pre_backward")
        d_flat = torch.empty(self.NP, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_reduce_rows(part.data_ptr(), rows, self.NP, self.NP, d_flat.data_ptr(), st),
                "qc_reduce_rows")
        return d_flat

    # ------------------------------------------------------------------ K outputs behind one shared network
    def forward_multi(self, X: torch.Tensor, w4k: torch.Tensor):
        """Six derivative channels of every output of a K-output model, [K, 6, B], from ONE pass of pre network,
        circuit and hidden layer (qc_post_multi mode 4).  ``w4k`` = [K, H + 1] rows (W4[k], b4[k]); the W4 / b4 slots of
        the flat vector are unused.  Returns (ujets, ajets, qjets)."""
        X = self._X(X)
        B, K = X.shape[0], int(w4k.shape[0])
        st = _stream(self.device)
        self.refresh_gates()
        w4k = _need(w4k, self.device, "last-layer rows")
        ajets = torch.empty(NCH, self.n, B, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_pre_forward(X.data_ptr(), self.flat.data_ptr(), self.H, self.n, self.n_theta,
                                        ajets.data_ptr(), B, NCH, st), "qc_pre_forward")
        qjets = self.circuit.forward_jets(ajets)
        uj = torch.empty(K, NCH, B, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_post_multi(4, self.flat.data_ptr(), self.H, self.n, self.n_theta, K, w4k.data_ptr(),
                                       qjets.data_ptr(), uj.data_ptr(), None, None, None, 0, None, 0, 0, B, st),
                "qc_post_multi(forward)")
        return uj, ajets, qjets

    def backward_multi(self, X, ajets, qjets, w4k, ubar):
        """Reverse of ``forward_multi``: ``ubar`` [K, 6, B] -> (d_flat (NP; zeros in the W4 / b4 slots), d_w4k [K, H + 1]),
        one circuit adjoint sweep and one pre-network reverse pass for all K outputs."""
        X = self._X(X)
        B, K = X.shape[0], int(w4k.shape[0])
        st = _stream(self.device)
        rows = (B + 63) // 64
        KW = K * (self.H + 1)
        part = torch.empty(rows, self.NP, dtype=torch.float32, device=self.device)
        partk = torch.empty(rows, KW, dtype=torch.float32, device=self.device)
        qbar = torch.empty_like(qjets)
        ub = _need(ubar.reshape(-1), self.device, "ubar")
        if ub.numel() != K * NCH * B:
            raise L.QcError("the cotangent must be a [K, 6, B] tensor")
        w4k = _need(w4k, self.device, "last-layer rows")
        L.check(self.lib.qc_post_multi(3, self.flat.data_ptr(), self.H, self.n, self.n_theta, K, w4k.data_ptr(),
                                       qjets.data_ptr(), None, ub.data_ptr(), qbar.data_ptr(), part.data_ptr(), self.NP,
                                       partk.data_ptr(), KW, 0, B, st), "qc_post_multi(backward)")
        abar = torch.empty_like(ajets)
        th = part.data_ptr() + 4 * self.theta_off
        c = self.circuit
        wp, wb = c.workspace(NCH, True, B)
        cin = c._amp_fwd(ajets, NCH) if c.amplitude else ajets
        L.check(self.lib.qc_backward_jets(c.handle, c.trig.data_ptr(), _ptr(c.umat), cin.data_ptr(), qbar.data_ptr(),
                                          abar.data_ptr(), th, self.NP, 0, B, wp, wb, st), "qc_backward_jets")
        if c.amplitude:
            abar = c._amp_bwd(ajets, abar, NCH)
        L.check(self.lib.qc_pre_backward(X.data_ptr(), self.flat.data_ptr(), self.H, self.n, self.n_theta,
                                         abar.data_ptr(), part.data_ptr(), self.NP, 0, B, NCH, st), "qc_pre_backward")
        d_flat = torch.empty(self.NP, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_reduce_rows(part.data_ptr(), rows, self.NP, self.NP, d_flat.data_ptr(), st), "qc_reduce_rows")
        d_w4k = torch.empty(KW, dtype=torch.float32, device=self.device)
        L.check(self.lib.qc_reduce_rows(partk.data_ptr(), rows, KW, KW, d_w4k.data_ptr(), st), "qc_reduce_rows")
        return d_flat, d_w4k.view(K, self.H + 1)

    # ------------------------------------------------------------------ fused training step
    def fused(self, B_res: int, n_ic: int, n_bc: int, opt: "OptimState", counts=None) -> "FusedStep":
        key = (B_res, n_ic, n_bc, id(opt), counts, self.problem, self.D, self.vx, self.vy, self.sigma, self.coeffs)
        if key not in self._fused:
            self._fused[key] = FusedStep(self, B_res, n_ic, n_bc, opt, counts)
        return self._fused[key]


class OptimState:
    """Device-resident Adam moments + the 64-byte {lr, best, num_bad, step, ...} record that the
    optimiser kernel advances (trainer/diffusion_train.py:81-90 without host round trips)."""

    def __init__(self, NP: int, lr: float, device, hist_cap: int = 0, betas=(0.9, 0.999), eps=1e-8,
                 max_norm=1.0, factor=0.9, patience=1000, threshold=1e-4, min_lr=0.0, sched_eps=1e-8):
        self.device = device
        self.m = torch.zeros(NP, dtype=torch.float32, device=device)
        self.v = torch.zeros(NP, dtype=torch.float32, device=device)
        rec = np.zeros(16, dtype=np.float32)
        rec[0] = lr
        rec[1] = np.inf                 # ReduceLROnPlateau mode "min": best starts at +inf
        self.state = torch.from_numpy(rec).to(device)
        self.hist = torch.zeros(max(hist_cap, 1), dtype=torch.float32, device=device)
        self.hist_cap = hist_cap
        self.hyper = L.QcOptHyper(betas[0], betas[1], eps, max_norm, factor, threshold, min_lr, sched_eps,
                                  patience, 2.0, 4.0, 2.0)

    def read(self) -> dict:
        raw = self.state.cpu().numpy()
        ints = raw.view(np.int32)
        return {"lr": float(raw[0]), "best": float(raw[1]), "num_bad_epochs": int(ints[2]), "step": int(ints[3]),
                "hist_base": int(ints[9]),
                "loss": float(raw[4]), "grad_norm": float(raw[5]), "loss_res": float(raw[6]),
                "loss_bc": float(raw[7]), "loss_ic": float(raw[8])}

    def write(self, lr=None, best=None, num_bad=None, step=None, hist_base=None) -> None:
        raw = self.state.cpu().numpy().copy()
        ints = raw.view(np.int32)
        if lr is not None:
            raw[0] = lr
        if best is not None:
            raw[1] = best
        if num_bad is not None:
            ints[2] = num_bad
        if step is not None:
            ints[3] = step
        if hist_base is not None:
            ints[9] = hist_base         # the history buffer starts at this (absolute) step count
        self.state.copy_(torch.from_numpy(raw))

    def loss_history(self, steps: Optional[int] = None):
        """Losses of the first ``steps`` steps of THIS history buffer (default: all taken so far)."""
        if steps is None:
            rec = self.read()
            steps = rec["step"] - rec["hist_base"]
        return self.hist[: max(0, min(steps, self.hist_cap))].cpu().tolist()


class FusedStep:
    """One ``qc_fused_pinn_residual_step`` descriptor over fixed-size resident batches.  The caller
    fills ``X_res`` / ``X_val`` (IC points first, then BC points) and calls ``run``."""

    def __init__(self, eng: SolverEngine, B_res: int, n_ic: int, n_bc: int, opt: OptimState, counts=None):
        self.eng, self.opt = eng, opt
        dev, n = eng.device, eng.n
        self.B_res, self.n_ic, self.n_bc = B_res, n_ic, n_bc
        B_val = n_ic + n_bc
        f = dict(dtype=torch.float32, device=dev)
        self.X_res = torch.zeros(max(B_res, 1), 3, **f)
        self.X_val = torch.zeros(max(B_val, 1), 3, **f)
        self.ws_res = torch.empty(4, NCH, n, max(B_res, 1), **f)
        self.ws_val = torch.empty(4, 1, n, max(B_val, 1), **f)
        rows = (B_res + 63) // 64 + (B_val + 63) // 64
        self.stride = eng.NP + 3
        self.part = torch.zeros(rows, self.stride, **f)
        self.flat_grad = torch.zeros(self.stride, **f)
        # global point counts (differ from the local ones under data parallelism)
        g_res, g_ic, g_bc = counts if counts is not None else (B_res, n_ic, n_bc)
        c = eng.circuit
        d = L.QcStepDesc()
        d.prog, d.trig_dev, d.umat_dev = c.handle, c.trig.data_ptr(), _ptr(c.umat)
        d.H, d.n, d.n_theta = eng.H, n, eng.n_theta
        d.params_dev, d.m_dev, d.v_dev = eng.flat.data_ptr(), opt.m.data_ptr(), opt.v.data_ptr()
        d.opt_state_dev = opt.state.data_ptr()
        d.hist_dev, d.hist_cap = (opt.hist.data_ptr() if opt.hist_cap > 0 else None), opt.hist_cap
        d.X_res_dev, d.B_res = self.X_res.data_ptr(), B_res
        d.X_val_dev, d.B_val = self.X_val.data_ptr(), B_val
        (d.ajets_res_dev, d.qjets_res_dev, d.qbar_res_dev, d.abar_res_dev) = [self.ws_res[i].data_ptr() for i in range(4)]
        (d.ajets_val_dev, d.qjets_val_dev, d.qbar_val_dev, d.abar_val_dev) = [self.ws_val[i].data_ptr() for i in range(4)]
        d.part_dev, d.part_stride, d.part_rows_cap = self.part.data_ptr(), self.stride, rows
        d.flat_dev = self.flat_grad.data_ptr()
        d.pde = eng._pde(max(g_res, 1), max(g_ic, 1), max(g_bc, 1), n_ic)
        d.hyper = opt.hyper
        need = int(eng.lib.qc_step_workspace_bytes(c.handle, B_res, B_val))
        self.step_ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        d.circ_ws_dev, d.circ_ws_bytes = (self.step_ws.data_ptr(), need) if need else (None, 0)
        d.n_ic = n_ic
        d.comm = None
        d.sample_off_res = d.sample_off_ic = d.sample_off_bc = 0
        d.sample_seed, d.sample_step = 0, 0
        d.sample_bc_face_points = 0
        self.desc = d

    def set_sampler(self, seed: int, off_res: int = 0, off_ic: int = 0, off_bc: int = 0,
                    bc_face_points: int = 0) -> None:
        """On-device batches (QC_PHASE_SAMPLE): Philox stream `seed`; off_* = global index of this
        rank's first point in each batch (data parallelism); bc_face_points > 0 spreads the boundary
        batch over the four faces x=0, x=1, y=0, y=1 (that many GLOBAL points per face)."""
        d = self.desc
        d.sample_bc_face_points = bc_face_points
        d.sample_seed = seed & 0xFFFFFFFFFFFFFFFF
        d.sample_off_res, d.sample_off_ic, d.sample_off_bc = off_res, off_ic, off_bc

    def set_comm(self, comm) -> None:
        """A communicator of ``qc_comm_create`` (or None): GRADS | UPDATE in one call then all-reduces the flat
        [gradient | 3 loss sums] vector across the ranks inside the library (RCCL, same stream)."""
        self.desc.comm = comm

    def run(self, phases: int = L.QC_PHASE_GRADS | L.QC_PHASE_UPDATE) -> None:
        if phases & L.QC_PHASE_SAMPLE:
            self.desc.sample_step += 1          # a fresh counter block per step
        L.check(self.eng.lib.qc_fused_pinn_residual_step(C.byref(self.desc), phases, _stream(self.eng.device)),
                "qc_fused_pinn_residual_step")
