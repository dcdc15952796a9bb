"""``DVQuantumLayer`` — drop-in for the reference module of the same name
(nn/DVQuantumLayer.py:9-380) on its *simulator* branch: same constructor signature and ``args``
keys, same ``params`` Parameter ``(num_quantum_layers, P)`` float32 with xavier-normal init, same
Haar-seed gating, ``forward((B, n)) -> (n, B)``.

What differs is only where the arithmetic runs: instead of a PennyLane ``default.qubit`` QNode
(``diff_method="backprop"``) the circuit is lowered once to a gate program (``circuits.py``) and
evaluated by the HIP statevector kernels (``csrc/qc_circuit_*.hip``) through ``libqcpinn_hip.so``;
its backward is the HIP adjoint sweep.  There is no CPU path: a CPU input raises.

``encoding == "amplitude"`` selects AmplitudeEmbedding(normalize=True, pad_with=0.0) (:177-180): the n
features, zero-padded and L2-normalised, are the initial statevector; anything else is the RX angle
embedding (:182), as in the reference.

Derivatives: the reverse pass w.r.t. angles and parameters is the HIP adjoint sweep; with
``create_graph=True`` (generic ``diffusion_operator`` on an arbitrary model) the reverse pass is
re-expressed through the exact trigonometric-interpolation form of the layer, so second and higher
input derivatives work as with the reference's backprop simulator (angle encoding, n <= 7).

Out of scope, by design: IBM Runtime devices / shot-based execution
(``use_ibm_hardware=True``) are refused with an explicit error.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import circuits
from ..hip import engine as _engine
from ..hip.lib import QcError


# Largest register count for which higher-order input derivatives (create_graph=True) are provided:
# the trigonometric-interpolation form below has 3^n coefficients per output wire.
TRIG_INTERP_MAX_QUBITS = 7


def _trig_nodes(n: int, device) -> torch.Tensor:
    """(3^n, n) grid of embedding angles {0, 2pi/3, 4pi/3}^n, wire 0 the slowest index."""
    a = torch.tensor([0.0, 2.0 * torch.pi / 3.0, 4.0 * torch.pi / 3.0], dtype=torch.float32, device=device)
    return torch.cartesian_prod(*([a] * n)).reshape(-1, n) if n > 1 else a.reshape(-1, 1)


def _trig_finv(device) -> torch.Tensor:
    a = torch.tensor([0.0, 2.0 * torch.pi / 3.0, 4.0 * torch.pi / 3.0], dtype=torch.float64)
    F = torch.stack([torch.ones_like(a), torch.cos(a), torch.sin(a)], 1)        # F[j, k] = f_k(alpha_j)
    return torch.linalg.inv(F).to(device)                                        # (3, 3) float64


def _mode_products(T: torch.Tensor, M: torch.Tensor, n: int) -> torch.Tensor:
    """T (..., 3, 3, ..., 3) with n trailing axes of size 3: contracts every one of them with M[k, j]."""
    for ax in range(T.dim() - n, T.dim()):
        T = torch.movedim(torch.tensordot(T, M, dims=([ax], [1])), -1, ax)
    return T


class _TrigCoeffFn(torch.autograd.Function):
    """theta (L, P) -> C (n, 3^n): coefficients of <Z_v>(a) = sum_k C[v, k] prod_w f_{k_w}(a_w) with
    f = (1, cos, sin).  Every embedding angle enters through ONE RX gate, so each <Z_v> is a degree-1
    trigonometric polynomial in each a_w: evaluating the circuit (HIP statevector kernels) on the 3^n grid
    {0, 2pi/3, 4pi/3}^n and inverting the 3x3 node matrix per wire gives the coefficients exactly.  The
    reverse pass is the HIP adjoint sweep on the same grid (first order in theta)."""

    @staticmethod
    def forward(ctx, params, layer, device):
        circ = layer._circuit_for(device)
        n = layer.num_qubits
        theta = params.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        nodes = _trig_nodes(n, device).t().contiguous()                      # (n, 3^n)
        circ.prepare(theta)
        E = circ.forward_expval(nodes)                                           # (n, 3^n)
        finv = _trig_finv(device)
        C = _mode_products(E.to(torch.float64).reshape((n,) + (3,) * n), finv, n)
        ctx.circ, ctx.n = circ, n
        ctx.save_for_backward(nodes, theta, finv)
        ctx.pshape, ctx.pdev = params.shape, params.device
        return C.reshape(n, -1).to(torch.float32)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gC):
        nodes, theta, finv = ctx.saved_tensors
        n = ctx.n
        gE = _mode_products(gC.to(torch.float64).reshape((n,) + (3,) * n), finv.t().contiguous(), n)
        ctx.circ.prepare(theta)
        _, d_theta = ctx.circ.backward_expval(nodes, gE.reshape(n, -1).to(torch.float32).contiguous())
        return d_theta.reshape(ctx.pshape).to(ctx.pdev), None, None


def _trig_features(x: torch.Tensor) -> torch.Tensor:
    """(B, n) angles -> (B, 3^n) products of (1, cos a_w, sin a_w), wire 0 the slowest index."""
    phi = torch.ones(x.shape[0], 1, dtype=x.dtype, device=x.device)
    for w in range(x.shape[1]):
        f = torch.stack([torch.ones_like(x[:, w]), torch.cos(x[:, w]), torch.sin(x[:, w])], 1)
        phi = (phi[:, :, None] * f[:, None, :]).reshape(x.shape[0], -1)
    return phi


class _LayerVjpFn(torch.autograd.Function):
    """(x, theta, gq) -> (gx, gtheta): the reverse pass of the layer as a function that is differentiable ONCE more.
    Serves ``create_graph=True`` where the trigonometric interpolant does not reach (num_qubits > 7, amplitude
    encoding): the second-order quantities come from the derivative-channel kernels.  With a cotangent cx of gx,
        s = sum(gx * cx) = sum_p gq_p . (J(a_p) cx_p)
    is the first-derivative channel of the circuit along the direction cx, so ONE forward / reverse pass of the jet
    kernels with (a, da = cx) and the cotangent gq on that channel returns ds/da (the Hessian-vector product),
    ds/dgq = J cx (the channel itself) and ds/dtheta (reference use: nn/pde.py:59-70 on a user-composed model).
    Third-order derivatives (a loss.backward() THROUGH second input derivatives) are not available on this path."""

    @staticmethod
    def forward(ctx, x, params, gq, layer):
        circ = layer._circuit_for(x.device)
        angles = x.detach().to(torch.float32).t().contiguous()
        theta = params.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        circ.prepare(theta)
        d_angles, d_theta = circ.backward_expval(angles, gq.detach().to(torch.float32).contiguous())
        ctx.circ, ctx.layer = circ, layer
        ctx.save_for_backward(x, params, gq)
        return d_angles.t().to(x.dtype), d_theta.reshape(params.shape).to(params.device)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, cx, cp):
        x, params, gq = ctx.saved_tensors
        circ = ctx.circ
        if cp is not None and bool((cp != 0).any()):
            raise NotImplementedError("second derivatives with respect to the circuit parameters (theta-Hessians) are not "
                                      "provided by DVQuantumLayer")
        n, B = x.shape[1], x.shape[0]
        theta = params.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        circ.prepare(theta)
        aj = torch.zeros(6, n, B, dtype=torch.float32, device=x.device)
        aj[0] = x.detach().to(torch.float32).t()
        if cx is not None:
            aj[1] = cx.detach().to(torch.float32).t()
        qj = circ.forward_jets(aj)                       # qj[1] = J cx
        w = torch.zeros_like(aj)
        w[1] = gq.detach().to(torch.float32)
        abar, d_theta = circ.backward_jets(aj, w)        # abar[0] = d s / d a, d_theta = d s / d theta
        return abar[0].t().to(x.dtype), d_theta.reshape(params.shape).to(params.device), qj[1].to(gq.dtype), None


class _ExpvalFn(torch.autograd.Function):
    """angles (B, n), theta (L, P) -> <Z> (n, B).  Forward and the ordinary reverse pass are the HIP
    statevector kernels.  Under ``create_graph=True`` (the reference's nn/pde.py:59-70 usage on an
    arbitrary model) the reverse pass is rebuilt from differentiable torch ops on the exact
    trigonometric-interpolation form of the same function (coefficients from the HIP kernels), so input
    derivatives of any order exist; theta stays first order."""

    @staticmethod
    def forward(ctx, x, params, layer):
        circ = layer._circuit_for(x.device)
        angles = x.detach().to(torch.float32).t().contiguous()
        theta = params.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        circ.prepare(theta)
        out = circ.forward_expval(angles)
        ctx.circ, ctx.layer = circ, layer
        ctx.save_for_backward(x, params)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, params = ctx.saved_tensors
        circ, layer = ctx.circ, ctx.layer
        if torch.is_grad_enabled():          # create_graph=True: the result must itself be differentiable
            if layer.encoding == "amplitude" or layer.num_qubits > TRIG_INTERP_MAX_QUBITS:
                # second-order input derivatives from the derivative-channel kernels (any size, both encodings); the
                # interpolant below additionally carries third and higher orders
                gx, gp = _LayerVjpFn.apply(x, params, grad_out, layer)
                return (gx if x.requires_grad else None), (gp if params.requires_grad else None), None
            q = layer.trig_interpolant(x, params)
            wanted = [t for t in (x, params) if t.requires_grad]
            got = list(torch.autograd.grad(q, wanted, grad_out.to(q.dtype), create_graph=True, allow_unused=True))
            gx = got.pop(0) if x.requires_grad else None
            gp = got.pop(0) if params.requires_grad else None
            return gx, gp, None
        angles = x.detach().to(torch.float32).t().contiguous()
        theta = params.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        circ.prepare(theta)          # another forward may have re-used the trig table since
        d_angles, d_theta = circ.backward_expval(angles, grad_out.to(torch.float32).contiguous())
        return d_angles.t().to(x.dtype), d_theta.reshape(params.shape).to(params.device), None


class DVQuantumLayer(nn.Module):
    def __init__(self, args, diff_method="parameter-shift"):
        super().__init__()
        self.num_qubits = args["num_qubits"]
        self.num_quantum_layers = args["num_quantum_layers"]
        self.shots = args.get("shots", 1024)
        self.q_ansatz = args["q_ansatz"]
        self.problem = args["problem"]
        self.encoding = args.get("encoding", "angle")
        self.use_ibm_hardware = args.get("use_ibm_hardware", False)
        self.ibm_token = args.get("ibm_token", None)
        self.ibm_backend = args.get("ibm_backend", "ibmq_qasm_simulator")
        self.ibm_instance = args.get("ibm_instance", None)
        self.diff_method = diff_method          # accepted and ignored, as in the reference (:10,:143-145)

        per_layer = circuits.params_per_layer(self.q_ansatz, self.num_qubits)   # ValueError on unknown ansatz
        self.params = nn.Parameter(torch.empty(self.num_quantum_layers, per_layer, dtype=torch.float32))
        torch.nn.init.xavier_normal_(self.params)

        seed = args.get("seed", None) if self.num_qubits >= 4 else None
        self.haar_seed1 = seed
        self.haar_seed2 = seed + 1 if seed is not None else None

        if self.use_ibm_hardware:
            raise NotImplementedError(
                "use_ibm_hardware=True selects the IBM Runtime / shot-based branch of the reference, which is "
                "outside the MI355X simulator path; set use_ibm_hardware=False (the analytic simulator branch)")
        self.use_batch_processing = True
        self.dev = "hip.statevector"
        # lowered once; raises the same IndexError the reference hits for over-indexed ansaetze
        self.program = circuits.build_program(self.q_ansatz, self.num_qubits, self.num_quantum_layers,
                                              self.haar_seed1 is not None)
        self._haar = circuits.haar_unitaries(self.haar_seed1, self.haar_seed2)
        self._circuits = {}

    def _circuit_for(self, device) -> "_engine.Circuit":
        device = torch.device(device)
        if device.type != "cuda":
            raise QcError("DVQuantumLayer runs on the GPU only (HIP kernels, no CPU fallback): "
                          f"got an input on {device}")
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        if key not in self._circuits:
            self._circuits[key] = _engine.Circuit(self.program, self._haar, torch.device(*key),
                                                  amplitude=(self.encoding == "amplitude"))
        return self._circuits[key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 2 or x.shape[1] != self.num_qubits:
            raise ValueError(f"expected angles of shape (B, {self.num_qubits}), got {tuple(x.shape)}")
        return _ExpvalFn.apply(x, self.params, self)

    def trig_interpolant(self, x: torch.Tensor, params: torch.Tensor = None) -> torch.Tensor:
        """The same (n, B) expectation values as ``forward`` written as a trigonometric polynomial of the
        embedding angles (coefficients from the HIP kernels, see ``_TrigCoeffFn``): built from torch ops, so
        derivatives with respect to ``x`` exist to any order.  Used for the create_graph=True reverse pass."""
        params = self.params if params is None else params
        C = _TrigCoeffFn.apply(params, self, x.device)                       # (n, 3^n)
        return C @ _trig_features(x.to(torch.float32)).t()

    def circuit(self, x):
        """The reference exposes the QNode as ``.circuit``; calling it returns the per-wire list."""
        out = self.forward(x if x.dim() == 2 else x[None, :])
        return [out[i] if x.dim() == 2 else out[i, 0] for i in range(self.num_qubits)]

    def describe(self) -> str:
        names = circuits.OP_NAMES
        lines = ["AmplitudeEmbedding(normalize, pad 0)" if self.encoding == "amplitude"
                 else f"AngleEmbedding RX on wires 0..{self.num_qubits - 1}"]
        for g in self.program.gates:
            w = f"[{g.a}]" if g.b < 0 else f"[{g.a},{g.b}]"
            lines.append(f"{names[g.op]}{w}" + (f" p{g.slot}" if g.op in circuits.PARAMETRIC else ""))
        lines.append("measure <Z> on every wire")
        return " | ".join(lines)
