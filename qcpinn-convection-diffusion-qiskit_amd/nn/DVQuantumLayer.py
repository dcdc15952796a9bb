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

Out of scope, by design: IBM Runtime devices / shot-based execution
(``use_ibm_hardware=True``) are refused with an explicit error.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import circuits
from ..hip import engine as _engine
from ..hip.lib import QcError


class _ExpvalFn(torch.autograd.Function):
    """angles (B, n), theta (L, P) -> <Z> (n, B); first-order differentiable in both."""

    @staticmethod
    def forward(ctx, x, params, layer):
        circ = layer._circuit_for(x.device)
        angles = x.detach().to(torch.float32).t().contiguous()
        theta = params.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        circ.prepare(theta)
        out = circ.forward_expval(angles)
        ctx.circ = circ
        ctx.save_for_backward(angles, theta)
        ctx.pshape = params.shape
        ctx.pdev = params.device
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        angles, theta = ctx.saved_tensors
        circ = ctx.circ
        circ.prepare(theta)          # another forward may have re-used the trig table since
        d_angles, d_theta = circ.backward_expval(angles, grad_out.to(torch.float32).contiguous())
        return d_angles.t(), d_theta.reshape(ctx.pshape).to(ctx.pdev), None


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
