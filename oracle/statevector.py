"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).

Restates, with torch complex128 tensors on the CPU, the arithmetic that the reference
delegates to PennyLane ``default.qubit`` (torch interface, ``diff_method="backprop"``) for the
DV quantum layer: ``nn/DVQuantumLayer.py:176-214`` (circuit order) and ``:246-371`` (ansatz
builders).  Every op is a differentiable torch op, so torch autograd (including
``create_graph=True`` double backward, as ``nn/pde.py:59-70`` uses) runs through it — the same
algorithm class as the reference's backprop simulator.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product path (``qcpinn-convection-diffusion-qiskit_amd``) never does.

PARITY UNPINNED: the arithmetic lives in PennyLane (third-party, unpinned core version; the
plugin pin ``pennylane-qiskit==0.44.1`` in ``requirements-dev.txt:1`` implies a 0.44-series core),
which is absent from this image, and the reference holds no tests, golden vectors or fixtures
for this path.  The conventions below are PennyLane's published definitions:
  * wire 0 is the most significant bit of the amplitude index; the initial state is |0...0>;
  * RX(t)=exp(-i t X/2), RY(t)=exp(-i t Y/2), RZ(t)=diag(e^{-it/2}, e^{+it/2});
  * CRX/CRZ(t, wires=[c, t]) = |0><0| (x) I + |1><1| (x) R(t);  CNOT(wires=[c, t]);
  * QubitUnitary(U, wires=[a, b]) indexes its 4x4 with ``a`` as the more significant bit;
  * AngleEmbedding(x, rotation="X") = RX(x[:, i]) on wire i, broadcast over the batch;
  * expval(PauliZ(i)) = sum_k |psi_k|^2 (1 - 2 bit_i(k)).
They are cross-checked by an independent dense-matrix implementation (``dense_unitary.py``) and
closed-form known answers in ``tests/test_oracle.py``.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

CDT = torch.complex128
RDT = torch.float64


def _as_c(x):
    return x.to(CDT) if torch.is_tensor(x) else torch.tensor(x, dtype=CDT)


def _mat_rx(theta):
    c = torch.cos(theta / 2).to(CDT)
    s = torch.sin(theta / 2).to(CDT)
    return torch.stack([torch.stack([c, -1j * s], -1), torch.stack([-1j * s, c], -1)], -2)


def _mat_ry(theta):
    c = torch.cos(theta / 2).to(CDT)
    s = torch.sin(theta / 2).to(CDT)
    return torch.stack([torch.stack([c, -s], -1), torch.stack([s, c], -1)], -2)


def _mat_rz(theta):
    e = torch.exp(-0.5j * theta.to(CDT))
    z = torch.zeros_like(e)
    return torch.stack([torch.stack([e, z], -1), torch.stack([z, e.conj()], -1)], -2)


_H = torch.tensor([[1, 1], [1, -1]], dtype=CDT) / math.sqrt(2.0)


class Simulator:
    """Batched statevector on ``n`` wires; ``state`` has shape (B, 2, 2, ..., 2), axis 1+w = wire w."""

    def __init__(self, n: int, batch: int):
        self.n = n
        st = torch.zeros((batch,) + (2,) * n, dtype=CDT)
        st[(slice(None),) + (0,) * n] = 1.0
        self.state = st

    # -- single-wire gate: ``mat`` is (2,2) or (B,2,2)
    def apply1(self, mat, w: int):
        st = torch.movedim(self.state, 1 + w, -1)                # (..., 2)
        if mat.dim() == 2:
            st = torch.einsum("ij,...j->...i", mat, st)
        else:
            shape = (mat.shape[0],) + (1,) * (self.n - 1) + (2, 2)
            st = (mat.reshape(shape) * st.unsqueeze(-2)).sum(-1)
        self.state = torch.movedim(st, -1, 1 + w)

    # -- two-wire gate on [a, b], ``a`` the more significant index of the 4x4
    def apply2(self, mat4, a: int, b: int):
        st = torch.movedim(self.state, (1 + a, 1 + b), (-2, -1))
        shp = st.shape
        st = st.reshape(shp[:-2] + (4,))
        st = torch.einsum("ij,...j->...i", mat4, st)
        self.state = torch.movedim(st.reshape(shp), (-2, -1), (1 + a, 1 + b))

    def controlled(self, mat2, c: int, t: int):
        """|0><0| (x) I + |1><1| (x) mat2 on wires [c, t]."""
        p0 = torch.tensor([[1, 0], [0, 0]], dtype=CDT)
        p1 = torch.tensor([[0, 0], [0, 1]], dtype=CDT)
        self.apply2(torch.kron(p0, torch.eye(2, dtype=CDT)) + torch.kron(p1, mat2), c, t)

    def RX(self, th, w): self.apply1(_mat_rx(th), w)
    def RY(self, th, w): self.apply1(_mat_ry(th), w)
    def RZ(self, th, w): self.apply1(_mat_rz(th), w)
    def H(self, w): self.apply1(_H, w)
    def CRX(self, th, c, t): self.controlled(_mat_rx(th), c, t)
    def CRZ(self, th, c, t): self.controlled(_mat_rz(th), c, t)

    def CNOT(self, c, t):
        x = torch.tensor([[0, 1], [1, 0]], dtype=CDT)
        self.controlled(x, c, t)

    def expval_z(self, w: int):
        prob = (self.state.real ** 2 + self.state.imag ** 2)
        prob = torch.movedim(prob, 1 + w, -1).reshape(prob.shape[0], -1, 2).sum(1)
        return prob[:, 0] - prob[:, 1]


# ---------------------------------------------------------------- ansatz builders
# each follows the reference builder of the same name, gate by gate.

def layered(sim: Simulator, p):                  # nn/DVQuantumLayer.py:246-262
    n = sim.n
    assert p is not None and len(p) == n * 4
    k = 0
    for q in range(n):
        sim.RZ(p[k], q); k += 1
        sim.RX(p[k], q); k += 1
    for q in range(n):
        sim.CNOT(q, (q + 1) % n)
    for q in range(n):
        sim.RX(p[k], q); k += 1
        sim.RZ(p[k], q); k += 1


def alternate(sim: Simulator, p):                # nn/DVQuantumLayer.py:264-285
    n = sim.n
    assert p is not None and len(p) == n * 4 - 4
    k = 0

    def tdcnot(c, t):
        nonlocal k
        sim.RY(p[k], c); k += 1
        sim.RY(p[k], t); k += 1
        sim.CNOT(c, t)
        sim.RZ(p[k], c); k += 1
        sim.RZ(p[k], t); k += 1

    for i in range(n - 1)[::2]:
        tdcnot(i, (i + 1) % n)
    for i in range(n)[1::2]:
        tdcnot(i, (i + 1) % n)


def cascade(sim: Simulator, p):                  # nn/DVQuantumLayer.py:287-305
    n = sim.n
    k = 0
    for q in range(n):
        sim.RX(p[k], q); k += 1
    for q in range(n):
        sim.RZ(p[k], q); k += 1
    sim.CRX(p[k], n - 1, 0); k += 1
    for q in reversed(range(1, n)):
        sim.CRX(p[k], q - 1, q); k += 1


def farhi(sim: Simulator, p):                    # nn/DVQuantumLayer.py:307-324
    n = sim.n
    if len(p) != 2 * n - 2:
        raise ValueError("Insufficient parameters for RXX and RZX gates")
    k = 0
    for q in range(n - 1):
        sim.CNOT(n - 1, q); sim.RX(p[k], n - 1); sim.CNOT(n - 1, q); k += 1
    for q in range(n - 1):
        sim.CNOT(n - 1, q); sim.RZ(p[k], n - 1); sim.CNOT(n - 1, q); k += 1


def sim_circ_15(sim: Simulator, p):              # nn/DVQuantumLayer.py:326-346
    n = sim.n
    if p is None or len(p) != 2 * n:
        raise ValueError("Insufficient parameters for RXX and RZX gates")
    k = 0
    for q in range(n):
        sim.RY(p[k], q); k += 1
    for q in reversed(range(n)):
        sim.CNOT(q, (q + 1) % n)
    for q in range(n):
        sim.RY(p[k], q); k += 1
    for q in range(n):
        c = (q + n - 1) % n
        sim.CNOT(c, (c + 3) % n)


def cross_mesh(sim: Simulator, p):               # nn/DVQuantumLayer.py:348-371
    n = sim.n
    k = 0
    for q in range(n):
        sim.RX(p[k], q); k += 1
    for q in range(n):
        sim.RZ(p[k], q); k += 1
    for c in range(n - 1, -1, -1):
        for t in range(n - 1, -1, -1):
            if t != c:
                sim.CRZ(p[k], c, t); k += 1
    for q in range(n):
        sim.RX(p[k], q); k += 1
    for q in range(n):
        sim.RZ(p[k], q); k += 1


ANSATZ = {"layered": layered, "alternate": alternate, "cascade": cascade,
          "farhi": farhi, "sim_circ_15": sim_circ_15, "cross_mesh": cross_mesh}


def haar_pair(seed1: Optional[int], seed2: Optional[int]):
    """nn/DVQuantumLayer.py:203-207 — drawn afresh on every circuit call in the reference."""
    if seed1 is None or seed2 is None:
        return None
    from scipy.stats import unitary_group
    u1 = unitary_group.rvs(4, random_state=np.random.RandomState(seed1))
    u2 = unitary_group.rvs(4, random_state=np.random.RandomState(seed2))
    return torch.tensor(u1, dtype=CDT), torch.tensor(u2, dtype=CDT)


def circuit_expvals(x: torch.Tensor, params: torch.Tensor, q_ansatz: str, n: int,
                    haar=None, encoding: str = "angle") -> torch.Tensor:
    """``DVQuantumLayer.forward`` batch branch (nn/DVQuantumLayer.py:151-154,176-214).

    x: (B, n) real angles; params: (L, P) real.  Returns (n, B) float64 — the ``torch.stack`` of
    the per-wire expectation values, i.e. the layout the reference's QNode hands back."""
    x = x.to(RDT)
    params = params.to(RDT)
    sim = Simulator(n, x.shape[0])
    if encoding == "amplitude":                              # AmplitudeEmbedding(normalize=True, pad_with=0.0)  (:177-180)
        padded = torch.zeros(x.shape[0], 1 << n, dtype=RDT)
        padded = torch.cat([x, padded[:, x.shape[1]:]], dim=1)
        padded = padded / padded.norm(dim=1, keepdim=True)
        sim.state = padded.to(CDT).reshape((x.shape[0],) + (2,) * n)
    else:
        for w in range(n):                                   # AngleEmbedding, rotation="X"  (:182)
            sim.RX(x[:, w], w)
    for layer in range(params.shape[0]):                     # :184-201
        ANSATZ[q_ansatz](sim, params[layer])
    if haar is not None:                                     # :203-209
        sim.apply2(haar[0], 0, 1)
        sim.apply2(haar[1], 2, 3)
    if n > 0:                                                # :211-212
        sim.H(n - 1)
    return torch.stack([sim.expval_z(w) for w in range(n)])  # :214
