"""Gate-program IR for the DV variational circuits.

A circuit is lowered ONCE on the host into a flat list of gate records
``(opcode, wire_a, wire_b, slot)`` that the HIP kernels interpret on device
(``csrc/qc_gates.h``).  The embedding (``RX(x_i)`` on wire ``i``) is not part
of the program: the kernels build the embedded product state, and its
derivative channels, directly in registers.

What each ansatz lowers to follows the gate order of the reference:
``nn/DVQuantumLayer.py:176-214`` (circuit order: ansatz layers, the two fixed
4x4 unitaries on wires [0,1] and [2,3], Hadamard on the last wire) and
``:246-371`` (the six ansatz builders).  Conventions (PennyLane ``default.qubit``):
wire 0 is the most significant bit of the amplitude index; controlled gates
take ``[control, target]``; a two-wire unitary on ``[a, b]`` uses ``a`` as the
more significant bit of its 4x4 index.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

# opcodes -- must match csrc/qc_gates.h
OP_RX, OP_RY, OP_RZ, OP_H, OP_CNOT, OP_CRX, OP_CRZ, OP_U4 = range(8)
OP_NAMES = ("RX", "RY", "RZ", "H", "CNOT", "CRX", "CRZ", "U4")
PARAMETRIC = (OP_RX, OP_RY, OP_RZ, OP_CRX, OP_CRZ)

ANSATZ_NAMES = ("layered", "alternate", "cascade", "farhi", "sim_circ_15", "cross_mesh")


def params_per_layer(q_ansatz: str, n: int) -> int:
    """Trainable angles per layer, reference ``nn/DVQuantumLayer.py:25-78``."""
    table = {
        "layered": 4 * n,
        "alternate": 4 * n - 4,
        "cascade": 3 * n,
        "farhi": 2 * n - 2,
        "sim_circ_15": 2 * n,
        "cross_mesh": 4 * n + n * (n - 1),
    }
    if q_ansatz not in table:
        # same failure the reference raises at :82-85
        raise ValueError("Parameters are not initialized. Check the q_ansatz value.")
    return table[q_ansatz]


@dataclass(frozen=True)
class Gate:
    op: int
    a: int          # target wire (1q gates), control wire (controlled), first wire (U4)
    b: int          # target wire (controlled), second wire (U4), -1 otherwise
    slot: int       # flat parameter index (layer*P + k) for parametric gates, U4 slot, else -1

    def as_row(self) -> Tuple[int, int, int, int]:
        return (self.op, self.a, self.b, self.slot)


class _Emitter:
    def __init__(self, n: int, base: int, count: int):
        self.n, self.base, self.count, self.k = n, base, count, 0
        self.gates: List[Gate] = []

    def _next(self) -> int:
        if self.k >= self.count:
            # the reference indexes a length-P tensor; running past it is an IndexError there too
            raise IndexError(
                f"index {self.k} is out of bounds for dimension 0 with size {self.count}")
        slot = self.base + self.k
        self.k += 1
        return slot

    def _wire(self, *ws):
        for w in ws:
            if not 0 <= w < self.n:
                raise ValueError(f"wire {w} outside 0..{self.n - 1}")
        if len(set(ws)) != len(ws):
            raise ValueError(f"a gate needs distinct wires, got {list(ws)}")

    def rot(self, op: int, w: int):
        self._wire(w)
        self.gates.append(Gate(op, w, -1, self._next()))

    def crot(self, op: int, c: int, t: int):
        self._wire(c, t)
        self.gates.append(Gate(op, c, t, self._next()))

    def cnot(self, c: int, t: int):
        self._wire(c, t)
        self.gates.append(Gate(OP_CNOT, c, t, -1))


def _layered(e: _Emitter):                      # reference :246-262
    n = e.n
    for q in range(n):
        e.rot(OP_RZ, q)
        e.rot(OP_RX, q)
    for q in range(n):
        e.cnot(q, (q + 1) % n)
    for q in range(n):
        e.rot(OP_RX, q)
        e.rot(OP_RZ, q)


def _alternate(e: _Emitter):                    # reference :264-285 (over-indexes for even n)
    n = e.n

    def block(c, t):
        e.rot(OP_RY, c)
        e.rot(OP_RY, t)
        e.cnot(c, t)
        e.rot(OP_RZ, c)
        e.rot(OP_RZ, t)

    for i in range(0, n - 1, 2):
        block(i, (i + 1) % n)
    for i in range(1, n, 2):
        block(i, (i + 1) % n)


def _cascade(e: _Emitter):                      # reference :287-305
    n = e.n
    for q in range(n):
        e.rot(OP_RX, q)
    for q in range(n):
        e.rot(OP_RZ, q)
    e.crot(OP_CRX, n - 1, 0)
    for q in range(n - 1, 0, -1):
        e.crot(OP_CRX, q - 1, q)


def _farhi(e: _Emitter):                        # reference :307-324 (CNOT-conjugated rotations)
    n = e.n
    for op in (OP_RX, OP_RZ):
        for q in range(n - 1):
            e.cnot(n - 1, q)
            e.rot(op, n - 1)
            e.cnot(n - 1, q)


def _sim_circ_15(e: _Emitter):                  # reference :326-346
    n = e.n
    for q in range(n):
        e.rot(OP_RY, q)
    for q in range(n - 1, -1, -1):
        e.cnot(q, (q + 1) % n)
    for q in range(n):
        e.rot(OP_RY, q)
    for q in range(n):
        c = (q + n - 1) % n
        e.cnot(c, (c + 3) % n)


def _cross_mesh(e: _Emitter):                   # reference :348-371
    n = e.n
    for q in range(n):
        e.rot(OP_RX, q)
    for q in range(n):
        e.rot(OP_RZ, q)
    for c in range(n - 1, -1, -1):
        for t in range(n - 1, -1, -1):
            if t != c:
                e.crot(OP_CRZ, c, t)
    for q in range(n):
        e.rot(OP_RX, q)
    for q in range(n):
        e.rot(OP_RZ, q)


_BUILDERS = {
    "layered": _layered, "alternate": _alternate, "cascade": _cascade,
    "farhi": _farhi, "sim_circ_15": _sim_circ_15, "cross_mesh": _cross_mesh,
}


@dataclass
class GateProgram:
    """Flat, device-ready description of one variational circuit (embedding excluded)."""
    n_qubits: int
    n_layers: int
    q_ansatz: str
    n_params: int                 # total trainable angles = n_layers * params_per_layer
    use_haar: bool
    gates: List[Gate]

    def rows(self) -> np.ndarray:
        """(n_gates, 4) int32 array handed to ``qc_program_create``."""
        return np.asarray([g.as_row() for g in self.gates], dtype=np.int32).reshape(-1, 4)

    @property
    def n_gates(self) -> int:
        return len(self.gates)

    def algorithmic_flops(self) -> int:
        """Real flops of one circuit evaluation under SURVEY.md §8(d)'s per-gate model
        (FMA = 2): embedding RX 6N each, RX/RY/RZ 6N, CRX/CRZ 3N, CNOT 0, H 4N,
        4x4 two-wire unitary 30N, <Z> for all wires (3+n)N, N = 2**n."""
        n, N = self.n_qubits, 1 << self.n_qubits
        per = {OP_RX: 6, OP_RY: 6, OP_RZ: 6, OP_H: 4, OP_CNOT: 0, OP_CRX: 3, OP_CRZ: 3, OP_U4: 30}
        return (6 * n + sum(per[g.op] for g in self.gates) + 3 + n) * N


def build_program(q_ansatz: str, n_qubits: int, n_layers: int, use_haar: bool) -> GateProgram:
    """Lower ``q_ansatz`` on ``n_qubits`` wires x ``n_layers`` into a gate program.

    ``use_haar`` mirrors the reference's gating (``nn/DVQuantumLayer.py:88-94,203-209``):
    the two fixed unitaries are present iff ``num_qubits >= 4`` and ``args['seed']`` is given.
    """
    P = params_per_layer(q_ansatz, n_qubits)
    gates: List[Gate] = []
    for layer in range(n_layers):
        e = _Emitter(n_qubits, layer * P, P)
        _BUILDERS[q_ansatz](e)
        gates += e.gates
    if use_haar:
        if n_qubits < 4:
            raise ValueError("the fixed two-wire unitaries act on wires 0-3: need num_qubits >= 4")
        gates.append(Gate(OP_U4, 0, 1, 0))
        gates.append(Gate(OP_U4, 2, 3, 1))
    if n_qubits > 0:
        gates.append(Gate(OP_H, n_qubits - 1, -1, -1))
    return GateProgram(n_qubits, n_layers, q_ansatz, n_layers * P, use_haar, gates)


def haar_unitaries(seed1: Optional[int], seed2: Optional[int]) -> Optional[np.ndarray]:
    """The two fixed 4x4 unitaries exactly as the reference draws them on every circuit call
    (``nn/DVQuantumLayer.py:203-207``): ``scipy.stats.unitary_group.rvs(4, RandomState(seed))``.
    Returns a (2,4,4) complex128 array, or None when the reference applies none."""
    if seed1 is None or seed2 is None:
        return None
    from scipy.stats import unitary_group
    u1 = unitary_group.rvs(4, random_state=np.random.RandomState(seed1))
    u2 = unitary_group.rvs(4, random_state=np.random.RandomState(seed2))
    return np.stack([u1, u2]).astype(np.complex128)
