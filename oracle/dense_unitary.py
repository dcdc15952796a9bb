"""CPU ORACLE, second opinion — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see statevector.py).

An independent numpy restatement of the same circuit (``nn/DVQuantumLayer.py:176-214,246-371``)
that shares no code with ``statevector.py``: every gate is expanded into a dense 2^n x 2^n
matrix by explicit bit arithmetic on basis-state indices and multiplied onto the state.
Used to cross-check wire order, control/target direction and the two-wire-unitary convention
for n <= 8.  Takes the *gate program* of the product package as input as well, so the IR
lowering (``circuits.build_program``) is checked against the hand-written builders here.
"""
from __future__ import annotations

import numpy as np

SQ2 = 1.0 / np.sqrt(2.0)


def _bit(k: int, wire: int, n: int) -> int:
    return (k >> (n - 1 - wire)) & 1


def _flip(k: int, wire: int, n: int) -> int:
    return k ^ (1 << (n - 1 - wire))


def one_wire(m2: np.ndarray, wire: int, n: int) -> np.ndarray:
    N = 1 << n
    U = np.zeros((N, N), dtype=np.complex128)
    for col in range(N):
        b = _bit(col, wire, n)
        for out in (0, 1):
            row = col if out == b else _flip(col, wire, n)
            U[row, col] += m2[out, b]
    return U


def controlled(m2: np.ndarray, ctl: int, tgt: int, n: int) -> np.ndarray:
    N = 1 << n
    U = np.zeros((N, N), dtype=np.complex128)
    for col in range(N):
        if _bit(col, ctl, n) == 0:
            U[col, col] = 1.0
            continue
        b = _bit(col, tgt, n)
        for out in (0, 1):
            row = col if out == b else _flip(col, tgt, n)
            U[row, col] += m2[out, b]
    return U


def two_wire(m4: np.ndarray, a: int, b: int, n: int) -> np.ndarray:
    N = 1 << n
    U = np.zeros((N, N), dtype=np.complex128)
    for col in range(N):
        cin = 2 * _bit(col, a, n) + _bit(col, b, n)
        base = col & ~((1 << (n - 1 - a)) | (1 << (n - 1 - b)))
        for rout in range(4):
            row = base | ((rout >> 1) << (n - 1 - a)) | ((rout & 1) << (n - 1 - b))
            U[row, col] += m4[rout, cin]
    return U


def rx(t):
    c, s = np.cos(t / 2), np.sin(t / 2)
    return np.array([[c, -1j * s], [-1j * s, c]])


def ry(t):
    c, s = np.cos(t / 2), np.sin(t / 2)
    return np.array([[c, -s], [s, c]], dtype=np.complex128)


def rz(t):
    return np.array([[np.exp(-0.5j * t), 0], [0, np.exp(0.5j * t)]])


HAD = np.array([[1, 1], [1, -1]], dtype=np.complex128) * SQ2
PX = np.array([[0, 1], [1, 0]], dtype=np.complex128)

# opcode numbering of the product IR (circuits.py / csrc/qc_gates.h), restated, not imported
RX_, RY_, RZ_, H_, CNOT_, CRX_, CRZ_, U4_ = range(8)


def program_unitary(rows: np.ndarray, params: np.ndarray, haar, n: int) -> np.ndarray:
    """Dense unitary of a gate program given as (n_gates, 4) rows (op, a, b, slot)."""
    flat = np.asarray(params, dtype=np.float64).reshape(-1)
    V = np.eye(1 << n, dtype=np.complex128)
    for op, a, b, slot in np.asarray(rows).tolist():
        if op == RX_:
            G = one_wire(rx(flat[slot]), a, n)
        elif op == RY_:
            G = one_wire(ry(flat[slot]), a, n)
        elif op == RZ_:
            G = one_wire(rz(flat[slot]), a, n)
        elif op == H_:
            G = one_wire(HAD, a, n)
        elif op == CNOT_:
            G = controlled(PX, a, b, n)
        elif op == CRX_:
            G = controlled(rx(flat[slot]), a, b, n)
        elif op == CRZ_:
            G = controlled(rz(flat[slot]), a, b, n)
        elif op == U4_:
            G = two_wire(np.asarray(haar[slot]), a, b, n)
        else:
            raise ValueError(f"opcode {op}")
        V = G @ V
    return V


def embed(x_row: np.ndarray, n: int) -> np.ndarray:
    """RX(x_i) on wire i applied to |0...0>."""
    psi = np.zeros(1 << n, dtype=np.complex128)
    psi[0] = 1.0
    for w in range(n):
        psi = one_wire(rx(float(x_row[w])), w, n) @ psi
    return psi


def expvals(psi: np.ndarray, n: int) -> np.ndarray:
    p = np.abs(psi) ** 2
    out = np.zeros(n)
    for w in range(n):
        sign = np.array([1 - 2 * _bit(k, w, n) for k in range(1 << n)])
        out[w] = float((p * sign).sum())
    return out


def program_expvals(rows, x: np.ndarray, params, haar, n: int) -> np.ndarray:
    """(n, B) expectation values of the gate program on angle-embedded inputs x (B, n)."""
    V = program_unitary(rows, params, haar, n)
    return np.stack([expvals(V @ embed(x[b], n), n) for b in range(x.shape[0])], axis=1)
