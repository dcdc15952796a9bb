"""The compile-time programs of the lanes-as-amplitudes family (6 <= n <= 8 qubits) do NOT run the reference circuit
(nn/DVQuantumLayer.py:176-214, ansatz builders :246-371) in program order: RZ / CRZ gates are moved as far as
commutation allows and the ones that meet become one phase-table multiply (csrc/qc_wave_sched.h, evaluated at compile
time by the kernels and at program creation by the host).  This test re-executes the schedule record
(``qc_wave_sched_describe``, host-only) with numpy on random statevectors and compares with the gate-by-gate program,
and checks the bookkeeping the kernels rely on (every gate exactly once, runs hold diagonal gates only).  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from conftest import pkg
from test_hbm_plan import OP_CRZ, OP_RZ, OP_U4, apply_gate, program_gates


def describe(prog):
    L = pkg("hip.lib")
    lib = L.load()
    rows = np.ascontiguousarray(prog.rows())
    need = lib.qc_wave_sched_describe(rows.ctypes.data_as(C.c_void_p), prog.n_gates, prog.n_qubits, prog.n_params, None, 0)
    assert need > 0
    buf = np.zeros(need, dtype=np.int32)
    got = lib.qc_wave_sched_describe(rows.ctypes.data_as(C.c_void_p), prog.n_gates, prog.n_qubits, prog.n_params,
                                     buf.ctypes.data_as(C.c_void_p), need)
    assert got == need
    it = iter(buf.tolist())
    n_items, n_runs = next(it), next(it)
    items = [next(it) for _ in range(n_items)]
    runs = []
    for _ in range(n_runs):
        cnt = next(it)
        runs.append([next(it) for _ in range(cnt)])
    assert next(it, None) is None
    return items, runs


# the registered compile-time programs (csrc/gen_static.py: WAVE_PROGRAMS) and a few the schedule has not been tuned on
CASES = [("layered", 8, 2), ("cascade", 6, 1), ("cross_mesh", 8, 1), ("layered", 7, 1), ("layered", 6, 1), ("layered", 8, 1),
         ("cross_mesh", 6, 2), ("sim_circ_15", 6, 1), ("farhi", 7, 1), ("alternate", 7, 1), ("layered", 5, 3)]


@pytest.mark.parametrize("ansatz,n,layers", CASES)
def test_schedule_is_equivalent_to_the_program(ansatz, n, layers):
    circuits = pkg("circuits")
    prog = circuits.build_program(ansatz, n, layers, use_haar=True)
    items, runs = describe(prog)
    gates = program_gates(prog)
    # every gate exactly once: as an item of its own or as a member of exactly one run that is itself scheduled once
    seen = [g for g in items if g >= 0] + [g for r in runs for g in r]
    assert sorted(seen) == list(range(prog.n_gates))
    assert sorted(-i - 1 for i in items if i < 0) == list(range(len(runs)))
    for r in runs:
        assert len(r) >= 2 and all(gates[g][0] in (OP_RZ, OP_CRZ) for g in r)
    rng = np.random.default_rng(n * 11 + layers)
    theta = rng.uniform(-2, 2, prog.n_params)
    U = [np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))[0] for _ in range(2)]
    psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)

    def run(order):
        psi = psi0.copy()
        for g in order:
            op, tb, cb, slot = gates[g]
            apply_gate(psi, op, tb, cb, theta[slot] if op != OP_U4 and slot >= 0 else 0.0, U[slot] if op == OP_U4 else None, n)
        return psi

    ref = run(range(prog.n_gates))
    order = []
    for i in items:
        order += [i] if i >= 0 else runs[-i - 1]
    out = run(order)
    assert np.abs(out - ref).max() < 1e-12 * np.abs(ref).max()


def test_layered_and_cross_mesh_collapse_as_designed():
    """The two shapes the schedule was written for: 4n RZ gates of two layered layers -> 3 runs (n, 2n, n);
    the cross-mesh layer's n RZ + n(n-1) CRZ -> one run, its trailing n RZ another."""
    circuits = pkg("circuits")
    items, runs = describe(circuits.build_program("layered", 8, 2, use_haar=True))
    assert [len(r) for r in runs] == [8, 16, 8]
    items, runs = describe(circuits.build_program("cross_mesh", 8, 1, use_haar=True))
    assert [len(r) for r in runs] == [8 + 56, 8]
