"""The execution plan of the HBM-resident kernel family (9 <= n <= 20 qubits, csrc/qc_hbm2_plan.h) is NOT the
program order of the reference circuit (nn/DVQuantumLayer.py:176-214, ansatz builders :246-371): gates on disjoint
wires and diagonal gates are commuted so that, e.g., the 16-qubit cross_mesh circuit crosses HBM in two stages.
This test re-executes the plan record (``qc_hbm_plan_describe``, host-only) with numpy on random statevectors and
compares with the gate-by-gate program; it also checks the locality bookkeeping the kernels rely on (which index
bits are tile-local / register-resident for every gate).  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from conftest import pkg

OP_RX, OP_RY, OP_RZ, OP_H, OP_CNOT, OP_CRX, OP_CRZ, OP_U4 = range(8)
K_REG1, K_REG2, K_PRED, K_PHASE, K_U4 = range(5)


def describe(prog):
    L = pkg("hip.lib")
    lib = L.load()
    rows = np.ascontiguousarray(prog.rows())
    need = lib.qc_hbm_plan_describe(rows.ctypes.data_as(C.c_void_p), prog.n_gates, prog.n_qubits, prog.n_params, None, 0)
    assert need > 0
    buf = np.zeros(need, dtype=np.int32)
    got = lib.qc_hbm_plan_describe(rows.ctypes.data_as(C.c_void_p), prog.n_gates, prog.n_qubits, prog.n_params,
                                   buf.ctypes.data_as(C.c_void_p), need)
    assert got == need
    return parse(buf)


def parse(d):
    it = iter(d.tolist())
    nx = lambda: next(it)
    assert nx() == 0x48324832
    plan = {"n": nx(), "absorb": nx()}
    n_stages, n_tables = nx(), nx()
    plan["rbits"] = nx()
    plan["stages"], plan["tables"] = [], []
    for _ in range(n_stages):
        st = {"nloc": nx(), "ngb": nx()}
        st["lb"] = [nx() for _ in range(st["nloc"])]
        st["gb"] = [nx() for _ in range(st["ngb"])]
        nr, st["np"], st["ntab"] = nx(), nx(), nx()
        st["rounds"] = []
        for _ in range(nr):
            rd = {"kind": nx(), "nrb": nx()}
            rd["rb"] = [nx() for _ in range(rd["nrb"])]
            rd["table"], rd["tslot"] = nx(), nx()
            rd["tab_pre"], rd["ts_pre"], rd["tab_post"], rd["ts_post"] = nx(), nx(), nx(), nx()
            ng = nx()
            rd["gates"] = []
            for _ in range(ng):
                v = [nx() for _ in range(10)]
                rd["gates"].append(dict(zip(("op", "kind", "tq", "cq", "tbit", "cbit", "gi", "slot", "pidx"), v)))
            st["rounds"].append(rd)
        plan["stages"].append(st)
    for _ in range(n_tables):
        ng = nx()
        plan["tables"].append([dict(zip(("op", "bt", "bc", "gi", "slot"), [nx() for _ in range(5)])) for _ in range(ng)])
    assert next(it, None) is None
    return plan


def mat1(op, th):
    c, s = np.cos(th / 2), np.sin(th / 2)
    return {OP_RX: np.array([[c, -1j * s], [-1j * s, c]]), OP_RY: np.array([[c, -s], [s, c]], dtype=complex),
            OP_RZ: np.array([[c - 1j * s, 0], [0, c + 1j * s]]),
            OP_H: np.array([[1, 1], [1, -1]], dtype=complex) / np.sqrt(2)}[op]


def apply_1q(psi, m, tb, n, cb=-1):
    idx = np.arange(1 << n)
    lo = idx[(idx >> tb) & 1 == 0]
    if cb >= 0:
        lo = lo[(lo >> cb) & 1 == 1]
    hi = lo | (1 << tb)
    a, b = psi[lo].copy(), psi[hi].copy()
    psi[lo] = m[0, 0] * a + m[0, 1] * b
    psi[hi] = m[1, 0] * a + m[1, 1] * b


def apply_gate(psi, op, tbit, cbit, th, U, n):
    if op in (OP_RX, OP_RY, OP_RZ, OP_H):
        apply_1q(psi, mat1(op, th), tbit, n)
    elif op == OP_CNOT:
        apply_1q(psi, np.array([[0, 1], [1, 0]], dtype=complex), tbit, n, cbit)
    elif op == OP_CRX:
        apply_1q(psi, mat1(OP_RX, th), tbit, n, cbit)
    elif op == OP_CRZ:
        apply_1q(psi, mat1(OP_RZ, th), tbit, n, cbit)
    elif op == OP_U4:          # tbit = high bit, cbit = low bit of the 4x4 index
        idx = np.arange(1 << n)
        base = idx[((idx >> tbit) & 1 == 0) & ((idx >> cbit) & 1 == 0)]
        sel = [base, base | (1 << cbit), base | (1 << tbit), base | (1 << tbit) | (1 << cbit)]
        x = np.stack([psi[s] for s in sel])
        y = U @ x
        for r, s in enumerate(sel):
            psi[s] = y[r]
    else:
        raise AssertionError(op)


def program_gates(prog):
    """(op, target bit, control bit, slot) per gate in program order, bit = n-1-wire (csrc/qc_common.h)."""
    n, out = prog.n_qubits, []
    for g in prog.gates:
        if g.op == OP_U4:
            out.append((g.op, n - 1 - g.a, n - 1 - g.b, g.slot))
        elif g.op in (OP_CNOT, OP_CRX, OP_CRZ):
            out.append((g.op, n - 1 - g.b, n - 1 - g.a, g.slot))
        else:
            out.append((g.op, n - 1 - g.a, -1, g.slot))
    return out


CASES = [("cross_mesh", 16, 1), ("cross_mesh", 10, 1), ("cross_mesh", 12, 2), ("layered", 9, 2), ("layered", 13, 1),
         ("cascade", 11, 2), ("farhi", 10, 1), ("sim_circ_15", 12, 1), ("alternate", 9, 1), ("cascade", 14, 1)]


@pytest.mark.parametrize("ansatz,n,layers", CASES)
def test_plan_is_equivalent_to_the_program(ansatz, n, layers):
    circuits = pkg("circuits")
    prog = circuits.build_program(ansatz, n, layers, use_haar=True)
    plan = describe(prog)
    assert plan["n"] == n
    rng = np.random.default_rng(n * 7 + layers)
    theta = rng.uniform(-2, 2, prog.n_params)
    U = [np.linalg.qr(rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)))[0] for _ in range(2)]
    gates = program_gates(prog)
    first = n if plan["absorb"] else 0
    assert plan["absorb"] == (1 if ansatz in ("cascade", "cross_mesh") else 0)
    psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    ref = psi0.copy()
    for op, tb, cb, slot in gates[first:]:
        apply_gate(ref, op, tb, cb, theta[slot] if op != OP_U4 and slot >= 0 else 0.0, U[slot] if op == OP_U4 else None, n)

    out = psi0.copy()
    seen = []
    T = 12 if n >= 12 else (10 if n >= 10 else n)
    for st in plan["stages"]:
        lb, gb = st["lb"], st["gb"]
        assert st["nloc"] == T and sorted(lb + gb) == list(range(n)) and lb == sorted(lb)
        assert lb[:4] == [0, 1, 2, 3]                       # >= 128-byte runs in HBM
        assert st["nloc"] % plan["rbits"] == 0
        pos = {b: j for j, b in enumerate(lb)}
        pidx_seen, ntab = [], 0
        tslots = []

        def run_table(k, ts):
            tslots.append(ts)
            for g in plan["tables"][k]:
                assert g["op"] in (OP_RZ, OP_CRZ) and gates[g["gi"]][0] == g["op"]
                assert (g["bt"], g["bc"]) == (gates[g["gi"]][1], gates[g["gi"]][2]) and g["slot"] == gates[g["gi"]][3]
                apply_gate(out, g["op"], g["bt"], g["bc"], theta[g["slot"]], None, n)
                seen.append(g["gi"])

        for rd in st["rounds"]:
            if rd["kind"] == 1:
                ntab += 1
                run_table(rd["table"], rd["tslot"])
                continue
            if rd["tab_pre"] >= 0:      # applied to the amplitudes as the round loads them
                ntab += 1
                run_table(rd["tab_pre"], rd["ts_pre"])
            rb = rd["rb"]
            assert len(rb) == plan["rbits"] and rb == sorted(rb) and all(0 <= p < st["nloc"] for p in rb)
            for g in rd["gates"]:
                op, tb, cb, slot = gates[g["gi"]]
                assert (g["op"], g["tbit"], g["cbit"], g["slot"]) == (op, tb, cb, slot)
                if g["kind"] in (K_REG1, K_REG2, K_PRED, K_U4):
                    assert rb[g["tq"]] == pos[tb]             # target is a register bit of this round
                if g["kind"] in (K_REG2, K_U4):
                    assert rb[g["cq"]] == pos[cb]
                if g["kind"] == K_PRED:
                    assert op in (OP_CNOT, OP_CRX) and pos.get(cb, -1) not in rb
                if g["kind"] == K_PHASE:
                    assert op in (OP_RZ, OP_CRZ)
                if op in (OP_RX, OP_RY, OP_RZ, OP_CRX, OP_CRZ):
                    pidx_seen.append(g["pidx"])
                else:
                    assert g["pidx"] == -1
                apply_gate(out, op, tb, cb, theta[slot] if op != OP_U4 and slot >= 0 else 0.0,
                           U[slot] if op == OP_U4 else None, n)
                seen.append(g["gi"])
            if rd["tab_post"] >= 0:     # applied before the round stores them
                ntab += 1
                run_table(rd["tab_post"], rd["ts_post"])
        assert sorted(pidx_seen) == list(range(st["np"])) and ntab == st["ntab"] <= 2
        assert sorted(tslots) == list(range(ntab))             # every table owns one t accumulator of the stage
    assert sorted(seen) == list(range(first, len(gates)))     # every gate exactly once
    assert np.abs(out - ref).max() < 1e-10 * np.abs(ref).max()


def test_cross_mesh_16_runs_in_two_stages():
    """BASELINE config 5: 307 gates after folding the leading RX layer -> 2 passes over HBM per direction."""
    circuits = pkg("circuits")
    plan = describe(circuits.build_program("cross_mesh", 16, 1, use_haar=True))
    assert len(plan["stages"]) == 2
    assert [len(t) for t in plan["tables"]][0] == 256           # the 240 CRZ + 16 RZ of the mesh are ONE phase table
    rounds = sum(1 for st in plan["stages"] for r in st["rounds"])
    assert rounds <= 10, rounds                                 # LDS round trips per statevector: <= 10, not one per gate (307)
