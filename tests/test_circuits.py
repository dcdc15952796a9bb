"""Gate-program lowering (host logic, no GPU)."""
import numpy as np
import pytest

from conftest import pkg


def test_params_per_layer_and_unknown_ansatz():
    c = pkg("circuits")
    assert [c.params_per_layer(a, 4) for a in ("layered", "alternate", "cascade", "farhi", "sim_circ_15", "cross_mesh")] \
        == [16, 12, 12, 6, 8, 28]
    assert c.params_per_layer("cross_mesh", 16) == 304
    with pytest.raises(ValueError):
        c.params_per_layer("nope", 4)


def test_gate_counts_and_flop_model_of_baseline_configs():
    c = pkg("circuits")
    p1 = c.build_program("cascade", 4, 1, True)
    assert p1.n_gates == 12 + 2 + 1 and p1.n_params == 12 and p1.algorithmic_flops() == 155 * 16
    p3 = c.build_program("layered", 8, 2, True)
    assert p3.n_gates == 2 * 40 + 2 + 1 and p3.n_params == 64 and p3.algorithmic_flops() == 507 * 256
    p5 = c.build_program("cross_mesh", 16, 1, True)
    assert p5.n_gates == 304 + 2 + 1 and p5.n_params == 304 and p5.algorithmic_flops() == 1283 * 65536


def test_reference_failure_modes_are_kept():
    c = pkg("circuits")
    with pytest.raises(IndexError):           # alternate over-indexes its 4n-4 angles for even n
        c.build_program("alternate", 4, 1, True)
    c.build_program("alternate", 5, 1, True)  # odd n is fine
    with pytest.raises(ValueError):           # sim_circ_15 at n=3 asks for CNOT[c, c]
        c.build_program("sim_circ_15", 3, 1, False)
    with pytest.raises(ValueError):           # Haar pair needs wires 0-3
        c.build_program("cascade", 3, 1, True)


def test_cascade_gate_order_matches_reference_listing():
    c = pkg("circuits")
    g = c.build_program("cascade", 4, 1, True).gates
    ops = [(c.OP_NAMES[x.op], x.a, x.b, x.slot) for x in g]
    assert ops[:4] == [("RX", i, -1, i) for i in range(4)]
    assert ops[4:8] == [("RZ", i, -1, 4 + i) for i in range(4)]
    assert ops[8:12] == [("CRX", 3, 0, 8), ("CRX", 2, 3, 9), ("CRX", 1, 2, 10), ("CRX", 0, 1, 11)]
    assert ops[12:] == [("U4", 0, 1, 0), ("U4", 2, 3, 1), ("H", 3, -1, -1)]


def test_rows_are_int32_n_by_4():
    c = pkg("circuits")
    r = c.build_program("layered", 4, 2, False).rows()
    assert r.dtype == np.int32 and r.shape == (2 * 20 + 1, 4)
