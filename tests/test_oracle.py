"""The CPU oracle against closed forms, its independent dense twin, the committed golden vectors
and the reference's own analytic functions (analytic.npz was produced by importing
/root/reference/data/diffusion_dataset.py)."""
import glob
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, pkg

from oracle import dense_unitary as du
from oracle import solver as osol
from oracle import statevector as sv


def test_rx_on_zero_gives_cos():
    a = torch.tensor([0.3, -1.2, 2.5], dtype=torch.float64)
    sim = sv.Simulator(1, 3)
    sim.RX(a, 0)
    assert torch.allclose(sim.expval_z(0), torch.cos(a), atol=1e-14)


def test_wire0_is_most_significant_bit_and_cnot_direction():
    sim = sv.Simulator(2, 1)
    sim.RX(torch.tensor([math.pi], dtype=torch.float64), 0)      # |10> up to phase
    assert abs(abs(sim.state.reshape(-1)[2]) - 1) < 1e-14
    sim.CNOT(0, 1)                                                # control = wire 0 -> |11>
    assert abs(abs(sim.state.reshape(-1)[3]) - 1) < 1e-14
    sim2 = sv.Simulator(2, 1)
    sim2.RX(torch.tensor([math.pi], dtype=torch.float64), 0)
    sim2.CNOT(1, 0)                                               # control = wire 1 (|0>) -> unchanged
    assert abs(abs(sim2.state.reshape(-1)[2]) - 1) < 1e-14


def test_crx_acts_only_when_control_is_one():
    th = torch.tensor(0.9, dtype=torch.float64)
    s0 = sv.Simulator(2, 1)
    s0.CRX(th, 0, 1)
    assert abs(s0.state.reshape(-1)[0] - 1) < 1e-14
    s1 = sv.Simulator(2, 1)
    s1.RX(torch.tensor([math.pi], dtype=torch.float64), 0)
    s1.CRX(th, 0, 1)
    assert abs(s1.expval_z(1).item() - math.cos(0.9)) < 1e-14


def test_two_wire_unitary_uses_first_wire_as_high_index():
    g = np.random.RandomState(0)
    A = torch.tensor(du.rx(0.7) @ du.rz(0.2), dtype=sv.CDT)
    Bm = torch.tensor(du.ry(1.1), dtype=sv.CDT)
    x = torch.tensor(g.randn(1, 3), dtype=torch.float64)
    s1 = sv.Simulator(3, 1)
    s2 = sv.Simulator(3, 1)
    for w in range(3):
        s1.RX(x[:, w], w)
        s2.RX(x[:, w], w)
    s1.apply2(torch.kron(A, Bm), 0, 2)
    s2.apply1(A, 0)
    s2.apply1(Bm, 2)
    assert torch.allclose(s1.state, s2.state, atol=1e-14)


def test_hadamard_last_wire_turns_z_into_x():
    a = torch.tensor([[0.4, 1.3]], dtype=torch.float64)
    sim = sv.Simulator(2, 1)
    sim.RY(a[0, 0], 0)
    sim.RY(a[0, 1], 1)
    sim.H(1)
    assert abs(sim.expval_z(1).item() - math.sin(1.3)) < 1e-14    # <X> of RY(a)|0> = sin a


@pytest.mark.parametrize("fname", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "expval_*.npz"))))
def test_oracle_reproduces_golden_expvals(fname):
    z = np.load(os.path.join(GOLDEN, fname))
    stem = fname[len("expval_"):-len(".npz")]
    ans, ntag, _ = stem.rsplit("_", 2)
    n = int(ntag[1:])
    seed = int(z["seed"])
    haar = sv.haar_pair(seed, seed + 1) if seed >= 0 and n >= 4 else None
    q = sv.circuit_expvals(torch.from_numpy(z["x"]), torch.from_numpy(z["params"]), ans, n, haar).numpy()
    assert np.abs(q - z["expval"]).max() < 1e-12
    assert np.abs(q).max() <= 1 + 1e-12


@pytest.mark.parametrize("ans,n,L", [("cascade", 4, 1), ("layered", 5, 2), ("cross_mesh", 4, 1), ("farhi", 6, 1),
                                     ("sim_circ_15", 5, 1), ("alternate", 5, 2), ("cascade", 2, 3), ("layered", 8, 1)])
def test_per_gate_oracle_equals_dense_unitary_oracle(ans, n, L):
    circuits = pkg("circuits")
    g = torch.Generator().manual_seed(n * 10 + L)
    P = circuits.params_per_layer(ans, n)
    params = torch.randn(L, P, generator=g, dtype=torch.float64)
    x = torch.randn(4, n, generator=g, dtype=torch.float64)
    use_haar = n >= 4
    haar = sv.haar_pair(3, 4) if use_haar else None
    q1 = sv.circuit_expvals(x, params, ans, n, haar).numpy()
    prog = circuits.build_program(ans, n, L, use_haar)
    q2 = du.program_expvals(prog.rows(), x.numpy(), params.numpy(),
                            None if haar is None else [h.numpy() for h in haar], n)
    assert np.abs(q1 - q2).max() < 1e-12


def test_two_term_shift_exact_for_rx_but_not_crx_four_term_is():
    """SURVEY hard part: +-pi/2 rule is exact for RX/RZ, controlled rotations need the 4-term rule."""
    n, ans = 4, "cascade"
    g = torch.Generator().manual_seed(9)
    params = torch.randn(1, 12, generator=g, dtype=torch.float64)
    x = torch.randn(3, n, generator=g, dtype=torch.float64)
    haar = sv.haar_pair(1, 2)

    def f(p):
        return sv.circuit_expvals(x, p, ans, n, haar).sum()

    p = params.clone().requires_grad_(True)
    f(p).backward()

    def shifted(k, s):
        q = params.clone()
        q[0, k] += s
        return f(q).item()

    k = 1                                                          # an RX parameter
    two = 0.5 * (shifted(k, math.pi / 2) - shifted(k, -math.pi / 2))
    assert abs(two - p.grad[0, k].item()) < 1e-12
    k = 9                                                          # a CRX parameter
    two = 0.5 * (shifted(k, math.pi / 2) - shifted(k, -math.pi / 2))
    c1, c2 = (math.sqrt(2) + 1) / (4 * math.sqrt(2)), (math.sqrt(2) - 1) / (4 * math.sqrt(2))
    four = c1 * (shifted(k, math.pi / 2) - shifted(k, -math.pi / 2)) - c2 * (shifted(k, 3 * math.pi / 2) - shifted(k, -3 * math.pi / 2))
    assert abs(four - p.grad[0, k].item()) < 1e-12
    assert abs(two - p.grad[0, k].item()) > 1e-6


def test_analytic_u_r_match_reference_dataset_functions():
    z = np.load(os.path.join(GOLDEN, "analytic.npz"))
    X = torch.from_numpy(z["X"])
    data = pkg("data.diffusion_dataset")
    for fn_u, fn_r in ((osol.analytic_u, osol.analytic_r), (data.u, data.r)):
        assert np.abs(fn_u(X).numpy() - z["u"]).max() < 1e-7
        assert np.abs(fn_r(X).numpy() - z["r"]).max() < 1e-5 * np.abs(z["r"]).max()


def test_forcing_term_quirk_is_true_residual_plus_4u():
    X = torch.rand(64, 3, dtype=torch.float64)
    t, x, y = (X[:, i:i + 1].clone().requires_grad_(True) for i in range(3))
    _, res = osol.diffusion_residual(lambda v: osol.analytic_u(v), t, x, y)
    r = osol.analytic_r(X)
    assert (res.detach() - r + 4 * osol.analytic_u(X)).abs().max() < 1e-10


def test_haar_unitaries_match_fixture_and_are_unitary():
    z = np.load(os.path.join(GOLDEN, "haar.npz"))
    circuits = pkg("circuits")
    for s1 in (1, 42):
        U = circuits.haar_unitaries(s1, s1 + 1)
        assert np.abs(U[0] - z[f"seed{s1}"]).max() < 1e-13
        assert np.abs(U[1] - z[f"seed{s1 + 1}"]).max() < 1e-13
        for k in range(2):
            assert np.abs(U[k] @ U[k].conj().T - np.eye(4)).max() < 1e-12
    assert abs(z["seed1"][0, 0] - (0.6793174318473825 - 0.07211119614799695j)) < 1e-13   # SURVEY §8c value


def test_amplitude_embedding_conventions():
    """AmplitudeEmbedding(normalize=True, pad_with=0): features fill amplitudes 0..n-1 of the index with
    wire 0 as MSB.  x = e_0 is |0...0>, i.e. the angle embedding at zero angles; scaling x changes nothing."""
    n, ans = 4, "cascade"
    g = torch.Generator().manual_seed(2)
    params = torch.randn(1, 12, generator=g, dtype=torch.float64)
    haar = sv.haar_pair(1, 2)
    e0 = torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=torch.float64)
    qa = sv.circuit_expvals(e0, params, ans, n, haar, "amplitude")
    qz = sv.circuit_expvals(torch.zeros(1, n, dtype=torch.float64), params, ans, n, haar, "angle")
    assert (qa - qz).abs().max() < 1e-14
    x = torch.randn(3, n, generator=g, dtype=torch.float64)
    q1 = sv.circuit_expvals(x, params, ans, n, haar, "amplitude")
    q2 = sv.circuit_expvals(3.7 * x, params, ans, n, haar, "amplitude")
    assert (q1 - q2).abs().max() < 1e-13
    # feature 1 lands on amplitude index 1 = |0001>: only the LAST wire is flipped before the ansatz
    sim = sv.Simulator(2, 1)
    q = sv.circuit_expvals(torch.tensor([[0.0, 1.0]], dtype=torch.float64), torch.zeros(1, 6, dtype=torch.float64), "cascade", 2, None, "amplitude")
    assert abs(q[0, 0].item() - 1.0) < 1e-14          # wire 0 still |0>; wire 1 = H|1> -> <Z> = 0
    assert abs(q[1, 0].item()) < 1e-14
