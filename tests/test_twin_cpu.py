"""Host logic of the second workload (package train_hybrid_qpinn.py; reference train_hybrid_qpinn.py:50-947):
argument surface, sampler boxes, exact solution, model construction and initialisation order.  The reference
file imports PennyLane at module level, so nothing of it can be imported here: these tests pin the product
module against the oracle restatement (oracle/solver.py, "parity unpinned") and against closed forms."""
import numpy as np
import torch

from conftest import pkg


def test_parse_args_defaults_and_flags():
    t = pkg("train_hybrid_qpinn")
    a = t.parse_args([])
    assert (a.device, a.num_qubits, a.ansatz, a.encoding, a.shots) == ("auto", 4, "cascade", "angle", 1024)
    assert (a.epochs, a.batch_size, a.lr, a.seed, a.hidden_dim, a.print_every) == (5000, 64, 0.005, 42, 50, 100)
    assert (a.output_dir, a.diffusion_coef, a.use_ibm, a.ibm_backend) == ("./outputs", 0.01, False, "ibm_torino")
    b = t.parse_args(["--num-qubits", "5", "--ansatz", "cross_mesh", "--encoding", "amplitude", "--epochs", "3",
                      "--batch-size", "96", "--diffusion-coef", "0.02"])
    assert (b.num_qubits, b.ansatz, b.encoding, b.epochs, b.batch_size, b.diffusion_coef) == (5, "cross_mesh", "amplitude", 3, 96, 0.02)


def test_samplers_boxes_and_targets():
    t = pkg("train_hybrid_qpinn")
    torch.manual_seed(0)
    ics, bcs, res, dom = t.create_samplers("cpu", D=0.01)
    X, u = ics.sample(50)
    assert X.shape == (50, 3) and (X[:, 0] == 0).all() and u.shape == (50, 1)
    assert torch.allclose(u, torch.sin(torch.pi * X[:, 1:2]) * torch.sin(torch.pi * X[:, 2:3]))
    fixed = [(1, 0.0), (1, 1.0), (2, 0.0), (2, 1.0)]                    # x=0, x=1, y=0, y=1
    for s, (col, val) in zip(bcs, fixed):
        Xb, ub = s.sample(20)
        assert (Xb[:, col] == val).all() and (ub == 0).all() and Xb.min() >= 0 and Xb.max() <= 1
    Xr, r = res.sample(30)
    assert (r == 0).all() and Xr.min() >= 0 and Xr.max() <= 1
    assert dom.tolist() == [[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]]


def test_exact_solution_solves_the_pde_and_matches_oracle():
    from oracle import solver as osolver
    t = pkg("train_hybrid_qpinn")
    X = torch.rand(40, 3, dtype=torch.float64)
    assert torch.allclose(t.analytical_solution_torch(X, 0.03), osolver.twin_analytic_u(X, 0.03))
    assert np.allclose(t.analytical_solution(X[:, 0].numpy(), X[:, 1].numpy(), X[:, 2].numpy(), 0.03),
                       osolver.twin_analytic_u(X, 0.03)[:, 0].numpy())

    class Exact(torch.nn.Module):
        def forward(self, X):
            return t.analytical_solution_torch(X, 0.03)

    cols = [X[:, i:i + 1].clone() for i in range(3)]
    u, res = t.diffusion_operator(Exact(), *cols, D=0.03)          # generic (autograd) branch
    assert res.abs().max() < 1e-12


def test_model_construction_follows_the_reference_draw_order():
    from oracle import solver as osolver
    t = pkg("train_hybrid_qpinn")
    args = t.parse_args(["--device", "cpu", "--seed", "7"])
    torch.manual_seed(7)
    model = t.HybridQPINN(args, torch.device("cpu"))
    after_product = torch.rand(1)
    torch.manual_seed(7)
    ref = osolver.OracleHybridQPINN(num_qubits=4, ansatz="cascade", hidden=50, lr=0.005, seed=7)
    after_oracle = torch.rand(1)
    assert after_product.item() == after_oracle.item()                # same RNG consumption
    assert model.quantum_layer.params.shape == (12,) and sum(p.numel() for p in model.parameters()) == 717
    assert model.scheduler.patience == 500 and model.optimizer.param_groups[0]["lr"] == 0.005
    sd, so = dict(model.named_parameters()), dict(ref.named_parameters())
    for k in sd:
        assert torch.equal(sd[k].detach().cpu().reshape(-1), so[k].detach().reshape(-1)), k
    for seq in (model.preprocessor, model.postprocessor):
        assert all((m.bias == 0).all() for m in seq if isinstance(m, torch.nn.Linear))
    assert model.quantum_layer.haar_seed1 == 7 and model.quantum_layer.haar_seed2 == 8
