"""DVQuantumLayer under generic autograd (the reference's duck type: nn/pde.py:53-72 on ANY model):
forward and first-order reverse are the HIP kernels; with create_graph=True the reverse pass is rebuilt
on the trigonometric-interpolation form, so the five autograd.grad passes of diffusion_operator and the
loss.backward() through them work on a user-composed model.  Checked against the float64 CPU oracle and
against the operator fixtures produced by the reference's own nn/pde.py (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, pkg
from test_gpu_solver import Log, base_args, flat_grad, load_weights

pytestmark = pytest.mark.gpu


class Composite(torch.nn.Module):
    """What a user of the reference writes around the layer (nn/DVPDESolver.py:81-110 arithmetic)."""

    def __init__(self, solver):
        super().__init__()
        self.pre, self.q, self.post = solver.preprocessor, solver.quantum_layer, solver.postprocessor

    def forward(self, X):
        q = self.q(self.pre(X)).to(torch.float32)
        return self.post(q.T.reshape(-1, self.q.num_qubits))


@pytest.mark.parametrize("ansatz,n,layers", [("cascade", 4, 1), ("layered", 3, 2), ("sim_circ_15", 5, 1)])
def test_layer_second_derivatives_match_oracle(ansatz, n, layers, gpu_device):
    from oracle import solver as osolver
    args = base_args(q_ansatz=ansatz, num_qubits=n, num_quantum_layers=layers)
    torch.manual_seed(3)
    layer = pkg("nn.DVQuantumLayer").DVQuantumLayer(args).to(gpu_device)
    ref = osolver.OracleQuantumLayer(args)
    with torch.no_grad():
        ref.params.copy_(layer.params.cpu())
    B = 37
    x0 = torch.rand(B, n, dtype=torch.float64) * 4.0 - 2.0
    w = torch.rand(n, B, dtype=torch.float64)

    def run(lay, x, wt):
        x = x.clone().requires_grad_(True)
        q = lay(x)
        g, = torch.autograd.grad((q * wt.to(q.dtype)).sum(), x, create_graph=True)
        h, = torch.autograd.grad((g * g).sum(), x, create_graph=True)
        lay.zero_grad()
        (h * h).sum().backward()
        return q.detach(), g.detach(), h.detach(), lay.params.grad.detach().clone()

    qo, go, ho, po = run(ref, x0, w)
    qh, gh, hh, ph = run(layer, x0.to(gpu_device, torch.float32), w.to(gpu_device, torch.float32))
    assert (qh.cpu().double() - qo).abs().max() < 1e-5
    assert (gh.cpu().double() - go).abs().max() < 2e-5 * max(1.0, go.abs().max().item())
    assert (hh.cpu().double() - ho).abs().max() < 1e-4 * max(1.0, ho.abs().max().item())
    assert (ph.cpu().double() - po.double()).abs().max() < 2e-4 * max(1.0, po.abs().max().item())


def test_interpolant_equals_statevector_forward(gpu_device):
    args = base_args(q_ansatz="cross_mesh")
    torch.manual_seed(5)
    layer = pkg("nn.DVQuantumLayer").DVQuantumLayer(args).to(gpu_device)
    x = (torch.rand(513, 4, device=gpu_device) * 6.0 - 3.0)
    with torch.no_grad():
        assert (layer(x) - layer.trig_interpolant(x)).abs().max() < 5e-6


def test_generic_diffusion_operator_on_composite_model(gpu_device, tmp_path):
    z = np.load(os.path.join(GOLDEN, "operator_cascade_n4.npz"))
    pde = pkg("nn.pde")
    data = pkg("data.diffusion_dataset")
    torch.manual_seed(1)
    solver = pkg("nn.DVPDESolver").DVPDESolver(base_args(), Log(tmp_path), device=gpu_device)
    load_weights(solver, z, "w__")
    model = Composite(solver)
    assert not hasattr(model, "residual")          # -> the reference's autograd formulation, not the fused path
    X = torch.from_numpy(z["X"]).to(gpu_device)
    t, x, y = X[:, 0:1].clone(), X[:, 1:2].clone(), X[:, 2:3].clone()
    u, res = pde.diffusion_operator(model, t, x, y)
    assert np.abs(u.detach().cpu().numpy() - z["u"]).max() < 2e-5
    scale = max(1.0, np.abs(z["residual"]).max())
    assert np.abs(res.detach().cpu().numpy() - z["residual"]).max() < 1e-4 * scale
    loss = 2.0 * torch.nn.functional.mse_loss(res, data.r(X))
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * max(1.0, float(z["loss"]))
    solver.zero_grad()
    loss.backward()
    g = flat_grad(solver).cpu().numpy()
    gs = max(1.0, np.abs(z["grad"]).max())
    assert np.abs(g - z["grad"]).max() < 2e-4 * gs


@pytest.mark.parametrize("over,B", [({"num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"}, 5),
                                    ({"encoding": "amplitude"}, 9),
                                    ({"num_qubits": 9, "q_ansatz": "cascade"}, 3)])
def test_second_derivatives_beyond_the_interpolant_come_from_the_derivative_channels(over, B, gpu_device):
    """num_qubits > 7 or amplitude encoding: the create_graph=True reverse pass of the layer is differentiable once more
    through the jet kernels (Hessian-vector products, their theta-gradient, J c).  Same quantities as the oracle's
    torch double backward (reference nn/pde.py:59-70 on a user-composed model); third order is refused loudly."""
    from oracle import solver as osolver
    args = base_args(**over)
    torch.manual_seed(3)
    layer = pkg("nn.DVQuantumLayer").DVQuantumLayer(args).to(gpu_device)
    ref = osolver.OracleQuantumLayer(args)
    with torch.no_grad():
        ref.params.copy_(layer.params.cpu())
    n = args["num_qubits"]
    g0 = torch.Generator().manual_seed(5)
    x0 = torch.rand(B, n, dtype=torch.float64, generator=g0) * 2.0 - 0.7
    w = torch.rand(n, B, dtype=torch.float64, generator=g0)

    def run(lay, x, wt):
        x = x.clone().requires_grad_(True)
        q = lay(x)
        g, = torch.autograd.grad((q * wt.to(q.dtype)).sum(), x, create_graph=True)
        lay.zero_grad()
        s = (g * g).sum()
        h, = torch.autograd.grad(s, x, retain_graph=True)
        s.backward()
        return q.detach(), g.detach(), h.detach(), lay.params.grad.detach().clone()

    qo, go, ho, po = run(ref, x0, w)
    qh, gh, hh, ph = run(layer, x0.to(gpu_device, torch.float32), w.to(gpu_device, torch.float32))
    assert (qh.cpu().double() - qo).abs().max() < 1e-5
    assert (gh.cpu().double() - go).abs().max() < 2e-5 * max(1.0, go.abs().max().item())
    assert (hh.cpu().double() - ho).abs().max() < 1e-4 * max(1.0, ho.abs().max().item())
    assert (ph.cpu().double() - po.double()).abs().max() < 2e-4 * max(1.0, po.abs().max().item())
    # third order (differentiating the Hessian-vector product again) is not provided on this path
    x = x0.to(gpu_device, torch.float32).requires_grad_(True)
    g, = torch.autograd.grad(layer(x).sum(), x, create_graph=True)
    h, = torch.autograd.grad((g * g).sum(), x, create_graph=True)
    with pytest.raises(RuntimeError):
        (h * h).sum().backward()
