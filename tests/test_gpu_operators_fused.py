"""Klein-Gordon, wave and Helmholtz operators (reference nn/pde.py:28-52,73-95) on a two-input DVPDESolver: second
derivatives from the fused derivative-channel HIP kernels (``DVPDESolver.second_order``), checked against
``tests/golden/other_operators.npz`` - outputs, loss and parameter gradient computed by the REFERENCE's own operator
functions driving the CPU oracle (tests/golden/make_golden.py).  The fixture's model is Linear(2,16)-Tanh-Linear(16,4)
-> <Z> -> Linear(4,16)-Tanh-Linear(16,1): exactly a DVPDESolver with classic_network [2, 16, 1]."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, pkg
from test_gpu_solver import Log, base_args

pytestmark = pytest.mark.gpu

NAME_MAP = {"pre": "preprocessor", "post": "postprocessor", "q": "quantum_layer"}


def _solver_from_fixture(z, name, device, tmp_path, network=(2, 16, 1)):
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    torch.manual_seed(0)
    model = Solver(base_args(classic_network=list(network)), Log(tmp_path), device=device)
    prefix = f"{name}__w__"
    sd = {}
    for k in z.files:
        if k.startswith(prefix):
            parts = k[len(prefix):].split("__")
            sd[".".join([NAME_MAP[parts[0]]] + parts[1:])] = torch.from_numpy(z[k])
    with torch.no_grad():
        for pname, p in model.named_parameters():
            p.copy_(sd[pname].to(p.device))
    return model, sd


def _ref_grad(z, name, model, n_in=2, n_out=1):
    """the fixture's gradient (composite order: pre, q, post) re-ordered to model.parameters() order"""
    g = z[f"{name}__grad"]
    shapes = {"pre.0.weight": (16, n_in), "pre.0.bias": (16,), "pre.2.weight": (4, 16), "pre.2.bias": (4,), "q.params": (1, 12),
              "post.0.weight": (16, 4), "post.0.bias": (16,), "post.2.weight": (n_out, 16), "post.2.bias": (n_out,)}
    off, by = 0, {}
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        first, rest = k.split(".", 1)
        by[NAME_MAP[first] + "." + rest] = g[off:off + n].reshape(shp)
        off += n
    assert off == g.size
    return {n_: by[n_] for n_, _ in model.named_parameters()}


@pytest.mark.parametrize("name", ["klein_gordon", "wave", "helmholtz"])
def test_second_order_operators_on_fused_channels_match_reference(name, gpu_device, tmp_path):
    z = np.load(os.path.join(GOLDEN, "other_operators.npz"))
    pde = pkg("nn.pde")
    model, _ = _solver_from_fixture(z, name, gpu_device, tmp_path)
    assert model.preprocessor[0].weight.shape == (16, 2)          # the reference's parameter shapes
    op = {"klein_gordon": pde.klein_gordon_operator, "wave": pde.wave_operator, "helmholtz": pde.helmholtz_operator}[name]
    X = torch.from_numpy(z[f"{name}__X"]).to(gpu_device)
    cols = [X[:, i:i + 1].clone() for i in range(2)]
    res = list(op(model, *cols))
    for i, r in enumerate(res):
        want = z[f"{name}__out{i}"]
        assert np.abs(r.detach().cpu().numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (name, i)
    loss = sum((r ** 2).mean() * (i + 1) for i, r in enumerate(res))
    assert abs(loss.item() - float(z[f"{name}__loss"])) < 1e-4 * max(1.0, float(z[f"{name}__loss"]))
    model.zero_grad()
    loss.backward()
    ref = _ref_grad(z, name, model)
    gs = max(1.0, max(np.abs(v).max() for v in ref.values()))
    for pname, p in model.named_parameters():
        assert np.abs(p.grad.cpu().numpy() - ref[pname]).max() < 2e-4 * gs, pname
    # the value path of the two-input model agrees with the operator's u
    with torch.no_grad():
        assert (model(X) - res[0]).abs().max().item() < 1e-6


def test_two_input_model_refuses_three_columns(gpu_device, tmp_path):
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    model = Solver(base_args(classic_network=[2, 16, 1]), Log(tmp_path), device=gpu_device)
    with pytest.raises(ValueError):
        model(torch.rand(4, 3, device=gpu_device))
    with pytest.raises(ValueError):
        model.residual(torch.rand(4, 3, device=gpu_device))


def test_navier_stokes_on_fused_channels_matches_reference(gpu_device, tmp_path):
    """Reference nn/pde.py:2-27 on a three-output model (u, v, p): the six derivative channels of every output come
    from the fused kernels (DVPDESolver.jets, qc_post modes 4 / 3 with a cotangent per channel), the products
    u u_x + v u_y of the momentum equations are formed in torch.  Fixture: the reference's own operator on the CPU
    oracle (Linear(3,16)-Tanh-Linear(16,4) -> <Z> -> Linear(4,16)-Tanh-Linear(16,3))."""
    name = "navier_stokes"
    z = np.load(os.path.join(GOLDEN, "other_operators.npz"))
    pde = pkg("nn.pde")
    model, _ = _solver_from_fixture(z, name, gpu_device, tmp_path, network=(3, 16, 3))
    assert model.postprocessor[2].weight.shape == (3, 16) and model.n_out == 3
    X = torch.from_numpy(z[f"{name}__X"]).to(gpu_device)
    cols = [X[:, i:i + 1].clone() for i in range(3)]
    res = list(pde.navier_stokes_2D_operator(model, *cols))
    assert len(res) == 3
    for i, r in enumerate(res):
        want = z[f"{name}__out{i}"]
        assert np.abs(r.detach().cpu().numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max()), i
    loss = sum((r ** 2).mean() * (i + 1) for i, r in enumerate(res))
    assert abs(loss.item() - float(z[f"{name}__loss"])) < 1e-4 * max(1.0, float(z[f"{name}__loss"]))
    model.zero_grad()
    loss.backward()
    ref = _ref_grad(z, name, model, n_in=3, n_out=3)
    gs = max(1.0, max(np.abs(v).max() for v in ref.values()))
    for pname, p in model.named_parameters():
        assert np.abs(p.grad.cpu().numpy() - ref[pname]).max() < 2e-4 * gs, pname
    # the model's forward is the value channel of the three outputs
    with torch.no_grad():
        f = model(X)
    assert f.shape == (X.shape[0], 3)
    J = [model.jets(X, o) for o in range(3)]
    assert max((f[:, o] - J[o][:, 0]).abs().max().item() for o in range(3)) < 1e-6
    # the operator takes all three outputs from ONE pass (jets_all): one circuit evaluation and one adjoint sweep, where
    # three single-output evaluations run three of each; same numbers
    eng = model._jet_engine(X.device)
    calls = {"fwd": 0, "bwd": 0}
    fwd0, bwd0 = eng.circuit.forward_jets, eng.lib.qc_backward_jets

    def count_fwd(a):
        calls["fwd"] += 1
        return fwd0(a)
    eng.circuit.forward_jets = count_fwd
    try:
        JA = model.jets_all(X)
        assert calls["fwd"] == 1
        assert JA.shape == (X.shape[0], 3, 6)
        assert max((JA[:, o] - J[o]).abs().max().item() for o in range(3)) < 2e-6
        w = torch.randn_like(JA)
        model.zero_grad()
        (JA * w).sum().backward()
        g_all = {k: v.grad.clone() for k, v in model.named_parameters()}
        model.zero_grad()
        sum((model.jets(X, o) * w[:, o]).sum() for o in range(3)).backward()
        assert calls["fwd"] == 4
        for k, v in model.named_parameters():
            assert (v.grad - g_all[k]).abs().max().item() < 1e-5 * max(1.0, v.grad.abs().max().item()), k
    finally:
        eng.circuit.forward_jets = fwd0
    with pytest.raises(NotImplementedError):
        model.residual(X)


def test_jets_of_a_single_output_model_agree_with_the_residual_path(gpu_device, tmp_path):
    """DVPDESolver.jets on the convection-diffusion model: u and the operator's linear combination of the channels
    equal DVPDESolver.residual (the same kernels behind qc_post modes 0 and 4), and so do the gradients."""
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    torch.manual_seed(3)
    model = Solver(base_args(), Log(tmp_path), device=gpu_device)
    X = torch.rand(100, 3, device=gpu_device)
    D, vx, vy = 0.01, 1.0, 1.0
    u0, r0 = model.residual(X, D=D, v_x=vx, v_y=vy)
    J = model.jets(X)
    r1 = J[:, 1:2] + vx * J[:, 2:3] + vy * J[:, 3:4] - D * (J[:, 4:5] + J[:, 5:6])
    assert (J[:, 0:1] - u0).abs().max().item() < 1e-6
    assert (r1 - r0).abs().max().item() < 1e-5 * max(1.0, r0.abs().max().item())
    w = torch.randn(100, 1, device=gpu_device)
    g0 = torch.autograd.grad((r0 * w).sum() + (u0 * w).sum(), list(model.parameters()))
    g1 = torch.autograd.grad((r1 * w).sum() + (J[:, 0:1] * w).sum(), list(model.parameters()))
    scale = max(1.0, max(g.abs().max().item() for g in g0))
    assert max((a - b).abs().max().item() for a, b in zip(g0, g1)) < 2e-5 * scale
