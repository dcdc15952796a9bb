"""Size-independent properties at BASELINE.json's full sizes (where the CPU oracle is too slow to be
the checker): determinism, batch additivity of the gradient/loss vector, permutation equivariance,
|<Z>| <= 1, and a finite-difference check of the derivative channels against the value channel."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def base_args(**kw):
    a = {"batch_size": 64, "epochs": 0, "lr": 0.005, "seed": 1, "print_every": 10 ** 9, "num_qubits": 4,
         "num_quantum_layers": 1, "classic_network": [3, 50, 1], "q_ansatz": "cascade", "shots": 1024,
         "problem": "diffusion", "solver": "DV", "encoding": "None", "use_ibm_hardware": False}
    a.update(kw)
    return a


class Log:
    def print(self, *a):
        pass

    def get_output_dir(self):
        return "/tmp"


def make(gpu_device, **kw):
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    torch.manual_seed(1)
    return Solver(base_args(**kw), Log(), device=gpu_device)


def grads_for(model, X_ic, X_bc, X_res):
    """flat [grad | L_r, L_bc, L_ic] of the fused step's gradient phase on the given batches."""
    L = pkg("hip.lib")
    engine = pkg("hip.engine")
    dev = model._flat.device
    eng = model._engine_for(dev)
    eng.refresh_gates()
    n_res, n_ic, n_bc = X_res.shape[0], X_ic.shape[0], X_bc.shape[0]
    opt = engine.OptimState(eng.NP, 0.005, dev)
    fs = engine.FusedStep(eng, n_res, n_ic, n_bc, opt, (max(n_res, 1), max(n_ic, 1), max(n_bc, 1)))
    if fs.step_ws.numel() >= 4:
        # poison the step workspace (kept-state stores; a device-side fill, ~20 ms for the 51 GB of config 5): nothing may
        # depend on what it held before the step, and rows of points beyond a ragged batch must be masked, not
        # multiplied by zero cotangents
        nfl = fs.step_ws.numel() // 4
        fs.step_ws[: 4 * nfl].view(torch.float32).fill_(float("nan"))
    if n_res:
        fs.X_res[:n_res] = X_res.to(dev)
    if n_ic:
        fs.X_val[:n_ic] = X_ic.to(dev)
    if n_bc:
        fs.X_val[n_ic:n_ic + n_bc] = X_bc.to(dev)
    fs.run(L.QC_PHASE_GRADS)
    torch.cuda.synchronize()
    return fs.flat_grad.clone()


def test_config2_full_batch_determinism_and_additivity(gpu_device):
    B = 65536
    n3 = B // 3
    g = torch.Generator().manual_seed(9)
    X_ic = torch.rand(n3, 3, generator=g) * torch.tensor([0.0, 1.0, 1.0])
    X_bc = torch.rand(n3, 3, generator=g) * torch.tensor([1.0, 0.0, 1.0])
    X_res = torch.rand(B, 3, generator=g)
    model = make(gpu_device)
    f1 = grads_for(model, X_ic, X_bc, X_res)
    f2 = grads_for(model, X_ic, X_bc, X_res)
    assert torch.equal(f1, f2)                                   # fixed-order reductions: bit-reproducible
    assert torch.isfinite(f1).all()
    NP = f1.numel() - 3
    # every term of the flat vector is a mean over its batch: equal halves recombine by averaging
    h = B // 2
    none = X_ic[:0]
    ra = grads_for(model, none, none, X_res[:h])
    rb = grads_for(model, none, none, X_res[h:])
    rfull = grads_for(model, none, none, X_res)
    scale = max(1.0, rfull.abs().max().item())
    assert (0.5 * (ra + rb) - rfull).abs().max().item() < 2e-5 * scale
    h3 = n3 // 2
    va = grads_for(model, X_ic[:h3], X_bc[:h3], X_res[:0])
    vb = grads_for(model, X_ic[h3:2 * h3], X_bc[h3:2 * h3], X_res[:0])
    vfull = grads_for(model, X_ic[:2 * h3], X_bc[:2 * h3], X_res[:0])
    scale = max(1.0, vfull.abs().max().item())
    assert (0.5 * (va + vb) - vfull).abs().max().item() < 2e-5 * scale
    # and the three pipelines add up to the whole step
    full_val = grads_for(model, X_ic, X_bc, X_res[:0])
    scale = max(1.0, f1.abs().max().item())
    assert (rfull + full_val - f1).abs().max().item() < 2e-5 * scale
    assert f1[NP:].min().item() >= 0.0


def test_config2_permutation_equivariance_and_bounds(gpu_device):
    B = 65536
    model = make(gpu_device)
    X = torch.rand(B, 3, device=gpu_device)
    perm = torch.randperm(B, device=gpu_device)
    with torch.no_grad():
        u1, r1 = model.residual(X)
        u2, r2 = model.residual(X[perm])
    assert torch.equal(u1[perm], u2) and torch.equal(r1[perm], r2)     # every point is independent
    eng = model._engine_for(gpu_device)
    _, _, ajets, qjets = eng.forward(X, 6)
    assert qjets[0].abs().max().item() <= 1.0 + 1e-5                   # <Z> of a normalised state


@pytest.mark.parametrize("over", [{}, {"num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"}])
def test_derivative_channels_match_finite_differences_of_value_channel(over, gpu_device):
    """u_t, u_x, u_y, u_xx, u_yy implied by the residual kernels vs central differences of model(X)."""
    model = make(gpu_device, **over)
    eng = model._engine_for(gpu_device)
    B = 4096 if not over else 1024
    X = (0.1 + 0.8 * torch.rand(B, 3, device=gpu_device)).double()
    h = 2e-2

    def u_at(Y):
        with torch.no_grad():
            return model(Y.float()).double()[:, 0]

    # residual = u_t + u_x + u_y - 0.01 (u_xx + u_yy): assemble the same combination from differences
    e = torch.eye(3, device=gpu_device, dtype=torch.float64) * h
    u0 = u_at(X)
    d1 = [(u_at(X + e[k]) - u_at(X - e[k])) / (2 * h) for k in range(3)]
    d2 = [(u_at(X + e[k]) - 2 * u0 + u_at(X - e[k])) / (h * h) for k in (1, 2)]
    fd = d1[0] + d1[1] + d1[2] - 0.01 * (d2[0] + d2[1])
    with torch.no_grad():
        _, res = model.residual(X.float())
    err = (res.double()[:, 0] - fd).abs()
    scale = max(1.0, fd.abs().max().item())
    # O(h^2) truncation + fp32 cancellation in the second difference: loose, but a wrong channel is O(1) off
    assert err.max().item() < 5e-2 * scale and err.median().item() < 5e-3 * scale


def test_config3_and_config5_sizes_run_and_are_permutation_equivariant(gpu_device):
    for over, B in (({"num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"}, 131072),
                    ({"num_qubits": 16, "q_ansatz": "cross_mesh"}, 256)):
        model = make(gpu_device, **over)
        X = torch.rand(B, 3, device=gpu_device)
        perm = torch.randperm(B, device=gpu_device)
        with torch.no_grad():
            u1 = model(X)
            u2 = model(X[perm])
        assert torch.isfinite(u1).all()
        if over["num_qubits"] == 8:
            assert torch.equal(u1[perm], u2)
        else:   # 64-point tiles share block-level reductions only for gradients; values are per point
            assert (u1[perm] - u2).abs().max().item() < 1e-6
