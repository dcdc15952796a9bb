"""GPU parity of the DVPDESolver path (modules -> C ABI -> HIP kernels) against fixtures produced by
the REFERENCE's own nn/pde.py and trainer/diffusion_train.py driving the CPU oracle
(tests/golden/make_golden.py).  Tolerances: 1e-5 on <Z>, 1e-4 on the PINN loss (north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, pkg

pytestmark = pytest.mark.gpu


def base_args(**kw):
    a = {"batch_size": 64, "epochs": 20, "lr": 0.005, "seed": 1, "print_every": 100,
         "num_qubits": 4, "num_quantum_layers": 1, "classic_network": [3, 50, 1],
         "q_ansatz": "cascade", "shots": 1024, "problem": "diffusion", "solver": "DV",
         "encoding": "None", "use_ibm_hardware": False}
    a.update(kw)
    return a


class Log:
    def __init__(self, d):
        self.d, self.lines = str(d), []

    def print(self, *a):
        self.lines.append(" ".join(str(v) for v in a))

    def get_output_dir(self):
        return self.d


def load_weights(model, z, prefix):
    sd = {k[len(prefix):].replace("__", "."): torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(sd[name].to(p.device))


def flat_grad(model):
    return torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in model.parameters()])


OPERATOR_CASES = [("cascade_n4", {}), ("cross_mesh_n4", {"q_ansatz": "cross_mesh"}),
                  ("layered_n8", {"num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"}),
                  ("cross_mesh_n16", {"num_qubits": 16, "q_ansatz": "cross_mesh"})]


@pytest.mark.parametrize("tag,over", OPERATOR_CASES)
def test_diffusion_operator_matches_reference_pde(tag, over, gpu_device, tmp_path):
    z = np.load(os.path.join(GOLDEN, f"operator_{tag}.npz"))
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    pde = pkg("nn.pde")
    data = pkg("data.diffusion_dataset")
    torch.manual_seed(1)
    model = Solver(base_args(**over), Log(tmp_path), device=gpu_device)
    load_weights(model, z, "w__")
    X = torch.from_numpy(z["X"]).to(gpu_device)
    # value path
    u_val = model(X)
    assert np.abs(u_val.detach().cpu().numpy() - z["u"]).max() < 2e-5
    # residual path, same call shape as the reference trainer
    t, x, y = X[:, 0:1].clone(), X[:, 1:2].clone(), X[:, 2:3].clone()
    u, res = pde.diffusion_operator(model, t, x, y)
    assert np.abs(u.detach().cpu().numpy() - z["u"]).max() < 2e-5
    scale = max(1.0, np.abs(z["residual"]).max())
    assert np.abs(res.detach().cpu().numpy() - z["residual"]).max() < 1e-4 * scale
    loss = 2.0 * torch.nn.functional.mse_loss(res, data.r(X))
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * max(1.0, float(z["loss"]))
    model.zero_grad()
    loss.backward()
    g = flat_grad(model).cpu().numpy()
    gs = max(1.0, np.abs(z["grad"]).max())
    assert np.abs(g - z["grad"]).max() < 2e-4 * gs


@pytest.mark.parametrize("tag,over", [("cascade_n4", {}), ("layered_n8", {"num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"}),
                                      ("cross_mesh_n16", {"num_qubits": 16, "q_ansatz": "cross_mesh"})])
def test_forward_is_differentiable_in_its_inputs_like_the_reference_module(tag, over, gpu_device, tmp_path):
    """The reference module is an ordinary autograd graph: trainer/diffusion_train.py:37-39 sets requires_grad on the
    inputs and nn/pde.py:59-70 differentiates model(X) five times.  Here that ALGORITHM (the generic branch of
    nn.pde.diffusion_operator: plain autograd.grad on model.forward, no fused dispatch) runs on the DVPDESolver and
    must reproduce the fixture the reference's own nn/pde.py produced - u, residual, loss and the parameter gradient
    of loss.backward() through the second derivatives."""
    z = np.load(os.path.join(GOLDEN, f"operator_{tag}.npz"))
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    pde = pkg("nn.pde")
    data = pkg("data.diffusion_dataset")
    torch.manual_seed(1)
    model = Solver(base_args(**over), Log(tmp_path), device=gpu_device)
    load_weights(model, z, "w__")
    X = torch.from_numpy(z["X"]).to(gpu_device)
    t, x, y = X[:, 0:1].clone(), X[:, 1:2].clone(), X[:, 2:3].clone()
    plain = lambda inp: model(inp)                       # hides .residual / .quantum_layer: the generic autograd branch
    u, res = pde.diffusion_operator(plain, t, x, y)
    assert np.abs(u.detach().cpu().numpy() - z["u"]).max() < 2e-5
    scale = max(1.0, np.abs(z["residual"]).max())
    assert np.abs(res.detach().cpu().numpy() - z["residual"]).max() < 1e-4 * scale
    loss = 2.0 * torch.nn.functional.mse_loss(res, data.r(X))
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * max(1.0, float(z["loss"]))
    model.zero_grad()
    loss.backward()
    g = flat_grad(model).cpu().numpy()
    assert np.abs(g - z["grad"]).max() < 2e-4 * max(1.0, np.abs(z["grad"]).max())
    # first derivatives alone, and what the six channels do not carry
    Xg = X.clone().requires_grad_(True)
    uu = model(Xg)
    g1, = torch.autograd.grad(uu.sum(), Xg, create_graph=True)
    jets = model.jets(X, 0)
    assert torch.allclose(g1.detach(), jets[:, 1:4].detach(), atol=1e-6)
    h_t, = torch.autograd.grad(g1[:, 0].sum(), Xg, retain_graph=True)
    assert torch.isnan(h_t).all()                        # u_tt, u_tx, u_ty: no channel -> NaN, never a silent zero
    h_x, = torch.autograd.grad(g1[:, 1].sum(), Xg, retain_graph=True)
    assert torch.allclose(h_x[:, 1], jets[:, 4].detach(), atol=1e-6) and torch.isnan(h_x[:, 0]).all()
    # without requires_grad on the input the value-only kernels run (same numbers)
    assert np.abs(model(X).detach().cpu().numpy() - u.detach().cpu().numpy()).max() < 2e-6


@pytest.mark.parametrize("tag,over", [("cascade_n4_sigma", {}),
                                      ("layered_n8_sigma", {"num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"})])
def test_diffusion_operator_with_sigma_scalings_matches_reference_pde(tag, over, gpu_device, tmp_path):
    """sigma_t, sigma_x, sigma_y != 1 (and non-default D, v_x, v_y): fixtures from the reference's own
    nn/pde.py:53-72 driving the oracle; here the scalings are folded into the coefficients of the fused kernels."""
    z = np.load(os.path.join(GOLDEN, f"operator_{tag}.npz"))
    kw = {k[4:]: float(z[k]) for k in z.files if k.startswith("op__")}
    assert kw["sigma_t"] != 1.0 and kw["sigma_x"] != 1.0 and kw["sigma_y"] != 1.0
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    pde = pkg("nn.pde")
    data = pkg("data.diffusion_dataset")
    torch.manual_seed(1)
    model = Solver(base_args(**over), Log(tmp_path), device=gpu_device)
    load_weights(model, z, "w__")
    X = torch.from_numpy(z["X"]).to(gpu_device)
    t, x, y = X[:, 0:1].clone(), X[:, 1:2].clone(), X[:, 2:3].clone()
    u, res = pde.diffusion_operator(model, t, x, y, **kw)
    assert np.abs(u.detach().cpu().numpy() - z["u"]).max() < 2e-5
    scale = max(1.0, np.abs(z["residual"]).max())
    assert np.abs(res.detach().cpu().numpy() - z["residual"]).max() < 1e-4 * scale
    loss = 2.0 * torch.nn.functional.mse_loss(res, data.r(X))
    assert abs(loss.item() - float(z["loss"])) < 1e-4 * max(1.0, float(z["loss"]))
    model.zero_grad()
    loss.backward()
    g = flat_grad(model).cpu().numpy()
    assert np.abs(g - z["grad"]).max() < 2e-4 * max(1.0, np.abs(z["grad"]).max())
    # the default call afterwards is unaffected by the scalings used above
    u1, res1 = pde.diffusion_operator(model, t, x, y)
    u2, res2 = model.residual(X)
    assert torch.equal(res1, res2)


TRAIN_CASES = [("cascade_n4_b64", {"epochs": 20}), ("cascade_n4_b128", {"epochs": 8}),
               ("layered_n8_b32", {"epochs": 5, "num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"})]


@pytest.mark.parametrize("tag,over", TRAIN_CASES)
def test_train_loop_matches_reference_train(tag, over, gpu_device, tmp_path):
    """Same seed, same batches: loss history of the fused HIP step vs the reference's train()."""
    z = np.load(os.path.join(GOLDEN, f"train_{tag}.npz"))
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    torch.manual_seed(1)
    model = Solver(base_args(**over), Log(tmp_path), device=gpu_device)
    # initial weights come from the same RNG stream as the reference construction order
    for name, p in model.named_parameters():
        ref = z["w0__" + name.replace(".", "__")]
        assert np.abs(p.detach().cpu().numpy() - ref).max() == 0.0, name
    B = int(z["batch_size"])
    steps = z["X_res"].shape[0]
    batches = [tuple(torch.from_numpy(z[k][it]) for k in ("X_ic", "X_bc", "X_res")) for it in range(steps)]
    trainer.train(model, batch_size=B, batches=batches)
    hist = np.array(model.loss_history)
    ref = z["loss_history"]
    assert hist.shape == ref.shape
    assert np.abs(hist - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
    # Adam divides by sqrt(v): a parameter whose gradient is at rounding-noise level still moves by ~lr per
    # step, so fp32-vs-fp64 noise can move single weights by a fraction of lr*steps; the bulk must agree.
    diffs = np.concatenate([np.abs(p.detach().cpu().numpy() - z["w1__" + name.replace(".", "__")]).reshape(-1)
                            for name, p in model.named_parameters()])
    assert np.median(diffs) < 1e-4 and diffs.max() < 0.25 * 0.005 * steps, (np.median(diffs), diffs.max())
    # optimiser / scheduler objects carry the device state back (checkpoint interchange)
    assert model.scheduler.last_epoch == steps
    assert abs(model.optimizer.param_groups[0]["lr"] - 0.005) < 1e-9


def test_first_step_gradient_matches_reference(gpu_device, tmp_path):
    z = np.load(os.path.join(GOLDEN, "train_cascade_n4_b64.npz"))
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    L = pkg("hip.lib")
    torch.manual_seed(1)
    model = Solver(base_args(), Log(tmp_path), device=gpu_device)
    tr = trainer.FusedTrainer(model, 64, capacity=4)
    tr.load_batches(*(torch.from_numpy(z[k][0]) for k in ("X_ic", "X_bc", "X_res")))
    tr.fs.run(L.QC_PHASE_GRADS)
    raw = tr.fs.flat_grad[: tr.eng.NP].cpu().numpy()
    assert np.abs(raw - z["grad_raw0"]).max() < 2e-4 * max(1.0, np.abs(z["grad_raw0"]).max())
    parts = tr.fs.flat_grad[tr.eng.NP:].cpu().numpy()          # L_r, L_bc, L_ic
    ref = z["parts"][0]                                          # loss, l_r, l_bc, l_ic
    assert np.abs(parts - ref[1:]).max() < 1e-4 * max(1.0, np.abs(ref).max())
    tr.fs.run(L.QC_PHASE_UPDATE)
    clipped = tr.fs.flat_grad[: tr.eng.NP].cpu().numpy()
    assert np.abs(clipped - z["grad_clipped0"]).max() < 2e-4
    assert abs(tr.opt.read()["loss"] - ref[0]) < 1e-4 * max(1.0, ref[0])


def test_value_vjp_through_autograd(gpu_device, tmp_path):
    """loss.backward() through model.forward (BC/IC style) vs the oracle on the same weights."""
    from oracle import solver as osol
    z = np.load(os.path.join(GOLDEN, "operator_cascade_n4.npz"))
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    data = pkg("data.diffusion_dataset")
    torch.manual_seed(1)
    model = Solver(base_args(), Log(tmp_path), device=gpu_device)
    load_weights(model, z, "w__")
    torch.manual_seed(1)
    om = osol.OracleSolver(base_args())
    load_weights(om, z, "w__")
    X = torch.from_numpy(z["X"])
    lo = 4.0 * torch.nn.functional.mse_loss(om(X), osol.analytic_u(X))
    om.zero_grad()
    lo.backward()
    Xg = X.to(gpu_device)
    lh = 4.0 * torch.nn.functional.mse_loss(model(Xg), data.u(Xg))
    model.zero_grad()
    lh.backward()
    assert abs(lh.item() - lo.item()) < 1e-5
    go = flat_grad(om).numpy()
    gh = flat_grad(model).cpu().numpy()
    assert np.abs(go - gh).max() < 1e-5 * max(1.0, np.abs(go).max())


def test_cpu_input_fails_loudly(gpu_device, tmp_path):
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    L = pkg("hip.lib")
    model = Solver(base_args(), Log(tmp_path), device=gpu_device)
    with pytest.raises(L.QcError):
        model(torch.rand(4, 3))


def test_device_sampler_boxes_and_sharding(gpu_device):
    """qc_sample_collocation: reference boxes (trainer/diffusion_train.py:9-20), uniform [0,1), and the
    data-parallel property: shards indexed by global point index reproduce the single-GPU batch."""
    L = pkg("hip.lib")
    lib = L.load()
    st = torch.cuda.current_stream(gpu_device).cuda_stream
    n_res, n_ic, n_bc = 4099, 1365, 1365
    Xr = torch.empty(n_res, 3, device=gpu_device)
    Xv = torch.empty(n_ic + n_bc, 3, device=gpu_device)
    L.check(lib.qc_sample_collocation(Xr.data_ptr(), n_res, 0, Xv.data_ptr(), n_ic, 0, n_bc, 0, 77, 5, st))
    xr, xv = Xr.cpu().numpy(), Xv.cpu().numpy()
    assert xr.min() >= 0.0 and xr.max() < 1.0 and abs(xr.mean() - 0.5) < 0.02
    assert np.all(xv[:n_ic, 0] == 0.0) and np.all(xv[n_ic:, 1] == 0.0)            # IC: t = 0, BC1: x = 0
    assert xv[:n_ic, 1:].std() > 0.2 and xv[n_ic:, [0, 2]].std() > 0.2
    assert len(np.unique(xr[:, 0])) > 0.99 * n_res
    # two "ranks" drawing their shards of the same global batch
    h = n_res // 2
    A = torch.empty(h, 3, device=gpu_device)
    Bm = torch.empty(n_res - h, 3, device=gpu_device)
    Va = torch.empty(700 + 600, 3, device=gpu_device)
    L.check(lib.qc_sample_collocation(A.data_ptr(), h, 0, Va.data_ptr(), 700, 0, 600, 0, 77, 5, st))
    L.check(lib.qc_sample_collocation(Bm.data_ptr(), n_res - h, h, Va.data_ptr(), 0, 0, 0, 0, 77, 5, st))
    assert np.array_equal(np.concatenate([A.cpu().numpy(), Bm.cpu().numpy()]), xr)
    assert np.array_equal(Va.cpu().numpy()[:700], xv[:700]) and np.array_equal(Va.cpu().numpy()[700:], xv[n_ic:n_ic + 600])
    # another step -> different points
    L.check(lib.qc_sample_collocation(Xr.data_ptr(), n_res, 0, Xv.data_ptr(), n_ic, 0, n_bc, 0, 77, 6, st))
    assert not np.array_equal(Xr.cpu().numpy(), xr)


@pytest.mark.parametrize("merge", ["1", "0"])
def test_fused_step_draws_the_same_points_as_the_sampler_entry_point(merge, gpu_device, tmp_path):
    """The merged step folds the sampler into its first stage: the points it leaves in X_res / X_val must be
    the ones qc_sample_collocation_faces draws for the same (seed, step, offsets); the two-stream form
    (QC_NO_MERGE=1, child process) launches k_sample itself."""
    if merge == "0":
        import subprocess, sys
        env = dict(os.environ, QC_NO_MERGE="1")
        code = ("import pytest,sys; sys.exit(pytest.main(['-q','-x','-m','gpu', "
                "'tests/test_gpu_solver.py::test_fused_step_draws_the_same_points_as_the_sampler_entry_point[1]']))")
        assert subprocess.run([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(__file__))).returncode == 0
        return
    L = pkg("hip.lib")
    lib = L.load()
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    torch.manual_seed(4)
    model = Solver(base_args(), Log(tmp_path), device=gpu_device)
    tr = trainer.FusedTrainer(model, 1000, capacity=4)
    for _ in range(2):
        tr.sample()
        tr.step()
    torch.cuda.synchronize()
    d = tr.fs.desc
    st = torch.cuda.current_stream(gpu_device).cuda_stream
    Xr = torch.empty_like(tr.fs.X_res)
    Xv = torch.empty_like(tr.fs.X_val)
    L.check(lib.qc_sample_collocation_faces(Xr.data_ptr(), d.B_res, d.sample_off_res, Xv.data_ptr(), d.n_ic, d.sample_off_ic,
                                            d.B_val - d.n_ic, d.sample_off_bc, d.sample_bc_face_points, d.sample_seed,
                                            d.sample_step, st))
    assert torch.equal(Xr, tr.fs.X_res) and torch.equal(Xv, tr.fs.X_val)


def test_training_with_device_sampler_reduces_loss(gpu_device, tmp_path):
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    torch.manual_seed(1)
    model = Solver(base_args(epochs=300, print_every=100), Log(tmp_path), device=gpu_device)
    trainer.train(model, batch_size=8192)
    h = np.array(model.loss_history)
    assert h.shape == (301,) and np.all(np.isfinite(h))
    assert h[-20:].mean() < 0.7 * h[:5].mean()
    assert os.path.exists(os.path.join(str(tmp_path), "model.pth"))               # saved at print_every


def test_grid_evaluation_and_checkpoint_roundtrip(gpu_device, tmp_path):
    """20^3-grid inference of (u, residual) + relative L2 errors, and a checkpoint
    written by save_state reloads into a fresh model that evaluates identically."""
    from oracle import solver as osol
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    ev = pkg("trainer.evaluate")
    z = np.load(os.path.join(GOLDEN, "train_cascade_n4_b64.npz"))
    torch.manual_seed(3)
    model = Solver(base_args(), Log(tmp_path), device=gpu_device)
    load_weights(model, z, "w1__")
    out = ev.evaluate(model, 20)
    assert out["u_pred"].shape == (20, 20, 20) and np.isfinite(out["error_u"]) and np.isfinite(out["error_f"])
    # same numbers from the oracle model (reference algorithm) on the same weights
    torch.manual_seed(3)
    om = osol.OracleSolver(base_args())
    load_weights(om, z, "w1__")
    X = torch.from_numpy(out["X"])
    idx = torch.arange(0, 8000, 97)
    Xs = X[idx]
    u_o, f_o = osol.diffusion_residual(om, Xs[:, 0:1].clone(), Xs[:, 1:2].clone(), Xs[:, 2:3].clone())
    assert np.abs(u_o.detach().numpy()[:, 0] - out["u_pred"].reshape(-1)[idx.numpy()]).max() < 2e-5
    fs = max(1.0, float(f_o.abs().max()))
    assert np.abs(f_o.detach().numpy()[:, 0] - out["f_pred"].reshape(-1)[idx.numpy()]).max() < 1e-4 * fs
    # checkpoint interchange: same keys, reload into a new model
    path = os.path.join(str(tmp_path), "ck.pth")
    model.save_state(path)
    st = Solver.load_state(path)
    m2 = Solver(base_args(), Log(tmp_path), device=gpu_device)
    m2.preprocessor.load_state_dict(st["preprocessor"])
    m2.postprocessor.load_state_dict(st["postprocessor"])
    m2.quantum_layer.load_state_dict(st["quantum_layer"])
    out2 = ev.evaluate(m2, 20)
    assert out2["error_u"] == out["error_u"] and out2["error_f"] == out["error_f"]


@pytest.mark.parametrize("over", [{"encoding": "amplitude"}, {"encoding": "amplitude", "num_qubits": 6, "q_ansatz": "layered"}])
def test_amplitude_encoding_training_matches_oracle(over, gpu_device, tmp_path):
    """Three fused training steps with encoding="amplitude" vs the CPU oracle running the reference loop."""
    from oracle import solver as osol
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    args = base_args(epochs=2, **over)
    torch.manual_seed(1)
    model = Solver(args, Log(tmp_path), device=gpu_device)
    torch.manual_seed(1)
    ref = osol.OracleSolver(args, device=torch.device("cpu"))
    torch.manual_seed(5)
    batches = [(osol.sample_box(osol.BOX_IC, 5), osol.sample_box(osol.BOX_BC1, 5), osol.sample_box(osol.BOX_DOM, 16))
               for _ in range(3)]
    trainer.train(model, batch_size=16, batches=batches)
    for b in batches:
        osol.train_step(ref, 16, b)
    got, want = np.array(model.loss_history), np.array(ref.loss_history)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (got, want)
