"""Second workload (pure-diffusion hybrid QPINN, reference train_hybrid_qpinn.py) on the HIP path against the
CPU oracle restatement of the same file (oracle/solver.py: parity unpinned, see its header): loss parts and
loss history of the fused training step on identical batches, the fused PDE operator, and the on-device
four-face boundary sampler.  Tolerances as for the first workload: 1e-4 on the loss."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _pair(gpu_device, seed=11, extra=()):
    from oracle import solver as osolver
    t = pkg("train_hybrid_qpinn")
    args = t.parse_args(["--seed", str(seed), "--epochs", "5", "--batch-size", "96", *extra])
    torch.manual_seed(seed)
    model = t.HybridQPINN(args, gpu_device)
    torch.manual_seed(seed)
    ref = osolver.OracleHybridQPINN(num_qubits=args.num_qubits, ansatz=args.ansatz, hidden=args.hidden_dim,
                                    lr=args.lr, seed=seed)
    return t, args, model, ref


@pytest.mark.parametrize("extra", [(), ("--ansatz", "layered", "--num-qubits", "3")])
def test_fused_training_matches_oracle_history(extra, gpu_device):
    from oracle import solver as osolver
    t, args, model, ref = _pair(gpu_device, extra=extra)
    B, D = args.batch_size, args.diffusion_coef
    torch.manual_seed(123)
    batches = [osolver.twin_sample(B) for _ in range(6)]
    tr = t.fused_trainer(model, B, D, capacity=6)
    for k, b in enumerate(batches):
        want = osolver.twin_train_step(ref, B, b, D)
        tr.load_batches(*b)
        tr.step()
        got, _ = tr.losses()
        for g, w in zip(got, want):
            assert abs(g - w) < 1e-4 * max(1.0, abs(w)), (k, got, want)
    hist = tr.opt.loss_history(6)
    assert np.abs(np.array(hist) - np.array(ref.loss_history)).max() < 1e-4 * max(1.0, max(ref.loss_history))
    tr.sync_to_torch()
    flat_o = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    flat_h = torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()])
    assert (flat_o - flat_h).abs().max() < 2e-3          # 6 Adam steps of lr 5e-3: sign-level noise only


def test_operator_fused_vs_oracle(gpu_device):
    from oracle import solver as osolver
    t, args, model, ref = _pair(gpu_device, seed=5)
    X = torch.rand(70, 3)
    cols = [X[:, i:i + 1].clone().to(gpu_device) for i in range(3)]
    u, res = t.diffusion_operator(model, *cols, D=0.02)
    uo, ro = osolver.diffusion_residual(ref, *[X[:, i:i + 1].clone() for i in range(3)], D=0.02, vx=0.0, vy=0.0)
    assert (u.detach().cpu() - uo.detach()).abs().max() < 2e-5
    assert (res.detach().cpu() - ro.detach()).abs().max() < 1e-4 * max(1.0, ro.abs().max().item())


def test_device_sampler_draws_the_four_faces(gpu_device):
    t, args, model, _ = _pair(gpu_device)
    B = 240
    tr = t.fused_trainer(model, B, args.diffusion_coef, capacity=4)
    tr.sample()
    tr.step()
    torch.cuda.synchronize()
    n_ic, q = B // 3, B // 12
    Xv, Xr = tr.fs.X_val.cpu(), tr.fs.X_res.cpu()
    assert Xv.shape[0] == n_ic + 4 * q and Xr.shape[0] == B
    assert (Xv[:n_ic, 0] == 0).all()
    bc = Xv[n_ic:]
    for f, (col, val) in enumerate([(1, 0.0), (1, 1.0), (2, 0.0), (2, 1.0)]):
        blk = bc[f * q:(f + 1) * q]
        assert (blk[:, col] == val).all()
        free = [c for c in (0, 1, 2) if c != col]
        assert blk[:, free].min() >= 0 and blk[:, free].max() < 1 and blk[:, free].std() > 0.1
    assert Xr.min() >= 0 and Xr.max() < 1
    first = Xv.clone()
    tr.sample()
    tr.step()
    torch.cuda.synchronize()
    assert not torch.equal(first, tr.fs.X_val.cpu())                    # a fresh batch every step


def test_train_and_evaluate_end_to_end(gpu_device, tmp_path, capsys):
    t = pkg("train_hybrid_qpinn")
    args = t.parse_args(["--epochs", "40", "--batch-size", "1200", "--print-every", "20", "--seed", "3"])
    torch.manual_seed(3)
    ics, bcs, res, dom = t.create_samplers(gpu_device, args.diffusion_coef)
    model = t.HybridQPINN(args, gpu_device)
    model = t.train(model, args, ics, bcs, res, str(tmp_path))
    assert len(model.loss_history) == 41 and model.loss_history[-1] < model.loss_history[0]
    assert "Epoch    20/40" in capsys.readouterr().out
    sd = torch.load(tmp_path / "model.pth", weights_only=True)
    assert sd["quantum_layer.params"].shape == (12,) and "preprocessor.0.weight" in sd
    ck = torch.load(tmp_path / "checkpoint.pth", weights_only=True)
    assert ck["epoch"] == 40 and len(ck["loss_history"]) == 41
    err = t.evaluate(model, args, dom, str(tmp_path))
    assert 0.0 < err < 5.0 and (tmp_path / "evaluation.json").exists()
