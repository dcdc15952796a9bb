"""Host-side mirror of the reference interface: construction, init parity, state-dict names,
error behaviour, generic (non-HIP) operator/loop, sharding arithmetic.  No GPU needed."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, pkg

from oracle import solver as osol


def base_args(**kw):
    a = {"batch_size": 64, "epochs": 2, "lr": 0.005, "seed": 1, "print_every": 100,
         "num_qubits": 4, "num_quantum_layers": 1, "classic_network": [3, 50, 1],
         "q_ansatz": "cascade", "shots": 1024, "problem": "diffusion", "solver": "DV",
         "encoding": "None", "use_ibm_hardware": False}
    a.update(kw)
    return a


class Log:
    def __init__(self, d="/tmp"):
        self.d, self.lines = str(d), []

    def print(self, *a):
        self.lines.append(" ".join(str(v) for v in a))

    def get_output_dir(self):
        return self.d


def test_quantum_layer_surface():
    QL = pkg("nn.DVQuantumLayer").DVQuantumLayer
    layer = QL(base_args())
    assert tuple(layer.params.shape) == (1, 12) and layer.params.dtype == torch.float32
    assert (layer.num_qubits, layer.num_quantum_layers, layer.q_ansatz, layer.shots) == (4, 1, "cascade", 1024)
    assert (layer.haar_seed1, layer.haar_seed2) == (1, 2) and layer.use_batch_processing is True
    small = QL(base_args(num_qubits=3))
    assert small.haar_seed1 is None and small.haar_seed2 is None           # Haar pair only for n >= 4
    noseed = QL({k: v for k, v in base_args().items() if k != "seed"})
    assert noseed.haar_seed1 is None
    with pytest.raises(ValueError):
        QL(base_args(q_ansatz="unknown"))
    with pytest.raises(KeyError):                                         # readme quick-start quirk Q5
        QL({k: v for k, v in base_args().items() if k != "problem"})
    with pytest.raises(NotImplementedError):
        QL(base_args(use_ibm_hardware=True))
    with pytest.raises(IndexError):
        QL(base_args(q_ansatz="alternate"))


@pytest.mark.parametrize("tag,over", [("cascade_n4_b64", {}), ("layered_n8_b32", {"num_qubits": 8, "num_quantum_layers": 2, "q_ansatz": "layered"})])
def test_solver_init_consumes_rng_like_the_reference(tag, over, tmp_path):
    z = np.load(os.path.join(GOLDEN, f"train_{tag}.npz"))
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    torch.manual_seed(1)
    model = Solver(base_args(**over), Log(tmp_path), device=torch.device("cpu"))
    names = [n for n, _ in model.named_parameters()]
    assert names == ["preprocessor.0.weight", "preprocessor.0.bias", "preprocessor.2.weight", "preprocessor.2.bias",
                     "postprocessor.0.weight", "postprocessor.0.bias", "postprocessor.2.weight", "postprocessor.2.bias",
                     "quantum_layer.params"]
    for n, p in model.named_parameters():
        assert np.array_equal(p.detach().numpy(), z["w0__" + n.replace(".", "__")]), n
    if not over:
        assert sum(p.numel() for p in model.parameters()) == 717
    lay = pkg("hip.engine").param_layout(50, model.num_qubits, model.quantum_layer.params.numel())
    off = 0
    for n, p in model.named_parameters():
        assert lay[n][0] == off
        off += p.numel()
    assert lay["__total__"][0] == off


def test_solver_members_and_checkpoint_keys(tmp_path):
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    model = Solver(base_args(), Log(tmp_path), device=torch.device("cpu"))
    assert isinstance(model.optimizer, torch.optim.Adam)
    assert isinstance(model.scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau)
    assert model.scheduler.factor == 0.9 and model.scheduler.patience == 1000
    assert isinstance(model.loss_fn, nn.MSELoss) and model.epochs == 2 and model.loss_history == []
    path = os.path.join(tmp_path, "m.pth")
    model.save_state(path)
    st = Solver.load_state(path)
    assert set(st) == {"args", "classic_network", "quantum_params", "preprocessor", "quantum_layer", "postprocessor",
                       "optimizer", "scheduler", "loss_history", "log_path"}
    assert list(st["quantum_layer"]) == ["params"] and list(st["preprocessor"]) == ["0.weight", "0.bias", "2.weight", "2.bias"]


def test_no_cpu_fallback(tmp_path):
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    L = pkg("hip.lib")
    log = Log(tmp_path)
    model = Solver(base_args(), log, device=torch.device("cpu"))
    with pytest.raises(L.QcError):
        model(torch.rand(5, 3))
    assert any("Forward pass failed" in s for s in log.lines)               # logged, then re-raised
    with pytest.raises(ValueError):
        model(torch.rand(5))
    with pytest.raises(L.QcError):
        pkg("nn.DVQuantumLayer").DVQuantumLayer(base_args())(torch.rand(5, 4))


@pytest.mark.parametrize("net", [[2, 50, 1], [3, 50, 3]])
def test_train_refuses_models_that_are_not_t_x_y_to_u(net, tmp_path):
    """The fused step is the (t, x, y) -> u convection-diffusion step (reference trainer/diffusion_train.py:30-49 would
    fail on the Linear(3, H) shape mismatch): a two-input or a K-output DVPDESolver must not be trained silently."""
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    model = Solver(base_args(classic_network=net, epochs=1), Log(tmp_path), device=torch.device("cpu"))
    with pytest.raises(ValueError, match="classic_network"):
        trainer.train(model, batch_size=16)


class Tiny(nn.Module):
    """A duck-typed classical model, like the reference's ClassicalSolver."""

    def __init__(self, logger):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(3, 8), nn.Tanh(), nn.Linear(8, 1))
        self.logger, self.device, self.epochs = logger, torch.device("cpu"), 3
        self.args = {"print_every": 2, "solver": "Classical"}
        self.optimizer = torch.optim.Adam(self.parameters(), lr=1e-3)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer)
        self.loss_fn, self.loss_history, self.saved = nn.MSELoss(), [], 0

    def forward(self, x):
        return self.net(x)

    def save_state(self):
        self.saved += 1


def test_generic_operator_and_loop_follow_the_reference_algorithm(tmp_path):
    pde = pkg("nn.pde")
    trainer = pkg("trainer.diffusion_train")
    torch.manual_seed(0)
    m = Tiny(Log(tmp_path))
    X = torch.rand(16, 3)
    t, x, y = (X[:, i:i + 1].clone() for i in range(3))
    u1, r1 = pde.diffusion_operator(m, t, x, y)
    t2, x2, y2 = (X[:, i:i + 1].clone() for i in range(3))
    u2, r2 = osol.diffusion_residual(m, t2, x2, y2)
    assert torch.allclose(u1, u2) and torch.allclose(r1, r2, atol=1e-6)
    assert t.requires_grad and x.requires_grad and y.requires_grad          # mutated like the reference
    trainer.train(m, batch_size=12)
    assert len(m.loss_history) == m.epochs + 1 and m.saved == 1             # epochs+1 iterations; saved at it=2


def test_shard_arithmetic():
    tr = pkg("trainer.diffusion_train")
    for total in (0, 1, 21, 64, 65536, 21845):
        for world in (1, 2, 3, 8):
            counts = [tr.shard_count(total, world, r) for r in range(world)]
            assert sum(counts) == total and max(counts) - min(counts) <= 1
            seen = []
            for r in range(world):
                s = tr.shard_slice(total, world, r)
                seen += list(range(total))[s]
            assert seen == list(range(total))


def test_logger_writes_one_line_per_print(tmp_path):
    Logging = pkg("utils.logger").Logging
    lg = Logging(str(tmp_path), experiment_name="t")
    lg.print("loss: ", 0.5, " it ", 3)
    lg.print("second")
    text = open(os.path.join(lg.get_output_dir(), "output.log")).read().splitlines()
    assert text == ["loss: 5.0000e-01 it 3", "second"]


def test_sampler_boxes_and_targets():
    data = pkg("data.diffusion_dataset")
    torch.manual_seed(0)
    ics, bcs, res = data.generate_training_dataset("cpu")
    X, Y = ics.sample(50)
    assert X.shape == (50, 3) and torch.all(X[:, 0] == 0) and torch.equal(Y, data.u(X))
    Xb, _ = bcs[0].sample(50)
    assert torch.all(Xb[:, 1] == 0) and Xb[:, 0].max() <= 1
    Xr, R = res.sample(50)
    assert torch.equal(R, data.r(Xr)) and Xr.min() >= 0 and Xr.max() <= 1
