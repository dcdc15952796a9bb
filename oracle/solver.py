"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for the quantum arithmetic
(see statevector.py); the classical parts below are pinned against the reference's own
importable files by ``tests/golden/make_golden.py``.

CPU restatement of the DV solver path around the quantum layer:
  * ``OracleQuantumLayer``  — nn/DVQuantumLayer.py:10-94,151-154,216-244 (shapes, init, Haar gating,
    (n,B) output) with ``statevector.circuit_expvals`` in place of the PennyLane QNode;
  * ``OracleSolver``        — nn/DVPDESolver.py:10-110 (MLPs, init asymmetry, cast + transpose);
  * ``diffusion_residual``  — nn/pde.py:53-72;
  * ``analytic_u`` / ``analytic_r`` / ``sample_box`` — data/diffusion_dataset.py:12-38 (incl. the -400
    constant of :31-34, which the reference's forcing term really uses);
  * ``train_step`` / ``train`` — trainer/diffusion_train.py:8-93.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import time

import torch
import torch.nn as nn

from . import statevector as sv

_P_PER_LAYER = {
    "layered": lambda n: 4 * n, "alternate": lambda n: 4 * n - 4, "cascade": lambda n: 3 * n,
    "farhi": lambda n: 2 * n - 2, "sim_circ_15": lambda n: 2 * n,
    "cross_mesh": lambda n: 4 * n + n * (n - 1),
}


class OracleQuantumLayer(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.num_qubits = args["num_qubits"]
        self.num_quantum_layers = args["num_quantum_layers"]
        self.q_ansatz = args["q_ansatz"]
        if self.q_ansatz not in _P_PER_LAYER:
            raise ValueError("Parameters are not initialized. Check the q_ansatz value.")
        P = _P_PER_LAYER[self.q_ansatz](self.num_qubits)
        self.params = nn.Parameter(torch.empty(self.num_quantum_layers, P, dtype=torch.float32))
        nn.init.xavier_normal_(self.params)                        # :216-244
        seed = args.get("seed", None) if self.num_qubits >= 4 else None   # :88-94
        self.haar_seed1 = seed
        self.haar_seed2 = seed + 1 if seed is not None else None
        self._haar = sv.haar_pair(self.haar_seed1, self.haar_seed2)
        self.encoding = args.get("encoding", "angle")

    def forward(self, x):
        return sv.circuit_expvals(x, self.params, self.q_ansatz, self.num_qubits, self._haar,
                                  "amplitude" if self.encoding == "amplitude" else "angle")


class _NullLogger:
    def __init__(self, out_dir="."):
        self._dir = out_dir
        self.lines = []

    def print(self, *a):
        self.lines.append(" ".join(str(v) for v in a))

    def get_output_dir(self):
        return self._dir


class OracleSolver(nn.Module):
    """Same construction order (hence same RNG consumption) as nn/DVPDESolver.py:10-76."""

    def __init__(self, args, logger=None, data=None, device=None):
        super().__init__()
        self.logger = logger if logger is not None else _NullLogger()
        self.device = device
        self.args = args
        self.num_qubits = args["num_qubits"]
        self.epochs = args["epochs"]
        self.loss_history = []
        cn = args["classic_network"]
        H = cn[-2]
        self.preprocessor = nn.Sequential(nn.Linear(cn[0], H), nn.Tanh(), nn.Linear(H, self.num_qubits))
        self.postprocessor = nn.Sequential(nn.Linear(self.num_qubits, H), nn.Tanh(), nn.Linear(H, cn[-1]))
        self.quantum_layer = OracleQuantumLayer(args)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=args["lr"])
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(
            self.optimizer, mode="min", factor=0.9, patience=1000)
        self.loss_fn = nn.MSELoss()
        for layer in self.preprocessor:                            # :69-76 (post keeps default init)
            if isinstance(layer, nn.Linear):
                nn.init.xavier_normal_(layer.weight)
                nn.init.zeros_(layer.bias)

    def forward(self, x):
        if x.dim() != 2:
            raise ValueError(f"Expected 2D input tensor, got shape {x.shape}")
        a = self.preprocessor(x)
        q = self.quantum_layer(a).to(torch.float32)                # :96
        if q.shape[0] == self.num_qubits and q.dim() == 2:         # :101-104
            q = q.T
        return self.postprocessor(q.reshape(-1, self.num_qubits))

    def save_state(self, path=None):                               # no file I/O in the oracle
        pass


# ------------------------------------------------------------------ data/diffusion_dataset.py
def analytic_u(txy):
    t, x, y = txy[:, 0:1], txy[:, 1:2], txy[:, 2:3]
    return torch.exp(-100.0 * ((x - 0.5) ** 2 + (y - 0.5) ** 2)) * torch.exp(-t)


def analytic_r(txy, D=0.01, vx=1.0, vy=1.0):
    """Forcing term as the reference evaluates it (data/diffusion_dataset.py:25-38): its u_xx/u_yy
    use -400 where calculus gives -200, so this equals the true residual of ``analytic_u`` plus
    4*u.  Reproduced on purpose: it is the training target."""
    x, y = txy[:, 1:2], txy[:, 2:3]
    uu = analytic_u(txy)
    ut = -uu
    ux = -200.0 * (x - 0.5) * uu
    uy = -200.0 * (y - 0.5) * uu
    uxx = (40000.0 * (x - 0.5) ** 2 - 400.0) * uu
    uyy = (40000.0 * (y - 0.5) ** 2 - 400.0) * uu
    return ut + vx * ux + vy * uy - D * (uxx + uyy)


BOX_IC = ((0.0, 0.0, 0.0), (0.0, 1.0, 1.0))       # trainer/diffusion_train.py:9-20
BOX_BC1 = ((0.0, 0.0, 0.0), (1.0, 0.0, 1.0))
BOX_DOM = ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0))


def sample_box(box, N, device=None):               # data/diffusion_dataset.py:12-19
    lo = torch.tensor([box[0]], dtype=torch.float32, device=device)
    hi = torch.tensor([box[1]], dtype=torch.float32, device=device)
    return lo + (hi - lo) * torch.rand(N, 3, device=device)


# ------------------------------------------------------------------ nn/pde.py:53-72
def diffusion_residual(model, t, x, y, D=0.01, vx=1.0, vy=1.0):
    for v in (t, x, y):
        v.requires_grad_(True)
    u = model(torch.cat((t, x, y), 1))
    one = torch.ones_like(u)
    u_t = torch.autograd.grad(u, t, one, create_graph=True)[0]
    u_x = torch.autograd.grad(u, x, one, create_graph=True)[0]
    u_y = torch.autograd.grad(u, y, one, create_graph=True)[0]
    u_xx = torch.autograd.grad(u_x, x, torch.ones_like(u_x), create_graph=True)[0]
    u_yy = torch.autograd.grad(u_y, y, torch.ones_like(u_y), create_graph=True)[0]
    return u, u_t + vx * u_x + vy * u_y - D * (u_xx + u_yy)


# ------------------------------------------------------------------ trainer/diffusion_train.py:30-49
def loss_on_batches(model, X_ic, X_bc, X_res):
    """Weighted loss on given batches; forward order BC -> IC -> residual as the reference (:40-43)."""
    u_bc = model.forward(X_bc)
    u_ic = model.forward(X_ic)
    t, x, y = X_res[:, 0:1], X_res[:, 1:2], X_res[:, 2:3]
    _, r_pred = diffusion_residual(model, t, x, y)
    l_r = model.loss_fn(r_pred, analytic_r(X_res))
    l_bc = model.loss_fn(u_bc, analytic_u(X_bc))
    l_ic = model.loss_fn(u_ic, analytic_u(X_ic))
    return 2.0 * l_r + 4.0 * l_bc + 2.0 * l_ic, l_r, l_bc, l_ic


def train_step(model, batch_size, batches=None):
    """One iteration of trainer/diffusion_train.py:52-90 (sampling order IC -> BC -> residual :34-36)."""
    model.optimizer.zero_grad()
    if batches is None:
        X_ic = sample_box(BOX_IC, batch_size // 3, model.device)
        X_bc = sample_box(BOX_BC1, batch_size // 3, model.device)
        X_res = sample_box(BOX_DOM, batch_size, model.device)
    else:
        X_ic, X_bc, X_res = batches
    loss, l_r, l_bc, l_ic = loss_on_batches(model, X_ic, X_bc, X_res)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1)      # :85 (DV branch)
    model.optimizer.step()
    model.scheduler.step(loss)
    model.loss_history.append(loss.item())
    return loss.item(), l_r.item(), l_bc.item(), l_ic.item()


def train(model, batch_size=128):
    """epochs+1 iterations, like ``for it in range(model.epochs + 1)`` (:52)."""
    t0 = time.time()
    for _ in range(model.epochs + 1):
        train_step(model, batch_size)
    return time.time() - t0


# ================================================================== second workload: train_hybrid_qpinn.py
# PARITY UNPINNED: the reference file imports PennyLane at module level (train_hybrid_qpinn.py:33), so none of
# it can be imported here and it holds no fixtures; the functions below restate it line by line.
TWIN_BOX_IC = ((0.0, 0.0, 0.0), (0.0, 1.0, 1.0))                       # train_hybrid_qpinn.py:162-164
TWIN_BOX_BC = (((0.0, 0.0, 0.0), (1.0, 0.0, 1.0)), ((0.0, 1.0, 0.0), (1.0, 1.0, 1.0)),       # :166-176 x=0, x=1,
               ((0.0, 0.0, 0.0), (1.0, 1.0, 0.0)), ((0.0, 0.0, 1.0), (1.0, 1.0, 1.0)))       #          y=0, y=1


def twin_analytic_u(X, D=0.01):                                         # :127-131
    t, x, y = X[:, 0:1], X[:, 1:2], X[:, 2:3]
    pi = torch.pi
    return torch.sin(pi * x) * torch.sin(pi * y) * torch.exp(-2 * pi ** 2 * D * t)


class OracleHybridQPINN(OracleSolver):
    """HybridQPINN of train_hybrid_qpinn.py:539-622 around the oracle layer: 1-D ``params = randn(P) * 0.1``
    (:416), xavier-normal weights / zero biases on ALL four Linear layers (:594-600), plateau patience 500
    (:583-585).  Construction order (hence RNG consumption) as there: pre, quantum params, post, re-init."""

    def __init__(self, num_qubits=4, ansatz="cascade", hidden=50, lr=0.005, seed=42, encoding="angle"):
        rng = torch.get_rng_state()
        args = {"num_qubits": num_qubits, "num_quantum_layers": 1, "q_ansatz": ansatz, "epochs": 0, "lr": lr,
                "classic_network": [3, hidden, 1], "seed": seed, "encoding": encoding}
        super().__init__(args)
        torch.set_rng_state(rng)
        twin_initialise(self, hidden)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=0.9,
                                                                    patience=500)


def twin_initialise(model, hidden):
    """Draws the initial weights in the order train_hybrid_qpinn.py:548-600 consumes the RNG and writes them
    into ``model`` (submodules ``preprocessor``, ``quantum_layer``, ``postprocessor``)."""
    n = model.num_qubits
    pre = [nn.Linear(3, hidden), nn.Linear(hidden, n)]
    theta = torch.randn(model.quantum_layer.params.numel()) * 0.1
    post = [nn.Linear(n, hidden), nn.Linear(hidden, 1)]
    for layer in pre + post:
        nn.init.xavier_normal_(layer.weight)
        nn.init.zeros_(layer.bias)
    with torch.no_grad():
        for dst, src in ((model.preprocessor[0], pre[0]), (model.preprocessor[2], pre[1]),
                         (model.postprocessor[0], post[0]), (model.postprocessor[2], post[1])):
            dst.weight.copy_(src.weight)
            dst.bias.copy_(src.bias)
        model.quantum_layer.params.copy_(theta.reshape(model.quantum_layer.params.shape))


def twin_sample(batch_size, device=None):
    """IC -> residual -> the four boundary faces (train_hybrid_qpinn.py:686-697)."""
    X_ic = sample_box(TWIN_BOX_IC, batch_size // 3, device)
    X_res = sample_box(BOX_DOM, batch_size, device)
    X_bc = torch.cat([sample_box(b, batch_size // 12, device) for b in TWIN_BOX_BC], 0)
    return X_ic, X_bc, X_res


def twin_loss_on_batches(model, X_ic, X_bc, X_res, D=0.01):
    """:704-719 — u on IC, u on BC (target 0), residual u_t - D (u_xx + u_yy) (target 0); 2, 4, 2 weights."""
    u_ic = model.forward(X_ic)
    u_bc = model.forward(X_bc)
    t, x, y = X_res[:, 0:1], X_res[:, 1:2], X_res[:, 2:3]
    _, r_pred = diffusion_residual(model, t, x, y, D=D, vx=0.0, vy=0.0)
    l_ic = model.loss_fn(u_ic, twin_analytic_u(X_ic, D))
    l_bc = model.loss_fn(u_bc, torch.zeros_like(u_bc))
    l_r = model.loss_fn(r_pred, torch.zeros_like(r_pred))
    return 2.0 * l_r + 4.0 * l_bc + 2.0 * l_ic, l_r, l_bc, l_ic


def twin_train_step(model, batch_size, batches=None, D=0.01):
    """One epoch of train_hybrid_qpinn.py:680-735."""
    model.optimizer.zero_grad()
    X_ic, X_bc, X_res = twin_sample(batch_size, model.device) if batches is None else batches
    loss, l_r, l_bc, l_ic = twin_loss_on_batches(model, X_ic, X_bc, X_res, D)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
    model.optimizer.step()
    model.scheduler.step(loss)
    model.loss_history.append(loss.item())
    return loss.item(), l_r.item(), l_bc.item(), l_ic.item()
