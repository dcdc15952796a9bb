"""Generates the golden fixtures in this directory.  Run ONCE in the build container
(``python tests/golden/make_golden.py``); needs /root/reference (read-only) on disk.

What is pinned by the *reference's own code* (imported from /root/reference, never copied):
  * ``analytic.npz``  — data/diffusion_dataset.py ``u`` and ``r`` on a fixed batch;
  * ``operator_*.npz`` — nn/pde.py ``diffusion_operator`` driving the oracle-backed solver;
  * ``train_*.npz``   — trainer/diffusion_train.py ``train`` driving the oracle-backed solver
    (loss history, final weights), plus the batches its RNG produced;
  * ``other_operators.npz`` — nn/pde.py Navier-Stokes / Klein-Gordon / wave / Helmholtz operators driving an
    oracle-backed composite model (``python tests/golden/make_golden.py operators``).
What is NOT pinned by the reference (PennyLane is absent here, the reference has no fixtures):
  * ``expval_*.npz``  — oracle <Z> vectors; they freeze the oracle against later drift and are the
    GPU parity targets, but stay "parity unpinned" w.r.t. PennyLane itself.
  * ``haar.npz``      — scipy.stats.unitary_group draws for the seeds the trainers use.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")

from oracle import statevector as sv                      # noqa: E402
from oracle import solver as osol                          # noqa: E402

import data.diffusion_dataset as ref_data                  # noqa: E402  (reference)
import nn.pde as ref_pde                                   # noqa: E402  (reference)
import trainer.diffusion_train as ref_train                # noqa: E402  (reference)


def base_args(**kw):
    a = {"batch_size": 64, "epochs": 20, "lr": 0.005, "seed": 1, "print_every": 100,
         "num_qubits": 4, "num_quantum_layers": 1, "classic_network": [3, 50, 1],
         "q_ansatz": "cascade", "shots": 1024, "problem": "diffusion", "solver": "DV",
         "encoding": "None", "use_ibm_hardware": False}
    a.update(kw)
    return a


def flat_params(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy()


def flat_grads(model):
    return torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad.detach()).reshape(-1)
                      for p in model.parameters()]).numpy()


def state_arrays(model, prefix=""):
    return {prefix + k.replace(".", "__"): v.detach().numpy().copy() for k, v in model.state_dict().items()}


def make_analytic():
    g = torch.Generator().manual_seed(7)
    X = torch.rand(256, 3, generator=g, dtype=torch.float32)
    np.savez(os.path.join(HERE, "analytic.npz"), X=X.numpy(),
             u=ref_data.u(X).numpy(), r=ref_data.r(X).numpy())


def make_haar():
    out = {}
    for s in (1, 2, 42, 43):
        from scipy.stats import unitary_group
        out[f"seed{s}"] = unitary_group.rvs(4, random_state=np.random.RandomState(s))
    np.savez(os.path.join(HERE, "haar.npz"), **out)


def make_expvals():
    cases = [("cascade", 4, 1, 64, 1), ("layered", 8, 2, 64, 1), ("cross_mesh", 16, 1, 4, 1),
             ("layered", 4, 2, 32, 42), ("cross_mesh", 4, 1, 32, 1), ("farhi", 4, 1, 32, 1),
             ("sim_circ_15", 4, 1, 32, 1), ("alternate", 5, 1, 32, 1), ("cascade", 3, 2, 32, None),
             ("cascade", 5, 1, 32, 1), ("cascade", 6, 1, 16, 1), ("cross_mesh", 8, 1, 16, 1),
             ("cascade", 2, 1, 16, None), ("layered", 10, 1, 8, 1)]
    for ans, n, L, B, seed in cases:
        g = torch.Generator().manual_seed(100 + n + L)
        P = osol._P_PER_LAYER[ans](n)
        params = (torch.randn(L, P, generator=g) * 0.7).to(torch.float32)
        x = (torch.randn(B, n, generator=g) * 1.3).to(torch.float32)
        haar = sv.haar_pair(seed, seed + 1) if (seed is not None and n >= 4) else None
        q = sv.circuit_expvals(x, params, ans, n, haar)
        np.savez(os.path.join(HERE, f"expval_{ans}_n{n}_L{L}.npz"), x=x.numpy(), params=params.numpy(),
                 expval=q.numpy(), seed=np.array(-1 if seed is None else seed))
        print("expval", ans, n, L, q.shape)


def make_operator(tag, args, B, **op_kw):
    """``op_kw``: keyword arguments of the reference's diffusion_operator (sigma_t, sigma_x, sigma_y, D, v_x, v_y)."""
    torch.manual_seed(1)
    model = osol.OracleSolver(args)
    g = torch.Generator().manual_seed(11)
    X = torch.rand(B, 3, generator=g)
    t, x, y = X[:, 0:1].clone(), X[:, 1:2].clone(), X[:, 2:3].clone()
    u, res = ref_pde.diffusion_operator(model, t, x, y, **op_kw)  # the reference's operator
    # derivative channels, for debugging the jet kernels
    ones = torch.ones_like(u)
    u_t = torch.autograd.grad(u, t, ones, create_graph=True)[0]
    u_x = torch.autograd.grad(u, x, ones, create_graph=True)[0]
    u_y = torch.autograd.grad(u, y, ones, create_graph=True)[0]
    u_xx = torch.autograd.grad(u_x, x, ones, create_graph=True)[0]
    u_yy = torch.autograd.grad(u_y, y, ones, create_graph=True)[0]
    r_ref = ref_data.r(X)
    loss = 2.0 * torch.nn.functional.mse_loss(res, r_ref)
    model.zero_grad()
    loss.backward()
    a = model.preprocessor(X)
    q = model.quantum_layer(a)
    np.savez(os.path.join(HERE, f"operator_{tag}.npz"), X=X.numpy(), u=u.detach().numpy(),
             residual=res.detach().numpy(), u_t=u_t.detach().numpy(), u_x=u_x.detach().numpy(),
             u_y=u_y.detach().numpy(), u_xx=u_xx.detach().numpy(), u_yy=u_yy.detach().numpy(),
             angles=a.detach().numpy(), expval=q.detach().numpy(),
             loss=np.array(loss.item()), grad=flat_grads(model), **state_arrays(model, "w__"),
             **{"op__" + k: np.array(v) for k, v in op_kw.items()})
    print("operator", tag, float(loss))


class _Log:
    def __init__(self): self.lines = []
    def print(self, *a): self.lines.append(a)
    def get_output_dir(self): return "/tmp"


def make_train(tag, args, batch_size):
    """The reference's train() on the oracle-backed solver; then replay its RNG to capture batches,
    and check the oracle's own train_step restatement reproduces the same history."""
    torch.manual_seed(1)
    model = osol.OracleSolver(args, _Log(), device=torch.device("cpu"))
    init = state_arrays(model, "w0__")
    rng_after_init = torch.get_rng_state()
    ref_train.train(model, batch_size=batch_size)                 # the reference's loop
    hist_ref = np.array(model.loss_history)
    final = state_arrays(model, "w1__")

    # replay the sampler RNG: IC -> BC -> residual per iteration (trainer/diffusion_train.py:34-36)
    torch.set_rng_state(rng_after_init)
    steps = args["epochs"] + 1
    n3 = batch_size // 3
    X_ic = np.zeros((steps, n3, 3), np.float32)
    X_bc = np.zeros((steps, n3, 3), np.float32)
    X_res = np.zeros((steps, batch_size, 3), np.float32)
    for it in range(steps):
        X_ic[it] = osol.sample_box(osol.BOX_IC, n3).numpy()
        X_bc[it] = osol.sample_box(osol.BOX_BC1, n3).numpy()
        X_res[it] = osol.sample_box(osol.BOX_DOM, batch_size).numpy()

    # oracle restatement of the loop on the captured batches must give the same history
    torch.manual_seed(1)
    m2 = osol.OracleSolver(args, _Log(), device=torch.device("cpu"))
    parts = []
    grads0 = None
    for it in range(steps):
        b = tuple(torch.from_numpy(v[it]) for v in (X_ic, X_bc, X_res))
        if it == 0:
            m2.optimizer.zero_grad()
            l, lr_, lb, li = osol.loss_on_batches(m2, *b)
            l.backward()
            g_raw = flat_grads(m2).copy()
            torch.nn.utils.clip_grad_norm_(m2.parameters(), max_norm=1)
            grads0 = flat_grads(m2).copy()
            m2.optimizer.zero_grad()
        parts.append(osol.train_step(m2, batch_size, b))
    hist_or = np.array(m2.loss_history)
    dev = np.abs(hist_or - hist_ref).max()
    print("train", tag, "ref-vs-oracle-restatement max|dloss| =", dev)
    assert dev < 1e-9, dev
    np.savez(os.path.join(HERE, f"train_{tag}.npz"), loss_history=hist_ref, parts=np.array(parts),
             X_ic=X_ic, X_bc=X_bc, X_res=X_res, grad_raw0=g_raw, grad_clipped0=grads0,
             batch_size=np.array(batch_size), **init, **final)


class OracleComposite(torch.nn.Module):
    """A user-style model around the quantum layer: Linear(d_in,H)-Tanh-Linear(H,n) -> <Z> -> Linear(n,H)-Tanh-
    Linear(H,d_out) (the arithmetic of nn/DVPDESolver.py:81-110 with free input/output widths)."""

    def __init__(self, args, d_in, d_out, hidden=16):
        super().__init__()
        n = args["num_qubits"]
        self.pre = torch.nn.Sequential(torch.nn.Linear(d_in, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, n))
        self.q = osol.OracleQuantumLayer(args)
        self.post = torch.nn.Sequential(torch.nn.Linear(n, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, d_out))
        self.n = n

    def forward(self, X):
        q = self.q(self.pre(X)).to(torch.float32)
        return self.post(q.T.reshape(-1, self.n))


def make_other_operators():
    """nn/pde.py:2-52,73-95 of the REFERENCE (imported) driving an oracle-backed composite model: outputs,
    a scalar loss on them and its parameter gradient."""
    args = base_args(q_ansatz="cascade", num_qubits=4)
    out = {}
    cases = [("navier_stokes", ref_pde.navier_stokes_2D_operator, 3, 3), ("klein_gordon", ref_pde.klein_gordon_operator, 2, 1),
             ("wave", ref_pde.wave_operator, 2, 1), ("helmholtz", ref_pde.helmholtz_operator, 2, 1)]
    for k, (name, fn, d_in, d_out) in enumerate(cases):
        torch.manual_seed(20 + k)
        model = OracleComposite(args, d_in, d_out)
        X = torch.rand(24, d_in, generator=torch.Generator().manual_seed(50 + k), dtype=torch.float32)
        cols = [X[:, i:i + 1].clone() for i in range(d_in)]
        res = fn(model, *cols)
        res = list(res)
        loss = sum((r ** 2).mean() * (i + 1) for i, r in enumerate(res))
        model.zero_grad()
        loss.backward()
        out[f"{name}__X"] = X.numpy()
        for i, r in enumerate(res):
            out[f"{name}__out{i}"] = r.detach().numpy()
        out[f"{name}__loss"] = np.array(loss.item())
        out[f"{name}__grad"] = flat_grads(model)
        out.update(state_arrays(model, f"{name}__w__"))
    np.savez(os.path.join(HERE, "other_operators.npz"), **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "operators":
    make_other_operators()
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "r2":
    # round 2: the fused training step of the HBM family (n >= 9: two 64-point tiles with a ragged tail) and of the
    # lanes-as-amplitudes family on three tiles; diffusion_operator with sigma != 1 (nn/pde.py:53-70)
    which = sys.argv[2:] or ["hbm", "wave", "sigma"]
    if "hbm" in which:
        make_train("cross_mesh_n10_b72", base_args(epochs=3, num_qubits=10, q_ansatz="cross_mesh"), 72)
    if "wave" in which:
        make_train("layered_n8_b136", base_args(epochs=3, num_qubits=8, num_quantum_layers=2, q_ansatz="layered"), 136)
    if "sigma" in which:
        make_operator("cascade_n4_sigma", base_args(), 48, sigma_t=2.0, sigma_x=0.5, sigma_y=0.5, D=0.02, v_x=0.7, v_y=1.3)
        make_operator("layered_n8_sigma", base_args(num_qubits=8, num_quantum_layers=2, q_ansatz="layered"), 16,
                      sigma_t=2.0, sigma_x=0.5, sigma_y=0.5)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "n16":
    # the 16-qubit cross_mesh model of BASELINE config 5 (2 221 parameters), tiny batches
    a16 = base_args(num_qubits=16, q_ansatz="cross_mesh")
    make_operator("cross_mesh_n16", a16, 2)     # ~14 min of CPU: double backward through 323 gates on 2^16 amplitudes
    # (a train_* fixture at n = 16 needs more memory than this container has for the autograd graph)
    sys.exit(0)

if __name__ == "__main__":
    make_analytic()
    make_haar()
    make_expvals()
    make_operator("cascade_n4", base_args(), 64)
    make_operator("layered_n8", base_args(num_qubits=8, num_quantum_layers=2, q_ansatz="layered"), 32)
    make_operator("cross_mesh_n4", base_args(q_ansatz="cross_mesh"), 32)
    make_train("cascade_n4_b64", base_args(epochs=20), 64)
    make_train("cascade_n4_b128", base_args(epochs=8), 128)
    make_train("layered_n8_b32", base_args(epochs=5, num_qubits=8, num_quantum_layers=2, q_ansatz="layered"), 32)
    make_other_operators()
