"""The reference's second, self-contained workload (``train_hybrid_qpinn.py``, 947 lines) on the same HIP
kernels: a hybrid quantum PINN for the pure diffusion problem

    u_t = D (u_xx + u_yy)  on [0,1]^3,   u(0,x,y) = sin(pi x) sin(pi y),   u = 0 on x,y in {0,1},
    exact solution  u = sin(pi x) sin(pi y) exp(-2 pi^2 D t)                       (reference :116-131)

with the same model family (Linear-Tanh-Linear -> n-qubit ansatz -> Linear-Tanh-Linear), the same names and
call signatures: ``parse_args`` (:50-110), ``analytical_solution[_torch]`` (:116-131), ``DataSampler`` /
``create_samplers`` (:134-203), ``QuantumLayer`` (:396-536), ``HybridQPINN`` (:539-622),
``diffusion_operator(model, t, x, y, D)`` (:625-658), ``train`` (:665-761), ``evaluate`` (:768-868), ``main``.

What differs from the first workload (trainer/diffusion_train.py) and is honoured here: 1-D quantum
parameters ``randn(P) * 0.1``; xavier-normal / zero-bias init of ALL four Linear layers; plateau patience 500;
IC batch ``B//3``, residual batch ``B``, FOUR boundary faces with ``B//12`` points each and target 0; residual
target 0 (no forcing term, no convection); checkpoint = plain ``state_dict`` files.

For a ``HybridQPINN`` one training epoch is one ``qc_fused_pinn_residual_step`` call (problem id
``QC_PROBLEM_PURE_DIFFUSION``: targets and the four-face sampler run on device).  Out of scope: ``--use-ibm``
(remote, shot-based; refused), matplotlib figures (metrics are written as JSON / npz instead).
"""
from __future__ import annotations

import argparse
import json
import os
import time
from datetime import datetime

import numpy as np
import torch
import torch.nn as nn

from .hip import lib as _lib
from .nn.DVPDESolver import DVPDESolver
from .nn.DVQuantumLayer import DVQuantumLayer
from .nn.pde import _grad

ANSATZ_CHOICES = ["cascade", "layered", "alternate", "farhi", "sim_circ_15", "cross_mesh"]


# ---------------------------------------------------------------------------------------------- 1. arguments
def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Hybrid Quantum PINN Trainer for 2D PDEs (MI355X HIP path)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--device", type=str, default="auto", choices=["auto", "cuda", "cpu"],
                   help="compute device (the HIP kernels need the GPU; 'cpu' only builds the model)")
    p.add_argument("--use-ibm", action="store_true", help="IBM Quantum hardware branch of the reference (refused here)")
    p.add_argument("--ibm-token", type=str, default=None)
    p.add_argument("--ibm-backend", type=str, default="ibm_torino")
    p.add_argument("--ibm-instance", type=str, default=None)
    p.add_argument("--num-qubits", type=int, default=4)
    p.add_argument("--ansatz", type=str, default="cascade", choices=ANSATZ_CHOICES)
    p.add_argument("--encoding", type=str, default="angle", choices=["angle", "amplitude"])
    p.add_argument("--shots", type=int, default=1024)
    p.add_argument("--epochs", type=int, default=5000)
    p.add_argument("--batch-size", type=int, default=64)
    p.add_argument("--lr", type=float, default=0.005)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--hidden-dim", type=int, default=50)
    p.add_argument("--print-every", type=int, default=100)
    p.add_argument("--output-dir", type=str, default="./outputs")
    p.add_argument("--diffusion-coef", type=float, default=0.01, help="diffusion coefficient D")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------- 2. data
def analytical_solution(t, x, y, D=0.01):
    return np.sin(np.pi * x) * np.sin(np.pi * y) * np.exp(-2 * np.pi ** 2 * D * t)


def analytical_solution_torch(X, D=0.01):
    t, x, y = X[:, 0:1], X[:, 1:2], X[:, 2:3]
    return torch.sin(torch.pi * x) * torch.sin(torch.pi * y) * torch.exp(-2 * torch.pi ** 2 * D * t)


class DataSampler:
    """Uniform points in the box ``coords = [lo; hi]`` with targets ``func(X[, D])`` (reference :134-156)."""

    def __init__(self, coords, func, device="cpu", D=0.01):
        self.coords, self.func, self.device, self.D = coords, func, device, D
        self.dim = coords.shape[1]

    def sample(self, N):
        X = self.coords[0:1, :] + (self.coords[1:2, :] - self.coords[0:1, :]) * torch.rand(N, self.dim, device=self.device)
        return X, (self.func(X, self.D) if self.D is not None else self.func(X))


_FACES = (((0.0, 0.0, 0.0), (1.0, 0.0, 1.0)), ((0.0, 1.0, 0.0), (1.0, 1.0, 1.0)),      # x = 0, x = 1
          ((0.0, 0.0, 0.0), (1.0, 1.0, 0.0)), ((0.0, 0.0, 1.0), (1.0, 1.0, 1.0)))      # y = 0, y = 1


def _zeros(X, D=None):
    return torch.zeros((X.shape[0], 1), device=X.device)


def create_samplers(device, D=0.01):
    """-> (ics_sampler, [4 bc_samplers], res_sampler, dom_coords), boxes of the reference (:159-203).  The
    returned objects carry ``standard = True``: ``train`` then draws the same boxes on device."""
    def box(lo_hi):
        return torch.tensor(lo_hi, dtype=torch.float32, device=device)

    ics = DataSampler(box(((0.0, 0.0, 0.0), (0.0, 1.0, 1.0))), analytical_solution_torch, device, D)
    bcs = [DataSampler(box(f), _zeros, device, None) for f in _FACES]
    dom = box(((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    res = DataSampler(dom, _zeros, device, None)
    for s in [ics, res] + bcs:
        s.standard = True
    return ics, bcs, res, dom


# ---------------------------------------------------------------------------------------------- 3./4. layer
class QuantumLayer(DVQuantumLayer):
    """The reference's ``QuantumLayer`` constructor signature (:399-401) over the HIP statevector kernels:
    one ansatz layer, ``params`` of shape ``(P,)`` drawn as ``randn * 0.1`` (:416), Haar seeds ``seed, seed+1`` for
    4+ qubits (:419-420).  ``forward((B, n)) -> (B, n)`` as there (:503-505)."""

    def __init__(self, num_qubits, ansatz_type="cascade", encoding="angle", use_ibm=False, ibm_token=None,
                 ibm_backend=None, ibm_instance=None, shots=1024, seed=42):
        if use_ibm:
            raise NotImplementedError("use_ibm=True selects the IBM Runtime / shot-based branch of the reference, "
                                      "which is outside the MI355X simulator path")
        super().__init__({"num_qubits": num_qubits, "num_quantum_layers": 1, "q_ansatz": ansatz_type,
                          "problem": "diffusion", "encoding": encoding, "shots": shots, "seed": seed,
                          "use_ibm_hardware": False})
        self.ansatz_type = ansatz_type
        self.use_ibm = False
        self.params = nn.Parameter(torch.randn(self.params.numel()) * 0.1)

    def forward(self, x):
        return super().forward(x).T


# ---------------------------------------------------------------------------------------------- 5. model
def _solver_args(args):
    return {"batch_size": args.batch_size, "epochs": args.epochs, "lr": args.lr, "seed": args.seed,
            "print_every": args.print_every, "num_qubits": args.num_qubits, "num_quantum_layers": 1,
            "classic_network": [3, args.hidden_dim, 1], "q_ansatz": args.ansatz, "shots": args.shots,
            "problem": "diffusion", "solver": "DV", "encoding": args.encoding, "use_ibm_hardware": False}


class _Quiet:
    def __init__(self, out_dir="."):
        self._dir = out_dir

    def print(self, *a):
        pass

    def get_output_dir(self):
        return self._dir


class HybridQPINN(DVPDESolver):
    """``HybridQPINN(args, device)`` of the reference (:539-622): ``args`` is the argparse namespace.  Built on
    ``DVPDESolver`` (flat parameter buffer, fused kernels); initial weights are drawn in the order the reference
    consumes the RNG (preprocessor, quantum parameters, postprocessor, then xavier-normal on all four Linear
    layers), so a seeded run starts from the weights the reference would start from."""

    def __init__(self, args, device):
        if getattr(args, "use_ibm", False):
            raise NotImplementedError("--use-ibm selects the IBM Runtime / shot-based branch of the reference, "
                                      "which is outside the MI355X simulator path")
        rng = torch.get_rng_state()
        super().__init__(_solver_args(args), _Quiet(getattr(args, "output_dir", ".")), device=None)
        torch.set_rng_state(rng)
        self.args_ns = args
        self.hidden_dim = args.hidden_dim
        self.quantum_layer = QuantumLayer(args.num_qubits, args.ansatz, args.encoding, False, None, None, None,
                                          args.shots, args.seed)
        n, H = args.num_qubits, args.hidden_dim
        torch.set_rng_state(rng)
        fresh_pre = [nn.Linear(3, H), nn.Linear(H, n)]                       # default init: consumes the RNG (:548-552)
        theta = torch.randn(self.quantum_layer.params.numel()) * 0.1         # (:416)
        fresh_post = [nn.Linear(n, H), nn.Linear(H, 1)]                      # (:568-572)
        for layer in fresh_pre + fresh_post:                                 # _init_weights (:594-600)
            nn.init.xavier_normal_(layer.weight)
            nn.init.zeros_(layer.bias)
        with torch.no_grad():
            for dst, src in ((self.preprocessor[0], fresh_pre[0]), (self.preprocessor[2], fresh_pre[1]),
                             (self.postprocessor[0], fresh_post[0]), (self.postprocessor[2], fresh_post[1])):
                dst.weight.copy_(src.weight)
                dst.bias.copy_(src.bias)
            self.quantum_layer.params.copy_(theta)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=args.lr)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=0.9,
                                                                    patience=500)
        self.draw_quantum_circuit_flag = False
        self.device = device
        target = self._resolve_device(device)
        if target is not None and target.type == "cuda":
            self.device = target
            self._pack(target)


# ---------------------------------------------------------------------------------------------- 6. operator
def diffusion_operator(model, t, x, y, D=0.01):
    """(u, residual) with residual = u_t - D (u_xx + u_yy)  (reference :625-658).  A ``HybridQPINN`` /
    ``DVPDESolver`` takes the fused derivative channels; any other model the autograd formulation."""
    t, x, y = t.requires_grad_(True), x.requires_grad_(True), y.requires_grad_(True)
    X = torch.cat([t, x, y], dim=1)
    fused = getattr(model, "residual", None)
    if fused is not None and hasattr(model, "quantum_layer"):
        return fused(X, D=D, v_x=0.0, v_y=0.0)
    u = model(X)
    u_t, u_x, u_y = _grad(u, t), _grad(u, x), _grad(u, y)
    return u, u_t - D * (_grad(u_x, x) + _grad(u_y, y))


# ---------------------------------------------------------------------------------------------- 7. training
def fused_trainer(model, batch_size, D, capacity, sampler="device"):
    from .trainer.diffusion_train import FusedTrainer
    return FusedTrainer(model, batch_size, capacity, sampler, n_bc=4 * (batch_size // 12), bc_faces=4,
                        pde={"D": D, "vx": 0.0, "vy": 0.0, "problem": _lib.QC_PROBLEM_PURE_DIFFUSION})


def _epoch_line(epoch, args, parts, lr, total_time, avg):
    loss, l_r, l_bc, l_ic = parts
    pct = 100 * epoch / args.epochs if args.epochs else 100.0
    return (f"Epoch {epoch:5d}/{args.epochs} [{pct:5.1f}%] | Loss: {loss:.2e} | Res: {l_r:.2e} | BC: {l_bc:.2e} | "
            f"IC: {l_ic:.2e} | LR: {lr:.2e} | Time: {total_time:.1f}s | ETA: {avg * (args.epochs - epoch):.1f}s")


def train(model, args, ics_sampler, bc_samplers, res_sampler, output_dir):
    """``epochs + 1`` epochs of the reference loop (:680-735): IC ``B//3``, residual ``B``, 4 x ``B//12`` boundary
    points; ``loss = 2 L_res + 4 L_bc + 2 L_ic``; clip 1.0; Adam; plateau scheduler; ``checkpoint.pth`` at
    ``print_every``, ``model.pth`` at the end.  Returns the model."""
    D, B = args.diffusion_coef, args.batch_size
    if not isinstance(model, HybridQPINN):
        return _train_autograd(model, args, ics_sampler, bc_samplers, res_sampler, output_dir)
    standard = all(getattr(s, "standard", False) for s in [ics_sampler, res_sampler] + list(bc_samplers))
    tr = fused_trainer(model, B, D, capacity=args.epochs + 1)
    t0 = time.time()
    done = 0
    for epoch in range(args.epochs + 1):
        if standard:
            tr.sample()
        else:                                   # caller-supplied boxes: coordinates from them, targets stay analytic
            X_ic, _ = ics_sampler.sample(B // 3)
            X_res, _ = res_sampler.sample(B)
            X_bc = torch.cat([s.sample(B // 12)[0] for s in bc_samplers], 0)
            tr.load_batches(X_ic, X_bc, X_res)
        tr.step()
        if epoch % args.print_every == 0 or epoch == 0:
            parts, lr = tr.losses()             # one device read-back per print_every epochs
            el = time.time() - t0
            print(_epoch_line(epoch, args, parts, lr, el, el / (epoch + 1)))
            if epoch > 0:
                hist = tr.opt.loss_history(epoch + 1)
                model.loss_history.extend(hist[done:])
                done = len(hist)
                tr.sync_to_torch()
                torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                            "optimizer_state_dict": model.optimizer.state_dict(), "loss": parts[0],
                            "loss_history": model.loss_history}, os.path.join(output_dir, "checkpoint.pth"))
    hist = tr.opt.loss_history(args.epochs + 1)
    model.loss_history.extend(hist[done:])
    tr.sync_to_torch()
    print(f"\nTraining completed in {time.time() - t0:.1f}s")
    torch.save(model.state_dict(), os.path.join(output_dir, "model.pth"))
    model._fused_trainer = tr
    return model


def _train_autograd(model, args, ics_sampler, bc_samplers, res_sampler, output_dir):
    """The same loop for any other model, on torch autograd (the reference's algorithm)."""
    D, B = args.diffusion_coef, args.batch_size
    for epoch in range(args.epochs + 1):
        model.optimizer.zero_grad()
        X_ic, u_ic = ics_sampler.sample(B // 3)
        X_res, _ = res_sampler.sample(B)
        drawn = [s.sample(B // 12) for s in bc_samplers]
        X_bc, u_bc = torch.cat([d[0] for d in drawn], 0), torch.cat([d[1] for d in drawn], 0)
        pred_ic, pred_bc = model(X_ic), model(X_bc)
        _, res = diffusion_operator(model, X_res[:, 0:1], X_res[:, 1:2], X_res[:, 2:3], D)
        l_ic, l_bc = model.loss_fn(pred_ic, u_ic), model.loss_fn(pred_bc, u_bc)
        l_r = model.loss_fn(res, torch.zeros_like(res))
        loss = 2.0 * l_r + 4.0 * l_bc + 2.0 * l_ic
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        model.optimizer.step()
        model.scheduler.step(loss)
        model.loss_history.append(loss.item())
    torch.save(model.state_dict(), os.path.join(output_dir, "model.pth"))
    return model


# ---------------------------------------------------------------------------------------------- 8. evaluation
def evaluate(model, args, dom_coords, output_dir):
    """Relative L2 error against the exact solution on the 20 x 20 (x, y) grid at t = 0.5 (:768-809); the
    fields and the loss history go to ``evaluation.npz`` / ``evaluation.json`` (no figures here)."""
    device, D, num = model.device, args.diffusion_coef, 20
    xs = torch.linspace(0, 1, num, device=device)
    Xm, Ym = torch.meshgrid(xs, xs, indexing="ij")
    X_eval = torch.stack([torch.full_like(Xm.flatten(), 0.5), Xm.flatten(), Ym.flatten()], dim=1)
    model.eval()
    with torch.no_grad():
        u_pred = model(X_eval)
    u_pred = u_pred.cpu().numpy().reshape(num, num)
    u_true = analytical_solution_torch(X_eval, D).cpu().numpy().reshape(num, num)
    error = float(np.linalg.norm(u_true - u_pred) / np.linalg.norm(u_true))
    print(f"Relative L2 Error at t=0.5: {error * 100:.4f}%")
    np.savez(os.path.join(output_dir, "evaluation.npz"), u_pred=u_pred, u_analytical=u_true,
             loss_history=np.asarray(model.loss_history, dtype=np.float64))
    with open(os.path.join(output_dir, "evaluation.json"), "w") as f:
        json.dump({"relative_l2_error_t0.5": error, "epochs": len(model.loss_history)}, f)
    return error


# ---------------------------------------------------------------------------------------------- 9. main
def main(argv=None):
    args = parse_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu") if args.device == "auto" else torch.device(args.device)
    print(f"Device: {device} | Qubits: {args.num_qubits} | Ansatz: {args.ansatz} | Encoding: {args.encoding} | "
          f"Epochs: {args.epochs} | Batch size: {args.batch_size} | LR: {args.lr} | D: {args.diffusion_coef}")
    output_dir = os.path.join(args.output_dir, datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, "config.txt"), "w") as f:
        for key, value in vars(args).items():
            f.write(f"{key}: {'****' if key == 'ibm_token' and value else value}\n")
    ics, bcs, res, dom = create_samplers(device, D=args.diffusion_coef)
    model = HybridQPINN(args, device)
    print(f"Total parameters: {sum(p.numel() for p in model.parameters())} | "
          f"Quantum parameters: {model.quantum_layer.params.numel()}")
    model = train(model, args, ics, bcs, res, output_dir)
    error = evaluate(model, args, dom, output_dir)
    print(f"Final L2 Error: {error * 100:.4f}% | results in {output_dir}")
    return error


if __name__ == "__main__":
    main()
