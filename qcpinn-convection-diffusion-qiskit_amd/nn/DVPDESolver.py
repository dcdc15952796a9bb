"""``DVPDESolver`` — drop-in for the reference solver module (nn/DVPDESolver.py:9-157):
``Linear(3,H)-Tanh-Linear(H,n)`` -> ``DVQuantumLayer`` -> ``Linear(n,H)-Tanh-Linear(H,1)`` with the
same constructor ``(args, logger, data=None, device=None)``, attribute names, sub-module /
state-dict names, construction order (so ``torch.manual_seed(s)`` gives the reference's initial
weights), Adam + ReduceLROnPlateau + MSELoss members and checkpoint dictionary.

The arithmetic of ``forward`` — both networks and the circuit — runs in the HIP kernels of
``libqcpinn_hip.so``.  To let the fused training step update everything in one kernel, all
parameters are views into ONE flat fp32 device buffer laid out in ``model.parameters()`` order.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ..hip import engine as _engine
from ..hip.lib import QcError
from .DVQuantumLayer import DVQuantumLayer


def _flat_param_list(model):
    return [p for p in model.parameters()]


class _ValueFn(torch.autograd.Function):
    """X (B,3) -> u (B,1); differentiable w.r.t. the solver parameters.  Inputs that require gradients take
    ``DVPDESolver._forward_wrt_inputs`` instead (value channel only here: one sixth of the work)."""

    @staticmethod
    def forward(ctx, X, solver, *params):
        eng = solver._engine_for(X.device)
        Xc = solver._pad_inputs(X.detach().to(torch.float32)).contiguous()
        u, _, ajets, qjets = eng.forward(Xc, 1)
        ctx.eng, ctx.solver = eng, solver
        ctx.save_for_backward(Xc, ajets, qjets)
        return u

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gu):
        Xc, ajets, qjets = ctx.saved_tensors
        ctx.eng.refresh_gates()
        d_flat = ctx.eng.backward(Xc, ajets, qjets, gu, None, 1)
        return (None, None) + ctx.solver._split_flat(d_flat)


class _ResidualFn(torch.autograd.Function):
    """X (B,3) -> (u, residual) of the convection-diffusion operator, both (B,1)."""

    @staticmethod
    def forward(ctx, X, solver, pde, *params):
        eng = solver._engine_for(X.device)
        eng.D, eng.vx, eng.vy, eng.sigma, eng.coeffs = pde
        Xc = solver._pad_inputs(X.detach().to(torch.float32)).contiguous()
        u, res, ajets, qjets = eng.forward(Xc, _engine.NCH)
        ctx.eng, ctx.solver, ctx.pde = eng, solver, pde
        ctx.save_for_backward(Xc, ajets, qjets)
        return u, res

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gu, gres):
        Xc, ajets, qjets = ctx.saved_tensors
        eng = ctx.eng
        eng.D, eng.vx, eng.vy, eng.sigma, eng.coeffs = ctx.pde
        eng.refresh_gates()
        d_flat = eng.backward(Xc, ajets, qjets, gu, gres, _engine.NCH)
        return (None, None, None) + ctx.solver._split_flat(d_flat)


class _JetsFn(torch.autograd.Function):
    """X (B,3) and ONE flat single-output parameter vector (kernel layout) -> the six derivative channels of that
    output, (B,6) = (u, u_t, u_x, u_y, u_xx, u_yy); the reverse pass takes a cotangent per channel (qc_post modes 4 / 3).
    The flat vector is assembled from the module's parameters by torch ops, so autograd carries its gradient on to
    them: a K-output model is K single-output evaluations that share everything but one row of the last layer."""

    @staticmethod
    def forward(ctx, X, solver, flat):
        eng = solver._jet_engine(X.device)
        Xc = solver._pad_inputs(X.detach().to(torch.float32)).contiguous()
        fl = flat.detach().to(torch.float32).contiguous()
        eng.flat.copy_(fl)
        uj, _, ajets, qjets = eng.forward(Xc, _engine.NCH, ujets=True)
        ctx.eng = eng
        ctx.save_for_backward(Xc, ajets, qjets, fl)
        return uj.t().contiguous()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        Xc, ajets, qjets, fl = ctx.saved_tensors
        eng = ctx.eng
        eng.flat.copy_(fl)              # (another output's evaluation may have used the engine in between)
        eng.refresh_gates()
        d_flat = eng.backward(Xc, ajets, qjets, g.to(torch.float32).t().contiguous(), None, _engine.NCH, ujets=True)
        return None, None, d_flat


class _JetsAllFn(torch.autograd.Function):
    """X (B,3), the flat vector of the SHARED parameters (single-output kernel layout, W4 / b4 slots unused) and the
    last layer's rows w4k [K, H + 1] -> (B, K, 6): the six derivative channels of every output from ONE pass through pre
    network, circuit and hidden layer (qc_post_multi); one adjoint sweep in the reverse pass.  Both parameter arguments
    are assembled from the module's parameters by torch ops, so autograd carries the gradients on to them."""

    @staticmethod
    def forward(ctx, X, solver, flat, w4k):
        eng = solver._jet_engine(X.device)
        Xc = solver._pad_inputs(X.detach().to(torch.float32)).contiguous()
        fl = flat.detach().to(torch.float32).contiguous()
        wk = w4k.detach().to(torch.float32).contiguous()
        eng.flat.copy_(fl)
        uj, ajets, qjets = eng.forward_multi(Xc, wk)
        ctx.eng = eng
        ctx.save_for_backward(Xc, ajets, qjets, fl, wk)
        return uj.permute(2, 0, 1).contiguous()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        Xc, ajets, qjets, fl, wk = ctx.saved_tensors
        eng = ctx.eng
        eng.flat.copy_(fl)
        eng.refresh_gates()
        d_flat, d_wk = eng.backward_multi(Xc, ajets, qjets, wk, g.to(torch.float32).permute(1, 2, 0).contiguous())
        return None, None, d_flat, d_wk


class _NoSecond(torch.autograd.Function):
    """Zero-valued guard of the input expansion (``DVPDESolver._forward_wrt_inputs``): contributes nothing to u and to its
    first derivatives, and makes every second derivative the kernels do NOT carry come out as NaN instead of a silent
    zero.  ``mask`` (k, k) marks the carried second derivatives (the diagonal entries of the coordinate slots that own
    a second-derivative channel)."""

    @staticmethod
    def forward(ctx, delta, mask):
        ctx.mask = mask
        ctx.save_for_backward(delta)
        return delta.new_zeros(delta.shape[0], 1)

    @staticmethod
    def backward(ctx, g):
        (delta,) = ctx.saved_tensors
        return _NoSecondGrad.apply(g, delta, ctx.mask), None


class _NoSecondGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, delta, mask):
        ctx.mask = mask
        return torch.zeros_like(delta)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, c):
        # c (B, k): cotangent of the first-derivative vector.  Entry j of the result is sum_i c_i d2u/dX_i dX_j of the
        # terms this guard stands for: NaN wherever a requested (i, j) is not carried
        mask = ctx.mask.to(c.device)
        asked = (c != 0).to(c.dtype)                       # (B, k)
        missing = asked @ (~mask).to(c.dtype)              # (B, k): > 0 where column j meets an uncarried (i, j)
        out = torch.where(missing > 0, torch.full_like(c, float("nan")), torch.zeros_like(c))
        return None, out, None


class DVPDESolver(nn.Module):
    def __init__(self, args, logger, data=None, device=None):
        super().__init__()
        self.logger = logger
        self.device = device
        self.args = args
        self.data = data
        self.batch_size = self.args["batch_size"]
        self.num_qubits = self.args["num_qubits"]
        self.epochs = self.args["epochs"]
        self.optimizer = None
        self.scheduler = None
        self.loss_history = []
        self.encoding = self.args.get("encoding", "angle")
        self.draw_quantum_circuit_flag = True
        self.classic_network = self.args["classic_network"]
        self.total_training_time = 0
        self.total_memory_peak = 0
        if self.classic_network[0] not in (2, 3) or self.classic_network[-1] < 1:
            raise ValueError("the DV path maps (t, x, y) -> u (or two coordinates -> u for the second-order operators of "
                             f"nn/pde.py): classic_network must be [3, H, K] or [2, H, K], got {self.classic_network}")
        self.input_dim = self.classic_network[0]
        # K > 1 outputs (Navier-Stokes' (u, v, p), nn/pde.py:2-27): served through jets(), one single-output
        # evaluation of the kernels per output; the fused TRAINING step and residual() are for K = 1
        self.n_out = self.classic_network[-1]
        hidden = self.classic_network[-2]

        # same construction order as the reference => same RNG consumption (nn/DVPDESolver.py:28-57)
        self.preprocessor = nn.Sequential(nn.Linear(self.classic_network[0], hidden), nn.Tanh(),
                                          nn.Linear(hidden, self.num_qubits))
        self.postprocessor = nn.Sequential(nn.Linear(self.num_qubits, hidden), nn.Tanh(),
                                           nn.Linear(hidden, self.classic_network[-1]))
        self.activation = nn.Tanh()
        self.quantum_layer = DVQuantumLayer(self.args)

        self.optimizer = torch.optim.Adam(filter(lambda p: p.requires_grad, self.parameters()), lr=self.args["lr"])
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode="min", factor=0.9,
                                                                    patience=1000)
        self.loss_fn = torch.nn.MSELoss()
        self.log_path = self.logger.get_output_dir() if self.logger is not None else "."
        for layer in self.preprocessor:          # only the preprocessor is re-initialised (:69-76)
            if isinstance(layer, nn.Linear):
                nn.init.xavier_normal_(layer.weight)
                nn.init.zeros_(layer.bias)

        self._flat = None
        self._engines = {}
        self._fused_opt = None
        self._jet_engines = {}
        target = self._resolve_device(device)
        if target is not None:
            self.device = target         # samplers of reference-style loops read model.device
            if self.n_out == 1:
                self._pack(target)
            else:
                self.to(target)

    # ------------------------------------------------------------------ device / flat storage
    @staticmethod
    def _resolve_device(device):
        """None (what the stock reference scripts end up passing) means "the GPU if
        there is one"; construction alone never needs a GPU."""
        if device is None:
            return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        device = torch.device(device)
        if device.type == "cuda" and not torch.cuda.is_available():
            return None
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        return device

    @property
    def hidden_width(self) -> int:
        return self.classic_network[-2]

    def _slots(self):
        """(parameter, flat slots it occupies) in ``model.parameters()`` order.  A two-input model keeps its first
        layer as W1[H][3] in the flat buffer with a zero column in front (the kernels' coordinate slots are (t, x, y);
        the two inputs ride in the x and y slots, the ones with second derivatives)."""
        out = []
        for p in _flat_param_list(self):
            pad = self.input_dim == 2 and p is self.preprocessor[0].weight
            out.append((p, p.shape[0] * 3 if pad else p.numel(), pad))
        return out

    def _pack(self, device) -> None:
        """Move every parameter into one flat buffer on ``device`` and re-point the Parameters at
        views of it (``model.parameters()`` order = kernel layout, see hip/engine.param_layout)."""
        slots = self._slots()
        parts = []
        for p, k, pad in slots:
            v = p.detach().to(device=device, dtype=torch.float32)
            if pad:
                v = torch.cat([torch.zeros(v.shape[0], 1, device=device), v], dim=1)
            parts.append(v.reshape(-1))
        flat = torch.cat(parts)
        off = 0
        for p, k, pad in slots:
            p.data = flat[off:off + k].view(p.shape[0], 3)[:, 1:3] if pad else flat[off:off + k].view(p.shape)
            off += k
        self._flat = flat
        self._engines = {}
        self._fused_opt = None

    def _packed_ok(self) -> bool:
        if self._flat is None:
            return False
        off = 0
        for p, k, pad in self._slots():
            if p.device != self._flat.device or p.data_ptr() != self._flat.data_ptr() + 4 * (off + (1 if pad else 0)):
                return False
            off += k
        return off == self._flat.numel()

    def _pad_inputs(self, X):
        """(B, 2) inputs of a two-input model -> the kernels' (t, x, y) slots: (0, X[:, 0], X[:, 1])."""
        if self.input_dim == 2:
            if X.shape[1] != 2:
                raise ValueError(f"this model takes (B, 2) inputs, got {tuple(X.shape)}")
            return torch.cat([torch.zeros_like(X[:, :1]), X], dim=1)
        return X

    def _jet_engine(self, device) -> "_engine.SolverEngine":
        """Engine over a scratch flat vector in the single-output kernel layout (see _JetsFn)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise QcError(f"DVPDESolver computes on the GPU only (HIP kernels, no CPU fallback); input is on {device}")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = device.index
        if key not in self._jet_engines:
            circ = self.quantum_layer._circuit_for(device)
            H, n = self.hidden_width, self.num_qubits
            NP = 3 * H + H + n * H + n + H * n + H + H + 1 + circ.n_params
            self._jet_engines[key] = _engine.SolverEngine(circ, H, torch.zeros(NP, dtype=torch.float32, device=device))
        return self._jet_engines[key]

    def _flat_for_output(self, out: int) -> torch.Tensor:
        """The parameters of output ``out`` as one flat vector in the kernels' single-output layout
        (W1[H][3] b1 W2 b2 | W3 b3 W4[out] b4[out] | theta), built by differentiable torch ops."""
        pre0, pre2 = self.preprocessor[0], self.preprocessor[2]
        post0, post2 = self.postprocessor[0], self.postprocessor[2]
        W1 = pre0.weight
        if self.input_dim == 2:
            W1 = torch.cat([torch.zeros_like(W1[:, :1]), W1], dim=1)
        parts = [W1.reshape(-1), pre0.bias, pre2.weight.reshape(-1), pre2.bias, post0.weight.reshape(-1), post0.bias,
                 post2.weight[out], post2.bias[out:out + 1]]
        parts += [p.reshape(-1) for p in self.quantum_layer.parameters()]
        return torch.cat([p.to(torch.float32) for p in parts])

    def jets_all(self, X: torch.Tensor) -> torch.Tensor:
        """(B, K, 6): (u_k, u_k,t, u_k,x, u_k,y, u_k,xx, u_k,yy) of every output k of a K-output model (K <= 4) from ONE
        evaluation of pre network, circuit and hidden layer, and one adjoint sweep in the reverse pass (what the
        Navier-Stokes operator of ``nn.pde`` uses for (u, v, p), reference nn/pde.py:2-27)."""
        if X.dim() != 2 or X.shape[1] != self.input_dim:
            raise ValueError(f"Expected points of shape (B, {self.input_dim}), got {tuple(X.shape)}")
        if self.n_out > 4:
            return torch.stack([self.jets(X, o) for o in range(self.n_out)], dim=1)
        self._jet_engine(X.device)
        post2 = self.postprocessor[2]
        w4k = torch.cat([post2.weight, post2.bias[:, None]], dim=1).to(torch.float32)
        flat = self._flat_for_output(0)          # the W4 / b4 slots are not read by the K-output post stage
        return _JetsAllFn.apply(X, self, flat, w4k)

    def jets(self, X: torch.Tensor, out: int = 0) -> torch.Tensor:
        """(B, 6) = (u, u_t, u_x, u_y, u_xx, u_yy) of output ``out`` at X (B, input_dim), from the fused derivative-channel
        kernels; differentiable w.r.t. the parameters.  For operators that are not linear in the channels
        (``nn.pde.navier_stokes_2D_operator``) and for K-output models."""
        if X.dim() != 2 or X.shape[1] != self.input_dim:
            raise ValueError(f"Expected points of shape (B, {self.input_dim}), got {tuple(X.shape)}")
        if not 0 <= out < self.n_out:
            raise ValueError(f"output index {out} outside [0, {self.n_out})")
        self._jet_engine(X.device)
        return _JetsFn.apply(X, self, self._flat_for_output(out))

    def _engine_for(self, device) -> "_engine.SolverEngine":
        if self.n_out != 1:
            raise NotImplementedError("residual() and the fused training step serve single-output models; a "
                                      f"{self.n_out}-output model is evaluated through jets(X, out)")
        device = torch.device(device)
        if device.type != "cuda":
            raise QcError(f"DVPDESolver computes on the GPU only (HIP kernels, no CPU fallback); input is on {device}")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if not self._packed_ok() or self._flat.device != device:
            self._pack(device)           # e.g. after model.to(...), load_state_dict into new tensors
        key = device.index
        if key not in self._engines:
            circ = self.quantum_layer._circuit_for(device)
            self._engines[key] = _engine.SolverEngine(circ, self.hidden_width, self._flat)
        return self._engines[key]

    def _split_flat(self, d_flat):
        out, off = [], 0
        for p, k, pad in self._slots():
            out.append(d_flat[off:off + k].view(p.shape[0], 3)[:, 1:3] if pad else d_flat[off:off + k].view(p.shape))
            off += k
        return tuple(out)

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        try:
            if x.dim() != 2:
                raise ValueError(f"Expected 2D input tensor, got shape {x.shape}")
            if self.draw_quantum_circuit_flag:
                self.draw_quantum_circuit(x)
                self.draw_quantum_circuit_flag = False
            if self.n_out != 1:       # (B, K): the value channel of every output
                if x.shape[1] != self.input_dim:
                    raise ValueError(f"Expected input of shape (B, {self.input_dim}), got {tuple(x.shape)}")
                return torch.cat([self.jets(x, o)[:, 0:1] for o in range(self.n_out)], dim=1)
            self._engine_for(x.device)
            if x.shape[1] != self.input_dim:
                raise ValueError(f"Expected input of shape (B, {self.input_dim}), got {tuple(x.shape)}")
            if x.requires_grad and torch.is_grad_enabled():
                return self._forward_wrt_inputs(x)
            return _ValueFn.apply(x, self, *_flat_param_list(self))
        except Exception as e:
            if self.logger is not None:
                self.logger.print(f"Forward pass failed: {str(e)}")
            raise

    def _forward_wrt_inputs(self, x: torch.Tensor) -> torch.Tensor:
        """u (B, 1) as an ordinary autograd graph in the INPUTS as well (reference nn/DVPDESolver.py:81-110 is a plain
        differentiable module, and trainer/diffusion_train.py:37-39 / nn/pde.py:59-70 differentiate it w.r.t. t, x, y).

        The kernels carry six derivative channels (u, u_t, u_x, u_y, u_xx, u_yy; for a two-input model u, u_a, u_b,
        u_aa, u_bb).  They are evaluated once by ``jets`` (differentiable w.r.t. the parameters with a cotangent per
        channel) and u is returned as its own second-order expansion around the input,
            u(X) + sum_k u_k d_k + 1/2 sum_k u_kk d_k^2,     d = X - X.detach()  (zero-valued, carries the graph),
        so ``autograd.grad(u, x, create_graph=True)`` is u_x, ``autograd.grad(u_x, x)`` is u_xx, and ``loss.backward()``
        through either reaches the parameters - exactly the derivatives the reference's operators request.  Second
        derivatives without a channel (u_tt of a three-input model, mixed ones) come out as NaN, not zero (_NoSecond)."""
        uj = self.jets(x, 0)                                # (B, 6), differentiable w.r.t. the parameters
        d = x - x.detach()
        k = self.input_dim
        first = uj[:, 1:4] if k == 3 else uj[:, 2:4]        # (t, x, y) slots, or (a, b) in the (x, y) slots
        second = uj[:, 4:6]
        dd = d[:, -2:]                                      # the coordinates that own a second-derivative channel
        u = uj[:, 0:1] + (first * d).sum(1, keepdim=True) + 0.5 * (second * dd * dd).sum(1, keepdim=True)
        mask = torch.zeros(k, k, dtype=torch.bool)
        mask[k - 2, k - 2] = mask[k - 1, k - 1] = True
        return u + _NoSecond.apply(d, mask)

    def residual(self, X: torch.Tensor, D=0.01, v_x=1.0, v_y=1.0, sigma=(1.0, 1.0, 1.0)):
        """(u, residual) at X (B,3) with the derivative channels carried through the HIP kernels —
        what ``nn.pde.diffusion_operator`` dispatches to for this model."""
        if X.dim() != 2 or X.shape[1] != 3 or self.input_dim != 3:
            raise ValueError(f"Expected collocation points of shape (B, 3), got {tuple(X.shape)}")
        self._engine_for(X.device)
        pde = (float(D), float(v_x), float(v_y), tuple(float(s) for s in sigma), None)
        return _ResidualFn.apply(X, self, pde, *_flat_param_list(self))

    def second_order(self, X: torch.Tensor, c_aa: float, c_bb: float):
        """(u, c_aa * u_aa + c_bb * u_bb) for a TWO-input model u(a, b) (X = (B, 2)): the second-derivative channels of
        the same fused kernels, with the two inputs in the coordinate slots that carry them.  What the Klein-Gordon,
        wave and Helmholtz operators of ``nn.pde`` dispatch to (reference nn/pde.py:28-52,73-95)."""
        if X.dim() != 2 or X.shape[1] != 2 or self.input_dim != 2:
            raise ValueError(f"second_order needs a two-input model and points of shape (B, 2), got {tuple(X.shape)}")
        self._engine_for(X.device)
        # residual = c_t u_t + c_x u_x + c_y u_y - (d_xx u_xx + d_yy u_yy)  with (a, b) in the (x, y) slots
        pde = (0.0, 0.0, 0.0, (1.0, 1.0, 1.0), (0.0, 0.0, 0.0, -float(c_aa), -float(c_bb)))
        return _ResidualFn.apply(X, self, pde, *_flat_param_list(self))

    # ------------------------------------------------------------------ checkpoint (same keys as :116-128)
    def save_state(self, path=None):
        sync = getattr(self, "_sync_fused_to_torch", None)
        if sync is not None:
            sync()
        state = {
            "args": self.args,
            "classic_network": self.classic_network,
            "quantum_params": self.quantum_layer.state_dict(),
            "preprocessor": self.preprocessor.state_dict(),
            "quantum_layer": self.quantum_layer.state_dict(),
            "postprocessor": self.postprocessor.state_dict(),
            "optimizer": self.optimizer.state_dict(),
            "scheduler": self.scheduler.state_dict(),
            "loss_history": self.loss_history,
            "log_path": self.log_path,
        }
        model_path = os.path.join(self.log_path, "model.pth") if path is None else path
        with open(model_path, "wb") as f:
            torch.save(state, f)
        if self.logger is not None:
            self.logger.print(f"Model state saved to {model_path}")

    @classmethod
    def load_state(cls, file_path, map_location=None):
        if map_location is None:
            map_location = torch.device("cpu")
        with open(file_path, "rb") as f:     # tensors, lists, numbers, strings, torch.device only
            return torch.load(f, map_location=map_location, weights_only=True)

    def draw_quantum_circuit(self, x):
        """The reference renders the QNode to circuit.pdf with matplotlib (:144-158); here the gate
        program is written as text to the log (plots are out of scope)."""
        if self.draw_quantum_circuit_flag and self.logger is not None:
            try:
                self.logger.print("The circuit used in the study:")
                self.logger.print(self.quantum_layer.describe())
            except Exception as e:
                self.logger.print(f"Failed to draw quantum circuit: {str(e)}")
