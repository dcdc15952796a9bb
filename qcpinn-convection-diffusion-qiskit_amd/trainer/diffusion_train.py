"""``train(model, nIter=10000, batch_size=128, log_NTK=False, update_lam=False)`` — the training
loop of the convection-diffusion DV path with the reference's signature and step semantics
(trainer/diffusion_train.py:8-93):

  per iteration (``epochs + 1`` of them, :52): sample IC (``batch_size//3``), BC1 (``batch_size//3``),
  residual (``batch_size``) points in that RNG order (:34-36); u on BC, u on IC, PDE residual (:40-43);
  ``loss = 2*MSE_res + 4*MSE_bc + 2*MSE_ic`` (:47); backward; ``clip_grad_norm_(1)`` (DV, :85); Adam;
  ``ReduceLROnPlateau.step(loss)``; ``loss_history.append`` (:86-90); log / checkpoint every
  ``print_every`` iterations (:56-79).  ``nIter``, ``log_NTK``, ``update_lam`` are unused, as there.

For a ``DVPDESolver`` the whole iteration is ONE call into ``libqcpinn_hip.so``
(``qc_fused_pinn_residual_step``): forward derivative channels, loss, adjoint sweep, row reduction,
clip + Adam + scheduler all run as HIP kernels on resident batches; the host never reads a value
back except at ``print_every``.  Under ``torch.distributed`` (one process per GPU, backend "nccl" =
RCCL) ``batch_size`` is the GLOBAL batch: every rank takes an equal shard of each of the three
batches, and the flat ``[gradient | 3 loss sums]`` vector is all-reduced once per step before the
(identical, replicated) optimiser update.

Any other model (the reference's duck type) runs the generic torch-autograd loop below.
"""
from __future__ import annotations

import os
import time

import torch

from ..data.diffusion_dataset import Sampler, box, r, u
from ..hip import engine as _engine
from ..hip import lib as _lib
from ..nn.pde import diffusion_operator


def fetch_minibatch(sampler, N):
    return sampler.sample(N)


# ---------------------------------------------------------------------------------------------
def _dist_info():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def shard_count(total: int, world: int, rank: int) -> int:
    """Points of a global batch of ``total`` that rank ``rank`` of ``world`` owns."""
    return total // world + (1 if rank < total % world else 0)


def shard_slice(total: int, world: int, rank: int) -> slice:
    start = sum(shard_count(total, world, r_) for r_ in range(rank))
    return slice(start, start + shard_count(total, world, rank))


def _log_line(model, it, parts, lr, step_time, total_elapsed):
    loss, l_r, l_bc, l_ic = parts
    remaining = model.epochs - it
    model.logger.print(
        "Epoch: %d/%d [%.1f%%] | Loss: %.2e | Loss_res: %.2e | Loss_bcs: %.2e | loss_ics: %.2e | lr: %.2e | "
        "Epoch_time: %.2fs | Total: %.1fs | ETA: %.1fs"
        % (it, model.epochs, 100.0 * it / model.epochs if model.epochs > 0 else 0, loss, l_r, l_bc, l_ic, lr,
           step_time, total_elapsed, step_time * remaining))


# ---------------------------------------------------------------------------------------------
class FusedTrainer:
    """Device-resident training state of one DVPDESolver: sampler boxes, optimiser record,
    the fused-step descriptor.  ``step()`` = one reference iteration, no host synchronisation.

    ``sampler="device"`` (default) draws the three batches inside the fused call with a counter-based
    Philox generator indexed by the global point index (seeded once from torch's CPU generator, so
    ``torch.manual_seed`` still controls the run); ``sampler="torch"`` uses three ``torch.rand`` calls
    like the reference (IC -> BC -> residual)."""

    def __init__(self, model, batch_size: int, capacity: int, sampler: str = "device", *, n_bc: int = None,
                 bc_faces: int = 1, pde: dict = None):
        """``n_bc`` (default ``batch_size // 3``): GLOBAL boundary points per step; ``bc_faces`` = 4 spreads them
        evenly over the faces x=0, x=1, y=0, y=1 (second workload, train_hybrid_qpinn.py:689-697) instead of
        the x=0 face; ``pde`` = {"D", "vx", "vy", "problem"} overrides the engine's operator / targets."""
        # the fused step is the 3-D convection-diffusion step of trainer/diffusion_train.py:30-49 on a (t, x, y) -> u model;
        # the reference fails on any other shape (Linear(3, H) weight mismatch), and so does this trainer: a two-input
        # model keeps a zero-padded t column in W1 that the step would train, a K-output model has K last-layer rows
        if getattr(model, "input_dim", 3) != 3 or getattr(model, "n_out", 1) != 1:
            raise ValueError("train() / FusedTrainer need classic_network = [3, H, 1] (got input_dim = %s, n_out = %s): the "
                             "convection-diffusion step takes (t, x, y) points and one output"
                             % (getattr(model, "input_dim", None), getattr(model, "n_out", None)))
        dev = model._resolve_device(model.device)
        if dev is None or dev.type != "cuda":
            raise _lib.QcError("training a DVPDESolver needs a GPU (HIP kernels, no CPU fallback)")
        self.model, self.device = model, dev
        self.eng = model._engine_for(dev)
        self.eng.sigma = (1.0, 1.0, 1.0)          # the trainers call the operator with its default scalings
        self.eng.coeffs = None
        if pde:
            self.eng.D, self.eng.vx, self.eng.vy = float(pde["D"]), float(pde["vx"]), float(pde["vy"])
            self.eng.problem = int(pde["problem"])
        self.world, self.rank = _dist_info()
        n3 = batch_size // 3
        nb = n3 if n_bc is None else int(n_bc)
        if bc_faces not in (1, 4) or (bc_faces == 4 and nb % 4):
            raise ValueError("bc_faces must be 1 or 4 (with n_bc divisible by 4)")
        self.bc_face_points = nb // 4 if bc_faces == 4 else 0
        self.global_counts = (batch_size, n3, nb)                    # residual, IC, BC
        self.B_res = shard_count(batch_size, self.world, self.rank)
        self.n_ic = shard_count(n3, self.world, self.rank)
        self.n_bc = shard_count(nb, self.world, self.rank)
        self.bc_start = shard_slice(nb, self.world, self.rank).start
        self.opt = self._make_opt_state(capacity)
        self.fs = self.eng.fused(self.B_res, self.n_ic, self.n_bc, self.opt, self.global_counts)
        self.lo = {k: box(k, dev)[0:1] for k in ("ics", "bc1", "dom")}
        self.span = {k: box(k, dev)[1:2] - box(k, dev)[0:1] for k in ("ics", "bc1", "dom")}
        self.eng.refresh_gates()
        model._sync_fused_to_torch = self.sync_to_torch
        if sampler not in ("device", "torch"):
            raise ValueError("sampler must be 'device' or 'torch'")
        self.sampler = sampler
        self._explicit = False
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())        # every rank must draw the same seed
        if self.world > 1:
            import torch.distributed as dist
            t = torch.tensor([seed], dtype=torch.int64, device=dev)
            dist.broadcast(t, 0)
            seed = int(t.item())
        self.fs.set_sampler(seed, shard_slice(batch_size, self.world, self.rank).start,
                            shard_slice(n3, self.world, self.rank).start, self.bc_start, self.bc_face_points)
        # QC_DP_COLLECTIVE=rccl: the all-reduce runs INSIDE the library call (qc_comm_*: RCCL on the step's own stream,
        # one host call per step); default: torch.distributed between the two phases
        self._comm = None
        if self.world > 1 and os.environ.get("QC_DP_COLLECTIVE", "torch") == "rccl":
            self._comm = self._make_comm()
            self.fs.set_comm(self._comm)

    def _make_comm(self):
        """One library-owned RCCL communicator over the ranks of the default process group (the 128-byte id travels
        through torch.distributed once, at construction)."""
        import ctypes as C
        import torch.distributed as dist
        lib = self.eng.lib
        buf = (C.c_ubyte * 128)()
        if self.rank == 0:
            _lib.check(lib.qc_comm_unique_id(buf), "qc_comm_unique_id")
        t = torch.tensor(list(buf), dtype=torch.uint8, device=self.device)
        dist.broadcast(t, 0)
        raw = bytes(t.cpu().tolist())
        comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.qc_comm_create(raw, self.world, self.rank, C.byref(comm)), "qc_comm_create")
        return comm

    def close(self):
        """Destroys the library-owned RCCL communicator (QC_DP_COLLECTIVE=rccl); idempotent."""
        comm, self._comm = getattr(self, "_comm", None), None
        if comm is not None:
            try:
                self.fs.set_comm(None)
                self.eng.lib.qc_comm_destroy(comm)
            except Exception:        # interpreter shutdown: the library may already be gone
                pass

    def __del__(self):
        self.close()

    # -- optimiser state: continue from the torch optimiser / scheduler objects of the model
    def _make_opt_state(self, capacity):
        model = self.model
        opt, sch = model.optimizer, model.scheduler
        g = opt.param_groups[0]
        st = _engine.OptimState(self.eng.NP, float(g["lr"]), self.device, hist_cap=capacity,
                                betas=tuple(g["betas"]), eps=float(g["eps"]), max_norm=1.0,
                                factor=float(sch.factor), patience=int(sch.patience),
                                threshold=float(sch.threshold), min_lr=float(sch.min_lrs[0]),
                                sched_eps=float(sch.eps))
        steps, off = 0, 0
        for p in model.parameters():
            s = opt.state.get(p, None)
            k = p.numel()
            if s:
                st.m[off:off + k] = s["exp_avg"].reshape(-1).to(self.device)
                st.v[off:off + k] = s["exp_avg_sq"].reshape(-1).to(self.device)
                steps = int(s["step"])
            off += k
        # a model trained before continues its Adam step count; the loss history of THIS run starts at index 0
        st.write(best=float(sch.best), num_bad=int(sch.num_bad_epochs), step=steps, hist_base=steps)
        return st

    def sync_to_torch(self):
        """Mirror the device optimiser record into model.optimizer / model.scheduler so that
        ``save_state`` writes a checkpoint interchangeable with the reference's."""
        model, rec = self.model, self.opt.read()
        off = 0
        for p in model.parameters():
            k = p.numel()
            model.optimizer.state[p] = {
                "step": torch.tensor(float(rec["step"])),
                "exp_avg": self.opt.m[off:off + k].view(p.shape),
                "exp_avg_sq": self.opt.v[off:off + k].view(p.shape),
            }
            off += k
        for g in model.optimizer.param_groups:
            g["lr"] = rec["lr"]
        sch = model.scheduler
        sch.best, sch.num_bad_epochs = rec["best"], rec["num_bad_epochs"]
        sch.last_epoch = rec["step"]
        sch._last_lr = [rec["lr"]]

    # -- batches
    def sample(self):
        """IC -> BC -> residual, uniform in the reference's boxes (trainer/diffusion_train.py:9-20,34-36).
        With the device sampler this only arms the next ``step()``."""
        self._explicit = False
        if self.sampler == "device":
            return
        self._explicit = True
        fs, dev = self.fs, self.device
        if self.n_ic:
            fs.X_val[: self.n_ic] = self.lo["ics"] + self.span["ics"] * torch.rand(self.n_ic, 3, device=dev)
        if self.n_bc:
            pts = self.lo["bc1"] + self.span["bc1"] * torch.rand(self.n_bc, 3, device=dev)
            if self.bc_face_points:      # faces x=0, x=1, y=0, y=1 by GLOBAL boundary-point index
                pts = torch.rand(self.n_bc, 3, device=dev)
                face = (self.bc_start + torch.arange(self.n_bc, device=dev)) // self.bc_face_points
                pts[:, 1] = torch.where(face == 0, 0.0, torch.where(face == 1, 1.0, pts[:, 1]))
                pts[:, 2] = torch.where(face == 2, 0.0, torch.where(face == 3, 1.0, pts[:, 2]))
            fs.X_val[self.n_ic: self.n_ic + self.n_bc] = pts
        if self.B_res:
            fs.X_res[: self.B_res] = self.lo["dom"] + self.span["dom"] * torch.rand(self.B_res, 3, device=dev)

    def load_batches(self, X_ic, X_bc, X_res):
        """Use given GLOBAL batches (parity tests): this rank takes its contiguous shard."""
        self._explicit = True
        fs, dev = self.fs, self.device
        s_ic = shard_slice(X_ic.shape[0], self.world, self.rank)
        s_bc = shard_slice(X_bc.shape[0], self.world, self.rank)
        s_rs = shard_slice(X_res.shape[0], self.world, self.rank)
        if self.n_ic:
            fs.X_val[: self.n_ic] = X_ic[s_ic].to(dev)
        if self.n_bc:
            fs.X_val[self.n_ic: self.n_ic + self.n_bc] = X_bc[s_bc].to(dev)
        if self.B_res:
            fs.X_res[: self.B_res] = X_res[s_rs].to(dev)

    def step(self):
        draw = 0 if self._explicit else _lib.QC_PHASE_SAMPLE
        if self.world == 1 or self._comm is not None:
            self.fs.run(draw | _lib.QC_PHASE_GRADS | _lib.QC_PHASE_UPDATE)
        else:
            import torch.distributed as dist
            self.fs.run(draw | _lib.QC_PHASE_GRADS)
            dist.all_reduce(self.fs.flat_grad)          # one small all-reduce: [grads | L_r, L_bc, L_ic]
            self.fs.run(_lib.QC_PHASE_UPDATE)

    def losses(self):
        rec = self.opt.read()
        return (rec["loss"], rec["loss_res"], rec["loss_bc"], rec["loss_ic"]), rec["lr"]


def _train_fused(model, batch_size, batches=None):
    steps = model.epochs + 1
    tr = FusedTrainer(model, batch_size, capacity=steps)
    t0 = time.time()
    model.logger.print(f"Starting training for {model.epochs} epochs...")
    model.logger.print(f"Batch size: {batch_size}")
    pe = model.args["print_every"]
    done = 0
    for it in range(steps):
        if batches is None:
            tr.sample()
        else:
            tr.load_batches(*batches[it])
        tr.step()
        if it % pe == 0 or it == 0:
            # the reference logs the loss of iteration `it` BEFORE its optimiser step; the fused step has
            # already applied it, and reports that same pre-update loss
            parts, lr = tr.losses()
            el = time.time() - t0
            _log_line(model, it, parts, lr, el / (it + 1), el)
            if it > 0 and it % pe == 0:
                hist = tr.opt.loss_history(it + 1)
                model.loss_history.extend(hist[done:])
                done = len(hist)
                model.save_state()
    hist = tr.opt.loss_history(steps)
    model.loss_history.extend(hist[done:])
    tr.sync_to_torch()
    total = time.time() - t0
    model.total_training_time += total
    model.logger.print(f"Training completed in {total:.2f} seconds ({total / 60:.2f} minutes)")
    return tr


# ---------------------------------------------------------------------------------------------
def _train_generic(model, batch_size):
    """The reference algorithm for any duck-typed model, on torch autograd."""
    dev = model.device
    ics = Sampler(3, box("ics", dev), u, name="Initial Condition", device=dev)
    bc1 = Sampler(3, box("bc1", dev), u, name="Dirichlet BC1", device=dev)
    res = Sampler(3, box("dom", dev), r, name="Forcing", device=dev)
    t0 = time.time()
    model.logger.print(f"Starting training for {model.epochs} epochs...")
    model.logger.print(f"Batch size: {batch_size}")
    fwd_times = []
    for it in range(model.epochs + 1):
        t_it = time.time()
        if model.optimizer is not None:
            model.optimizer.zero_grad()
        X_ic, u_ic = fetch_minibatch(ics, batch_size // 3)
        X_bc, u_bc = fetch_minibatch(bc1, batch_size // 3)
        X_rs, r_rs = fetch_minibatch(res, batch_size)
        pred_bc = model.forward(X_bc)
        pred_ic = model.forward(X_ic)
        _, pred_r = diffusion_operator(model, X_rs[:, 0:1], X_rs[:, 1:2], X_rs[:, 2:3])
        l_r, l_bc, l_ic = model.loss_fn(pred_r, r_rs), model.loss_fn(pred_bc, u_bc), model.loss_fn(pred_ic, u_ic)
        loss = 2.0 * l_r + 4.0 * l_bc + 2.0 * l_ic
        fwd_times.append(time.time() - t_it)
        if it % model.args["print_every"] == 0 or it == 0 or model.args.get("use_ibm_hardware", False):
            lr = model.optimizer.param_groups[0]["lr"] if model.optimizer else 0.0
            _log_line(model, it, (loss.item(), l_r.item(), l_bc.item(), l_ic.item()), lr,
                      sum(fwd_times) / len(fwd_times), time.time() - t0)
            if it > 0 and it % model.args["print_every"] == 0:
                model.save_state()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=0.1 if model.args["solver"] == "CV" else 1)
        if model.optimizer is not None:
            model.optimizer.step()
        if model.scheduler is not None:
            model.scheduler.step(loss)
        model.loss_history.append(loss.item())
    total = time.time() - t0
    model.logger.print(f"Training completed in {total:.2f} seconds ({total / 60:.2f} minutes)")


def train(model, nIter=10000, batch_size=128, log_NTK=False, update_lam=False, *, batches=None):
    """``batches`` (keyword-only, not in the reference): a per-iteration list of
    ``(X_ic, X_bc, X_res)`` tensors to use instead of sampling — for parity tests."""
    if hasattr(model, "_engine_for") and hasattr(model, "quantum_layer"):
        _train_fused(model, batch_size, batches)
    else:
        if batches is not None:
            raise ValueError("explicit batches are only supported for DVPDESolver models")
        _train_generic(model, batch_size)
