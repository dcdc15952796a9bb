"""On-device optimiser block of the step (trainer/diffusion_train.py:81-90; nn/DVPDESolver.py:59-64): gradient
clipping, Adam and ReduceLROnPlateau, including the learning-rate REDUCTION branch the reference reaches in its
20 000-epoch runs (trainer/diffusion_hybrid_trainer.py:46) — compared step by step with torch's own
``clip_grad_norm_`` / ``Adam`` / ``ReduceLROnPlateau`` on the CPU."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pkg
from test_gpu_solver import Log, base_args

pytestmark = pytest.mark.gpu


def _loss_sequence(steps, rng):
    """Weighted losses that fall, stall (> patience bad steps in a row, several times) and jitter."""
    base = np.concatenate([np.linspace(3.0, 1.0, 6), np.full(9, 1.0), np.linspace(0.99, 0.8, 5), np.full(steps, 0.8)])[:steps]
    return base * (1.0 + 1e-5 * rng.standard_normal(steps))


@pytest.mark.parametrize("NP", [717, 4000])
def test_adam_clip_plateau_kernels_match_torch(NP, gpu_device):
    """qc_adam_step (k_adam_fast for NP + 3 <= 3072, k_adam beyond) on synthetic gradients: parameters, moments,
    lr, best, num_bad_epochs after EVERY step vs torch.  patience 2, scheduler eps chosen so that the third
    reduction is refused by the ``lr - new_lr <= eps`` guard.  (The fold + update launch of the fused step,
    k_adam_fast<FOLD>, is covered by the training test below.)"""
    L = pkg("hip.lib")
    engine = pkg("hip.engine")
    lib = L.load()
    rng = np.random.default_rng(5)
    steps = 40
    lr0, patience, factor, sched_eps, min_lr = 0.005, 2, 0.9, 4.2e-4, 0.0
    p0 = rng.standard_normal(NP).astype(np.float32)
    losses = _loss_sequence(steps, rng)
    # torch side
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt_t = torch.optim.Adam([pt], lr=lr0)
    sch_t = torch.optim.lr_scheduler.ReduceLROnPlateau(opt_t, mode="min", factor=factor, patience=patience,
                                                       eps=sched_eps, min_lr=min_lr)
    # device side
    prm = torch.from_numpy(p0.copy()).to(gpu_device)
    opt = engine.OptimState(NP, lr0, gpu_device, hist_cap=steps, patience=patience, factor=factor,
                            sched_eps=sched_eps, min_lr=min_lr)
    flat = torch.zeros(NP + 3, device=gpu_device)
    st = torch.cuda.current_stream(gpu_device).cuda_stream
    lrs = []
    for k in range(steps):
        scale = 3.0 if k % 3 == 0 else 0.01          # clipped and unclipped steps
        g = (scale * rng.standard_normal(NP)).astype(np.float32)
        l_r, l_bc, l_ic = 0.25 * losses[k], 0.0625 * losses[k], 0.125 * losses[k]   # 2, 4, 2 weights -> losses[k]
        # torch
        opt_t.zero_grad()
        pt.grad = torch.from_numpy(g.copy())
        torch.nn.utils.clip_grad_norm_([pt], max_norm=1)
        opt_t.step()
        loss_t = torch.tensor(2.0 * np.float32(l_r) + 4.0 * np.float32(l_bc) + 2.0 * np.float32(l_ic))
        sch_t.step(loss_t)
        # device
        vec = np.concatenate([g, np.array([l_r, l_bc, l_ic], np.float32)])
        flat.copy_(torch.from_numpy(vec))
        L.check(lib.qc_adam_step(flat.data_ptr(), NP, prm.data_ptr(), opt.m.data_ptr(), opt.v.data_ptr(),
                                 opt.state.data_ptr(), C.byref(opt.hyper), opt.hist.data_ptr(), opt.hist_cap,
                                 None, 0, None, st), "qc_adam_step")
        rec = opt.read()
        lrs.append(rec["lr"])
        assert rec["step"] == k + 1
        assert rec["num_bad_epochs"] == sch_t.num_bad_epochs, (k, rec, sch_t.num_bad_epochs)
        assert abs(rec["lr"] - opt_t.param_groups[0]["lr"]) < 1e-9 + 1e-6 * lr0, (k, rec["lr"], opt_t.param_groups[0]["lr"])
        assert abs(rec["best"] - float(sch_t.best)) < 1e-6 * max(1.0, abs(float(sch_t.best))), (k, rec["best"], sch_t.best)
        assert abs(rec["loss"] - loss_t.item()) < 1e-6 * max(1.0, loss_t.item())
        dp = (prm.cpu() - pt.detach()).abs().max().item()
        assert dp < 2e-6 * (k + 1), (k, dp)
        assert (opt.m.cpu() - opt_t.state[pt]["exp_avg"]).abs().max().item() < 1e-6
        assert (opt.v.cpu() - opt_t.state[pt]["exp_avg_sq"]).abs().max().item() < 1e-6
    # the trajectory really exercised the branches under test
    distinct = sorted(set(round(v, 9) for v in lrs), reverse=True)
    assert len(distinct) == 3, distinct                      # two reductions applied, ...
    assert abs(distinct[-1] - lr0 * factor ** 2) < 1e-8
    assert lr0 * factor ** 2 - lr0 * factor ** 3 <= sched_eps  # ... the third refused by the eps guard
    assert len(opt.loss_history()) == steps


@pytest.mark.parametrize("split", [False, True])
def test_plateau_scheduler_trajectory_in_training_matches_oracle(split, gpu_device, tmp_path):
    """35 fused training steps with patience 2 on fresh batches of 64 points (sampling noise makes the loss
    stall repeatedly): lr / num_bad_epochs / best after every step and the loss history vs the CPU oracle running
    the reference loop with torch's scheduler.  ``split``: the two-call form of the step that data-parallel runs
    use (QC_PHASE_GRADS, [all-reduce], QC_PHASE_UPDATE -> qc_adam_step)."""
    from oracle import solver as osol
    L = pkg("hip.lib")
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    steps = 35
    args = base_args(epochs=steps - 1)
    torch.manual_seed(1)
    model = Solver(args, Log(tmp_path), device=gpu_device)
    torch.manual_seed(1)
    ref = osol.OracleSolver(args, device=torch.device("cpu"))
    for m in (model, ref):
        m.scheduler.patience = 2
        m.scheduler.eps = 4.2e-4            # third reduction refused: 0.00405 - 0.003645 <= eps
    torch.manual_seed(11)
    batches = [(osol.sample_box(osol.BOX_IC, 21), osol.sample_box(osol.BOX_BC1, 21), osol.sample_box(osol.BOX_DOM, 64))
               for _ in range(steps)]
    tr = trainer.FusedTrainer(model, 64, capacity=steps)
    lr_seen = set()
    for it in range(steps):
        osol.train_step(ref, 64, batches[it])
        tr.load_batches(*batches[it])
        if split:
            tr.fs.run(L.QC_PHASE_GRADS)
            tr.fs.run(L.QC_PHASE_UPDATE)
        else:
            tr.step()
        rec = tr.opt.read()
        sch = ref.scheduler
        assert rec["num_bad_epochs"] == sch.num_bad_epochs, (it, rec, sch.num_bad_epochs)
        assert abs(rec["lr"] - ref.optimizer.param_groups[0]["lr"]) < 1e-8, (it, rec["lr"])
        assert abs(rec["best"] - float(sch.best)) < 1e-4 * max(1.0, float(sch.best))
        lr_seen.add(round(rec["lr"], 9))
    got, want = np.array(tr.opt.loss_history()), np.array(ref.loss_history)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (got, want)
    assert len(lr_seen) >= 2, "no learning-rate reduction happened: the test would not cover the branch"


def test_second_train_call_continues_history_and_optimiser(gpu_device, tmp_path):
    """train() twice on one model: Adam / scheduler state continue (step count, moments), and loss_history is the
    concatenation of both runs — equal to the oracle stepping through all the batches once."""
    from oracle import solver as osol
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    args = base_args(epochs=4)
    torch.manual_seed(1)
    model = Solver(args, Log(tmp_path), device=gpu_device)
    torch.manual_seed(1)
    ref = osol.OracleSolver(args, device=torch.device("cpu"))
    torch.manual_seed(21)
    batches = [(osol.sample_box(osol.BOX_IC, 21), osol.sample_box(osol.BOX_BC1, 21), osol.sample_box(osol.BOX_DOM, 64))
               for _ in range(10)]
    trainer.train(model, batch_size=64, batches=batches[:5])
    assert len(model.loss_history) == 5
    trainer.train(model, batch_size=64, batches=batches[5:])
    for b in batches:
        osol.train_step(ref, 64, b)
    got, want = np.array(model.loss_history), np.array(ref.loss_history)
    assert got.shape == want.shape == (10,)
    assert np.abs(got - want).max() < 1e-4 * max(1.0, np.abs(want).max()), (got, want)
    assert model.scheduler.last_epoch == 10
    assert int(next(iter(model.optimizer.state.values()))["step"]) == 10
