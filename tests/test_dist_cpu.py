"""world_size-2 data-parallel logic on CPU (gloo): the sharding + global-normaliser + single
all-reduce of the flat [gradient | L_r, L_bc, L_ic] vector that trainer/diffusion_train.FusedTrainer
uses on GPUs (backend "nccl" = RCCL there) must reproduce the single-process step.  The per-shard
arithmetic is done by the CPU oracle here (test infrastructure); what is under test is the product's
shard arithmetic, normalisation convention and collective pattern."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def base_args():
    return {"batch_size": 64, "epochs": 0, "lr": 0.005, "seed": 1, "print_every": 100, "num_qubits": 4,
            "num_quantum_layers": 1, "classic_network": [3, 50, 1], "q_ansatz": "cascade", "shots": 1024,
            "problem": "diffusion", "solver": "DV", "encoding": "None", "use_ibm_hardware": False}


def shard_flat(model, X_ic, X_bc, X_res, world, rank, tr):
    """What one rank contributes: gradient of its shard's error sums divided by the GLOBAL counts
    (engine._pde: w = 2*weight/N_global), followed by its three loss sums / N_global."""
    from oracle import solver as osol
    s_ic, s_bc, s_rs = (tr.shard_slice(v.shape[0], world, rank) for v in (X_ic, X_bc, X_res))
    xi, xb, xr = X_ic[s_ic], X_bc[s_bc], X_res[s_rs]
    zero = torch.zeros((), dtype=torch.float32)
    l_bc = ((model(xb) - osol.analytic_u(xb)) ** 2).sum() / X_bc.shape[0] if xb.shape[0] else zero
    l_ic = ((model(xi) - osol.analytic_u(xi)) ** 2).sum() / X_ic.shape[0] if xi.shape[0] else zero
    if xr.shape[0]:
        _, res = osol.diffusion_residual(model, xr[:, 0:1].clone(), xr[:, 1:2].clone(), xr[:, 2:3].clone())
        l_r = ((res - osol.analytic_r(xr)) ** 2).sum() / X_res.shape[0]
    else:
        l_r = zero
    model.zero_grad()
    (2.0 * l_r + 4.0 * l_bc + 2.0 * l_ic).backward()
    g = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in model.parameters()])
    return torch.cat([g, torch.stack([l_r.detach(), l_bc.detach(), l_ic.detach()])])


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import solver as osol
    tr = pkg("trainer.diffusion_train")
    torch.manual_seed(1)
    model = osol.OracleSolver(base_args(), device=torch.device("cpu"))
    g = torch.Generator().manual_seed(5)
    X_ic = torch.rand(21, 3, generator=g) * torch.tensor([0.0, 1.0, 1.0])
    X_bc = torch.rand(21, 3, generator=g) * torch.tensor([1.0, 0.0, 1.0])
    X_res = torch.rand(64, 3, generator=g)
    assert tr._dist_info() == (world, rank)
    flat = shard_flat(model, X_ic, X_bc, X_res, world, rank, tr)
    dist.all_reduce(flat)                                   # the one collective of the step
    np.save(os.path.join(out_dir, f"flat_{rank}.npy"), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    f0, f1 = (np.load(os.path.join(tmp_path, f"flat_{r}.npy")) for r in range(2))
    assert np.array_equal(f0, f1)                           # replicated result on both ranks
    # single process, whole batch
    from oracle import solver as osol
    tr = pkg("trainer.diffusion_train")
    torch.manual_seed(1)
    model = osol.OracleSolver(base_args(), device=torch.device("cpu"))
    g = torch.Generator().manual_seed(5)
    X_ic = torch.rand(21, 3, generator=g) * torch.tensor([0.0, 1.0, 1.0])
    X_bc = torch.rand(21, 3, generator=g) * torch.tensor([1.0, 0.0, 1.0])
    X_res = torch.rand(64, 3, generator=g)
    whole = shard_flat(model, X_ic, X_bc, X_res, 1, 0, tr).numpy()
    assert np.abs(whole - f0).max() < 1e-5 * max(1.0, np.abs(whole).max())
    # and the whole-batch vector is the reference step's gradient / loss parts
    model.zero_grad()
    loss, l_r, l_bc, l_ic = osol.loss_on_batches(model, X_ic, X_bc, X_res)
    assert abs((2 * whole[-3] + 4 * whole[-2] + 2 * whole[-1]) - loss.item()) < 1e-5


def test_bench_self_launch_reaches_its_ranks_without_a_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE must spawn its ranks as child processes (not exit with a
    'use torch.distributed.run' message).  Without a GPU every rank stops at the loud 'needs a GPU' exit, which
    the launcher relays as a non-zero return code."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered by tests/test_gpu_dist.py on a GPU box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode != 0                                   # the child ranks' failure is the launcher's return code
    assert "needs a GPU" in p.stdout and "launch N>1 with" not in p.stdout
    assert not any(ln.startswith("{") and '"metric"' in ln for ln in p.stdout.splitlines())   # and no result line
