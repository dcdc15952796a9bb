"""Data-parallel FusedTrainer on the GPU: two ranks (both on cuda:0, gloo as the transport, since a
single-GPU box cannot host two RCCL ranks) must reproduce the single-process loss history on the same
global batches, and the device sampler must give both ranks disjoint shards of one global batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, pkg

pytestmark = pytest.mark.gpu


def base_args(**kw):
    a = {"batch_size": 64, "epochs": 20, "lr": 0.005, "seed": 1, "print_every": 1000, "num_qubits": 4,
         "num_quantum_layers": 1, "classic_network": [3, 50, 1], "q_ansatz": "cascade", "shots": 1024,
         "problem": "diffusion", "solver": "DV", "encoding": "None", "use_ibm_hardware": False}
    a.update(kw)
    return a


class Log:
    def print(self, *a):
        pass

    def get_output_dir(self):
        return "/tmp"


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    z = np.load(os.path.join(GOLDEN, "train_cascade_n4_b64.npz"))
    Solver = pkg("nn.DVPDESolver").DVPDESolver
    trainer = pkg("trainer.diffusion_train")
    torch.manual_seed(1)
    model = Solver(base_args(), Log(), device=dev)
    steps = z["X_res"].shape[0]
    batches = [tuple(torch.from_numpy(z[k][it]) for k in ("X_ic", "X_bc", "X_res")) for it in range(steps)]
    trainer.train(model, batch_size=64, batches=batches)
    np.save(os.path.join(out_dir, f"hist_{rank}.npy"), np.array(model.loss_history))
    np.save(os.path.join(out_dir, f"w_{rank}.npy"), torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy())
    # device sampler: shards of one global batch
    torch.manual_seed(5)
    tr = trainer.FusedTrainer(model, 640, capacity=2)
    tr.sample()
    tr.step()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"xres_{rank}.npy"), tr.fs.X_res[: tr.B_res].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_matches_reference_history(tmp_path, gpu_device):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(GOLDEN, "train_cascade_n4_b64.npz"))
    h0, h1 = (np.load(os.path.join(tmp_path, f"hist_{r}.npy")) for r in range(2))
    assert np.array_equal(h0, h1)                                     # replicated optimiser state
    assert np.abs(h0 - z["loss_history"]).max() < 1e-4 * max(1.0, np.abs(z["loss_history"]).max())
    w0, w1 = (np.load(os.path.join(tmp_path, f"w_{r}.npy")) for r in range(2))
    assert np.array_equal(w0, w1)
    # sampler shards: same seed on both ranks, disjoint contiguous shards of the global index space
    L = pkg("hip.lib")
    x0, x1 = (np.load(os.path.join(tmp_path, f"xres_{r}.npy")) for r in range(2))
    assert x0.shape == (320, 3) and x1.shape == (320, 3)
    assert not np.array_equal(x0, x1)
    both = np.concatenate([x0, x1])
    assert len(np.unique(both[:, 0])) > 630


def test_bench_launches_its_own_ranks(gpu_device):
    """`python bench.py --gpus 2` with no launcher around it: the script starts its ranks as a child process group
    and relays rank 0's JSON line.  Rehearsed on this one-GPU box with gloo as the transport (both ranks on
    cuda:0); on a multi-GPU node the same path runs one rank per GPU over RCCL."""
    import json
    import subprocess
    env = dict(os.environ, QC_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["per_gpu_batch"] == 131072 and out["config"]["global_batch"] == 262144   # config 4's shard
    assert out["value"] > 0 and np.isfinite(out["config"]["final_loss"])
    assert out["roofline"]["bound"] == "valu" and 0 < out["roofline"]["frac"] < 1


def test_library_collective_single_rank(gpu_device, tmp_path):
    """qc_comm_* / qc_allreduce_grads (RCCL behind the C ABI) on a one-rank communicator: the all-reduce is the
    identity, and the fused step with the communicator in its descriptor (gradients -> all-reduce -> optimiser in ONE
    library call) takes exactly the step it takes without it.  (More ranks need more GPUs than this box has.)"""
    import ctypes as C
    L = pkg("hip.lib")
    lib = L.load()
    ident = (C.c_ubyte * 128)()
    L.check(lib.qc_comm_unique_id(ident), "qc_comm_unique_id")
    comm = C.c_void_p()
    with torch.cuda.device(gpu_device):
        L.check(lib.qc_comm_create(bytes(ident), 1, 0, C.byref(comm)), "qc_comm_create")
    try:
        x = torch.randn(720, device=gpu_device)
        y = x.clone()
        st = torch.cuda.current_stream(gpu_device).cuda_stream
        L.check(lib.qc_allreduce_grads(y.data_ptr(), y.numel(), comm, st), "qc_allreduce_grads")
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        Solver = pkg("nn.DVPDESolver").DVPDESolver
        trainer = pkg("trainer.diffusion_train")
        z = np.load(os.path.join(GOLDEN, "train_cascade_n4_b64.npz"))
        hist = []
        for use_comm in (False, True):
            torch.manual_seed(1)
            model = Solver(base_args(), Log(), device=gpu_device)
            tr = trainer.FusedTrainer(model, 64, capacity=8)
            if use_comm:
                tr.fs.set_comm(comm)
            for it in range(4):
                tr.load_batches(*(torch.from_numpy(z[k][it]) for k in ("X_ic", "X_bc", "X_res")))
                tr.fs.run(L.QC_PHASE_GRADS | L.QC_PHASE_UPDATE)
            hist.append((np.array(tr.opt.loss_history()), model._flat.clone()))
        assert np.abs(hist[0][0] - z["loss_history"][:4]).max() < 1e-4 * max(1.0, np.abs(z["loss_history"]).max())
        assert np.array_equal(hist[0][0], hist[1][0])
        assert (hist[0][1] - hist[1][1]).abs().max().item() < 1e-6     # fold+update in one kernel vs reduce, all-reduce, update
    finally:
        L.check(lib.qc_comm_destroy(comm), "qc_comm_destroy")
