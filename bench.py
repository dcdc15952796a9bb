"""Benchmark of the hot path: residual collocation points per second through one FULL loss step
(sample -> BC/IC forward -> PDE residual -> backward -> clip -> Adam -> scheduler; reference
trainer/diffusion_train.py:30-49,81-90) on BASELINE.json config 2: 4-qubit cascade ansatz, 1 layer,
H=50, 65 536 residual points per GPU (+ 2 x 21 845 value points), synthetic uniform batches.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; weak scaling (per-GPU batch fixed, global batch = N x 65 536); the only
collective is one all-reduce (RCCL) of the flat [gradient | 3 loss sums] vector per step.
Prints ONE JSON line on rank 0.  ``roofline`` prices the slowest kernel of the step against the
fp32 vector/matrix peak (157.3 TFLOP/s: the state is register-resident, HBM is not the bound);
``cpu_baseline`` times the CPU oracle (a torch complex128 restatement of the reference's
PennyLane path, kind "port") on a bounded sample on this box's host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "qcpinn-convection-diffusion-qiskit_amd"

PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector == fp32 matrix (MFMA f32) peak
PEAK_HBM_GBS = 8000.0


def base_args(n=4, layers=1, ansatz="cascade", hidden=50):
    return {"batch_size": 64, "epochs": 0, "lr": 0.005, "seed": 1, "print_every": 10 ** 9, "num_qubits": n,
            "num_quantum_layers": layers, "classic_network": [3, hidden, 1], "q_ansatz": ansatz, "shots": 1024,
            "problem": "diffusion", "solver": "DV", "encoding": "None", "use_ibm_hardware": False}


class Log:
    def print(self, *a):
        pass

    def get_output_dir(self):
        return "/tmp"


def algorithmic_flops_per_point(prog, hidden, n, v=2.0 / 3.0):
    """per residual point and step, 6 forward + 6 adjoint channel-evaluations of the
    circuit, plus (1 fwd + 1 adj) for each of the 2*(1/3) value points; MLPs 2*(3H+Hn+nH+H) per
    channel-evaluation.  Returned per kernel family, per residual point."""
    F = prog.algorithmic_flops()
    mlp_pre = 2 * (3 * hidden + hidden * n)
    mlp_post = 2 * (n * hidden + hidden)
    # v: value (BC + IC) points per residual point, 2 * (B // 3) / B
    return {
        "circuit_jets_fwd": 6 * F, "circuit_jets_bwd": 6 * F,
        "pre_fwd": 6 * mlp_pre, "pre_bwd": 6 * mlp_pre, "post": 12 * mlp_post,
        # the fused step's merged launches: residual tiles (6 channels) + value tiles (1 channel) in one kernel
        "stage_circuit_fwd": (6 + v) * F, "stage_circuit_bwd": (6 + v) * F,
        "stage_pre_fwd": (6 + v) * mlp_pre, "stage_pre_bwd": (6 + v) * mlp_pre, "stage_post": 2 * (6 + v) * mlp_post,
        "step_total": (13 + 1 / 3) * (F + mlp_pre + mlp_post),
    }


def time_kernels(tr, reps=20):
    """Average duration (ms) of each kernel of the step, measured with HIP events on the stream the
    kernels are launched on (torch's current stream), each kernel launched back-to-back `reps` times
    on the resident step buffers through its own C-ABI entry point."""
    import ctypes as C
    eng, fs, lib = tr.eng, tr.fs, tr.eng.lib
    d, c = fs.desc, eng.circuit
    st = torch.cuda.current_stream(eng.device).cuda_stream
    NPo = eng.theta_off
    rows_res = (fs.B_res + 63) // 64
    pde = C.byref(d.pde)
    th = d.part_dev + 4 * NPo

    keep = bool(d.circ_ws_dev) and 2 <= d.n <= 5    # the fused step's no-recompute adjoint pair
    calls = {
        "pre_fwd": lambda: lib.qc_pre_forward(d.X_res_dev, d.params_dev, d.H, d.n, d.n_theta, d.ajets_res_dev, d.B_res, 6, st),
        "circuit_jets_fwd": (lambda: lib.qc_forward_jets_keep(d.prog, d.trig_dev, d.umat_dev, d.ajets_res_dev, d.qjets_res_dev,
                                                              d.B_res, d.circ_ws_dev, st)) if keep else
                            (lambda: lib.qc_forward_jets(d.prog, d.trig_dev, d.umat_dev, d.ajets_res_dev, d.qjets_res_dev, d.B_res,
                                                         d.circ_ws_dev, d.circ_ws_bytes, st)),
        "post": lambda: lib.qc_post(2, d.X_res_dev, d.params_dev, d.H, d.n, d.n_theta, pde, d.qjets_res_dev, d.abar_res_dev,
                                    d.abar_res_dev + 4 * d.B_res, None, None, d.qbar_res_dev, d.part_dev, d.part_stride, 0,
                                    d.B_res, 6, st),
        "circuit_jets_bwd": (lambda: lib.qc_backward_jets_kept(d.prog, d.trig_dev, d.umat_dev, d.ajets_res_dev, d.qbar_res_dev,
                                                               d.abar_res_dev, th, d.part_stride, 0, d.B_res, d.circ_ws_dev, st))
                            if keep else
                            (lambda: lib.qc_backward_jets(d.prog, d.trig_dev, d.umat_dev, d.ajets_res_dev, d.qbar_res_dev,
                                                          d.abar_res_dev, th, d.part_stride, 0, d.B_res, d.circ_ws_dev,
                                                          d.circ_ws_bytes, st)),
        "pre_bwd": lambda: lib.qc_pre_backward(d.X_res_dev, d.params_dev, d.H, d.n, d.n_theta, d.abar_res_dev, d.part_dev,
                                               d.part_stride, 0, d.B_res, 6, st),
        "value_pre_fwd": lambda: lib.qc_pre_forward(d.X_val_dev, d.params_dev, d.H, d.n, d.n_theta, d.ajets_val_dev, d.B_val, 1, st),
        "value_circuit_fwd": lambda: lib.qc_forward_expval(d.prog, d.trig_dev, d.umat_dev, d.ajets_val_dev, d.qjets_val_dev, d.B_val,
                                                           d.circ_ws_dev, d.circ_ws_bytes, st),
        "value_post": lambda: lib.qc_post(2, d.X_val_dev, d.params_dev, d.H, d.n, d.n_theta, pde, d.qjets_val_dev, d.abar_val_dev,
                                          None, None, None, d.qbar_val_dev, d.part_dev, d.part_stride, rows_res, d.B_val, 1, st),
        "value_circuit_bwd": lambda: lib.qc_backward_expval(d.prog, d.trig_dev, d.umat_dev, d.ajets_val_dev, d.qbar_val_dev,
                                                            d.abar_val_dev, th, d.part_stride, rows_res, d.B_val,
                                                            d.circ_ws_dev, d.circ_ws_bytes, st),
        "value_pre_bwd": lambda: lib.qc_pre_backward(d.X_val_dev, d.params_dev, d.H, d.n, d.n_theta, d.abar_val_dev, d.part_dev,
                                                     d.part_stride, rows_res, d.B_val, 1, st),
        "reduce_rows": lambda: lib.qc_reduce_rows(d.part_dev, d.part_rows_cap, d.part_stride, eng.NP + 3, d.flat_dev, st),
    }
    # when the step runs its merged form (one launch per stage over value + residual tiles) those are the kernels
    # it executes: time them through qc_fused_step_stage
    stages = {"stage_pre_fwd": 0, "stage_circuit_fwd": 1, "stage_post": 2, "stage_circuit_bwd": 3, "stage_pre_bwd": 4}
    if lib.qc_fused_step_stage(C.byref(d), 0, st) == 0:
        for name, sid in stages.items():
            calls[name] = (lambda sid=sid: lib.qc_fused_step_stage(C.byref(d), sid, st))
    out = {}
    for name, fn in calls.items():
        for _ in range(3):
            assert fn() == 0, name
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        out[name] = e0.elapsed_time(e1) / reps
    return out


def cpu_baseline(budget_s=24.0):
    """The CPU oracle's full training step (oracle/solver.py: torch complex128 per-gate statevector +
    torch double backward, the algorithm class of the reference's PennyLane default.qubit/backprop
    path) on this box's host cores, same model, bounded sample: residual batch 4096 (+2x1365).
    torch's intra-op thread count is picked by a short probe (tiny per-gate tensors scale badly on
    very many threads), and the one used is reported as `cores`."""
    import warnings
    from oracle import solver as osol
    warnings.filterwarnings("ignore", message="Converting a tensor with requires_grad=True")
    B = 4096
    default_threads = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, default_threads) if t <= (os.cpu_count() or 1)})
    best_t, best_dt = cands[0], float("inf")
    for t in cands:
        torch.set_num_threads(t)
        torch.manual_seed(1)
        m = osol.OracleSolver(base_args(), device=torch.device("cpu"))
        osol.train_step(m, B)                     # warm-up
        t0 = time.time()
        osol.train_step(m, B)
        dt = time.time() - t0
        if dt < best_dt:
            best_t, best_dt = t, dt
    torch.set_num_threads(best_t)
    torch.manual_seed(1)
    m = osol.OracleSolver(base_args(), device=torch.device("cpu"))
    osol.train_step(m, B)
    t0, steps = time.time(), 0
    while True:
        osol.train_step(m, B)
        steps += 1
        if time.time() - t0 > budget_s * 0.6 or steps >= 50:
            break
    dt = time.time() - t0
    torch.set_num_threads(default_threads)
    return {"value": steps * B / dt, "unit": "residual collocation points/s", "cores": best_t,
            "kind": "port", "host_cpus": os.cpu_count(),
            "sample": f"{steps} full training steps at residual batch {B} (+2x{B // 3} value points), "
                      f"{dt / steps * 1e3:.0f} ms/step, {best_t} torch threads (best of {cands}), "
                      f"torch {torch.__version__} CPU complex128 oracle"}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_traffic.json:
    FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE), or None."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm (xGMI).  QC_BENCH_BACKEND=gloo exists only to rehearse the multi-rank
        # control flow on a single-GPU box (all ranks share cuda:0).
        backend = os.environ.get("QC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    Solver = importlib.import_module(PKG + ".nn.DVPDESolver").DVPDESolver
    trainer = importlib.import_module(PKG + ".trainer.diffusion_train")
    args = base_args()
    torch.manual_seed(1)
    model = Solver(args, Log(), device=dev)
    torch.manual_seed(1234 + rank)                 # each rank draws its own shard of the global batch
    global_batch = a.batch_per_gpu * world
    tr = trainer.FusedTrainer(model, global_batch, capacity=a.steps + a.warmup)

    def one_step():
        tr.sample()
        tr.step()

    for _ in range(a.warmup):
        one_step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    t_enqueue = time.perf_counter() - t0           # host time to enqueue the K steps (no sync inside a step)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    rec = tr.opt.read()
    if rank == 0:
        prog = model.quantum_layer.program
        n, H = args["num_qubits"], args["classic_network"][1]
        flops = algorithmic_flops_per_point(prog, H, n, (tr.n_ic + tr.n_bc) / max(tr.B_res, 1))
        kt = time_kernels(tr)
        merged = "stage_circuit_bwd" in kt       # the kernels the step actually launches
        dom = max((k for k in kt if k in flops and k.startswith("stage_") == merged), key=lambda k: kt[k])
        ach = flops[dom] * tr.B_res / (kt[dom] * 1e-3) / 1e12
        step_ms = dt / a.steps * 1e3
        value = a.steps * global_batch / dt
        # algorithmic HBM bytes of the step: what must cross HBM per residual point if every stage were
        # fused (X in: 12 B) vs what the staged pipeline moves (4 jet buffers of 6n floats, written+read)
        staged_bytes = (2 * 4 * 6 * n * 4 + 2 * 12) * tr.B_res + (2 * 4 * n * 4 + 2 * 12) * (tr.n_ic + tr.n_bc)
        out = {
            "metric": "collocation-points/sec (PDE+BC+IC loss step), 4-qubit cascade",
            "value": value, "unit": "residual collocation points/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: DV solver, 4 qubits, cascade, 1 layer, H=50, "
                                   f"{a.batch_per_gpu} residual + 2x{a.batch_per_gpu // 3} BC/IC points per GPU",
                       "global_batch": global_batch, "parallelism": f"dp{world}",
                       "total_points_per_s": a.steps * (global_batch + 2 * (global_batch // 3)) / dt,
                       "final_loss": rec["loss"], "host_enqueue_ms_per_step": t_enqueue / a.steps * 1e3},
            "roofline": {"bound": "mfma", "pipe": "fp32 VALU (no MFMA used; fp32 vector peak == fp32 MFMA peak)",
                         "kernel": dom, "achieved": ach, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_F32_TFLOPS, "traffic": measured_traffic(dom),
                         "kernel_ms": kt[dom], "algorithmic_flops_per_launch": flops[dom] * tr.B_res,
                         "step_frac": flops["step_total"] * value / world / 1e12 / PEAK_F32_TFLOPS,
                         "hbm_GBps_staged_pipeline": staged_bytes / (step_ms * 1e-3) / 1e9,
                         "hbm_frac": staged_bytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "kernels_ms": kt,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
