"""Benchmark of the hot path: residual collocation points per second through one FULL loss step
(sample -> BC/IC forward -> PDE residual -> backward -> clip -> Adam -> scheduler; reference
trainer/diffusion_train.py:30-49,81-90) on the 4-qubit cascade model (1 layer, H=50), synthetic uniform batches.

    python bench.py [--gpus N --steps K --warmup W]

N = 1 is BASELINE.json config 2 (65 536 residual + 2 x 21 845 value points).  N > 1 is config 4's shard size,
131 072 residual points per GPU (1 048 576 over 8 GPUs), weak scaling; the N = 1 line also reports the step at
that shard size (``other_configs``) so a scaling curve has its own single-GPU base.  With N > 1 and no
WORLD_SIZE in the environment the script launches its own ranks (``python -m torch.distributed.run`` as a CHILD
process, before anything touches the GPU) and relays rank 0's JSON line; under ``torch.distributed.run`` it is
one rank.  One process per GPU; the only collective is one all-reduce (RCCL) of the flat
[gradient | 3 loss sums] vector per step.

Prints ONE JSON line on rank 0.  ``roofline`` prices the slowest kernel of the step against the fp32 vector
peak (157.3 TFLOP/s: the state is register-resident, HBM is not the bound); ``other_configs`` (N = 1) times a
few full-size steps of BASELINE configs 1, 3, 4-shard and 5; ``cpu_baseline`` times the CPU oracle (a torch
complex128 restatement of the reference's PennyLane path, kind "port") on a bounded sample on this box's host
cores.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "qcpinn-convection-diffusion-qiskit_amd"

PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector peak (the kernels use no MFMA)
PEAK_HBM_GBS = 8000.0
TRAFFIC_FILE = os.path.join("profiles", "r03_traffic.json")


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
    d = fs.desc
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
    # the stages IN SEQUENCE (step order, every kernel behind its true predecessor: the cache state and clocks of a real
    # step).  One event pair around `reps` repetitions of the whole five-stage sequence - an event between consecutive
    # stages costs 3-4 us of queue time, as much as a third of a stage - and the sequence total is distributed over the
    # stages in proportion to their isolated durations.
    seq = {}
    if "stage_pre_fwd" in calls:
        order = ["stage_pre_fwd", "stage_circuit_fwd", "stage_post", "stage_circuit_bwd", "stage_pre_bwd"]
        for _ in range(3):
            for name in order:
                calls[name]()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for name in order:
                calls[name]()
        e1.record()
        e1.synchronize()
        total = e0.elapsed_time(e1) / reps
        iso = sum(out[name] for name in order)
        for name in order:
            seq[name] = out[name] * total / iso
        seq["_sequence_total"] = total
        seq["_isolated_total"] = iso
    return out, seq


def _cpu_steps(osol, B, budget_s, max_steps):
    torch.manual_seed(1)
    m = osol.OracleSolver(base_args(), device=torch.device("cpu"))
    osol.train_step(m, B)                     # warm-up
    t0, steps = time.time(), 0
    while True:
        osol.train_step(m, B)
        steps += 1
        if time.time() - t0 > budget_s or steps >= max_steps:
            break
    return steps, time.time() - t0


def cpu_baseline(budget_s=24.0):
    """The CPU oracle's full training step (oracle/solver.py: torch complex128 per-gate statevector +
    torch double backward, the algorithm class of the reference's PennyLane default.qubit/backprop
    path) on this box's host cores, same model, bounded samples.  ``value`` is the CPU path at its most
    favourable batch (4 096 residual + 2 x 1 365 value points); ``config1`` is the reference's own
    configuration (batch 64, and 128 — the effective batch of the stock script, quirk Q1).
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
        _, dt = _cpu_steps(osol, B, 0.0, 1)
        if dt < best_dt:
            best_t, best_dt = t, dt
    torch.set_num_threads(best_t)
    steps, dt = _cpu_steps(osol, B, budget_s * 0.5, 50)
    small = {}
    torch.set_num_threads(min(8, best_t))       # 16 amplitudes x 64 points: more threads only add overhead
    for b in (64, 128):
        s, d = _cpu_steps(osol, b, budget_s * 0.12, 40)
        small[f"b{b}"] = {"residual_points_per_s": s * b / d, "ms_per_step": d / s * 1e3, "steps": s,
                          "threads": min(8, best_t)}
    torch.set_num_threads(default_threads)
    return {"value": steps * B / dt, "unit": "residual collocation points/s", "cores": best_t,
            "kind": "port", "host_cpus": os.cpu_count(), "config1": small,
            "sample": f"{steps} full training steps at residual batch {B} (+2x{B // 3} value points), "
                      f"{dt / steps * 1e3:.0f} ms/step, {best_t} torch threads (best of {cands}), "
                      f"torch {torch.__version__} CPU complex128 oracle; config1 = the same step at the reference's "
                      f"own batch 64 / 128"}


def measured_traffic():
    """HBM bytes per launch per kernel from the committed rocprofv3 PMC passes of this command (separate
    --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), or {}."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


PREROLL_MS = 60.0              # untimed GPU time before the timed region (steady clocks)
PREROLL_MAX_STEPS = 4096


def timed_steps(model_args, global_batch, steps, warmup, dev, dist=None, preroll_ms=0.0, windows=0):
    """`steps` full training steps of one configuration on this rank; returns (seconds, trainer, model, host_s)."""
    Solver = importlib.import_module(PKG + ".nn.DVPDESolver").DVPDESolver
    trainer = importlib.import_module(PKG + ".trainer.diffusion_train")
    torch.manual_seed(1)
    model = Solver(model_args, Log(), device=dev)
    tr = trainer.FusedTrainer(model, global_batch, capacity=steps * (1 + windows) + warmup + PREROLL_MAX_STEPS)
    for _ in range(warmup):
        tr.sample()
        tr.step()
    # clock pre-roll (untimed, beyond the W warm-up steps): a short timed region right after an idle GPU measures the
    # clock ramp, not the kernels.  A 16-step probe gives the pace; the count derived from it is made the same on
    # every rank (the steps contain a collective).
    preroll = 0
    if preroll_ms > 0:
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        for _ in range(16):
            tr.sample()
            tr.step()
        torch.cuda.synchronize()
        per = (time.perf_counter() - w0) / 16 * 1e3
        if dist is not None:
            t = torch.tensor([per], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            per = float(t.item())
        more = int(min(PREROLL_MAX_STEPS - 16, max(0, round(preroll_ms / max(per, 1e-3)) - 16)))
        for _ in range(more):
            tr.sample()
            tr.step()
        preroll = 16 + more
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.sample()
        tr.step()
    t_enqueue = time.perf_counter() - t0           # host time to enqueue the K steps (no sync inside a step)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # further windows of K steps each (extra keys of the line: spread of the same measurement)
    win = []
    for _ in range(windows):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        for _ in range(steps):
            tr.sample()
            tr.step()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        win.append((time.perf_counter() - w0) / steps * 1e3)
    tr.bench_info = {"preroll_steps": preroll, "window_ms_per_step": win}
    return dt, tr, model, t_enqueue


def other_configs(dev):
    """A few full training steps of the other BASELINE.json configurations on one GPU (they are parity-test cases
    of tests/, timed here so the driver's record holds them): each with the roofline that bounds it."""
    H = 50
    cases = [
        ("config 1: 4-qubit cascade, batch 64 (the reference's CPU-runnable case)", base_args(), 64, 200, 20),
        ("config 1 at the stock script's effective batch 128 (quirk Q1)", base_args(), 128, 200, 20),
        ("config 4 shard: 4-qubit cascade, 131072 residual points on one GPU", base_args(), 131072, 50, 5),
        ("config 3: 8-qubit layered x2, batch 131072", base_args(8, 2, "layered"), 131072, 5, 2),
        ("config 5: 16-qubit cross_mesh, batch 8192 (HBM-resident statevectors)", base_args(16, 1, "cross_mesh"), 8192, 3, 1),
    ]
    out = []
    for name, margs, B, steps, warmup in cases:
        try:
            dt, tr, model, _ = timed_steps(margs, B, steps, warmup, dev)
        except Exception as e:          # a failure here must not lose the headline line
            out.append({"workload": name, "error": f"{type(e).__name__}: {e}"})
            continue
        n = margs["num_qubits"]
        F = model.quantum_layer.program.algorithmic_flops()
        flops_pt = (13 + 1 / 3) * (F + 2 * (3 * H + H * n) + 2 * (n * H + H))
        rate = steps * B / dt
        row = {"workload": name, "ms_per_step": dt / steps * 1e3, "points_per_s": rate, "steps": steps,
               "final_loss": tr.opt.read()["loss"]}
        if n >= 9:      # statevectors live in HBM: 13 1/3 channel-evaluations x (one write + one read of 2^n complex64)
            bytes_pt = (13 + 1 / 3) * 2 * (1 << n) * 8
            row.update(bound="hbm", roofline_frac=bytes_pt * rate / (PEAK_HBM_GBS * 1e9), achieved=bytes_pt * rate / 1e9,
                       peak=PEAK_HBM_GBS, unit="GB/s", algorithmic_bytes_per_point=bytes_pt)
        else:           # register / lane resident statevectors: fp32 vector pipe
            row.update(bound="valu", roofline_frac=flops_pt * rate / (PEAK_F32_TFLOPS * 1e12), achieved=flops_pt * rate / 1e12,
                       peak=PEAK_F32_TFLOPS, unit="TFLOP/s", algorithmic_flops_per_point=flops_pt)
        out.append(row)
        del tr, model
        torch.cuda.empty_cache()
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process group (never exec: this
    process may not be replaced once a GPU runtime is loaded, and it has not touched the GPU yet), relay rank 0's
    JSON line and the return code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup)]
    if a.batch_per_gpu:
        cmd += ["--batch-per-gpu", str(a.batch_per_gpu)]
    if a.no_cpu_baseline:
        cmd += ["--no-cpu-baseline"]
    cmd += ["--collective", a.collective, "--windows", str(a.windows)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return p.returncode if p.returncode != 0 or line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=0,
                    help="residual points per GPU (default: 65536 = config 2 at N = 1, 131072 = config 4's shard at N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--windows", type=int, default=4, help="extra timed windows of K steps (min / median reported beside the line)")
    ap.add_argument("--collective", choices=("torch", "rccl"), default=os.environ.get("QC_DP_COLLECTIVE", "torch"),
                    help="N > 1: the step's all-reduce through torch.distributed (default) or inside the library call "
                         "(qc_step_desc.comm: RCCL on the step's own stream, one host call per step)")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))             # before any GPU call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        sys.exit(f"--gpus {a.gpus} does not match WORLD_SIZE={world}")
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

    os.environ["QC_DP_COLLECTIVE"] = a.collective          # read by FusedTrainer
    per_gpu = a.batch_per_gpu or (65536 if world == 1 else 131072)
    args = base_args()
    global_batch = per_gpu * world
    dt, tr, model, t_enqueue = timed_steps(args, global_batch, a.steps, a.warmup, dev, dist, preroll_ms=PREROLL_MS, windows=a.windows)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    rec = tr.opt.read()
    if rank == 0:
        prog = model.quantum_layer.program
        n, H = args["num_qubits"], args["classic_network"][1]
        flops = algorithmic_flops_per_point(prog, H, n, (tr.n_ic + tr.n_bc) / max(tr.B_res, 1))
        kt_iso, kt_seq = time_kernels(tr)
        merged = "stage_circuit_bwd" in kt_iso   # the kernels the step actually launches
        # kernel durations used for the roofline: in-sequence (step order) where the step runs its merged stages - every
        # stage is ONE launch (the post stage, too, since round 3) - else each kernel back to back on warm buffers
        kt = dict(kt_iso)
        kt.update({k: v for k, v in kt_seq.items() if not k.startswith("_")})
        dom = max((k for k in kt if k in flops and k.startswith("stage_") == merged), key=lambda k: kt[k])
        stages = {k: {"ms": kt[k], "ms_isolated": kt_iso[k], "launches": 1,
                      "frac": flops[k] * tr.B_res / (kt[k] * 1e-3) / 1e12 / PEAK_F32_TFLOPS}
                  for k in kt if k in flops and k.startswith("stage_") == merged}
        ach = flops[dom] * tr.B_res / (kt[dom] * 1e-3) / 1e12
        step_ms = dt / a.steps * 1e3
        value = a.steps * global_batch / dt
        wins = [step_ms] + list(tr.bench_info["window_ms_per_step"])     # the timed region + the extra K-step windows
        traffic = measured_traffic() if (world == 1 and per_gpu == 65536) else {}
        step_bytes = sum(v for k, v in traffic.items() if k.startswith("stage_") or k in ("fold_rows", "adam")) or None
        cfg_name = ("BASELINE config 2" if (world == 1 and per_gpu == 65536) else
                    "BASELINE config 4" if global_batch == 1048576 else "BASELINE config 2 model")
        out = {
            "metric": "collocation-points/sec (PDE+BC+IC loss step), 4-qubit cascade",
            "value": value, "unit": "residual collocation points/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": step_ms, "ms_per_step_min": min(wins), "ms_per_step_median": sorted(wins)[len(wins) // 2],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: DV solver, 4 qubits, cascade, 1 layer, H=50, "
                                   f"{per_gpu} residual + 2x{per_gpu // 3} BC/IC points per GPU",
                       "global_batch": global_batch, "per_gpu_batch": per_gpu, "parallelism": f"dp{world}",
                       "collective": (a.collective if world > 1 else None),
                       "total_points_per_s": a.steps * (global_batch + 2 * (global_batch // 3)) / dt,
                       "final_loss": rec["loss"], "host_enqueue_ms_per_step": t_enqueue / a.steps * 1e3,
                       "preroll_steps_untimed": tr.bench_info["preroll_steps"], "windows_ms_per_step": wins},
            "roofline": {"bound": "valu", "pipe": "fp32 VALU (register-resident statevectors: no MFMA, no dense contraction)",
                         "kernel": dom, "achieved": ach, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_F32_TFLOPS, "traffic": traffic.get(dom),
                         "traffic_source": (TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)")
                         if traffic.get(dom) else None,
                         "kernel_ms": kt[dom], "kernel_ms_isolated": kt_iso[dom],
                         "kernel_timing": "HIP events around the five stages run back to back in step order; the sequence total distributed "
                                          "in proportion to the stages' isolated durations" if kt_seq
                                          else "HIP events around back-to-back launches of the kernel",
                         "algorithmic_flops_per_launch": flops[dom] * tr.B_res,
                         "stages": stages,
                         "step_frac": flops["step_total"] * value / world / 1e12 / PEAK_F32_TFLOPS,
                         "hbm_bytes_per_step_measured": step_bytes,
                         "hbm_GBps_measured": (step_bytes / (step_ms * 1e-3) / 1e9) if step_bytes else None,
                         "hbm_frac": (step_bytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if step_bytes else None},
            "kernels_ms": kt_iso, "stages_ms_in_sequence": kt_seq,
        }
        del tr, model
        torch.cuda.empty_cache()
        if world == 1 and not a.no_other_configs:
            out["other_configs"] = other_configs(dev)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
