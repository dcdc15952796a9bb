"""Times one full training step of the other BASELINE.json configurations (they are parity-test cases,
not the bench line): config 3 (8-qubit layered x2, B = 131 072) and config 5 (16-qubit cross_mesh,
B = 8 192 by default; pass a smaller --b5 for a quick look).  Prints one JSON object per config with
ms/step, residual points/s and the achieved fraction of the bounding roofline."""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "qcpinn-convection-diffusion-qiskit_amd"


class Log:
    def print(self, *a):
        pass

    def get_output_dir(self):
        return "/tmp"


def run(name, n, ansatz, layers, B, steps, warmup):
    Solver = importlib.import_module(PKG + ".nn.DVPDESolver").DVPDESolver
    trainer = importlib.import_module(PKG + ".trainer.diffusion_train")
    args = {"batch_size": 64, "epochs": 0, "lr": 0.005, "seed": 1, "print_every": 10 ** 9, "num_qubits": n,
            "num_quantum_layers": layers, "classic_network": [3, 50, 1], "q_ansatz": ansatz, "shots": 1024,
            "problem": "diffusion", "solver": "DV", "encoding": "None", "use_ibm_hardware": False}
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = Solver(args, Log(), device=dev)
    tr = trainer.FusedTrainer(model, B, capacity=steps + warmup)
    for _ in range(warmup):
        tr.sample()
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.sample()
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prog = model.quantum_layer.program
    F = prog.algorithmic_flops()
    flops_pt = (13 + 1 / 3) * (F + 2 * (3 * 50 + 50 * n) + 2 * (n * 50 + 50))
    out = {"config": name, "n_qubits": n, "ansatz": ansatz, "layers": layers, "B_res": B, "ms_per_step": dt * 1e3,
           "residual_points_per_s": B / dt, "loss": tr.opt.read()["loss"],
           "valu_frac_of_157.3TF": flops_pt * B / dt / 157.3e12}
    if n >= 9:
        S = (1 << n) * 8
        out["algorithmic_hbm_bytes_per_point"] = (13 + 1 / 3) * 2 * S
        out["hbm_frac_of_8TBps"] = (13 + 1 / 3) * 2 * S * B / dt / 8.0e12
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--b3", type=int, default=131072)
    ap.add_argument("--b5", type=int, default=8192)
    ap.add_argument("--skip3", action="store_true")
    ap.add_argument("--skip5", action="store_true")
    ap.add_argument("--steps5", type=int, default=2)
    ap.add_argument("--steps3", type=int, default=5)
    ap.add_argument("--extra", default="", help="one more workload: ansatz,n_qubits,layers,batch (e.g. cross_mesh,8,1,131072)")
    a = ap.parse_args()
    if a.extra:
        ans, n, L, B = a.extra.split(",")
        run(f"{ans} n={n} L={L}", int(n), ans, int(L), int(B), steps=a.steps3, warmup=2)
    if not a.skip3:
        run("config 3", 8, "layered", 2, a.b3, steps=a.steps3, warmup=2 if a.steps3 > 1 else 1)
    if not a.skip5:
        run("config 5", 16, "cross_mesh", 1, a.b5, steps=a.steps5, warmup=1)
