"""Command-line entry of the second workload (pure-diffusion hybrid QPINN), same flags as the reference's
script of this name; the implementation lives in the package (qcpinn-convection-diffusion-qiskit_amd/train_hybrid_qpinn.py)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    importlib.import_module("qcpinn-convection-diffusion-qiskit_amd.train_hybrid_qpinn").main()
