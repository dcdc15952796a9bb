"""The Navier-Stokes / Klein-Gordon / wave / Helmholtz operators of nn/pde.py (reference :2-52,73-95) against
``tests/golden/other_operators.npz``, which the REFERENCE's own functions produced on an oracle-backed
composite model.  CPU: the operator formulas on the same oracle-backed model (host logic).  GPU: the same
operators on a model around the HIP DVQuantumLayer (forward, create_graph reverse, loss.backward)."""
import os

import numpy as np
import pytest
import torch

from composite_model import OPERATOR_SHAPES, Composite, run_case
from conftest import GOLDEN, pkg

ARGS = {"batch_size": 64, "epochs": 20, "lr": 0.005, "seed": 1, "print_every": 100, "num_qubits": 4,
        "num_quantum_layers": 1, "classic_network": [3, 50, 1], "q_ansatz": "cascade", "shots": 1024,
        "problem": "diffusion", "solver": "DV", "encoding": "None", "use_ibm_hardware": False}
OPS = {"navier_stokes": "navier_stokes_2D_operator", "klein_gordon": "klein_gordon_operator",
       "wave": "wave_operator", "helmholtz": "helmholtz_operator"}


@pytest.mark.parametrize("name", sorted(OPS))
def test_operator_formulas_on_oracle_model(name):
    from oracle import solver as osolver
    z = np.load(os.path.join(GOLDEN, "other_operators.npz"))
    d_in, d_out = OPERATOR_SHAPES[name]
    model = Composite(osolver.OracleQuantumLayer(ARGS), d_in, d_out)
    e_out, e_loss, e_grad = run_case(z, name, model, getattr(pkg("nn.pde"), OPS[name]))
    assert e_out < 1e-6 and e_loss < 1e-6 and e_grad < 1e-6, (e_out, e_loss, e_grad)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(OPS))
def test_operators_on_hip_layer_model(name, gpu_device):
    z = np.load(os.path.join(GOLDEN, "other_operators.npz"))
    d_in, d_out = OPERATOR_SHAPES[name]
    layer = pkg("nn.DVQuantumLayer").DVQuantumLayer(ARGS)
    model = Composite(layer, d_in, d_out).to(gpu_device)
    e_out, e_loss, e_grad = run_case(z, name, model, getattr(pkg("nn.pde"), OPS[name]), gpu_device)
    assert e_out < 1e-4 and e_loss < 1e-4 and e_grad < 2e-4, (e_out, e_loss, e_grad)
