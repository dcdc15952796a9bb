"""HBM-resident kernel family (9 <= n <= 20 qubits), round-structured plan (csrc/qc_circuit_hbm2.hip): parity
against the CPU oracle on programs whose plan has several stages, several tiles per statevector, predicated
controls, phase-multiply gates, tables with gradients from Walsh-Hadamard coefficients, more than one 64-point
tile per launch, and a workspace too small to keep the batch resident.  Tolerances as tests/test_gpu_circuit.py
(1e-5 on <Z>, north_star)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import pkg
from test_gpu_circuit import TOL_Z, _circuit, jets_inputs, jets_oracle, vjp_inputs, vjp_oracle

from oracle import jets as ojets
from oracle import statevector as sv

pytestmark = pytest.mark.gpu

VJP_CASES = [("cross_mesh", 12, 1, 1, 70), ("cross_mesh", 13, 1, 1, 3), ("cascade", 14, 1, 1, 66), ("layered", 13, 1, 1, 3), ("sim_circ_15", 11, 1, 1, 5),
             ("farhi", 10, 1, 1, 9), ("alternate", 9, 1, 1, 130), ("layered", 12, 2, 1, 2), ("cross_mesh", 9, 1, 1, 65)]


@pytest.mark.parametrize("ans,n,L,seed,B", VJP_CASES)
def test_value_channel_and_vjp_match_oracle(ans, n, L, seed, B, gpu_device):
    params, x, cot = vjp_inputs(ans, n, L, B, 11)
    circ, haar = _circuit(ans, n, L, seed, gpu_device)
    o = vjp_oracle(ans, n, L, seed, B, params, x, cot, tag="h2vjp")
    circ.prepare(params.to(gpu_device))
    ang = x.t().contiguous().to(gpu_device)
    qh = circ.forward_expval(ang)
    assert np.abs(qh.cpu().double().numpy() - o["q"]).max() < TOL_Z
    d_ang, d_theta = circ.backward_expval(ang, cot.to(gpu_device))
    assert np.abs(d_ang.t().cpu().double().numpy() - o["dx"]).max() < 2e-5
    scale = max(1.0, np.abs(o["dp"]).max())
    assert np.abs(d_theta.cpu().double().numpy() - o["dp"].reshape(-1)).max() < 1e-5 * scale * np.sqrt(B)
    # a workspace that holds ONE tile: the same batch in ceil(B / 64) launches, forward recomputed in the adjoint pass
    lib = circ.lib
    one = int(lib.qc_circuit_workspace_bytes(circ.handle, 1, 1))
    ws = torch.empty(one, dtype=torch.uint8, device=gpu_device)
    L_ = pkg("hip.lib")
    st = torch.cuda.current_stream(gpu_device).cuda_stream
    rows = (B + 63) // 64
    part = torch.zeros(rows, max(circ.n_params, 1), device=gpu_device)
    d2 = torch.empty_like(ang)
    L_.check(lib.qc_backward_expval(circ.handle, circ.trig.data_ptr(), None if circ.umat is None else circ.umat.data_ptr(),
                                    ang.data_ptr(), cot.to(gpu_device).data_ptr(), d2.data_ptr(), part.data_ptr(),
                                    part.shape[1], 0, B, ws.data_ptr(), one, st))
    assert torch.equal(d2, d_ang)                              # same kernels, same order: bit-identical
    assert np.abs(part.sum(0)[: circ.n_params].cpu().double().numpy() - o["dp"].reshape(-1)).max() < 1e-5 * scale * np.sqrt(B)


SIX_CASES = [("cross_mesh", 10, 1, 1, 2), ("cascade", 11, 1, 1, 1), ("layered", 9, 2, 1, 2),
             ("cross_mesh", 12, 1, 1, 2), ("cross_mesh", 13, 1, 1, 1)]   # 12 and 13: compile-time stage programs, one and two stages


@pytest.mark.parametrize("ans,n,L,seed,B", SIX_CASES)
def test_six_channels_and_cotangents_match_oracle(ans, n, L, seed, B, gpu_device):
    params, ajets, w = jets_inputs(ans, n, L, B)
    circ, haar = _circuit(ans, n, L, seed, gpu_device)
    o = jets_oracle(ans, n, L, seed, B, params, ajets, w, tag="h2jets")
    circ.prepare(params.to(gpu_device))
    aj = ajets.to(gpu_device)
    qh = circ.forward_jets(aj)
    err = np.abs(qh.cpu().double().numpy() - o["q"])
    assert err[0].max() < TOL_Z
    assert err.max() < 1e-5 * max(1.0, np.abs(o["q"]).max())
    abar, d_theta = circ.backward_jets(aj, w.to(gpu_device))
    sa = max(1.0, np.abs(o["da"]).max())
    assert np.abs(abar.cpu().double().numpy() - o["da"]).max() < 2e-5 * sa
    st = max(1.0, np.abs(o["dp"]).max())
    assert np.abs(d_theta.cpu().double().numpy() - o["dp"].reshape(-1)).max() < 2e-5 * st


@pytest.mark.parametrize("env,sel,count", [
    ({"QC_H2_RB": "4"}, "value_channel and (cascade-14 or layered-13 or alternate-9 or layered-12)", 4),
    ({"QC_NO_STATIC": "1"}, "(value_channel or six_channels) and (cross_mesh-12 or cross_mesh-13)", 4),
    ({"QC_H2S_RB": "3"}, "(value_channel or six_channels) and cross_mesh-12", 2),
    ({"QC_NO_ABSORB": "1"}, "value_channel and (cross_mesh-12 or cascade-14 or layered-13 or alternate-9)", 4)],
    ids=["interp_rb4", "no_static", "static_rb3", "no_absorb"])   # (ids without the selection text: the child must not select this test)
def test_plan_variants_pass_the_same_checks(env, sel, count):
    """QC_H2_RB=4: sixteen amplitudes per thread, 256 threads, in the plan interpreter (read once at load, hence the child
    process); QC_NO_STATIC=1: the plan interpreter instead of the generated stage programs; QC_H2S_RB=3: the generated
    program of the other tile geometry; QC_NO_ABSORB=1: leading RX layer kept as gates (no generated program matches)."""
    here = os.path.dirname(os.path.abspath(__file__))
    files = [os.path.join(here, "test_gpu_hbm2.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-q", "-x", "-k", f"({sel}) and not plan_variants"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"{count} passed" in r.stdout
