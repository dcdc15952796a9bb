"""GPU parity of the variational-circuit kernels (through the C ABI) against the CPU oracle and
the committed golden <Z> vectors.  Tolerance: 1e-5 absolute on expectation values (north_star)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, cached_oracle, pkg

from oracle import jets as ojets
from oracle import statevector as sv

pytestmark = pytest.mark.gpu

TOL_Z = 1e-5


def _circuit(ans, n, L, seed, device):
    circuits = pkg("circuits")
    engine = pkg("hip.engine")
    use_haar = seed is not None and n >= 4
    prog = circuits.build_program(ans, n, L, use_haar)
    haar = circuits.haar_unitaries(seed, seed + 1) if use_haar else None
    return engine.Circuit(prog, haar, device), (sv.haar_pair(seed, seed + 1) if use_haar else None)


# ---- seeded inputs and float64 oracle outputs of the parity cases (the oracle part is cached: conftest.cached_oracle;
# tests/golden/make_oracle_cache.py calls these same functions in the build container)
def vjp_inputs(ans, n, L, B, salt):
    g = torch.Generator().manual_seed(salt + n + B)
    P = pkg("circuits").params_per_layer(ans, n)
    params = torch.randn(L, P, generator=g) * 0.8
    x = torch.randn(B, n, generator=g) * 1.1
    cot = torch.randn(n, B, generator=g)
    return params, x, cot


def vjp_oracle(ans, n, L, seed, B, params, x, cot, encoding="angle", tag="vjp"):
    haar = sv.haar_pair(seed, seed + 1) if (seed is not None and n >= 4) else None

    def compute():
        xo = x.double().requires_grad_(True)
        po = params.double().requires_grad_(True)
        q = sv.circuit_expvals(xo, po, ans, n, haar) if encoding == "angle" else sv.circuit_expvals(xo, po, ans, n, haar, encoding)
        (q * cot.double()).sum().backward()
        return {"q": q.detach().numpy(), "dx": xo.grad.numpy(), "dp": po.grad.numpy()}
    return cached_oracle(f"{tag}_{ans}_n{n}_L{L}_s{seed}_B{B}", (params.numpy(), x.numpy(), cot.numpy()), compute)


def jets_inputs(ans, n, L, B, salt=77):
    g = torch.Generator().manual_seed(salt + n + B)
    P = pkg("circuits").params_per_layer(ans, n)
    params = torch.randn(L, P, generator=g) * 0.8
    ajets = torch.randn(6, n, B, generator=g) * 0.9
    w = torch.randn(6, n, B, generator=g)
    return params, ajets, w


def jets_oracle(ans, n, L, seed, B, params, ajets, w, encoding="angle", tag="jets"):
    haar = sv.haar_pair(seed, seed + 1) if (seed is not None and n >= 4) else None

    def compute():
        ao = ajets.double().requires_grad_(True)
        po = params.double().requires_grad_(True)
        qo = ojets.qjets_from_ajets(ao, po, ans, n, haar) if encoding == "angle" else ojets.qjets_from_ajets(ao, po, ans, n, haar, encoding)
        (qo * w.double()).sum().backward()
        return {"q": qo.detach().numpy(), "da": ao.grad.numpy(), "dp": po.grad.numpy()}
    return cached_oracle(f"{tag}_{ans}_n{n}_L{L}_s{seed}_B{B}", (params.numpy(), ajets.numpy(), w.numpy()), compute)


def _golden_cases(max_n):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "expval_*.npz"))):
        z = np.load(f)
        n = z["x"].shape[1]
        if n <= max_n:
            out.append(os.path.basename(f))
    return out


SUPPORTED_N = 16


@pytest.mark.parametrize("fname", _golden_cases(SUPPORTED_N))
def test_expval_matches_golden(fname, gpu_device):
    z = np.load(os.path.join(GOLDEN, fname))
    stem = fname[len("expval_"):-len(".npz")]
    ans, ntag, ltag = stem.rsplit("_", 2)
    n, L = int(ntag[1:]), int(ltag[1:])
    seed = int(z["seed"]) if int(z["seed"]) >= 0 else None
    circ, _ = _circuit(ans, n, L, seed, gpu_device)
    params = torch.from_numpy(z["params"]).to(gpu_device)
    x = torch.from_numpy(z["x"]).to(gpu_device)
    circ.prepare(params)
    q = circ.forward_expval(x.t().contiguous()).cpu().numpy()
    assert q.shape == z["expval"].shape
    assert np.abs(q - z["expval"]).max() < TOL_Z


VJP_CASES = [
    ("cascade", 4, 1, 1, 200), ("layered", 4, 2, 1, 70), ("cross_mesh", 4, 1, 1, 65), ("farhi", 4, 1, 1, 64),
    ("sim_circ_15", 4, 1, 1, 33), ("alternate", 5, 1, 1, 40), ("cascade", 3, 2, None, 50), ("cascade", 2, 1, None, 10),
    ("cascade", 5, 1, 1, 129), ("cascade", 4, 1, None, 300),
    ("cascade", 6, 1, 1, 70), ("layered", 7, 1, 1, 20), ("layered", 8, 2, 1, 37), ("cross_mesh", 8, 1, 1, 5),
    ("farhi", 6, 1, 1, 18), ("cascade", 9, 1, 1, 70), ("layered", 10, 1, 1, 5), ("cross_mesh", 16, 1, 1, 2),
]


@pytest.mark.parametrize("ans,n,L,seed,B", VJP_CASES)
def test_expval_vjp_matches_oracle_autograd(ans, n, L, seed, B, gpu_device):
    """Backward of DVQuantumLayer: d/d(angles) and d/d(theta) of sum(cot * <Z>)."""
    params, x, cot = vjp_inputs(ans, n, L, B, 5)
    circ, haar = _circuit(ans, n, L, seed, gpu_device)
    o = vjp_oracle(ans, n, L, seed, B, params, x, cot)
    # HIP
    circ.prepare(params.to(gpu_device))
    ang = x.t().contiguous().to(gpu_device)
    qh = circ.forward_expval(ang)
    assert np.abs(qh.cpu().double().numpy() - o["q"]).max() < TOL_Z
    d_ang, d_theta = circ.backward_expval(ang, cot.to(gpu_device))
    assert np.abs(d_ang.t().cpu().double().numpy() - o["dx"]).max() < 2e-5
    scale = max(1.0, np.abs(o["dp"]).max())
    assert np.abs(d_theta.cpu().double().numpy() - o["dp"].reshape(-1)).max() < 1e-5 * scale * np.sqrt(B)


JETS_CASES = [
    ("cascade", 4, 1, 1, 70), ("layered", 4, 1, 1, 9), ("cross_mesh", 4, 1, 1, 6), ("cascade", 3, 1, None, 8),
    ("cascade", 2, 1, None, 5), ("alternate", 5, 1, 1, 5),
    ("cascade", 6, 1, 1, 3), ("layered", 7, 1, 1, 2), ("layered", 8, 1, 1, 1), ("sim_circ_15", 6, 1, 1, 2),
    ("cascade", 7, 1, 1, 1),   # no generated static program: the run-time interpreter of the wave family
    ("cascade", 9, 1, 1, 1),
    ("layered", 8, 2, 1, 3), ("layered", 10, 1, 1, 2),   # (the float64 oracle's jets cost ~1 min per case: cached)
    ("cross_mesh", 8, 1, 1, 2),   # compile-time program whose 56 CRZ + 8 RZ gates are ONE phase-table run (qc_wave_sched.h)
]


@pytest.mark.parametrize("ans,n,L,seed,B", JETS_CASES)
def test_jets_forward_and_vjp_match_oracle(ans, n, L, seed, B, gpu_device):
    """Six derivative channels through the circuit and their cotangents (angle jets + theta)."""
    params, ajets, w = jets_inputs(ans, n, L, B)
    circ, haar = _circuit(ans, n, L, seed, gpu_device)
    o = jets_oracle(ans, n, L, seed, B, params, ajets, w)
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


def test_wave_family_agrees_with_oracle_at_small_n():
    """The lanes-as-amplitudes family is the generic twin of the register family: force n <= 5 through
    it (QC_FORCE_WAVE=1 is read once at library load, hence the child process) and re-run the n <= 5
    parity cases above."""
    import subprocess
    import sys
    env = dict(os.environ, QC_FORCE_WAVE="1")
    here = os.path.dirname(os.path.abspath(__file__))
    sel = ("(golden or vjp or jets) and not wave_family and not "
           "(n6 or n7 or n8 or n10 or n16 or -6- or -7- or -8- or -9- or -10- or -16-)")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_circuit.py"), "-m", "gpu", "-q",
                        "-x", "-k", sel], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_plan_interpreter_agrees_with_the_oracle_like_the_generated_programs():
    """n >= 9: gate programs with a generated compile-time stage program run it by default; QC_NO_STATIC=1 keeps the
    run-time plan interpreter for them (and the run-time gate interpreters of the other families).  Both must pass
    the same golden / oracle checks (child process: the switch is read once at library load)."""
    import subprocess
    import sys
    env = dict(os.environ, QC_NO_STATIC="1")
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "(golden and (n10 or n16)) or (expval_vjp and (cascade-9 or layered-10 or cross_mesh-16))"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_circuit.py"), "-m", "gpu", "-q",
                        "-x", "-k", sel], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "5 passed" in r.stdout


AMP_CASES = [("cascade", 4, 1, 1, 70), ("layered", 5, 1, 1, 9), ("layered", 6, 1, 1, 5),
             ("layered", 8, 1, 1, 2), ("cascade", 9, 1, 1, 2), ("cascade", 2, 1, None, 6)]


def amp_inputs(ans, n, L, B):
    g = torch.Generator().manual_seed(31 + n + B)
    P = pkg("circuits").params_per_layer(ans, n)
    params = torch.randn(L, P, generator=g) * 0.8
    x = torch.randn(B, n, generator=g) + 0.3
    cot = torch.randn(n, B, generator=g)
    ajets = torch.randn(6, n, B, generator=g) * 0.9
    ajets[0] += 0.3
    w = torch.randn(6, n, B, generator=g)
    return params, x, cot, ajets, w


@pytest.mark.parametrize("ans,n,L,seed,B", AMP_CASES)
def test_amplitude_encoding_matches_oracle(ans, n, L, seed, B, gpu_device):
    """encoding="amplitude" (AmplitudeEmbedding, nn/DVQuantumLayer.py:177-180): <Z>, its vjp, the six
    derivative channels and their cotangents, in all three kernel families."""
    circuits = pkg("circuits")
    engine = pkg("hip.engine")
    use_haar = seed is not None and n >= 4
    prog = circuits.build_program(ans, n, L, use_haar)
    haar_np = circuits.haar_unitaries(seed, seed + 1) if use_haar else None
    circ = engine.Circuit(prog, haar_np, gpu_device, amplitude=True)
    params, x, cot, ajets, w = amp_inputs(ans, n, L, B)
    o = vjp_oracle(ans, n, L, seed, B, params, x, cot, "amplitude", "ampvjp")
    circ.prepare(params.to(gpu_device))
    ang = x.t().contiguous().to(gpu_device)
    qh = circ.forward_expval(ang)
    assert np.abs(qh.cpu().double().numpy() - o["q"]).max() < TOL_Z
    d_ang, d_theta = circ.backward_expval(ang, cot.to(gpu_device))
    assert np.abs(d_ang.t().cpu().double().numpy() - o["dx"]).max() < 2e-5 * max(1.0, np.abs(o["dx"]).max())
    assert np.abs(d_theta.cpu().double().numpy() - o["dp"].reshape(-1)).max() < 2e-5 * max(1.0, np.abs(o["dp"]).max()) * np.sqrt(B)
    # derivative channels
    oj = jets_oracle(ans, n, L, seed, B, params, ajets, w, "amplitude", "ampjets")
    aj = ajets.to(gpu_device)
    qj = circ.forward_jets(aj)
    assert np.abs(qj.cpu().double().numpy() - oj["q"]).max() < 2e-5 * max(1.0, np.abs(oj["q"]).max())
    abar, d_theta = circ.backward_jets(aj, w.to(gpu_device))
    assert np.abs(abar.cpu().double().numpy() - oj["da"]).max() < 5e-5 * max(1.0, np.abs(oj["da"]).max())
    assert np.abs(d_theta.cpu().double().numpy() - oj["dp"].reshape(-1)).max() < 5e-5 * max(1.0, np.abs(oj["dp"]).max())
