"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see statevector.py).

Derivative channels of the quantum layer computed the plain way: along each input direction k the
angles are a curve  a(e) = a + e*a_k + e^2/2*a_kk  and the channels are ordinary torch-autograd
derivatives of <Z>(a(e)) w.r.t. the scalar e at e = 0 (batched: one e per point),
    q_k = d/de <Z>,   q_kk = d2/de2 <Z>,
which is what nn/pde.py:59-70 obtains with autograd.grad(create_graph=True) through the simulator
when a(X) is the pre-network.  Everything stays on the autograd graph, so gradients of any function
of the channels w.r.t. every angle-jet entry and circuit parameter follow by .backward().
Channels: 0 value, 1 d/dt, 2 d/dx, 3 d/dy, 4 d2/dx2, 5 d2/dy2.
"""
import torch

from . import statevector as sv


def qjets_from_ajets(ajets, params, q_ansatz, n, haar=None, encoding="angle"):
    """ajets: (6, n, B) float64 (may require grad); returns (6, n, B) float64, differentiable."""
    B = ajets.shape[2]
    a0 = ajets[0].T                                                       # (B, n)
    chans = {0: sv.circuit_expvals(a0, params, q_ansatz, n, haar, encoding)}        # (n, B)
    for k in (1, 2, 3):
        eps = torch.zeros(B, dtype=torch.float64, requires_grad=True)
        curve = a0 + eps[:, None] * ajets[k].T
        if k >= 2:
            curve = curve + 0.5 * eps[:, None] ** 2 * ajets[k + 2].T
        q = sv.circuit_expvals(curve, params, q_ansatz, n, haar, encoding)
        q1 = torch.stack([torch.autograd.grad(q[i].sum(), eps, create_graph=True)[0] for i in range(n)])
        chans[k] = q1
        if k >= 2:
            chans[k + 2] = torch.stack(
                [torch.autograd.grad(q1[i].sum(), eps, create_graph=True)[0] for i in range(n)])
    return torch.stack([chans[c] for c in range(6)])
