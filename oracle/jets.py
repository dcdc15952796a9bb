"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see statevector.py).

Derivative channels of the quantum layer computed the slow, obviously-correct way: per-point
Jacobian and Hessian of <Z>(angles) by nested torch autograd through ``statevector.circuit_expvals``,
then the chain rule
    q_k  = J a_k ,      q_kk = J a_kk + a_k^T H a_k .
This is what nn/pde.py:59-70 obtains implicitly with autograd.grad(create_graph=True) through the
simulator; the HIP jet kernels must reproduce it (and, through autograd on this construction, its
gradients w.r.t. every angle-jet entry and circuit parameter).
Channels: 0 value, 1 d/dt, 2 d/dx, 3 d/dy, 4 d2/dx2, 5 d2/dy2.
"""
import torch

from . import statevector as sv


def qjets_from_ajets(ajets, params, q_ansatz, n, haar=None):
    """ajets: (6, n, B) float64 (may require grad); returns (6, n, B) float64, differentiable."""
    B = ajets.shape[2]
    outs = []
    for p in range(B):
        a = ajets[0, :, p]

        def f(v):
            return sv.circuit_expvals(v[None, :], params, q_ansatz, n, haar)[:, 0]

        q = f(a)
        J = torch.autograd.functional.jacobian(f, a, create_graph=True)            # (n, n)
        H = torch.autograd.functional.jacobian(
            lambda v: torch.autograd.functional.jacobian(f, v, create_graph=True), a, create_graph=True)  # (n,n,n)
        ch = [q]
        for k in (1, 2, 3):
            ch.append(J @ ajets[k, :, p])
        for kk, k in ((4, 2), (5, 3)):
            d = ajets[k, :, p]
            ch.append(J @ ajets[kk, :, p] + torch.einsum("ijl,j,l->i", H, d, d))
        outs.append(torch.stack(ch))                                                 # (6, n)
    return torch.stack(outs, dim=2)
