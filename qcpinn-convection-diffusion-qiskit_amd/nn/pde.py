"""PDE operator of the convection-diffusion DV path: ``diffusion_operator`` with the reference's
signature and return value (nn/pde.py:53-72):

    residual = u_t/s_t + v_x u_x/s_x + v_y u_y/s_y - D (u_xx/s_x^2 + u_yy/s_y^2),   returns (u, residual)

For a ``DVPDESolver`` the five ``torch.autograd.grad(create_graph=True)`` passes of the reference
are replaced by forward-mode derivative channels carried through the HIP kernels (one launch
chain, see ``DVPDESolver.residual``); the returned tensors are attached to the autograd graph, so
``loss.backward()`` fills the parameters' ``.grad`` exactly as with the reference.  Any other
callable model (the reference's duck type, e.g. a classical solver) goes through the generic
autograd formulation below, which is the reference's algorithm.

The other operators of the reference file (Navier-Stokes, Klein-Gordon, wave, Helmholtz) are not
used by the DV trainers and are out of scope.
"""
import torch


def _grad(out, wrt):
    return torch.autograd.grad(out, wrt, torch.ones_like(out), create_graph=True)[0]


def diffusion_operator(model, t, x, y, sigma_t=1.0, sigma_x=1.0, sigma_y=1.0, D=0.01, v_x=1.0, v_y=1.0):
    t.requires_grad = True          # the reference mutates its inputs the same way (:56-58)
    x.requires_grad = True
    y.requires_grad = True
    fused = getattr(model, "residual", None)
    if fused is not None and hasattr(model, "quantum_layer"):
        # fold the sigma scalings into the coefficients: u_k/s_k and u_kk/s_k^2, whole thing / s_t on u_t
        if sigma_t != 1.0:
            raise NotImplementedError("sigma_t != 1 is not supported on the fused HIP path")
        if sigma_x != sigma_y:
            raise NotImplementedError("sigma_x != sigma_y is not supported on the fused HIP path")
        s = float(sigma_x)
        if s != 1.0:
            raise NotImplementedError("sigma_x, sigma_y != 1 are not supported on the fused HIP path")
        return fused(torch.cat((t, x, y), 1), D=D, v_x=v_x, v_y=v_y)
    u = model(torch.cat((t, x, y), 1))
    u_t = _grad(u, t) / sigma_t
    u_x = _grad(u, x) / sigma_x
    u_y = _grad(u, y) / sigma_y
    u_xx = _grad(u_x, x) / sigma_x
    u_yy = _grad(u_y, y) / sigma_y
    return u, u_t + v_x * u_x + v_y * u_y - D * (u_xx + u_yy)
