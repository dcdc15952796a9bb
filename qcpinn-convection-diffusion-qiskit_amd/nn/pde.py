"""PDE operator of the convection-diffusion DV path: ``diffusion_operator`` with the reference's
signature and return value (nn/pde.py:53-72):

    residual = u_t/s_t + v_x u_x/s_x + v_y u_y/s_y - D (u_xx/s_x^2 + u_yy/s_y^2),   returns (u, residual)

For a ``DVPDESolver`` the five ``torch.autograd.grad(create_graph=True)`` passes of the reference
are replaced by forward-mode derivative channels carried through the HIP kernels (one launch
chain, see ``DVPDESolver.residual``); the returned tensors are attached to the autograd graph, so
``loss.backward()`` fills the parameters' ``.grad`` exactly as with the reference.  Any other
callable model (the reference's duck type, e.g. a classical solver) goes through the generic
autograd formulation below, which is the reference's algorithm.

The other operators of the reference file (nn/pde.py:2-52,73-95: Navier-Stokes 2-D, Klein-Gordon,
wave, Helmholtz) are provided in the same autograd formulation, with the reference's signatures,
constants and return shapes.  None of the DV trainers uses them; they run on any model, including
user-composed models around ``DVQuantumLayer`` (whose ``create_graph=True`` reverse pass supplies the
second derivatives).  For a two-input ``DVPDESolver`` (``classic_network = [2, H, 1]``) the three scalar
second-order operators (Klein-Gordon, wave, Helmholtz) take their second derivatives from the fused
derivative-channel kernels (``DVPDESolver.second_order``: any qubit count, both encodings); for a three-output
``DVPDESolver`` (``classic_network = [3, H, 3]``) the Navier-Stokes operator takes the six derivative channels of
u, v and p from the same kernels (``DVPDESolver.jets``) and forms the momentum equations' products in torch.
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
        # the sigma scalings are folded into the operator coefficients the kernels take (qc_pde.c_*, d_*)
        return fused(torch.cat((t, x, y), 1), D=D, v_x=v_x, v_y=v_y, sigma=(sigma_t, sigma_x, sigma_y))
    u = model(torch.cat((t, x, y), 1))
    u_t = _grad(u, t) / sigma_t
    u_x = _grad(u, x) / sigma_x
    u_y = _grad(u, y) / sigma_y
    u_xx = _grad(u_x, x) / sigma_x
    u_yy = _grad(u_y, y) / sigma_y
    return u, u_t + v_x * u_x + v_y * u_y - D * (u_xx + u_yy)


def _track(*coords):
    for c in coords:
        c.requires_grad_(True)      # the reference sets .requires_grad = True on its inputs as well


def _second(out, wrt):
    first = _grad(out, wrt)
    return first, _grad(first, wrt)


def navier_stokes_2D_operator(model, t, x, y, min_x=0, max_x=1):
    """Reference nn/pde.py:2-27.  model: (B,3) -> (B,3) = (u, v, p); returns [continuity, f_u, f_v]
    with viscosity 0.00345 and density 1056 (the reference's constants)."""
    viscosity, density = 0.00345, 1056.0
    _track(t, x, y)
    jets = getattr(model, "jets", None)
    if jets is not None and getattr(model, "n_out", 0) == 3 and getattr(model, "input_dim", 0) == 3:
        # a three-output DVPDESolver: the six derivative channels of u, v and p from the fused kernels, the
        # products of the momentum equations formed here (columns: value, t, x, y, xx, yy)
        X = torch.cat((t, x, y), 1)
        J = model.jets_all(X)                       # (B, 3, 6): one pass for the three outputs
        U, V, P = J[:, 0], J[:, 1], J[:, 2]
        c = lambda J, k: J[:, k:k + 1]
        u, v = c(U, 0), c(V, 0)
        f_u = c(U, 1) + (u * c(U, 2) + v * c(U, 3)) + c(P, 2) / density - viscosity * (c(U, 4) + c(U, 5))
        f_v = c(V, 1) + (u * c(V, 2) + v * c(V, 3)) + c(P, 3) / density - viscosity * (c(V, 4) + c(V, 5))
        return [c(U, 2) + c(V, 3), f_u, f_v]
    fields = model(torch.cat((t, x, y), 1))
    u, v, p = fields[:, 0:1], fields[:, 1:2], fields[:, 2:3]
    mom = []
    for w, p_k in ((u, _grad(p, x)), (v, _grad(p, y))):
        w_t = _grad(w, t)
        w_x, w_xx = _second(w, x)
        w_y, w_yy = _second(w, y)
        mom.append((w_x, w_y, w_t + (u * w_x + v * w_y) + p_k / density - viscosity * (w_xx + w_yy)))
    (u_x, _, f_u), (_, v_y, f_v) = mom
    return [u_x + v_y, f_u, f_v]


def _fused_second_order(model):
    """A two-input DVPDESolver: second derivatives come from the fused derivative-channel kernels."""
    return getattr(model, "second_order", None) if getattr(model, "input_dim", 0) == 2 else None


def klein_gordon_operator(fluid_model, t, x, x_min=0.0, x_max=1.0):
    """Reference nn/pde.py:28-41: u_tt - u_xx + 0*u + u^3; returns (u, residual)."""
    alpha, beta, gamma, power = -1.0, 0.0, 1.0, 3
    _track(t, x)
    fused = _fused_second_order(fluid_model)
    if fused is not None:
        u, lin = fused(torch.cat((t, x), 1), 1.0, alpha)          # u_tt + alpha u_xx in the kernels
        return u, lin + beta * u + gamma * u ** power
    u = fluid_model(torch.cat((t, x), 1))
    _, u_tt = _second(u, t)
    _, u_xx = _second(u, x)
    return u, u_tt + alpha * u_xx + beta * u + gamma * u ** power


def wave_operator(model, t, x, sigma_t=1.0, sigma_x=1.0):
    """Reference nn/pde.py:42-52: u_tt - c^2 u_xx with c = 2 (the sigma arguments are accepted and, as in
    the reference, unused); returns (u, residual)."""
    speed = 2
    _track(t, x)
    fused = _fused_second_order(model)
    if fused is not None:
        return fused(torch.cat((t, x), 1), 1.0, -float(speed ** 2))
    u = model(torch.cat((t, x), 1))
    _, u_tt = _second(u, t)
    _, u_xx = _second(u, x)
    return u, u_tt - speed ** 2 * u_xx


def helmholtz_operator(fluid_model, x1, x2):
    """Reference nn/pde.py:73-95: u_x1x1 + u_x2x2 + lambda*u with lambda = 1; returns [u, residual]."""
    lam = 1.0
    _track(x1, x2)
    fused = _fused_second_order(fluid_model)
    if fused is not None:
        u, lin = fused(torch.cat((x1, x2), 1), 1.0, 1.0)
        return [u, lin + lam * u]
    u = fluid_model(torch.cat((x1, x2), 1))
    _, u_11 = _second(u, x1)
    _, u_22 = _second(u, x2)
    return [u, u_11 + u_22 + lam * u]
