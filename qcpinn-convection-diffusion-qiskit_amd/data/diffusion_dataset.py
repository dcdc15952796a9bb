"""Collocation samplers and the analytic solution / forcing term of the convection-diffusion
problem, with the reference's names and semantics (data/diffusion_dataset.py:12-56).

``u = exp(-100((x-.5)^2+(y-.5)^2)) exp(-t)``.  ``r`` is the forcing term *as the reference
evaluates it*: its second derivatives carry ``-400`` where calculus gives ``-200`` (:31-34), i.e.
``r = true_residual(u) + 4 u``; that is the training target, so it is reproduced, not "fixed".
The fused HIP step evaluates the same two functions on device (csrc/qc_mlp.hip); these torch
versions serve callers that want the targets as tensors (evaluation, generic models).
"""
import torch

default_D = 0.01
default_v_x = 1.0
default_v_y = 1.0


class Sampler:
    """Uniform points in the box ``coords = [[lo...],[hi...]]`` plus ``func`` of them."""

    def __init__(self, dim, coords, func, name=None, device="cpu"):
        self.dim, self.coords, self.func, self.name, self.device = dim, coords, func, name, device

    def sample(self, N):
        lo, hi = self.coords[0:1, :], self.coords[1:2, :]
        pts = lo + (hi - lo) * torch.rand(N, self.dim, device=self.device)
        return pts, self.func(pts.to(self.device))


def _parts(txy):
    """(u, x - 1/2, y - 1/2) of the Gaussian pulse u = exp(-100 |(x, y) - (1/2, 1/2)|^2) * exp(-t)."""
    dx = txy[:, 1:2] - 0.5
    dy = txy[:, 2:3] - 0.5
    val = torch.exp(-100 * (dx ** 2 + dy ** 2)) * torch.exp(-txy[:, 0:1])
    return val, dx, dy


def u(txy):
    return _parts(txy)[0]


def u_t(txy):
    return -_parts(txy)[0]


def u_x(txy):
    val, dx, _ = _parts(txy)
    return -200 * dx * val


def u_y(txy):
    val, _, dy = _parts(txy)
    return -200 * dy * val


# Second derivatives as the reference writes them: the constant is -400 where d2/dx2 of the pulse
# gives -200 (data/diffusion_dataset.py:31-34).  Kept on purpose: `r` below is the training target.
_REF_CONST = 400


def u_xx(txy):
    val, dx, _ = _parts(txy)
    return (40000 * dx ** 2 - _REF_CONST) * val


def u_yy(txy):
    val, _, dy = _parts(txy)
    return (40000 * dy ** 2 - _REF_CONST) * val


def r(txy, Diffusion=default_D, v_x=default_v_x, v_y=default_v_y):
    val, dx, dy = _parts(txy)
    first = -val - 200 * val * (v_x * dx + v_y * dy)
    second = (40000 * (dx ** 2 + dy ** 2) - 2 * _REF_CONST) * val
    return first - Diffusion * second


BOXES = {  # [[t,x,y] low, high]  (trainer/diffusion_train.py:9-20)
    "ics": ((0.0, 0.0, 0.0), (0.0, 1.0, 1.0)),
    "bc1": ((0.0, 0.0, 0.0), (1.0, 0.0, 1.0)),
    "bc2": ((0.0, 1.0, 0.0), (1.0, 1.0, 1.0)),
    "dom": ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)),
}


def box(name, device):
    return torch.tensor(BOXES[name], dtype=torch.float32, device=device)


def generate_training_dataset(device):
    ics = Sampler(3, box("ics", device), u, name="Initial Condition", device=device)
    bcs = [Sampler(3, box("bc1", device), u, name="Dirichlet BC1", device=device),
           Sampler(3, box("bc2", device), u, name="Dirichlet BC2", device=device)]
    res = Sampler(3, box("dom", device), r, name="Forcing", device=device)
    return [ics, bcs, res]
