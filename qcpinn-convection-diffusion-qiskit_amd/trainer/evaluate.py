"""Grid evaluation of a trained solver — the step that follows training in the reference's entry
script (trainer/diffusion_hybrid_trainer.py:126-184): a ``num_points``^3 grid over [0,1]^3 goes
through ``diffusion_operator`` (u and PDE residual in one fused HIP pass for a ``DVPDESolver``) and
is compared with the analytic ``u`` / forcing term ``r``:

    error_u = ||u - u_pred||_2 / ||u||_2 * 100,   error_f = ||r - f_pred||_2 / ||r + 1e-9||_2 * 100

Contour plotting (utils/ContourPlotter.py) is out of scope; the grids are returned instead.
"""
from __future__ import annotations

import numpy as np
import torch

from ..data.diffusion_dataset import r, u
from ..nn.pde import diffusion_operator


def evaluation_grid(num_points: int = 20, device=None) -> torch.Tensor:
    """(num_points^3, 3) points, t-major then x then y ('ij' meshgrid of three linspace(0, 1))."""
    axis = torch.linspace(0.0, 1.0, num_points, dtype=torch.float32, device=device)
    t, x, y = torch.meshgrid(axis, axis, axis, indexing="ij")
    return torch.stack((t.flatten(), x.flatten(), y.flatten()), dim=1).contiguous()


def evaluate(model, num_points: int = 20, device=None) -> dict:
    device = device if device is not None else getattr(model, "device", None)
    X = evaluation_grid(num_points, device)
    u_pred, f_pred = diffusion_operator(model, X[:, 0:1].clone(), X[:, 1:2].clone(), X[:, 2:3].clone())
    u_pred = u_pred.detach().cpu().numpy()
    f_pred = f_pred.detach().cpu().numpy()
    u_ref = u(X).cpu().numpy()
    f_ref = r(X).cpu().numpy()
    err_u = np.linalg.norm(u_ref - u_pred, 2) / np.linalg.norm(u_ref, 2) * 100.0
    err_f = np.linalg.norm(f_ref - f_pred, 2) / np.linalg.norm(f_ref + 1e-9, 2) * 100.0
    logger = getattr(model, "logger", None)
    if logger is not None:
        logger.print("Relative L2 error_u: {:.2e}".format(err_u))
        logger.print("Relative L2 error_f: {:.2e}".format(err_f))
    shape = (num_points,) * 3
    return {"error_u": float(err_u), "error_f": float(err_f), "X": X.cpu().numpy(),
            "u_pred": u_pred.reshape(shape), "f_pred": f_pred.reshape(shape),
            "u_analytic": u_ref.reshape(shape), "f_analytic": f_ref.reshape(shape)}
