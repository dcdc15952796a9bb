"""A user-style model around a quantum layer (any class with the DVQuantumLayer interface):
Linear(d_in,H)-Tanh-Linear(H,n) -> <Z> (n,B) -> Linear(n,H)-Tanh-Linear(H,d_out), i.e. the arithmetic of the
reference's nn/DVPDESolver.py:81-110 with free input/output widths.  tests/golden/make_golden.py builds the
same module around the CPU oracle layer to produce ``other_operators.npz``."""
import torch


class Composite(torch.nn.Module):
    def __init__(self, layer, d_in, d_out, hidden=16):
        super().__init__()
        n = layer.num_qubits
        self.pre = torch.nn.Sequential(torch.nn.Linear(d_in, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, n))
        self.q = layer
        self.post = torch.nn.Sequential(torch.nn.Linear(n, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, d_out))
        self.n = n

    def forward(self, X):
        q = self.q(self.pre(X)).to(torch.float32)
        return self.post(q.T.reshape(-1, self.n))


OPERATOR_SHAPES = {"navier_stokes": (3, 3), "klein_gordon": (2, 1), "wave": (2, 1), "helmholtz": (2, 1)}


def load_case(z, name, model):
    prefix = f"{name}__w__"
    sd = {k[len(prefix):].replace("__", "."): torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    with torch.no_grad():
        for pname, p in model.named_parameters():
            p.copy_(sd[pname].to(p.device))
    return torch.from_numpy(z[f"{name}__X"])


def run_case(z, name, model, operator, device="cpu"):
    """-> (max abs output error relative to scale, |loss error|, max grad error relative to scale)"""
    X = load_case(z, name, model).to(device)
    cols = [X[:, i:i + 1].clone() for i in range(X.shape[1])]
    res = list(operator(model, *cols))
    loss = sum((r ** 2).mean() * (i + 1) for i, r in enumerate(res))
    model.zero_grad()
    loss.backward()
    g = torch.cat([(torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1) for p in model.parameters()])
    e_out = 0.0
    for i, r in enumerate(res):
        ref = torch.from_numpy(z[f"{name}__out{i}"])
        e_out = max(e_out, float((r.detach().cpu() - ref).abs().max() / max(1.0, float(ref.abs().max()))))
    ref_g = torch.from_numpy(z[f"{name}__grad"])
    e_g = float((g.detach().cpu() - ref_g).abs().max() / max(1.0, float(ref_g.abs().max())))
    ref_l = float(z[f"{name}__loss"])
    return e_out, abs(loss.item() - ref_l) / max(1.0, abs(ref_l)), e_g
