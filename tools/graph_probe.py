"""Experiment: replay the fused step as a captured HIP graph (same batch every replay - timing only)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
PKG = bench.PKG
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
Solver = importlib.import_module(PKG + ".nn.DVPDESolver").DVPDESolver
trainer = importlib.import_module(PKG + ".trainer.diffusion_train")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(1)
model = Solver(bench.base_args(), bench.Log(), device=dev)
tr = trainer.FusedTrainer(model, B, capacity=100000)
def one():
    tr.sample(); tr.step()
for _ in range(20): one()
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K): one()
torch.cuda.synchronize()
print("stream launches ms/step", (time.perf_counter() - t0) / K * 1e3)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): one()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    one()
torch.cuda.synchronize()
for _ in range(20): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): g.replay()
torch.cuda.synchronize()
print("graph replay ms/step", (time.perf_counter() - t0) / K * 1e3)
