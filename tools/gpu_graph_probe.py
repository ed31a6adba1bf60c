"""Per-launch cost of small dependent kernels: eager ctypes launches vs one hipGraph replay (torch.cuda.CUDAGraph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import hip

dev = "cuda"
x = torch.zeros(4096, device=dev)
N = 1000


def body():
    for i in range(N):
        hip.fill(x, float(i))


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter(); body(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("eager: enqueue %.2f us/launch, finish %.2f us/launch" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graph: %.2f us/launch" % ((t2 - t0) / 5 / N * 1e6))
print("x[0] =", float(x[0]))
