"""Achievable HBM bandwidth of plain streaming kernels on COLD operands (rotating buffer sets > the 256 MB memory-side cache):
the yardstick for the HBM-bound conv / depthwise kernels.  python tools/gpu_hbm_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.gpu_microbench import timeit
from lm_net_amd import hip
dev = "cuda"
NSET = 8
n = 8 * 352 * 352 * 24          # one level-0 E-wide tensor: 95 MB
xs = [torch.randn(n, device=dev) for _ in range(NSET)]
ys = [torch.empty(n, device=dev) for _ in range(NSET)]
hs = [torch.randn(n // 2, device=dev) for _ in range(NSET)]
c = [0]
def rot():
    c[0] = (c[0] + 1) % NSET
    return c[0]
def copy():
    i = rot(); ys[i].copy_(xs[i])
def scale():
    i = rot(); torch.mul(xs[i], 2.0, out=ys[i])
def read():
    i = rot(); xs[i].sum()
def fill():
    i = rot(); hip.fill(ys[i], 0.0)
def expand():    # read n/2, write n (the byte mix of a 12 -> 24 1x1 conv)
    i = rot(); torch.cat([hs[i], hs[i]], out=ys[i])
for name, f, by in (("copy (1 read : 1 write)", copy, 2 * n * 4), ("scale (1:1)", scale, 2 * n * 4), ("sum (read only)", read, n * 4),
                    ("fill (write only)", fill, n * 4), ("cat (1 read : 2 write)", expand, 1.5 * n * 4)):
    t = timeit(f, iters=40, warm=8)
    print("%-28s %7.1f us  %6.0f GB/s" % (name, t * 1e6, by / t / 1e9))
