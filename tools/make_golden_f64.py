"""Gradient goldens at the BASELINE image sizes from the REAL reference run in float64 (build container only).

    python tools/make_golden_f64.py 352 2 31     -> tests/golden/train_f64_352_b2.npz
    python tools/make_golden_f64.py 512 2 33     -> tests/golden/train_f64_512_b2.npz

Why float64: at 352x352 / 512x512 every weight gradient is a sum of 10^5..10^6 terms that passes through BatchNorm
cancellations; the reference's own fp32 CPU autograd is only good to ~1e-2 on single elements there (measured:
dconv3.1.expand_conv.1.weight, 512x512: fp32 CPU vs HIP 1.3e-2, fp64 CPU vs HIP 1.4e-3), so an fp32 golden would force a
tolerance that hides real kernel errors.  The same reference code in float64 is the ground truth both fp32 sides are
compared with.  Stored per parameter (514): [max|g|, ||g||_2] and 128 sampled elements (a fixed stride through the flat
gradient), plus 32768 sampled logits, the input gradient's digest and the BatchNorm running statistics after the step.
Weights / inputs: tools/detweights.py (seed, key names) -- nothing but the recipe and the expected numbers is stored.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.detweights import det_input, fill_module  # noqa: E402
from tools.ref_import import import_reference_lmnet  # noqa: E402

NS = 128


def sample_index(n, ns=NS):
    """The fixed sample of a flat tensor of n elements (shared with tests/test_configs_gpu.py)."""
    if n <= ns:
        return np.arange(n)
    return (np.arange(ns, dtype=np.int64) * (n - 1)) // (ns - 1)


def main():
    size, B, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    torch.set_num_threads(8)
    LM_Net = import_reference_lmnet()
    m = LM_Net(3, 2)
    fill_module(m, seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m = m.double().train()
    key = "f64_%d_b%d" % (size, B)
    x = det_input((B, 3, size, size), key + "/x").double().requires_grad_(True)
    t0 = time.time()
    y = m(x)
    G = det_input(tuple(y.shape), key + "/G").double()
    (y * G).sum().backward()
    print("reference fp64 step: %.1f s" % (time.time() - t0), flush=True)
    yf = y.detach().flatten()
    out = {"logits/stat": np.array([yf.abs().max().item(), yf.norm().item()]),
           "logits/sample": yf[torch.from_numpy(sample_index(yf.numel(), 32768))].numpy(),
           "meta": np.array([size, B, seed], dtype=np.int64)}
    gx = x.grad.detach().flatten()
    out["gx/stat"] = np.array([gx.abs().max().item(), gx.norm().item()])
    out["gx/sample"] = gx[torch.from_numpy(sample_index(gx.numel()))].numpy()
    for k, p in m.named_parameters():
        g = p.grad.detach().flatten()
        out["gstat/" + k] = np.array([g.abs().max().item(), g.norm().item()])
        out["gsamp/" + k] = g[torch.from_numpy(sample_index(g.numel()))].numpy()
    for k, v in m.state_dict().items():
        if "running_" in k:
            out["state/" + k] = v.detach().float().numpy()
    path = os.path.join(ROOT, "tests", "golden", "train_%s.npz" % key)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
