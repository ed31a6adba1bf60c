"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
            m.p = 0.0


def rel_err(a, b):
    """max|a-b| / max|b| (the tolerance north_star states is <=1e-4 rel fp32)."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def digest(t):
    t = torch.as_tensor(t).detach().double().flatten().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])


# parameters whose exact gradient is 0 in train mode (a bias feeding straight into a batch-stat
# BatchNorm): compare those on an absolute scale instead of relative to their own magnitude.
def is_pre_bn_bias(key):
    return key.endswith("expand_conv.0.bias") or key.endswith("fuse_conv.0.bias")
