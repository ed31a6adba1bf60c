"""Deterministic, name-keyed weights and inputs (test/bench infrastructure).

Every tensor of a ``state_dict`` is filled from a counter-based integer hash of
its KEY NAME and element index (FNV-1a of the name -> splitmix64 stream), so any
process -- the reference import in the build container, the oracle, the HIP
product on the GPU box -- regenerates bit-identical weights from the key list
alone.  Nothing depends on torch RNG state, init order or library version.
"""
import numpy as np
import torch

GAIN = 0.9          # conv/linear weight gain; keeps stage activations O(1) through the net
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [0,1) determined by (key, seed) only."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        bits = _splitmix64(base + idx * np.uint64(0xD1342543DE82EF95))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def fill_tensor(key: str, t: torch.Tensor, seed: int = 0) -> torch.Tensor:
    """Value law by key suffix; fan-in scaled weights keep activations O(1)."""
    n = t.numel()
    if key.endswith("num_batches_tracked"):
        return torch.zeros_like(t)
    u = uniform(key, n, seed)
    if key.endswith("running_var"):
        v = 0.5 + 1.5 * u
    elif key.endswith("running_mean"):
        v = 0.4 * u - 0.2
    elif t.dim() == 1 and key.endswith(".weight"):            # BN / LN gamma
        v = 0.5 + u
    elif key.endswith(".bias"):
        v = 0.2 * u - 0.1
    elif key.endswith("rpb"):
        v = 1.0 * u - 0.5
    else:                                                       # conv / linear weight
        fan_in = max(1, int(np.prod(t.shape[1:])))
        v = (2.0 * u - 1.0) * (3.0 / fan_in) ** 0.5 * GAIN
    return torch.from_numpy(v.astype(np.float32)).reshape(t.shape).to(t.dtype)


@torch.no_grad()
def fill_module(module: torch.nn.Module, seed: int = 0) -> None:
    sd = module.state_dict()
    module.load_state_dict({k: fill_tensor(k, v, seed) for k, v in sd.items()})


def det_input(shape, key: str = "input", seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    """Approximately N(0,1) input (sum of 4 uniforms, centred/scaled)."""
    n = int(np.prod(shape))
    u = sum(uniform("%s/%d" % (key, i), n, seed) for i in range(4))
    v = (u - 2.0) * (3.0 ** 0.5) * scale
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def disc_labels(batch: int, h: int, w: int, seed: int = 1234) -> torch.Tensor:
    """[B,H,W] int64 in {0,1}: one filled disc per image, ~25 % foreground."""
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((batch, h, w), dtype=np.int64)
    for i in range(batch):
        u = uniform("label/%d" % i, 3, seed)
        r = (0.22 + 0.1 * u[0]) * min(h, w)
        cy = r + (h - 2 * r) * u[1]
        cx = r + (w - 2 * r) * u[2]
        out[i] = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
    return torch.from_numpy(out)
