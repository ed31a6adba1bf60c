"""Attribute the small torch-side launches (fill / copy / elementwise) of one training step to source lines."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
from lm_net_amd import LM_Net  # noqa: E402


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("zero", "fill", "copy", "clone", "zeros", "add", "mul", "cat", "contiguous")):
            fr = [f for f in traceback.extract_stack() if "lm_net_amd" in f.filename or "bench" in f.filename]
            where = "%s:%d" % (os.path.basename(fr[-1].filename), fr[-1].lineno) if fr else "?"
            self.c[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    torch.manual_seed(0)
    m = LM_Net(3, 2).cuda().train()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    for _ in range(2):
        m(x).sum().backward()
    with Counter() as c:
        m(x).sum().backward()
    for (n, w), k in sorted(c.c.items(), key=lambda kv: -kv[1])[:40]:
        print("%4d  %-40s %s" % (k, n, w))


if __name__ == "__main__":
    main()
