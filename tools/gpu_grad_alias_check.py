import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net
m = LM_Net(3, 2, filters=[12] * 5).cuda().train()
x = torch.randn(2, 3, 32, 48, device="cuda")
m(x).square().mean().backward()
flat = m._grad_flat
lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
n_alias = sum(1 for p in m.parameters() if lo <= p.grad.data_ptr() < hi)
print("params", len(list(m.parameters())), "grads aliasing the flat buffer:", n_alias)
