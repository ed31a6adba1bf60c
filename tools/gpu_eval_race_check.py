import os, sys
sys.path.insert(0, "/root/repo")
import torch
from lm_net_amd import LM_Net, hip
from tools.detweights import det_input, fill_module
for dtype in ("fp32", "bf16"):
    for (B, S) in ((8, 352), (3, 96)):
        x = det_input((B, 3, S, S), "evr/x").cuda()
        def run(cfg, deploy=False):
            m = LM_Net(3, 2); fill_module(m, 47); m = m.cuda().eval()
            if deploy: m.structural_reparam()
            m.deterministic = True; m.compute_dtype = dtype
            for k, v in cfg.items(): setattr(m._engine, k, v)
            with torch.no_grad():
                y = m(x)
            torch.cuda.synchronize()
            return y.clone()
        for deploy in (False, True):
            ref = run(dict(branch_overlap=False, overlap_wgrad=False), deploy)
            bad = sum(0 if torch.equal(ref, run({}, deploy)) else 1 for _ in range(6))
            print("%s B=%d %d deploy=%s: %d of 6 four-stream eval forwards differ from the serial one" % (dtype, B, S, deploy, bad), flush=True)
hip.set_deterministic(False)
