"""Deterministic-mode race localiser: ONE forward/backward (no optimizer) of the serial schedule vs repeated runs of a multi-stream
configuration; lists the gradient tensors (and the logits) that differ bitwise, per repetition.
    DTYPE=bf16 BATCH=8 python tools/gpu_race_locate.py [reps] [cfg]     cfg: four | eager | branch | wgrad"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net, hip
from tools.detweights import det_input, fill_module
from tests.helpers import no_dropout

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfgname = sys.argv[2] if len(sys.argv) > 2 else "four"
CFG = dict(four={}, eager=dict(lazy_wgrad=False), branch=dict(overlap_wgrad=False), wgrad=dict(branch_overlap=False))[cfgname]
B, S = int(os.environ.get("BATCH", "8")), int(os.environ.get("SIZE", "352"))
x = det_input((B, 3, S, S), "race/x").cuda()
G = det_input((B, 2, S, S), "race/G").cuda()
names = [n for n, _ in LM_Net(3, 2).named_parameters()]


def run(cfg, plans=False):
    m = LM_Net(3, 2)
    fill_module(m, 43)
    no_dropout(m)
    m = m.cuda().train()
    m.deterministic = True
    m.compute_dtype = os.environ.get("DTYPE", "fp32")
    for k, v in cfg.items():
        setattr(m._engine, k, v)
    kept = {}
    if os.environ.get("KEEP") == "1":
        # engine.probe: called in stream order; clone what the report compares, and repeat the attention backward in place (equal
        # inputs, equal kernel -> equal output?) into a scratch buffer (bias gradient into a dummy)
        def probe(tag, mod, t):
            d = kept.setdefault(mod, {})
            if tag == "nat_bwd:na":
                d["do"] = t["do"].clone()
                d["dqkv_now"] = t["dqkv"].clone()
                dq2 = torch.empty_like(t["dqkv"])
                hip.na_bwd(t["qkv"], t["rpb"], t["do"], dq2, torch.zeros_like(t["rpb"]), t["heads"])
                d["dqkv_second"] = dq2
                d["do_after"] = t["do"].clone()
            elif tag == "nat_bwd":
                d.update({k: v.clone() for k, v in t.items()})
        m._engine.probe = probe
    y = m(x)
    (y.float() * G).sum().backward()
    torch.cuda.synchronize()
    keep = {}
    if m._engine.probe is not None:
        for name in ("natt4", "natt3", "natt2", "natt1"):
            keep[name] = kept.get(getattr(m, name), {})
    return y.detach().clone(), [p.grad.detach().clone() for p in m.parameters()], keep


ref = run(dict(branch_overlap=False, overlap_wgrad=False))
for r in range(reps):
    y, g, keep = run(CFG)
    bad = [(names[i], float((u.float() - v.float()).abs().max()), float(u.float().abs().max())) for i, (u, v) in enumerate(zip(ref[1], g)) if not torch.equal(u, v)]
    print("rep %d %s: logits %s; %d of %d gradient tensors differ" % (r, cfgname, "equal" if torch.equal(y, ref[0]) else "DIFFER", len(bad), len(names)), flush=True)
    for n, d, mx in bad[:400]:
        print("      %-50s max|diff| %.3e (max|ref| %.3e)" % (n, d, mx))
    for blk, kk in keep.items():
        if "dqkv_second" in kk and not torch.equal(kk["dqkv_second"], kk["dqkv_now"]):
            d = (kk["dqkv_second"].float() - kk["dqkv_now"].float()).abs()
            idx = torch.nonzero(d > 0)
            Cc = d.shape[-1] // 3
            parts = torch.bincount(idx[:, 3] // Cc, minlength=3).tolist()
            pix = torch.unique(idx[:, :3], dim=0)
            print("        dq/dk/dv elements %s; %d pixels; y range %d..%d x range %d..%d; first pixels %s" % (parts, pix.shape[0], int(pix[:, 1].min()), int(pix[:, 1].max()),
                  int(pix[:, 2].min()), int(pix[:, 2].max()), pix[:12].tolist()))
            print("      SAME-RUN %s: second na_bwd call differs from the first in %d elements (max %.3e); first == serial ref: %s, second == ref: %s"
                  % (blk, int((d > 0).sum()), float(d.max()), torch.equal(kk["dqkv_now"], ref[2][blk]["dqkv_now"]), torch.equal(kk["dqkv_second"], ref[2][blk]["dqkv_now"])))
    if bad:
        for blk, kk in keep.items():
            for k, v in kk.items():
                r0 = ref[2][blk][k]
                if not torch.equal(r0, v):
                    d = (r0.float() - v.float()).abs()
                    nz = torch.nonzero(d.reshape(d.shape[0], -1).amax(1) > 0).flatten().tolist()
                    print("      KEPT %s.%s differs: max %.3e, %d elements, images %s" % (blk, k, float(d.max()), int((d > 0).sum()), nz))
hip.set_deterministic(False)
