"""Measured distances of the bf16 mixed-precision path to the reference's fp32 goldens (tolerance calibration)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from helpers import load_golden, no_dropout, rel_err, is_pre_bn_bias
from tools.detweights import det_input, disc_labels, fill_module
from tools.metrics_ref import dice_iou
from lm_net_amd import LM_Net

def l2(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

g = load_golden("default_64x96.npz")
m = LM_Net(3, 2); fill_module(m, 0); no_dropout(m); m = m.cuda()
for mode in ("fp32", "bf16-mma", "bf16"):
    m.compute_dtype = mode
    m.eval()
    x = det_input((2, 3, 64, 96), "d64/x").cuda()
    with torch.no_grad():
        y = m(x)
    print(mode, "eval logits max-rel %.2e l2 %.2e" % (rel_err(y, g["logits"]), l2(y, g["logits"])))
    m.train()
    for p in m.parameters(): p.grad = None
    xg = x.clone().requires_grad_(True)
    yt = m(xg)
    print(mode, "train logits max-rel %.2e l2 %.2e" % (rel_err(yt, g["train_logits"]), l2(yt, g["train_logits"])))
    (yt * det_input(tuple(yt.shape), "d64/G").cuda()).sum().backward()
    print(mode, "dx max-rel %.2e l2 %.2e" % (rel_err(xg.grad, g["train_grad_input"]), l2(xg.grad, g["train_grad_input"])))
    rows = []
    for k, p in m.named_parameters():
        if is_pre_bn_bias(k) or "grad/" + k not in g: continue
        rows.append((l2(p.grad, g["grad/" + k]), rel_err(p.grad, g["grad/" + k]), k))
    rows.sort(reverse=True)
    print(mode, "worst grads (l2, maxrel):", [(round(a, 4), round(b, 4), k) for a, b, k in rows[:6]], "median l2 %.2e" % rows[len(rows)//2][0])
    rs = [rel_err(v, g["state/" + k]) for k, v in m.state_dict().items() if "running_" in k]
    print(mode, "running stats worst %.2e" % max(rs))
    for bn in [mm for mm in m.modules() if isinstance(mm, torch.nn.BatchNorm2d)]:
        pass
    fill_module(m, 0); m = m.cuda()
g3 = load_golden("default_352.npz")
for mode in ("fp32", "bf16-mma", "bf16"):
    m.compute_dtype = mode
    m.eval()
    x3 = det_input((1, 3, 352, 352), "d352/x").cuda()
    with torch.no_grad(): y3 = m(x3)
    pred = y3.argmax(1).cpu()
    d, i = dice_iou(pred, disc_labels(1, 352, 352))
    print(mode, "352 logits max-rel %.2e l2 %.2e dice %.6f ref %.6f iou %.6f ref %.6f pred_sum %d ref %d" % (
        rel_err(y3, g3["logits"]), l2(y3, g3["logits"]), d, float(g3["dice"][0]), i, float(g3["iou"][0]), int(pred.sum()), int(g3["pred_sum"][0])))
