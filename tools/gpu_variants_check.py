"""Crash / sanity check of less common configurations: class counts, input channels, batch 1, non-square sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lm_net_amd import LM_Net
from lm_net_amd.loss import SegLoss
from lm_net_amd.metrics import ConfusionMeter
from lm_net_amd.optim import FusedAdamW

for (cin, ncls, B, H, W) in [(3, 3, 2, 64, 96), (1, 2, 1, 32, 32), (4, 4, 3, 48, 80), (3, 2, 1, 352, 352)]:
    torch.manual_seed(0)
    m = LM_Net(cin, ncls).cuda().train()
    opt = FusedAdamW(m, lr=1e-3)
    crit = SegLoss(ce_weight=[1.0] * ncls, dice_weight=[1.0] * ncls, label_smoothing=0.01).cuda()
    x = torch.randn(B, cin, H, W, device="cuda")
    y = torch.randint(0, ncls, (B, H, W), device="cuda")
    for _ in range(3):
        out = m(x)
        loss = crit(out, y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    meter = ConfusionMeter(ncls)
    m.eval()
    with torch.no_grad():
        meter.update(m(x), y)
    r = meter.compute()
    assert torch.isfinite(loss) and all(torch.isfinite(p).all() for p in m.parameters())
    print("cin=%d classes=%d B=%d %dx%d: loss %.4f acc %.3f ok" % (cin, ncls, B, H, W, float(loss), r["accuracy"]))
