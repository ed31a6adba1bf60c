"""Whole-model parity on the GPU box: lm_net_amd.LM_Net (HIP) vs the CPU oracle, stage by stage.

    python tools/gpu_model_check.py [tiny|default] [H W B]
Prints per-stage forward errors (eval + train mode), and per-parameter gradient errors."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lm_net_amd import LM_Net  # noqa: E402
from oracle.lmnet_ref import LM_Net as Oracle  # noqa: E402
from tools.detweights import det_input, fill_module  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if not torch.isfinite(a).all():
        return float("inf")
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def nodrop(m):
    for mm in m.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
        if hasattr(mm, "p") and isinstance(getattr(mm, "p"), float):
            mm.p = 0.0


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    H, W, B = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 48, 2)
    f64 = len(sys.argv) > 5 and sys.argv[5] == "f64"      # oracle in float64: separates GPU error from fp32-CPU noise
    filters = [12] * 5 if cfg == "tiny" else [12, 24, 48, 96, 192]
    ora = Oracle(3, 2, filters=filters)
    fill_module(ora, seed=1)
    nodrop(ora)
    net = LM_Net(3, 2, filters=filters)
    net.load_state_dict(ora.state_dict())
    net.cuda()
    nodrop(net)
    net._keep_taps = True
    x = det_input((B, 3, H, W), "mc/x")
    if f64:
        ora = ora.double()
    worst = 0.0
    for mode in ("eval", "train"):
        ora.train(mode == "train"); net.train(mode == "train")
        taps = {}
        xo = (x.double() if f64 else x.clone()).requires_grad_(True)
        yo = ora(xo, taps)
        xg = x.cuda().requires_grad_(True)
        t0 = time.time()
        yg = net(xg)
        torch.cuda.synchronize()
        print("[%s] forward %.3fs logits rel err %.3e" % (mode, time.time() - t0, rel(yg, yo)))
        for k, v in net._taps.items():
            e = rel(v.permute(0, 3, 1, 2), taps[k])
            worst = max(worst, e)
            print("   stage %-4s %.3e %s" % (k, e, "" if e < 1e-4 else "<<<<"))
        Gm = det_input(tuple(yo.shape), "mc/G")
        (yo * (Gm.double() if f64 else Gm)).sum().backward()
        t0 = time.time()
        (yg * Gm.cuda()).sum().backward()
        torch.cuda.synchronize()
        print("[%s] backward %.3fs dx rel err %.3e" % (mode, time.time() - t0, rel(xg.grad, xo.grad)))
        gmax = max(float(p.grad.abs().max()) for p in ora.parameters())
        bad = 0
        for (k, po), (k2, pg) in zip(ora.named_parameters(), net.named_parameters()):
            assert k == k2
            e_abs = float((pg.grad.detach().cpu().double() - po.grad.double()).abs().max())
            e_rel = e_abs / (float(po.grad.abs().max()) + 1e-30)
            ok = e_rel < 5e-4 or e_abs < 1e-5 * gmax
            if not ok:
                bad += 1
                print("   grad %-55s rel %.3e abs %.3e (|g|max %.3e) <<<<" % (k, e_rel, e_abs, float(po.grad.abs().max())))
        print("[%s] %d / %d parameter gradients out of tolerance" % (mode, bad, len(list(ora.parameters()))))
        if mode == "train":
            nb = 0
            for (k, bo), (k2, bg) in zip(ora.named_buffers(), net.named_buffers()):
                if rel(bg.float(), bo.float()) > 1e-5:
                    nb += 1
                    print("   buffer %-50s %.3e <<<<" % (k, rel(bg.float(), bo.float())))
            print("[train] %d buffers (BN running stats) differ" % nb)
        ora.zero_grad(); net.zero_grad()
    print("WORST STAGE ERR %.3e" % worst)


if __name__ == "__main__":
    main()
