"""Generate tests/golden/* by running the REAL reference (/root/reference) on CPU.

Runs only in the build container (the reference tree cannot travel).  The
reference's ``core.LM_Net`` is imported with the three stubbed third-party
modules of ``tools/ref_import.py`` (natten -> oracle/natten_ref.py, our
restatement of the published NA semantics).  Weights/inputs come from the
name-keyed generator in ``tools/detweights.py`` so every consumer regenerates
them from key names; fixtures hold only inputs' recipe + expected outputs.

    python tools/make_golden.py            # rewrites tests/golden/*.npz, keys.json

Fixtures (all float32, compressed):
  keys.json                 state_dict key -> shape, train form (766) and deploy form (510)
  tiny_eval.npz / tiny_train.npz
                            LM_Net(3,2,[12]*5) @ 1x3x32x48: logits, input grad, every param grad of
                            L = sum(logits * G); train variant uses batch-stat BN (dropout p=0) and
                            also stores the updated BN running stats
  default_64x96.npz         LM_Net(3,2) @ 2x3x64x96 eval: logits + all stage activations; deploy logits;
                            train-mode (batch-stat BN, p=0) logits + per-parameter grad digests
  default_352.npz           LM_Net(3,2) @ 1x3x352x352 eval: logits (full), stage digests, argmax mask
                            checksum, Dice / IoU against the synthetic disc label
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.detweights import det_input, disc_labels, fill_module  # noqa: E402
from tools.ref_import import import_reference_lmnet  # noqa: E402
from tools.metrics_ref import dice_iou  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
STAGES = ["x1", "x2", "x3", "x4", "xd4", "x5", "xs1", "xs2", "xs3", "xs4",
          "x46", "x37", "x28", "x19", "x6", "x7", "x8", "x9"]


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def stage_hooks(model, store):
    """Capture the named activations of core/LM_Net.py:96-120 from the reference via forward hooks."""
    names = dict(conv1="x1", conv2="x2", conv3="x3", conv4="x4", down4="xd4", gft="x5",
                 skip1="xs1", skip2="xs2", skip3="xs3", skip4="xs4", natt1="x46", natt2="x37",
                 natt3="x28", natt4="x19", dconv1="x6", dconv2="x7", dconv3="x8", dconv4="x9")
    hs = []
    for mod, tag in names.items():
        hs.append(getattr(model, mod).register_forward_hook(
            lambda m, i, o, tag=tag: store.__setitem__(tag, o.detach().clone())))
    return hs


def digest(t):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()], dtype=np.float64)


def grads_fixture(model, x, gkey):
    model.zero_grad(set_to_none=True)
    x = x.clone().requires_grad_(True)
    y = model(x)
    G = det_input(tuple(y.shape), gkey)
    (y * G).sum().backward()
    return y.detach(), x.grad.detach(), {k: p.grad.detach() for k, p in model.named_parameters()}


def main():
    os.makedirs(GOLD, exist_ok=True)
    Ref = import_reference_lmnet()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- keys
    m = Ref(3, 2)
    keys = {"train": {k: list(v.shape) for k, v in m.state_dict().items()}}
    m.structural_reparam()
    keys["deploy"] = {k: list(v.shape) for k, v in m.state_dict().items()}
    keys["num_parameters"] = 3966566
    with open(os.path.join(GOLD, "keys.json"), "w") as f:
        json.dump(keys, f, indent=0)

    # ---------------------------------------------------------------- tiny config, full grads
    for mode in ("eval", "train"):
        m = Ref(3, 2, filters=[12, 12, 12, 12, 12])
        fill_module(m)
        no_dropout(m)
        m.train(mode == "train")
        x = det_input((1, 3, 32, 48), "tiny/x")
        y, gx, gp = grads_fixture(m, x, "tiny/G")
        out = {"logits": y.numpy(), "grad_input": gx.numpy()}
        out.update({"grad/" + k: v.numpy() for k, v in gp.items()})
        if mode == "train":
            out.update({"state/" + k: v.numpy() for k, v in m.state_dict().items()
                        if "running_" in k or "num_batches" in k})
        np.savez_compressed(os.path.join(GOLD, "tiny_%s.npz" % mode), **out)
        print("tiny", mode, float(y.std()), len(gp))

    # ---------------------------------------------------------------- default config @ 2x3x64x96
    m = Ref(3, 2)
    fill_module(m)
    no_dropout(m)
    m.eval()
    x = det_input((2, 3, 64, 96), "d64/x")
    st = {}
    hs = stage_hooks(m, st)
    with torch.no_grad():
        y = m(x)
    for h in hs:
        h.remove()
    out = {"logits": y.numpy()}
    out.update({"stage/" + k: st[k].numpy() for k in STAGES})
    # train-mode forward/backward (batch-stat BN, dropout off)
    m.train()
    yt, gx, gp = grads_fixture(m, x, "d64/G")
    out["train_logits"] = yt.numpy()
    out["train_grad_input"] = gx.numpy()
    for k, v in gp.items():
        out["gdig/" + k] = digest(v)
        if v.numel() <= 4096:
            out["grad/" + k] = v.numpy()
    for k, v in m.state_dict().items():
        if "running_" in k:
            out["state/" + k] = v.numpy()
    # deploy form
    m2 = Ref(3, 2)
    fill_module(m2)
    m2.eval()
    m2.structural_reparam()
    with torch.no_grad():
        out["deploy_logits"] = m2(x).numpy()
    np.savez_compressed(os.path.join(GOLD, "default_64x96.npz"), **out)
    print("d64 eval/deploy maxdiff", float(np.abs(out["deploy_logits"] - out["logits"]).max()))

    # ---------------------------------------------------------------- default config @ 1x3x352x352 (BASELINE config 1)
    m = Ref(3, 2)
    fill_module(m)
    m.eval()
    x = det_input((1, 3, 352, 352), "d352/x")
    st = {}
    hs = stage_hooks(m, st)
    with torch.no_grad():
        y = m(x)
    for h in hs:
        h.remove()
    lab = disc_labels(1, 352, 352)
    pred = y.argmax(1)
    dice, iou = dice_iou(pred, lab)
    out = {"logits": y.numpy(), "pred_sum": np.array([int(pred.sum())]),
           "dice": np.array([dice]), "iou": np.array([iou])}
    out.update({"sdig/" + k: digest(st[k]) for k in STAGES})
    out.update({"scrop/" + k: st[k][:, :4, :8, :8].numpy() for k in STAGES})
    np.savez_compressed(os.path.join(GOLD, "default_352.npz"), **out)
    print("d352 logits std", float(y.std()), "pred fg", int(pred.sum()), "dice", dice, "iou", iou)
    for f in sorted(os.listdir(GOLD)):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
