"""CPU: the oracle (oracle/lmnet_ref.py) against golden vectors produced by the REAL reference
(tools/make_golden.py imports /root/reference).  Tolerance: 1e-5 rel on CPU-vs-CPU fp32."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.lmnet_ref import LM_Net
from tools.detweights import det_input, disc_labels, fill_module
from tools.metrics_ref import dice_iou
from helpers import GOLDEN, digest, is_pre_bn_bias, load_golden, no_dropout, rel_err

TOL = 1e-5


def test_state_dict_keys_match_reference():
    keys = json.load(open(os.path.join(GOLDEN, "keys.json")))
    m = LM_Net(3, 2)
    sd = m.state_dict()
    assert len(sd) == 766 and list(sd.keys()) == list(keys["train"].keys())
    assert all(list(v.shape) == keys["train"][k] for k, v in sd.items())
    assert sum(p.numel() for p in m.parameters()) == keys["num_parameters"] == 3966566
    m.structural_reparam()
    sd = m.state_dict()
    assert len(sd) == 510 and list(sd.keys()) == list(keys["deploy"].keys())
    assert all(list(v.shape) == keys["deploy"][k] for k, v in sd.items())


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_tiny_config_logits_and_grads(mode):
    g = load_golden("tiny_%s.npz" % mode)
    m = LM_Net(3, 2, filters=[12] * 5)
    fill_module(m)
    no_dropout(m)
    m.train(mode == "train")
    x = det_input((1, 3, 32, 48), "tiny/x").requires_grad_(True)
    y = m(x)
    assert rel_err(y, g["logits"]) < TOL
    (y * det_input(tuple(y.shape), "tiny/G")).sum().backward()
    assert rel_err(x.grad, g["grad_input"]) < 1e-4
    gmax = max(float(np.abs(g["grad/" + k]).max()) for k, _ in m.named_parameters())
    for k, p in m.named_parameters():
        ref = g["grad/" + k]
        if mode == "train" and is_pre_bn_bias(k):
            assert float(np.abs(p.grad.numpy() - ref).max()) < 1e-4 * gmax, k
        else:
            assert rel_err(p.grad, ref) < 2e-4, k
    if mode == "train":
        for k, v in m.state_dict().items():
            if "running_" in k:
                assert rel_err(v, g["state/" + k]) < TOL, k
            if "num_batches" in k:
                assert int(v) == int(g["state/" + k]) == 1


def test_default_config_64x96_eval_stages_deploy_train():
    g = load_golden("default_64x96.npz")
    m = LM_Net(3, 2)
    fill_module(m)
    no_dropout(m)
    m.eval()
    x = det_input((2, 3, 64, 96), "d64/x")
    taps = {}
    with torch.no_grad():
        y = m(x, taps)
    assert rel_err(y, g["logits"]) < TOL
    for k in [k for k in g if k.startswith("stage/")]:
        assert rel_err(taps[k[6:]], g[k]) < TOL, k
    m.train()
    xg = x.clone().requires_grad_(True)
    yt = m(xg)
    assert rel_err(yt, g["train_logits"]) < TOL
    (yt * det_input(tuple(yt.shape), "d64/G")).sum().backward()
    assert rel_err(xg.grad, g["train_grad_input"]) < 1e-4
    for k, p in m.named_parameters():
        if is_pre_bn_bias(k):
            continue
        d, r = digest(p.grad), g["gdig/" + k]
        assert abs(d[2] - r[2]) <= 2e-4 * r[2] + 1e-9, k          # L2 norm of the gradient
        if "grad/" + k in g:
            assert rel_err(p.grad, g["grad/" + k]) < 5e-4, k
    m2 = LM_Net(3, 2)
    fill_module(m2)
    m2.eval()
    m2.structural_reparam()
    with torch.no_grad():
        yd = m2(x)
    assert rel_err(yd, g["deploy_logits"]) < TOL
    assert rel_err(yd, g["logits"]) < 1e-5                            # the reparam invariant


def test_default_config_352_logits_dice():
    """BASELINE.json configs[0]: LM-Net forward, 1x3x352x352 on PyTorch CPU."""
    g = load_golden("default_352.npz")
    m = LM_Net(3, 2)
    fill_module(m)
    m.eval()
    x = det_input((1, 3, 352, 352), "d352/x")
    taps = {}
    with torch.no_grad():
        y = m(x, taps)
    assert rel_err(y, g["logits"]) < TOL
    for k in [k for k in g if k.startswith("sdig/")]:
        d, r = digest(taps[k[5:]]), g[k]
        assert abs(d[2] - r[2]) <= 1e-5 * r[2], k
        assert rel_err(taps[k[5:]][:, :4, :8, :8], g["scrop/" + k[5:]]) < 1e-4, k
    pred = y.argmax(1)
    dice, iou = dice_iou(pred, disc_labels(1, 352, 352))
    assert int(pred.sum()) == int(g["pred_sum"][0])
    assert round(dice, 4) == round(float(g["dice"][0]), 4) and round(iou, 4) == round(float(g["iou"][0]), 4)
