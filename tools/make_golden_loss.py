"""tests/golden/loss_metrics.npz from the REAL reference classes (build container only):

  * utils/loss.py::DiceLoss (the loss of utils/train_eval_utils.py:141, with nn.CrossEntropyLoss(weight=[1,4],
    label_smoothing) as train.py:157 builds it): loss value and d loss / d logits in float64;
  * utils/train_eval_utils.py::Evaluator (confusion-matrix Dice / IoU / accuracy, :55-118) on argmax(logits).

Third-party imports the two modules make but the code under test never reaches are stubbed (torchvision.ops.focal_loss,
cv2, skimage, sklearn, tqdm -- none is installed here).  Inputs: tools/detweights.py recipes, so only the recipe and
the expected numbers are stored.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.detweights import det_input, disc_labels  # noqa: E402

REF = os.environ.get("LMNET_REFERENCE_ROOT", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules.setdefault(name, m)


def main():
    for n in ("torchvision", "torchvision.ops"):
        _stub(n)
    _stub("torchvision.ops.focal_loss", sigmoid_focal_loss=None)
    _stub("cv2")
    _stub("skimage")
    _stub("skimage.metrics", hausdorff_distance=None)
    _stub("sklearn")
    _stub("sklearn.metrics", accuracy_score=None, precision_score=None, recall_score=None, f1_score=None)
    _stub("tqdm", tqdm=lambda x, **k: x)
    sys.path.insert(0, REF)
    from utils.loss import DiceLoss                      # noqa: E402  (the reference's class)
    from utils.train_eval_utils import Evaluator          # noqa: E402
    out = {}
    for tag, (B, H, W, scale) in {"a": (3, 40, 56, 3.0), "b": (2, 64, 48, 1.0)}.items():
        lg = (det_input((B, 2, H, W), "loss/%s" % tag) * scale).double().requires_grad_(True)
        y = disc_labels(B, H, W, seed=77)
        ce = torch.nn.CrossEntropyLoss(weight=torch.tensor([1.0, 4.0], dtype=torch.float64), label_smoothing=0.001)
        loss = ce(lg, y) + DiceLoss(2)(lg, y.unsqueeze(1).float(), weight=[1.0, 4.0])     # train_eval_utils.py:141
        loss.backward()
        out["%s/meta" % tag] = np.array([B, H, W, scale])
        out["%s/loss" % tag] = np.array([float(loss)])
        out["%s/dlogits" % tag] = lg.grad.float().numpy()
        ev = Evaluator(2)
        pred = lg.detach().argmax(1).numpy()
        gt = y.numpy()
        # Evaluator.add_batch / _generate_matrix (train_eval_utils.py): confusion[label, pred]
        if hasattr(ev, "add_batch"):
            ev.add_batch(gt, pred)
        else:
            raise RuntimeError("reference Evaluator has no add_batch")
        out["%s/confusion" % tag] = np.asarray(ev.confusion_matrix, dtype=np.float64)
        out["%s/dice_fg" % tag] = np.array([float(ev.Dice())])
        out["%s/mean_dice" % tag] = np.array([float(ev.Mean_Dice())])
        out["%s/miou" % tag] = np.array([float(ev.Mean_Intersection_over_Union())])
        out["%s/acc" % tag] = np.array([float(ev.Accuracy())])
    path = os.path.join(ROOT, "tests", "golden", "loss_metrics.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), {k: v.tolist() for k, v in out.items() if v.size <= 4})


if __name__ == "__main__":
    main()
