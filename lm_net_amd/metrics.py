"""On-device segmentation metrics.

The reference moves every batch to the CPU and updates a torchmetrics collection (``utils/train_eval_utils.py:150-156``,
``train.py:165-174``).  ``ConfusionMeter`` keeps the confusion matrix of ``argmax(logits, 1)`` against the labels on the
GPU (one kernel per batch, no sync) and derives Dice ``2TP/(2TP+FP+FN)`` and IoU ``TP/(TP+FP+FN)`` per class
(``train_eval_utils.py:78-95``) when ``compute()`` is called.
"""
import torch

from . import hip


class ConfusionMeter:
    def __init__(self, n_classes=2, device="cuda"):
        self.n = n_classes
        self.total = torch.zeros(n_classes, n_classes, device=device, dtype=torch.float64)

    def reset(self):
        self.total.zero_()

    @torch.no_grad()
    def update(self, logits, target):
        if not logits.is_cuda:
            raise RuntimeError("lm_net_amd.ConfusionMeter: device tensors required (the HIP path has no CPU fallback)")
        counts = torch.zeros(self.n, self.n, device=logits.device)      # exact: < 2^24 per cell and launch
        hip.confusion(logits.contiguous().float(), target.contiguous().long(), counts)
        self.total += counts.double()

    def compute(self):
        """{'dice': [per class], 'iou': [per class], 'accuracy': float, 'confusion': [[...]]} -- rows = label."""
        m = self.total.cpu()
        tp = m.diag()
        fp, fn = m.sum(0) - tp, m.sum(1) - tp
        dice = (2 * tp / (2 * tp + fp + fn).clamp_min(1)).tolist()
        iou = (tp / (tp + fp + fn).clamp_min(1)).tolist()
        return dict(dice=dice, iou=iou, accuracy=float(tp.sum() / m.sum().clamp_min(1)), confusion=m.long().tolist())
