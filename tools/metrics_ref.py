"""Foreground Dice / IoU from the 2x2 confusion matrix (test/bench infrastructure).

Formulas follow the reference's numpy evaluator: Dice = 2TP / (2TP + FP + FN)
(utils/train_eval_utils.py:78-82), IoU = TP / (TP + FP + FN) (the foreground term
of :92-95).  ``pred``/``label`` are integer masks in {0,1}.
"""
import torch


def confusion(pred: torch.Tensor, label: torch.Tensor):
    pred, label = pred.reshape(-1).long(), label.reshape(-1).long()
    tp = int(((pred == 1) & (label == 1)).sum())
    fp = int(((pred == 1) & (label == 0)).sum())
    fn = int(((pred == 0) & (label == 1)).sum())
    tn = int(((pred == 0) & (label == 0)).sum())
    return tp, fp, fn, tn


def dice_iou(pred: torch.Tensor, label: torch.Tensor):
    tp, fp, fn, _ = confusion(pred, label)
    dice = 2.0 * tp / max(2 * tp + fp + fn, 1)
    iou = tp / max(tp + fp + fn, 1)
    return dice, iou
