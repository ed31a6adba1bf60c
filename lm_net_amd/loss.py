"""The reference's training loss as ONE fused HIP pass per direction.

``utils/train_eval_utils.py:141``::

    loss = criterion(output, labels) + criterion_dice(output, labels.unsqueeze(1).float(), weight=[1.0, 4.0])

with ``criterion = CrossEntropyLoss(weight=[1, 4], label_smoothing=args.smoothing)`` (``train.py:157``) and
``criterion_dice = DiceLoss(num_classes)`` (``utils/loss.py:170-206``).  ``SegLoss`` computes the same scalar from the
``[B, C, H, W]`` logits with two kernels (batch sums, then a one-block finish) and the gradient with one more; the
loss value stays on the device, so nothing synchronises the step.
"""
import torch

from . import hip


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, w_ce, w_dice, label_smoothing, smooth):
        if not logits.is_cuda:
            raise RuntimeError("lm_net_amd.SegLoss: device tensors required (the HIP path has no CPU fallback)")
        logits = logits.contiguous()
        target = target.contiguous()
        Cn = logits.shape[1]
        sums = torch.empty(3 + 3 * Cn, device=logits.device)
        coef = torch.empty(3 + 2 * Cn, device=logits.device)
        loss = torch.empty(1, device=logits.device)
        hip.segloss_fwd(logits, target, w_ce, w_dice, label_smoothing, smooth, sums, coef, loss)
        ctx.save_for_backward(logits, target, w_ce, coef)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        logits, target, w_ce, coef = ctx.saved_tensors
        d = torch.empty_like(logits)
        hip.segloss_bwd(logits, target, w_ce, coef, g.reshape(1).contiguous().float(), d)
        return d, None, None, None, None, None


class SegLoss(torch.nn.Module):
    """``CrossEntropyLoss(weight=ce_weight, label_smoothing) + DiceLoss(n_classes)(…, weight=dice_weight)``."""

    def __init__(self, ce_weight=(1.0, 4.0), dice_weight=(1.0, 4.0), label_smoothing=0.0, smooth=1e-5):
        super().__init__()
        self.register_buffer("ce_weight", torch.tensor(ce_weight, dtype=torch.float32))
        self.register_buffer("dice_weight", torch.tensor(dice_weight, dtype=torch.float32))
        self.label_smoothing, self.smooth = float(label_smoothing), float(smooth)

    def forward(self, logits, target):
        if logits.shape[1] != self.ce_weight.numel():
            raise ValueError("SegLoss: %d classes in the logits, %d weights" % (logits.shape[1], self.ce_weight.numel()))
        if target.dim() == logits.dim():          # the reference passes labels.unsqueeze(1) to the Dice term
            target = target[:, 0]
        return _SegLossFn.apply(logits, target.long(), self.ce_weight, self.dice_weight, self.label_smoothing, self.smooth)
