"""lm_net_amd -- MI355X-native (gfx950) implementation of the LM-Net forward/backward hot path.

    from lm_net_amd import LM_Net          # drop-in for ``from core.LM_Net import LM_Net``

Around the path (SURVEY.md section 8f "next" rows): ``lm_net_amd.loss.SegLoss`` (fused CE + Dice),
``lm_net_amd.optim.FusedAdamW`` (one-launch AdamW), ``lm_net_amd.metrics.ConfusionMeter`` (on-device Dice / IoU),
``lm_net_amd.ddp.DistributedLMNet`` (bucketed RCCL gradient all-reduce).
"""
from .LM_Net import LM_Net  # noqa: F401

__all__ = ["LM_Net"]
