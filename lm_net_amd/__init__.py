"""lm_net_amd -- MI355X-native (gfx950) implementation of the LM-Net forward/backward hot path.

    from lm_net_amd import LM_Net          # drop-in for ``from core.LM_Net import LM_Net``
"""
from .LM_Net import LM_Net  # noqa: F401

__all__ = ["LM_Net"]
