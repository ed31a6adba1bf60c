"""Import the real reference ``core.LM_Net`` from /root/reference in THIS container.

Used only by ``tools/make_golden.py`` (fixture generation) and by
``tests/test_oracle_vs_reference.py`` (skipped when /root/reference is absent,
i.e. on the GPU box).  Three third-party modules the reference imports are not
installable here (no network): ``timm``, ``torchvision.ops.*`` and ``natten``.
They are injected as stubs:

  * timm.models.layers      -> to_2tuple, trunc_normal_, DropPath
  * torchvision.ops.deform_conv / ps_roi_pool -> only reached by dead classes;
    the star-import is also where the reference gets ``math`` from
  * natten.NeighborhoodAttention2D -> oracle.natten_ref (our restatement of the
    published semantics; the reference has no source for it)
"""
import math
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("LMNET_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "core", "LM_Net.py"))


def _install_stubs():
    import torch.nn as nn
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle.natten_ref import NeighborhoodAttention2D

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    if "timm" not in sys.modules:
        mod("timm")
        mod("timm.models")
        mod("timm.models.layers", to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
            trunc_normal_=nn.init.trunc_normal_, DropPath=DropPath)
    try:
        import torchvision  # noqa: F401
    except Exception:
        class DeformConv2d(nn.Module):
            pass
        mod("torchvision")
        mod("torchvision.ops")
        dc = mod("torchvision.ops.deform_conv", math=math, DeformConv2d=DeformConv2d)
        dc.__all__ = ["math", "DeformConv2d"]
        pr = mod("torchvision.ops.ps_roi_pool")
        pr.__all__ = []
    mod("natten", NeighborhoodAttention2D=NeighborhoodAttention2D)


def import_reference_lmnet():
    """Returns the reference's ``LM_Net`` class (real reference code)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the reference package is called 'core'; make sure we do not pick up
    # anything else of that name
    for name in [n for n in sys.modules if n == "core" or n.startswith("core.")]:
        f = getattr(sys.modules[name], "__file__", "") or ""
        if not f.startswith(REFERENCE_ROOT):
            del sys.modules[name]
    from core.LM_Net import LM_Net  # type: ignore
    return LM_Net
