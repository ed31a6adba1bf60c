"""Parameter containers mirroring the reference's module tree (same attribute names, parameter shapes
and default initialisation => identical ``state_dict`` keys, so reference checkpoints load).

These classes hold parameters/buffers ONLY.  They carry no PyTorch compute path: the arithmetic of
every block lives in ``engine.py`` and runs as hand-written HIP kernels.  Calling ``forward`` on a
container directly raises.  Reference: core/modules.py (line ranges per class) and core/LM_Net.py:6-87.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

NUM_HEADS = 12  # core/LM_Net.py:56,81-84


class _Container(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("%s is a parameter container of the HIP path; call LM_Net.forward instead" % type(self).__name__)


def _conv(cin, cout, k, stride=1, groups=1, bias=True):
    pad = (k // 2) if isinstance(k, int) else (k[0] // 2, k[1] // 2)
    return nn.Conv2d(cin, cout, k, stride, pad, groups=groups, bias=bias)


class SE(_Container):  # core/modules.py:1020-1044
    def __init__(self, ch, reduction=4):
        super().__init__()
        self.fc1 = _conv(ch, ch // reduction, 1)
        self.fc2 = _conv(ch // reduction, ch, 1)
        nn.init.kaiming_normal_(self.fc1.weight)
        nn.init.kaiming_normal_(self.fc2.weight)


def _conv_bn(conv, ch):
    return nn.Sequential(OrderedDict([("conv", conv), ("bn", nn.BatchNorm2d(ch))]))


class ReparamConv(_Container):  # core/modules.py:525-657
    def __init__(self, cin, cexp, cout, large_k=5, small_k=3):
        super().__init__()
        assert large_k == 5 and small_k == 3, "the HIP stencil is built for the 5/3 kernels LM_Net uses"
        self.cin, self.cexp, self.cout = cin, cexp, cout
        self.deploy = False
        self.se = SE(cexp)
        self.expand_conv = nn.Sequential(_conv(cin, cexp, 1), nn.BatchNorm2d(cexp), nn.Hardswish())
        self.large_conv = _conv_bn(_conv(cexp, cexp, 5, groups=cexp, bias=False), cexp)
        self.square_conv = _conv_bn(_conv(cexp, cexp, 3, groups=cexp, bias=False), cexp)
        self.ver_conv = _conv_bn(_conv(cexp, cexp, (3, 1), groups=cexp, bias=False), cexp)
        self.hor_conv = _conv_bn(_conv(cexp, cexp, (1, 3), groups=cexp, bias=False), cexp)
        self.pointwise_conv = nn.Sequential(_conv(cexp, cout, 1))
        self.shortcut = nn.Sequential(_conv(cin, cout, 1))

    def branches(self):
        return (self.large_conv, self.square_conv, self.ver_conv, self.hor_conv)

    @torch.no_grad()
    def switch_to_deploy(self):
        """Fold the four BN'd branches into one 5x5 depthwise conv with bias (modules.py:622-657).
        Weight-only host-side folding (row A12); the stencil itself still runs in HIP."""
        if self.deploy:
            return
        w = torch.zeros_like(self.large_conv.conv.weight)
        b = torch.zeros_like(self.large_conv.bn.bias)
        for br in self.branches():
            s = br.bn.weight / torch.sqrt(br.bn.running_var + br.bn.eps)
            wb = br.conv.weight * s.view(-1, 1, 1, 1)
            kh, kw = wb.shape[2:]
            w[:, :, 2 - kh // 2:2 - kh // 2 + kh, 2 - kw // 2:2 - kw // 2 + kw] += wb
            b += br.bn.bias - br.bn.running_mean * s
        self.fuse_conv = _conv(self.cexp, self.cexp, 5, groups=self.cexp, bias=True).to(w.device)
        self.fuse_conv.weight.data = w
        self.fuse_conv.bias.data = b
        self.deploy = True
        del self.square_conv, self.hor_conv, self.ver_conv   # the reference keeps large_conv (modules.py:654-657)


class M3Skip(_Container):  # core/modules.py:83-107
    def __init__(self, ch):
        super().__init__()
        cl, cm, cs = ch
        self.convl = nn.Sequential(_conv(cl, cm, 3, 2))
        self.convm = nn.Sequential(_conv(cm, cm, 3))
        self.convs = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True), _conv(cs, cm, 3))
        self.fuse_conv = nn.Sequential(_conv(3 * cm, cm, 3), nn.BatchNorm2d(cm), nn.GELU())


class M2Skip(_Container):  # core/modules.py:109-143
    def __init__(self, ch, model_type="bottom"):
        super().__init__()
        c0, c1 = ch
        self.model_type = model_type
        if model_type == "bottom":
            self.convl = nn.Sequential(_conv(c0, c1, 3, 2))
            self.convs = nn.Sequential(_conv(c1, c1, 3))
            cf = c1
        else:
            self.convl = nn.Sequential(_conv(c0, c0, 3))
            self.convs = nn.Sequential(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True), _conv(c1, c0, 3))
            cf = c0
        self.fuse_conv = nn.Sequential(_conv(2 * cf, cf, 3), nn.BatchNorm2d(cf), nn.GELU())


class OverlapPatchEmbed(_Container):  # core/modules.py:22-40
    def __init__(self, cin, cout):
        super().__init__()
        self.patch_embeddings = _conv(cin, cout, 3)


class Mlp(_Container):  # core/modules.py:42-56
    def __init__(self, cin, chid, cout):
        super().__init__()
        self.fc1 = nn.Linear(cin, chid)
        self.fc2 = nn.Linear(chid, cout)
        self.act_fn = nn.GELU()
        self.dropout = nn.Dropout(0.1)


class GlobalAttention(_Container):  # core/modules.py:235-279
    def __init__(self, dim, num_heads):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        for m in (self.qkv, self.proj):
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.zeros_(m.bias)


class GFT(_Container):  # core/modules.py:329-347
    def __init__(self, cin, expand, cout, num_heads):
        super().__init__()
        self.patchembedding = OverlapPatchEmbed(cin, cin)
        self.norm1 = nn.LayerNorm(cin)
        self.attention = GlobalAttention(cin, num_heads)
        self.norm2 = nn.LayerNorm(cin)
        self.mlp = Mlp(cin, expand * cin, cin)
        self.conv = nn.Sequential(_conv(cin, cout, 1))


class PyramidPool(_Container):  # core/modules.py:454-498 (no parameters)
    pass


class NeighborhoodAttention2D(_Container):
    """natten.NeighborhoodAttention2D (external; constructed at core/modules.py:509 with kernel_size=3)."""

    def __init__(self, dim, num_heads, kernel_size=3):
        super().__init__()
        assert kernel_size % 2 == 1 and 3 <= kernel_size <= 13, kernel_size   # (the reference constructs 3: the LDS-tiled kernels)
        assert dim % num_heads == 0
        self.num_heads, self.head_dim, self.kernel_size = num_heads, dim // num_heads, kernel_size
        self.qkv = nn.Linear(dim, 3 * dim)
        self.rpb = nn.Parameter(torch.zeros(num_heads, 2 * kernel_size - 1, 2 * kernel_size - 1))
        nn.init.trunc_normal_(self.rpb, std=0.02, mean=0.0, a=-2.0, b=2.0)
        self.proj = nn.Linear(dim, dim)


class NeighborhoodTransformer(_Container):  # core/modules.py:504-521
    def __init__(self, ch, num_heads, kernel_size=3):
        super().__init__()
        self.patchembedding = OverlapPatchEmbed(ch, ch)
        self.norm1 = nn.LayerNorm(ch)
        self.att1 = NeighborhoodAttention2D(ch, num_heads, kernel_size)   # (the reference: always 3, core/modules.py:509)
        self.norm2 = nn.LayerNorm(ch)
        self.mlp = Mlp(ch, 2 * ch, ch)


def stage(cin, cexp, cout):
    return nn.Sequential(ReparamConv(cin, cexp, cout), ReparamConv(cout, cexp, cout))
