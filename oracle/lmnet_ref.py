"""TEST INFRASTRUCTURE ONLY -- CPU oracle (pure PyTorch fp32) for ``core.LM_Net``.

A from-scratch restatement of the reference forward graph, written against the
reference's semantics (file:line cited per class) with the SAME ``state_dict``
key names and tensor shapes, so one weight set drives the reference, this
oracle and the HIP product.  It is pinned against the real reference code by
``tests/golden/*`` (made by ``tools/make_golden.py``, which imports
/root/reference here) and, when /root/reference is present, directly by
``tests/test_oracle_vs_reference.py``.  The neighborhood-attention core is
external to the reference (natten, unpinned) -- see ``oracle/natten_ref.py``.

Used by: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Never
imported by the product package.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .natten_ref import NeighborhoodAttention2D

NUM_HEADS = 12          # core/LM_Net.py:56,81-84 (last ctor arg of GFT / NeighborhoodTransformer)


def _conv(cin, cout, k, stride=1, groups=1, bias=True, padding=None):
    if padding is None:
        padding = (k // 2) if isinstance(k, int) else (k[0] // 2, k[1] // 2)
    return nn.Conv2d(cin, cout, k, stride, padding, groups=groups, bias=bias)


def _up2(x):
    # nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)  core/LM_Net.py:59, modules.py:94,129
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


class SE(nn.Module):
    """Squeeze-excite gate.  core/modules.py:1020-1044."""

    def __init__(self, channels, reduction=4):
        super().__init__()
        self.fc1 = _conv(channels, channels // reduction, 1)
        self.fc2 = _conv(channels // reduction, channels, 1)
        for m in (self.fc1, self.fc2):
            nn.init.kaiming_normal_(m.weight)           # modules.py:1038-1041

    def gate(self, g):
        m = g.mean(dim=(2, 3), keepdim=True)
        return F.hardsigmoid(self.fc2(F.relu(self.fc1(m))))

    def forward(self, g):
        return g * self.gate(g)


class _ConvBN(nn.Sequential):
    def __init__(self, conv, ch):
        super().__init__(OrderedDict([("conv", conv), ("bn", nn.BatchNorm2d(ch))]))


class ReparamConv(nn.Module):
    """Expand 1x1+BN+Hardswish -> 4 BN'd depthwise branches (5x5,3x3,3x1,1x3) summed
    -> GELU -> SE -> 1x1, plus a 1x1 shortcut.  core/modules.py:525-600.
    Deploy form (single 5x5 depthwise with bias): modules.py:602-657.
    """

    def __init__(self, cin, cexp, cout, large_k=5, small_k=3):
        super().__init__()
        self.cexp, self.large_k, self.small_k = cexp, large_k, small_k
        self.deploy = False
        self.se = SE(cexp)
        self.expand_conv = nn.Sequential(_conv(cin, cexp, 1), nn.BatchNorm2d(cexp), nn.Hardswish())
        self.large_conv = _ConvBN(_conv(cexp, cexp, large_k, groups=cexp, bias=False), cexp)
        self.square_conv = _ConvBN(_conv(cexp, cexp, small_k, groups=cexp, bias=False), cexp)
        self.ver_conv = _ConvBN(_conv(cexp, cexp, (small_k, 1), groups=cexp, bias=False), cexp)
        self.hor_conv = _ConvBN(_conv(cexp, cexp, (1, small_k), groups=cexp, bias=False), cexp)
        self.pointwise_conv = nn.Sequential(_conv(cexp, cout, 1))
        self.shortcut = nn.Sequential(_conv(cin, cout, 1))

    def depthwise(self, x1):
        if self.deploy:
            return self.fuse_conv(x1)
        return self.large_conv(x1) + self.square_conv(x1) + self.ver_conv(x1) + self.hor_conv(x1)

    def forward(self, x):
        x1 = self.expand_conv(x)
        g = F.gelu(self.depthwise(x1))
        return self.pointwise_conv(self.se(g)) + self.shortcut(x)

    # ---- structural re-parameterisation (eval-time folding), modules.py:602-657
    @staticmethod
    def _fold(branch):
        bn = branch.bn
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return branch.conv.weight * s.view(-1, 1, 1, 1), bn.bias - bn.running_mean * s

    def equivalent_kernel_bias(self):
        K, k = self.large_k, self.small_k
        w, b = self._fold(self.large_conv)
        w = w.clone()
        c = K // 2
        for br in (self.square_conv, self.ver_conv, self.hor_conv):
            wb, bb = self._fold(br)
            kh, kw = wb.shape[2:]
            w[:, :, c - kh // 2:c - kh // 2 + kh, c - kw // 2:c - kw // 2 + kw] += wb
            b = b + bb
        return w, b

    @torch.no_grad()
    def switch_to_deploy(self):
        w, b = self.equivalent_kernel_bias()
        self.fuse_conv = _conv(self.cexp, self.cexp, self.large_k, groups=self.cexp, bias=True)
        self.fuse_conv.weight.data = w
        self.fuse_conv.bias.data = b
        self.deploy = True
        # the reference deletes the three small branches and keeps large_conv (modules.py:654-657)
        del self.square_conv, self.hor_conv, self.ver_conv


class M3Skip(nn.Module):
    """3-scale fusion: large(stride-2 3x3), mid(3x3), small(up x2 -> 3x3); cat(l,m,s) -> 3x3+BN+GELU.
    core/modules.py:83-107."""

    def __init__(self, ch):
        super().__init__()
        cl, cm, cs = ch
        self.convl = nn.Sequential(_conv(cl, cm, 3, 2))
        self.convm = nn.Sequential(_conv(cm, cm, 3))
        self.convs = nn.Sequential(nn.Identity(), _conv(cs, cm, 3))      # index 1 = conv, as in the reference
        self.fuse_conv = nn.Sequential(_conv(3 * cm, cm, 3), nn.BatchNorm2d(cm), nn.GELU())

    def forward(self, xl, xm, xs):
        cat = torch.cat([self.convl(xl), self.convm(xm), self.convs[1](_up2(xs))], dim=1)
        return self.fuse_conv(cat)


class M2Skip(nn.Module):
    """2-scale fusion.  'bottom': large(stride-2), small(3x3) ; 'top': large(3x3), small(up x2 -> 3x3).
    core/modules.py:109-143."""

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
            self.convs = nn.Sequential(nn.Identity(), _conv(c1, c0, 3))
            cf = c0
        self.fuse_conv = nn.Sequential(_conv(2 * cf, cf, 3), nn.BatchNorm2d(cf), nn.GELU())

    def forward(self, xl, xs):
        if self.model_type == "bottom":
            s = self.convs(xs)
        else:
            s = self.convs[1](_up2(xs))
        return self.fuse_conv(torch.cat([self.convl(xl), s], dim=1))


class OverlapPatchEmbed(nn.Module):
    """3x3 stride-1 conv; returns NHWC ('nat') or tokens [B,N,C].  core/modules.py:22-40."""

    def __init__(self, cin, cout, patch=3, stride=1):
        super().__init__()
        self.patch_embeddings = _conv(cin, cout, patch, stride)

    def forward(self, x):
        return self.patch_embeddings(x).permute(0, 2, 3, 1)               # NHWC


class Mlp(nn.Module):
    """fc1 -> GELU -> Dropout(0.1) -> fc2 -> Dropout(0.1).  core/modules.py:42-56."""

    def __init__(self, cin, chid, cout, p=0.1):
        super().__init__()
        self.fc1 = nn.Linear(cin, chid)
        self.fc2 = nn.Linear(chid, cout)
        self.p = p

    def forward(self, x):
        x = F.dropout(F.gelu(self.fc1(x)), self.p, self.training)
        return F.dropout(self.fc2(x), self.p, self.training)


class GlobalAttention(nn.Module):
    """Dense multi-head self-attention over all tokens.  core/modules.py:235-279."""

    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        for m in (self.qkv, self.proj):                                   # modules.py:250-254
            nn.init.trunc_normal_(m.weight, std=0.02)
            nn.init.zeros_(m.bias)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).view(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        p = torch.softmax((q @ k.transpose(-2, -1)) * self.scale, dim=-1)
        return self.proj((p @ v).transpose(1, 2).reshape(B, N, C))


class GFT(nn.Module):
    """Patch-embed 3x3 -> pre-LN ViT block on HxW tokens -> 1x1 conv.  core/modules.py:329-347."""

    def __init__(self, cin, expand, cout, num_heads):
        super().__init__()
        self.patchembedding = OverlapPatchEmbed(cin, cin)
        self.norm1 = nn.LayerNorm(cin)
        self.attention = GlobalAttention(cin, num_heads)
        self.norm2 = nn.LayerNorm(cin)
        self.mlp = Mlp(cin, expand * cin, cin)
        self.conv = nn.Sequential(_conv(cin, cout, 1))

    def forward(self, x):
        B, C, H, W = x.shape
        e = self.patchembedding(x).reshape(B, H * W, C)
        a = self.attention(self.norm1(e)) + e
        y = self.mlp(self.norm2(a)) + a
        return self.conv(y.reshape(B, H, W, C).permute(0, 3, 1, 2))


class PyramidPool(nn.Module):
    """avg-pool x1..x4 to x5's grid, cat [x1,x2,x3,x4,x5].  core/modules.py:481-498."""

    def forward(self, x1, x2, x3, x4, x5):
        hw = x5.shape[2:]
        return torch.cat([F.adaptive_avg_pool2d(t, hw) for t in (x1, x2, x3, x4)] + [x5], dim=1)


class NeighborhoodTransformer(nn.Module):
    """3x3 patch-embed -> pre-LN block with neighborhood attention (K=3) + Mlp(C,2C,C).
    core/modules.py:504-521."""

    def __init__(self, ch, num_heads, kernel_size=3):
        super().__init__()
        self.patchembedding = OverlapPatchEmbed(ch, ch)
        self.norm1 = nn.LayerNorm(ch)
        self.att1 = NeighborhoodAttention2D(dim=ch, num_heads=num_heads, kernel_size=kernel_size)   # (the reference: 3)
        self.norm2 = nn.LayerNorm(ch)
        self.mlp = Mlp(ch, 2 * ch, ch)

    def forward(self, x):
        e = self.patchembedding(x)                                         # NHWC
        a = self.att1(self.norm1(e)) + e
        y = self.mlp(self.norm2(a)) + a
        return y.permute(0, 3, 1, 2).contiguous()


def _stage(cin, cexp, cout):
    return nn.Sequential(ReparamConv(cin, cexp, cout), ReparamConv(cout, cexp, cout))


class LM_Net(nn.Module):
    """core/LM_Net.py:5-123 (ctor 6-87, structural_reparam 90-93, forward 95-123)."""

    def __init__(self, channel, n_classes=2, filters=(12, 24, 48, 96, 192), deep_supervision=False, na_kernel_size=3):
        super().__init__()
        f = list(filters)
        self.filters, self.deep_supervision = f, deep_supervision
        self.conv1 = _stage(channel, f[1], f[0]); self.down1 = nn.Sequential(_conv(f[0], f[1], 3, 2))
        self.conv2 = _stage(f[1], f[2], f[1]);    self.down2 = nn.Sequential(_conv(f[1], f[2], 3, 2))
        self.conv3 = _stage(f[2], f[3], f[2]);    self.down3 = nn.Sequential(_conv(f[2], f[3], 3, 2))
        self.conv4 = _stage(f[3], f[4], f[3]);    self.down4 = nn.Sequential(_conv(f[3], f[4], 3, 2))
        self.dconv1 = _stage(f[3], f[4], f[3])
        self.dconv2 = _stage(f[2], f[3], f[2])
        self.dconv3 = _stage(f[1], f[2], f[1])
        self.dconv4 = _stage(f[0], f[1], f[0])
        self.pyramidpool = PyramidPool()
        self.gft = GFT(sum(f), 2, f[4], NUM_HEADS)
        self.up1 = nn.Sequential(nn.Identity(), _conv(f[4], f[3], 3))
        self.up2 = nn.Sequential(nn.Identity(), _conv(f[3], f[2], 3))
        self.up3 = nn.Sequential(nn.Identity(), _conv(f[2], f[1], 3))
        self.up4 = nn.Sequential(nn.Identity(), _conv(f[1], f[0], 3))
        self.skip1 = M2Skip([f[2], f[3]], "bottom")
        self.skip2 = M3Skip([f[1], f[2], f[3]])
        self.skip3 = M3Skip([f[0], f[1], f[2]])
        self.skip4 = M2Skip([f[0], f[1]], "top")
        self.natt1 = NeighborhoodTransformer(f[3], NUM_HEADS, na_kernel_size)
        self.natt2 = NeighborhoodTransformer(f[2], NUM_HEADS, na_kernel_size)
        self.natt3 = NeighborhoodTransformer(f[1], NUM_HEADS, na_kernel_size)
        self.natt4 = NeighborhoodTransformer(f[0], NUM_HEADS, na_kernel_size)
        self.output_layer = _conv(f[0], n_classes, 1)

    def structural_reparam(self):
        for m in list(self.modules()):
            if hasattr(m, "switch_to_deploy") and not m.deploy:
                m.switch_to_deploy()

    def forward(self, x, taps=None):
        """``taps`` (optional dict) receives the named stage activations of core/LM_Net.py:96-120."""
        x1 = self.conv1(x);  xd1 = self.down1(x1)
        x2 = self.conv2(xd1); xd2 = self.down2(x2)
        x3 = self.conv3(xd2); xd3 = self.down3(x3)
        x4 = self.conv4(xd3); xd4 = self.down4(x4)
        x5 = self.gft(self.pyramidpool(x1, x2, x3, x4, xd4))
        xs1 = self.skip1(x3, x4)
        xs2 = self.skip2(x2, x3, x4)
        xs3 = self.skip3(x1, x2, x3)
        xs4 = self.skip4(x1, x2)
        x46, x37, x28, x19 = self.natt1(xs1), self.natt2(xs2), self.natt3(xs3), self.natt4(xs4)
        x6 = self.dconv1(self.up1[1](_up2(x5)) + x46)
        x7 = self.dconv2(self.up2[1](_up2(x6)) + x37)
        x8 = self.dconv3(self.up3[1](_up2(x7)) + x28)
        x9 = self.dconv4(self.up4[1](_up2(x8)) + x19)
        out = self.output_layer(x9)
        if taps is not None:
            taps.update(x1=x1, x2=x2, x3=x3, x4=x4, xd4=xd4, x5=x5, xs1=xs1, xs2=xs2, xs3=xs3, xs4=xs4,
                        x46=x46, x37=x37, x28=x28, x19=x19, x6=x6, x7=x7, x8=x8, x9=x9, out=out)
        return out
