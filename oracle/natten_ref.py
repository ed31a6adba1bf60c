"""TEST INFRASTRUCTURE ONLY -- CPU oracle for 2-D neighborhood attention.

Restates the published semantics of ``natten.NeighborhoodAttention2D`` (PyPI
``natten`` / SHI-Labs NATTEN, the rpb-carrying 0.14.x API; NOT vendored in the
reference and version-unpinned there -- see SURVEY.md section 8c).  The
reference only touches it at ``core/modules.py:18`` (import), ``:509``
(``dim=C, num_heads=12, kernel_size=3``) and ``:517`` (call on an NHWC tensor).

**Parity of the NA core is UNPINNED by the reference** (it holds no tests or
golden vectors at this boundary).  It is pinned here by (i) the brute-force
per-pixel loop ``na2d_bruteforce`` below, (ii) hand-checkable known answers in
``tests/test_oracle_na.py``.

Semantics (kernel_size K, neighborhood NS = K // 2, dilation 1, non-causal):
  * window start along an axis of length L for query index i:
        start(i) = clamp(i - NS, 0, L - K)          (window slides inward at
    borders; it is never zero padded)
  * relative-position-bias index of window slot k for query i:
        (start(i) + k - i) + (K - 1)                 in [0, 2K-2]
  * attn[b,h,i,j,ki*K+kj] = scale * q[b,h,i,j,:] . k[b,h,start(i)+ki,start(j)+kj,:]
                            + rpb[h, bias_idx(i,ki), bias_idx(j,kj)]
  * out[b,h,i,j,:] = sum_n softmax_n(attn)[n] * v[b,h,start(i)+ki,start(j)+kj,:]
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def window_start(length: int, ksize: int, device=None) -> torch.Tensor:
    """start(i) for every query index along one axis."""
    ns = ksize // 2
    idx = torch.arange(length, device=device)
    return (idx - ns).clamp(0, length - ksize)


def _axis_tables(length: int, ksize: int, device=None):
    """(nbr[L,K] absolute neighbour index, bias[L,K] rpb index) for one axis."""
    start = window_start(length, ksize, device)
    k = torch.arange(ksize, device=device)
    nbr = start[:, None] + k[None, :]
    bias = nbr - torch.arange(length, device=device)[:, None] + (ksize - 1)
    return nbr, bias


def na2d_qkrpb(q, k, rpb, ksize: int):
    """q, k: [B, h, H, W, d] (q already scaled); rpb: [h, 2K-1, 2K-1] or None.

    Returns attn logits [B, h, H, W, K*K].
    """
    B, h, H, W, d = q.shape
    assert H >= ksize and W >= ksize, "feature map smaller than the NA window"
    rn, rb = _axis_tables(H, ksize, q.device)
    cn, cb = _axis_tables(W, ksize, q.device)
    out = []
    for ki in range(ksize):
        krow = k[:, :, rn[:, ki]]                      # [B,h,H,W,d] rows gathered
        for kj in range(ksize):
            kk = krow[:, :, :, cn[:, kj]]              # cols gathered
            a = (q * kk).sum(-1)
            if rpb is not None:
                a = a + rpb[:, rb[:, ki]][:, :, cb[:, kj]][None]
            out.append(a)
    return torch.stack(out, dim=-1)


def na2d_av(attn, v, ksize: int):
    """attn: [B,h,H,W,K*K] (post-softmax); v: [B,h,H,W,d] -> [B,h,H,W,d]."""
    B, h, H, W, d = v.shape
    rn, _ = _axis_tables(H, ksize, v.device)
    cn, _ = _axis_tables(W, ksize, v.device)
    out = torch.zeros_like(v)
    n = 0
    for ki in range(ksize):
        vrow = v[:, :, rn[:, ki]]
        for kj in range(ksize):
            out = out + attn[..., n:n + 1] * vrow[:, :, :, cn[:, kj]]
            n += 1
    return out


def na2d_bruteforce(q, k, v, rpb, ksize: int):
    """O(HW*K*K) python-loop restatement, fp64, for small cross-checks only.

    q (already scaled), k, v: [B,h,H,W,d].  Returns out [B,h,H,W,d] (fp64).
    """
    q, k, v = q.double(), k.double(), v.double()
    rpb = None if rpb is None else rpb.double()
    B, h, H, W, d = q.shape
    ns = ksize // 2
    out = torch.zeros_like(q)
    for i in range(H):
        si = min(max(i - ns, 0), H - ksize)
        for j in range(W):
            sj = min(max(j - ns, 0), W - ksize)
            logits = torch.empty(B, h, ksize * ksize, dtype=torch.float64)
            for ki in range(ksize):
                for kj in range(ksize):
                    a = (q[:, :, i, j] * k[:, :, si + ki, sj + kj]).sum(-1)
                    if rpb is not None:
                        a = a + rpb[:, si + ki - i + ksize - 1, sj + kj - j + ksize - 1][None]
                    logits[:, :, ki * ksize + kj] = a
            p = torch.softmax(logits, -1)
            acc = torch.zeros(B, h, d, dtype=torch.float64)
            for ki in range(ksize):
                for kj in range(ksize):
                    acc += p[:, :, ki * ksize + kj, None] * v[:, :, si + ki, sj + kj]
            out[:, :, i, j] = acc
    return out


class NeighborhoodAttention2D(nn.Module):
    """Module-level restatement (state_dict keys: rpb, qkv.*, proj.*).

    Input/output NHWC ``[B, H, W, C]`` exactly as natten's module; the
    reference feeds it ``LayerNorm(e)`` at ``core/modules.py:516-517``.
    Defaults follow the rpb-carrying natten API: qkv_bias=True, bias (rpb)=True,
    dilation=1, attn_drop=proj_drop=0, scale=head_dim**-0.5.
    """

    def __init__(self, dim, num_heads, kernel_size=3, qkv_bias=True, rpb=True):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.kernel_size = kernel_size
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        if rpb:
            self.rpb = nn.Parameter(torch.zeros(num_heads, 2 * kernel_size - 1, 2 * kernel_size - 1))
            nn.init.trunc_normal_(self.rpb, std=0.02, mean=0.0, a=-2.0, b=2.0)
        else:
            self.register_parameter("rpb", None)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, H, W, C = x.shape
        qkv = self.qkv(x).reshape(B, H, W, 3, self.num_heads, self.head_dim)
        q, k, v = qkv.permute(3, 0, 4, 1, 2, 5).unbind(0)       # each [B,h,H,W,d]
        attn = na2d_qkrpb(q * self.scale, k, self.rpb, self.kernel_size)
        attn = F.softmax(attn, dim=-1)
        o = na2d_av(attn, v, self.kernel_size)                  # [B,h,H,W,d]
        o = o.permute(0, 2, 3, 1, 4).reshape(B, H, W, C)
        return self.proj(o)
