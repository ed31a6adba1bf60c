"""CPU: pins the neighborhood-attention oracle (external natten semantics; parity UNPINNED by the
reference) with a brute-force loop and hand-checkable known answers."""
import torch

from oracle.natten_ref import (NeighborhoodAttention2D, na2d_av, na2d_bruteforce, na2d_qkrpb,
                               window_start)


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def test_window_start_is_clamped_not_padded():
    assert window_start(6, 3).tolist() == [0, 0, 1, 2, 3, 3]
    assert window_start(3, 3).tolist() == [0, 0, 0]
    assert window_start(7, 5).tolist() == [0, 0, 0, 1, 2, 2, 2]


def test_vectorised_matches_bruteforce():
    for (H, W, d, K) in [(3, 3, 2, 3), (4, 7, 1, 3), (6, 5, 4, 3), (7, 8, 2, 5)]:
        q, k, v = (_rand(2, 3, H, W, d, seed=s) for s in (1, 2, 3))
        rpb = _rand(3, 2 * K - 1, 2 * K - 1, seed=4)
        attn = torch.softmax(na2d_qkrpb(q, k, rpb, K), -1)
        out = na2d_av(attn, v, K)
        ref = na2d_bruteforce(q, k, v, rpb, K)
        assert (out.double() - ref).abs().max() < 1e-5


def test_uniform_query_gives_window_mean():
    H, W, d = 5, 6, 3
    v = _rand(1, 1, H, W, d, seed=7)
    q = torch.zeros(1, 1, H, W, d)
    attn = torch.softmax(na2d_qkrpb(q, _rand(1, 1, H, W, d), None, 3), -1)
    out = na2d_av(attn, v, 3)
    # corner (0,0) attends rows {0,1,2} x cols {0,1,2}; interior (2,3) rows {1,2,3} x cols {2,3,4}
    assert torch.allclose(out[0, 0, 0, 0], v[0, 0, 0:3, 0:3].reshape(9, d).mean(0), atol=1e-6)
    assert torch.allclose(out[0, 0, 2, 3], v[0, 0, 1:4, 2:5].reshape(9, d).mean(0), atol=1e-6)
    assert torch.allclose(out[0, 0, 4, 5], v[0, 0, 2:5, 3:6].reshape(9, d).mean(0), atol=1e-6)


def test_rpb_indices_corner_and_interior():
    # q = 0 -> logits == bias; bias[(nbr - i) + 2]
    H = W = 4
    rpb = torch.arange(25.).reshape(1, 5, 5)
    z = torch.zeros(1, 1, H, W, 1)
    a = na2d_qkrpb(z, z, rpb, 3)[0, 0]
    assert a[0, 0].tolist() == [rpb[0, r, c].item() for r in (2, 3, 4) for c in (2, 3, 4)]
    assert a[1, 2].tolist() == [rpb[0, r, c].item() for r in (1, 2, 3) for c in (1, 2, 3)]
    assert a[3, 3].tolist() == [rpb[0, r, c].item() for r in (0, 1, 2) for c in (0, 1, 2)]


def test_k3_on_3x3_map_is_global_attention():
    q, k, v = (_rand(1, 2, 3, 3, 4, seed=s) for s in (1, 2, 3))
    attn = torch.softmax(na2d_qkrpb(q, k, None, 3), -1)
    out = na2d_av(attn, v, 3)
    qf, kf, vf = (t.reshape(1, 2, 9, 4) for t in (q, k, v))
    ref = torch.softmax(qf @ kf.transpose(-1, -2), -1) @ vf
    assert torch.allclose(out.reshape(1, 2, 9, 4), ref, atol=1e-6)


def test_module_shapes_and_state_dict_keys():
    m = NeighborhoodAttention2D(24, 12, 3)
    assert list(m.state_dict().keys()) == ["rpb", "qkv.weight", "qkv.bias", "proj.weight", "proj.bias"]
    assert m.rpb.shape == (12, 5, 5) and m.qkv.weight.shape == (72, 24)
    assert m(_rand(2, 5, 7, 24)).shape == (2, 5, 7, 24)
