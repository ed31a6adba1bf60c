"""GPU: lmn_preprocess_u8 through lm_net_amd.data.DevicePreprocess against the oracle (row N4): labels and the
resized uint8 values bit-exact (the normalised floats are a fixed function of them: compared exactly)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_ref as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,hs,ws,h,w", [(3, 37, 53, 32, 48), (2, 512, 620, 256, 256), (2, 100, 80, 352, 352), (1, 64, 64, 64, 64),
                                         (8, 530, 622, 352, 352)])
def test_preprocess_matches_oracle(B, hs, ws, h, w):
    from lm_net_amd.data import DevicePreprocess
    rng = np.random.default_rng(B * 1000 + hs)
    img = rng.integers(0, 256, (B, hs, ws, 3), dtype=np.uint8)
    mask = rng.integers(0, 256, (B, hs, ws), dtype=np.uint8)
    flips = rng.integers(0, 4, (B,), dtype=np.uint8)
    xr, yr = P.preprocess(img, mask, (h, w), flips=flips)
    x, y = DevicePreprocess((h, w))(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), torch.from_numpy(flips).cuda())
    assert x.shape == (B, 3, h, w) and y.shape == (B, h, w) and y.dtype == torch.int64
    assert np.array_equal(y.cpu().numpy(), yr)
    assert np.array_equal(x.cpu().numpy(), xr)
    # no flips / images only / masks only
    x2, none = DevicePreprocess((h, w))(torch.from_numpy(img).cuda())
    assert none is None and np.array_equal(x2.cpu().numpy(), P.preprocess(img, mask, (h, w))[0])
    none, y2 = DevicePreprocess((h, w))(None, torch.from_numpy(mask).cuda())
    assert none is None and np.array_equal(y2.cpu().numpy(), P.preprocess(img, mask, (h, w))[1])


def test_preprocess_feeds_the_model():
    from lm_net_amd import LM_Net
    from lm_net_amd.data import DevicePreprocess
    rng = np.random.default_rng(5)
    img = torch.from_numpy(rng.integers(0, 256, (2, 90, 120, 3), dtype=np.uint8)).cuda()
    mask = torch.from_numpy(rng.integers(0, 256, (2, 90, 120), dtype=np.uint8)).cuda()
    x, y = DevicePreprocess((64, 96))(img, mask)
    out = LM_Net(3, 2).cuda().eval()(x)
    assert out.shape == (2, 2, 64, 96) and torch.isfinite(out).all() and int(y.max()) <= 1
