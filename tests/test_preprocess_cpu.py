"""CPU: the oracle of the device-side input pipeline (row N4) against hand-checkable answers; host-side errors."""
import numpy as np
import pytest
import torch

from oracle import preprocess_ref as P


def test_identity_size_returns_the_image():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    assert np.array_equal(P.resize_linear_u8(img, 19, 23), img)
    m = rng.integers(0, 2, (19, 23), dtype=np.uint8)
    assert np.array_equal(P.resize_nearest(m, 19, 23), m)


def test_half_size_is_the_rounded_2x2_mean():
    rng = np.random.default_rng(2)
    big = rng.integers(0, 256, (16, 24, 3), dtype=np.uint8).astype(np.int64)
    ref = (big[0::2, 0::2] + big[1::2, 0::2] + big[0::2, 1::2] + big[1::2, 1::2] + 2) >> 2
    assert np.array_equal(P.resize_linear_u8(big.astype(np.uint8), 8, 12), ref.astype(np.uint8))
    m = np.arange(16 * 24).reshape(16, 24)
    assert np.array_equal(P.resize_nearest(m, 8, 12), m[0::2, 0::2])      # INTER_NEAREST takes floor(d * scale)


def test_constant_image_stays_constant_under_any_scale():
    for hs, ws, h, w in [(7, 9, 32, 48), (50, 40, 16, 16), (33, 65, 32, 64)]:
        img = np.full((hs, ws, 3), 137, dtype=np.uint8)
        assert np.all(P.resize_linear_u8(img, h, w) == 137)


def test_upscale_interpolates_between_neighbours_and_clamps_the_border():
    row = np.array([[0, 100]], dtype=np.uint8).reshape(1, 2, 1)
    out = P.resize_linear_u8(row, 1, 4)[0, :, 0]
    assert list(out) == [0, 25, 75, 100]          # centres at -0.25, 0.25, 0.75, 1.25: border clamp, 1/4, 3/4, clamp


def test_normalize_and_flips_and_threshold():
    img = np.zeros((1, 4, 6, 3), dtype=np.uint8)
    img[0, 0, 0] = (255, 0, 128)
    mask = np.zeros((1, 4, 6), dtype=np.uint8)
    mask[0, 0, 0], mask[0, 3, 5] = 128, 127
    x, y = P.preprocess(img, mask, (4, 6), flips=[3])
    assert y[0, 3, 5] == 1 and y.sum() == 1                       # 128 > 127 -> 1, moved by both flips; 127 -> 0
    ref = (np.array([255, 0, 128], dtype=np.float64) - np.array([0.485, 0.456, 0.406]) * 255) / (np.array([0.229, 0.224, 0.225]) * 255)
    assert np.allclose(x[0, :, 3, 5], ref, rtol=1e-6)
    assert x.dtype == np.float32 and y.dtype == np.int64


def test_device_preprocess_has_no_cpu_path():
    from lm_net_amd.data import DevicePreprocess
    with pytest.raises(RuntimeError):
        DevicePreprocess((32, 32))(torch.zeros(1, 40, 40, 3, dtype=torch.uint8), torch.zeros(1, 40, 40, dtype=torch.uint8))
