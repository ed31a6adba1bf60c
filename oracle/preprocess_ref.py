"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the input pipeline of row N4 (never imported by the product path).

Follows `dataset/data_loading.py:203-206` (A.Resize(256,256) -> A.Normalize() -> ToTensorV2), `:213-214` (flips) and
`:235-237` (cv2.imread, mask threshold).  The arithmetic lives in two third-party packages that are NOT in
/root/reference and not installed here: OpenCV (`cv2.resize`, version unpinned by the reference) and albumentations
(`A.Normalize`, unpinned).  Restated from their published algorithms:

  * cv2.resize(INTER_LINEAR) on uint8 (modules/imgproc/src/resize.cpp, resizeGeneric_ / HResizeLinear /
    VResizeLinear<uchar>): src coordinate f = (d + 0.5) * (n_src / n_dst) - 0.5 in float32, s = floor(f); columns
    zero the fraction at the borders, rows are clamped; coefficients saturate_cast<short>(w * 2048) (round half to
    even); horizontal pass in int32 at scale 2^11; vertical pass
    ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, saturated to uint8;
  * cv2.resize(INTER_NEAREST): s = min(floor(d * n_src / n_dst), n_src - 1);
  * A.Normalize: img.astype(float32); img -= mean * 255 (float64); img *= reciprocal(std * 255) (float64) -- numpy
    computes each in-place step in float64 and stores float32.

PARITY UNPINNED against OpenCV itself: the reference holds no fixtures for its data pipeline and cv2 cannot be
imported in this environment; pinned here by hand-checkable cases (identity size, exact 2x2 averaging at scale 2,
constant images, flips) in tests/test_preprocess_cpu.py.
"""
import numpy as np


def _axis(n_dst, n_src, clamp_frac):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * (float(n_src) / float(n_dst)) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo, hi = s < 0, s >= n_src - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, n_src - 1, s))
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1), a0, a1


def resize_linear_u8(img, H, W):
    """img uint8 [Hs,Ws,C] -> uint8 [H,W,C], cv2.resize(img, (W, H), interpolation=cv2.INTER_LINEAR)."""
    Hs, Ws = img.shape[:2]
    x0, x1, ax0, ax1 = _axis(W, Ws, True)
    y0, y1, ay0, ay1 = _axis(H, Hs, False)
    src = img.astype(np.int64)
    hor = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]          # [Hs, W, C], scale 2^11
    r0, r1 = hor[y0], hor[y1]
    v = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_nearest(mask, H, W):
    Hs, Ws = mask.shape[:2]
    sy = np.minimum(np.floor(np.arange(H) * (float(Hs) / H)).astype(np.int64), Hs - 1)
    sx = np.minimum(np.floor(np.arange(W) * (float(Ws) / W)).astype(np.int64), Ws - 1)
    return mask[sy][:, sx]


def normalize(img_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    m = np.array(mean, dtype=np.float64) * 255.0
    d = np.reciprocal(np.array(std, dtype=np.float64) * 255.0)
    out = img_u8.astype(np.float32)
    out -= m
    out *= d
    return out


def preprocess(images, masks, size, flips=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """images uint8 [B,Hs,Ws,3], masks uint8 [B,Hs,Ws] -> (float32 [B,3,H,W], int64 [B,H,W])."""
    H, W = size
    xs, ys = [], []
    for b in range(images.shape[0]):
        im = resize_linear_u8(images[b], H, W)
        mk = resize_nearest((masks[b] > 127).astype(np.uint8), H, W)
        fl = int(flips[b]) if flips is not None else 0
        if fl & 1:
            im, mk = im[:, ::-1], mk[:, ::-1]
        if fl & 2:
            im, mk = im[::-1], mk[::-1]
        xs.append(normalize(np.ascontiguousarray(im), mean, std).transpose(2, 0, 1))
        ys.append(mk.astype(np.int64))
    return np.stack(xs), np.stack(ys)
