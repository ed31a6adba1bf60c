"""Device-side input pipeline.

The reference prepares every sample on the CPU with albumentations (`dataset/data_loading.py:203-229`):
`cv2.imread` (uint8 HWC, channel order as read), mask `cv2.threshold(127, 1)`, then `A.Resize(256, 256)`,
`A.Normalize()` and `ToTensorV2()` for validation; the training transform adds flips and colour / geometric
augmentations in front of the same Normalize.  `DevicePreprocess` runs the resize + (optional) flips + normalise +
layout change for a whole batch of raw uint8 frames already in HBM as ONE kernel (`lmn_preprocess_u8`), so decoded
frames can go to the GPU as bytes (3 B/pixel over PCIe instead of 12) and an 8-GPU loop is not fed by a CPU
albumentations pool.  The colour / elastic augmentations of the training transform stay on the CPU side (out of scope).
"""
import torch

from . import hip


class DevicePreprocess:
    """`x, y = DevicePreprocess((256, 256))(images_u8, masks_u8, flips=None)`.

    images_u8: uint8 [B,Hs,Ws,3] on the GPU; masks_u8: uint8 [B,Hs,Ws] or None; flips: uint8 [B] or None
    (bit 0 horizontal, bit 1 vertical -- `A.HorizontalFlip` / `A.VerticalFlip`, drawn by the caller).
    Returns fp32 [B,3,H,W] (what `LM_Net.forward` takes) and int64 [B,H,W] labels in {0,1}."""

    def __init__(self, size=(256, 256), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        self.size = (int(size[0]), int(size[1]))
        self.mean, self.std = tuple(mean), tuple(std)     # A.Normalize() defaults, max_pixel_value = 255

    def __call__(self, images, masks=None, flips=None):
        ref = images if images is not None else masks
        if ref is None:
            raise ValueError("DevicePreprocess: images or masks required")
        if not ref.is_cuda:
            raise RuntimeError("DevicePreprocess runs on the HIP device only (got %s); there is no CPU path" % ref.device)
        B, (H, W) = ref.shape[0], self.size
        x = torch.empty(B, 3, H, W, device=ref.device, dtype=torch.float32) if images is not None else None
        y = torch.empty(B, H, W, device=ref.device, dtype=torch.int64) if masks is not None else None
        hip.preprocess_u8(images, masks, flips, x, y, self.mean, self.std)
        return x, y
