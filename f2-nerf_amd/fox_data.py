"""The reference's example scene (data/example/ngp_fox) as committed fixtures: cameras / octree / warps in
tests/golden/fox_state.npz (960x540 intrinsics = dataset.factor 2 of confs/wanjinyou.yaml) and the photographs themselves.

Pixels come in two resolutions:
  factor 2 (540 x 960, what wanjinyou.yaml trains on): tests/golden/fox_images_f2_jpeg.npz holds the reference's own
            images_2/*.jpg files byte for byte; they are decoded here with PIL (the reference decodes the same files with
            stb_image through Utils::ReadImageTensor and divides by 255);
  factor 8 (135 x 240): tests/golden/fox_images_f8.npz, box-filtered, for quick runs.
File IO is outside the hot path: this module only turns bytes into the [C,H,W,3] fp32 tensor that the
host `Dataset` keeps resident in HBM.
"""
import io
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_state():
    return dict(np.load(os.path.join(GOLDEN, "fox_state.npz")))


def load_images(factor=2):
    """-> (uint8 [C,H,W,3], factor relative to the 960x540 intrinsics of fox_state.npz)."""
    if factor == 2:
        from PIL import Image
        z = np.load(os.path.join(GOLDEN, "fox_images_f2_jpeg.npz"))
        buf, off = z["jpeg"], z["offsets"]
        imgs = [np.asarray(Image.open(io.BytesIO(buf[off[i]:off[i + 1]].tobytes())).convert("RGB"), np.uint8)
                for i in range(len(off) - 1)]
        return np.stack(imgs), 1.0
    if factor == 8:
        z = np.load(os.path.join(GOLDEN, "fox_images_f8.npz"))
        return z["images"], float(z["factor_vs_state"])
    raise ValueError("fox images are committed at factor 2 and factor 8 only")


def scene(factor=2):
    """-> (state dict whose intrinsics / image size match the pixels, images fp32 [C,H,W,3] in [0,1] as a torch tensor)."""
    import torch
    st = load_state()
    u8, f = load_images(factor)
    images = torch.from_numpy(u8.astype(np.float32) / np.float32(255.))
    sc = dict(st)
    sc["image_hw"] = np.array(images.shape[1:3])
    sc["intri"] = st["intri"].copy()
    sc["intri"][:, :2, :] /= np.float32(f)
    return sc, images
