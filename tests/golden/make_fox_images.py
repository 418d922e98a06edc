#!/usr/bin/env python3
"""Generates tests/golden/fox_images_f8.npz: the 50 ngp_fox photographs of the reference's example data set
(/root/reference/data/example/ngp_fox/images_2, 960x540) box-filtered by 4 to 240x135 (= factor 8 of the originals),
uint8.  Data, not source: it lets the GPU box (which has no /root/reference) train on real pixels and report a PSNR
(tools/train_fox.py).  Image order = sorted file names = the order of cams_meta.npy (Dataset.cpp:86-93)."""
import glob
import os
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/example/ngp_fox/images_2"
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fox_images_f8.npz")
files = sorted(glob.glob(os.path.join(src, "*.jpg")))
assert len(files) == 50, len(files)
imgs = []
for f in files:
    im = Image.open(f).convert("RGB")
    w, h = im.size
    imgs.append(np.asarray(im.resize((w // 4, h // 4), Image.BOX), np.uint8))
arr = np.stack(imgs)
np.savez_compressed(out, images=arr, factor_vs_state=np.float32(4.0))
print(out, arr.shape, os.path.getsize(out) / 1e6, "MB")

# ---- factor 2 (confs/wanjinyou.yaml: dataset.factor 2 -> 960x540): the reference's own images_2 JPEG files, byte for byte,
# packed into one archive (decoded with PIL where they are used: f2-nerf_amd/fox_data.py).  9.6 MB of data, no source.
out2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fox_images_f2_jpeg.npz")
blobs = [np.frombuffer(open(f, "rb").read(), np.uint8) for f in files]
offsets = np.cumsum([0] + [len(b) for b in blobs]).astype(np.int64)
np.savez(out2, jpeg=np.concatenate(blobs), offsets=offsets, names=np.array([os.path.basename(f) for f in files]),
         factor_vs_state=np.float32(1.0))
print(out2, len(blobs), os.path.getsize(out2) / 1e6, "MB")
