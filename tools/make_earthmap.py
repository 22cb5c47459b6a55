"""Decode the reference's earth texture once and store it as a binary PPM.

The reference loads resources/earthmap.jpg through the `image` crate
(src/texture.rs:169-181).  Neither this build's C++ host nor the GPU box has a
JPEG decoder to link against, and JPEG decoders differ by an LSB or two, so the
decode is done once, here, with PIL, and the raw RGB is committed as data:
resources/earthmap.ppm (1024x512, P6).

    python tools/make_earthmap.py /root/reference/resources/earthmap.jpg resources/earthmap.ppm
"""
import sys

import numpy as np
from PIL import Image


def main(src, dst):
    rgb = np.asarray(Image.open(src).convert("RGB"), dtype=np.uint8)
    h, w, _ = rgb.shape
    with open(dst, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(rgb.tobytes())
    print(f"{dst}: {w}x{h}, mean RGB = {rgb.reshape(-1, 3).mean(axis=0)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
