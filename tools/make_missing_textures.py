#!/usr/bin/env python3
"""Decode the textures whose .ppm the reference checkout lacks (.MISSING_LARGE_BLOBS) into P6 files.

The reference converts diffuse.jpg -> diffuse.ppm with ffmpeg at load time
(src/materials/texture.rs:22-39); ffmpeg is absent here, so PIL decodes them once, in the build
container, and the results are committed as data under assets/textures/.  Texel parity with the
author's ffmpeg output is unpinned (PIL and ffmpeg differ by a few LSB on JPEGs).
mio_block3/diffuse.jpg is a WebP file with a .jpg name; PIL sniffs the real format.

usage: make_missing_textures.py <reference_root> <repo_root>
"""
import os
import sys

from PIL import Image


def main(ref, repo):
    for name in ("mio_grass", "mio_block3"):
        src = os.path.join(ref, "assets", "textures", name, "diffuse.jpg")
        dst_dir = os.path.join(repo, "assets", "textures", name)
        os.makedirs(dst_dir, exist_ok=True)
        img = Image.open(src).convert("RGB")
        with open(os.path.join(dst_dir, "diffuse.ppm"), "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % img.size)
            f.write(img.tobytes())
        print(name, img.size)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
