#!/usr/bin/env python3
"""Regenerates tests/golden/.

The reference is Rust and cannot be run in this pipeline, so the .felics files
written here are SELF-GOLDEN: produced by oracle/ (the CPU restatement) after it
passed every known-answer test of the reference (tests/test_oracle_kat.py).  What
IS pinned by the reference itself is recorded in pins.json:
  * the byte size of house/tree/lena_color_256 (DOC.md:469-477),
  * the folder totals 8 529 509 B and 7 543 288 B (DOC.md:385-396).

Fixture inputs are DATA copied from the reference's image-suite (USC-SIPI test
images): a few small TIFFs, enough to exercise II/MM byte order, 8/16 bit and RGB.
Run from the repo root in the build container: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from felics_amd import synth  # noqa: E402
from tests import oracle_lib  # noqa: E402

SUITE = "/root/reference/image-suite"
COPY = [
    "grayscale/8bit/6.3.09.tiff", "grayscale/8bit/5.1.09.tiff", "grayscale/8bit/6.1.01.tiff",
    "grayscale/16bit/aerial.tiff", "grayscale/16bit/man.tiff",
    "rgb/8bit/house.tiff", "rgb/8bit/tree.tiff", "rgb/8bit/lena_color_256.tif",
    # little-endian multi-strip files (the readers' strip logic; lena_color_256.tif above is one too): RGB in 74 strips without a
    # PlanarConfiguration tag, 16-bit gray of odd size (1081 x 1081) in 3 strips
    "rgb/8bit/mandril_color.tif", "grayscale/16bit/heightmap.tiff",
]
# Originals of the reference's integration corpus at full size (tests/compress.rs:73-103 round-trips all of image-suite/): the
# four 1024 x 1024 gray8 files, two >= 1000 x 1000 gray16 ones, three 1024 x 1024 RGB8 ones, the little-endian multi-strip
# 512 x 512 RGB8 file, and a 512 x 512 file of each kind -> tests/golden/suite/ (data; tests/test_gpu_parity.py encodes each on
# the GPU, singly and in same-shape batches of mixed content)
SUITE_COPY = [
    "grayscale/8bit/3.2.25.tiff", "grayscale/8bit/5.3.01.tiff", "grayscale/8bit/5.3.02.tiff", "grayscale/8bit/7.2.01.tiff",
    "grayscale/16bit/bands.tiff", "grayscale/16bit/octagon.tiff",
    "rgb/8bit/2.2.01.tiff", "rgb/8bit/2.2.02.tiff", "rgb/8bit/2.2.03.tiff", "rgb/8bit/lena_color_512.tif",
    "grayscale/8bit/5.2.08.tiff", "grayscale/16bit/boat.tiff", "rgb/8bit/2.1.01.tiff",
]
DOC_SIZE_PINS = {"house.tiff": 105741, "tree.tiff": 122246, "lena_color_256.tif": 110707, "mandril_color.tif": 617524}


def main():
    o = oracle_lib.load()
    pins = {"doc_size_pins": DOC_SIZE_PINS, "files": {}, "synthetic": {}}
    for rel in COPY:
        name = os.path.basename(rel)
        dst = os.path.join(HERE, name)
        shutil.copyfile(os.path.join(SUITE, rel), dst)
        os.chmod(dst, 0o644)
        img = np.array(Image.open(dst))
        data = o.compress(img)
        with open(os.path.join(HERE, name + ".felics"), "wb") as f:
            f.write(data)
        pins["files"][name] = {"shape": list(img.shape), "dtype": str(img.dtype), "size": len(data),
                               "sha256": hashlib.sha256(data).hexdigest()}
    os.makedirs(os.path.join(HERE, "suite"), exist_ok=True)
    pins["suite"] = {}
    for rel in SUITE_COPY:
        name = os.path.basename(rel)
        dst = os.path.join(HERE, "suite", name)
        shutil.copyfile(os.path.join(SUITE, rel), dst)
        os.chmod(dst, 0o644)
        img = np.array(Image.open(dst))
        data = o.compress(img)
        assert (o.decompress(data) == img).all(), name
        with open(dst + ".felics", "wb") as f:
            f.write(data)
        pins["suite"][name] = {"shape": list(img.shape), "dtype": str(img.dtype), "size": len(data),
                               "sha256": hashlib.sha256(data).hexdigest(), "from": "image-suite/" + rel}
    # hand-derived vector of SURVEY.md §8(c) (derived by reading the reference, not by running it)
    pins["hand_vector"] = {
        "pixels": [[10, 12, 11], [13, 9, 12]],
        "hex": "464c4353000000000003000000020000000a0000000c900090",
    }
    # synthetic frames: sha of the oracle stream for a few shapes/kinds
    for kind in ("S1", "S2", "S3"):
        for (w, h) in ((64, 48), (333, 77), (1024, 256)):
            img = synth.gray8(w, h, 0, kind)
            data = o.compress(img)
            pins["synthetic"]["gray8_%s_%dx%d" % (kind, w, h)] = {
                "size": len(data), "sha256": hashlib.sha256(data).hexdigest()}
    for (w, h) in ((64, 48), (333, 77)):
        data = o.compress(synth.rgb8(w, h, 0))
        pins["synthetic"]["rgb8_%dx%d" % (w, h)] = {"size": len(data), "sha256": hashlib.sha256(data).hexdigest()}
        data = o.compress(synth.gray16(w, h, 0))
        pins["synthetic"]["gray16_%dx%d" % (w, h)] = {"size": len(data), "sha256": hashlib.sha256(data).hexdigest()}
    with open(os.path.join(HERE, "pins.json"), "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
    print("wrote", len(pins["files"]), "fixtures and", len(pins["suite"]), "full-size suite originals")


if __name__ == "__main__":
    main()
