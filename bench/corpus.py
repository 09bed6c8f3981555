#!/usr/bin/env python3
"""Corpus benchmark through the command lines, after the reference's bench/benchmark-small-corpus.py:39-102:
every image of a directory goes through `cfelics -i .. -o ..` (wall clock over the whole loop, size of the output
directory as `du -m -s` reports it), then every .fel file through `dfelics` back to TIFF, timed the same way.  The
reference compares with ImageMagick `convert` (PNG, QOI) and `cwebp`, none of which exist in this image; here the
comparison column is PNG written by this repository's own writer (imgconv, zlib level 6).  Unlike the reference's
script this one also checks the round trip (decoded pixels == input pixels) and prints a JSON summary.

    python bench/corpus.py [--corpus DIR] [--out DIR] [--device N]

Default corpus: tests/golden (eight files of the reference's image-suite); the reference's own corpus is
bench/tiff_files in its repository.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
from time import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "felics_amd", "_build")


def get_disk_usage(path):
    """Size of the directory on disk in MB, as the reference measures it (`du -m -s`)."""
    out = subprocess.check_output(["du", "-m", "-s", path])
    return int(out.split()[0].decode())


def bytes_in(path):
    return sum(os.path.getsize(os.path.join(path, f)) for f in os.listdir(path))


def run_all(files, src_dir, dst_dir, ext, command):
    start = time()
    for f in files:
        name, _ = os.path.splitext(f)
        cmd = command(os.path.join(src_dir, f), os.path.join(dst_dir, name + ext))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("%s failed: %s" % (" ".join(cmd), r.stdout + r.stderr))
    return time() - start


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--corpus", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--out", default=None, help="working directory (default: a temporary one)")
    ap.add_argument("--device", default="0")
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    from felics_amd import build

    build.build()
    files = sorted(f for f in os.listdir(args.corpus) if f.lower().endswith((".tiff", ".tif", ".png", ".pgm", ".ppm")))
    if not files:
        raise SystemExit("no images in " + args.corpus)
    work = args.out or tempfile.mkdtemp(prefix="felics_corpus_")
    dirs = {k: os.path.join(work, k) for k in ("to_felics", "from_felics", "to_png", "from_png")}
    for d in dirs.values():
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
    cfelics, dfelics, imgconv = (os.path.join(BUILD, t) for t in ("cfelics", "dfelics", "imgconv"))
    print("Benchmarking compression for: .fel")
    t_fel = run_all(files, args.corpus, dirs["to_felics"], ".fel", lambda i, o: [cfelics, "-i", i, "-o", o, "--device", args.device])
    print("Benchmarking compression for: .png")
    t_png = run_all(files, args.corpus, dirs["to_png"], ".png", lambda i, o: [imgconv, "-i", i, "-o", o])
    fel = sorted(os.listdir(dirs["to_felics"]))
    print("Benchmarking decompression for: .fel")
    t_dfel = run_all(fel, dirs["to_felics"], dirs["from_felics"], ".tiff", lambda i, o: [dfelics, "-i", i, "-o", o])
    print("Benchmarking decompression for: .png")
    t_dpng = run_all(sorted(os.listdir(dirs["to_png"])), dirs["to_png"], dirs["from_png"], ".tiff", lambda i, o: [imgconv, "-i", i, "-o", o])
    # round trip: the decoded TIFF holds the input's pixels (PNM of both through the same reader)
    import numpy as np
    from PIL import Image

    for f in files:
        name, _ = os.path.splitext(f)
        a = np.array(Image.open(os.path.join(args.corpus, f)))
        b = np.array(Image.open(os.path.join(dirs["from_felics"], name + ".tiff")))
        if a.shape != b.shape or not (a == b).all():
            raise SystemExit("round trip changed " + f)
    usages = {".fel": get_disk_usage(dirs["to_felics"]), ".png": get_disk_usage(dirs["to_png"])}
    print("Compression times: ", [(".fel", t_fel), (".png", t_png)])
    print("Memory usages: ", list(usages.items()))
    print("Decompression times: ", [(".fel", t_dfel), (".png", t_dpng)])
    in_bytes = sum(os.path.getsize(os.path.join(args.corpus, f)) for f in files)
    print(json.dumps({"files": len(files), "input_bytes": in_bytes, "felics_bytes": bytes_in(dirs["to_felics"]),
                      "png_bytes": bytes_in(dirs["to_png"]), "ratio_felics": round(in_bytes / bytes_in(dirs["to_felics"]), 4),
                      "ratio_png": round(in_bytes / bytes_in(dirs["to_png"]), 4), "seconds": {"cfelics": round(t_fel, 3), "dfelics": round(t_dfel, 3),
                                                                                              "to_png": round(t_png, 3), "from_png": round(t_dpng, 3)},
                      "round_trip": "ok", "note": "one process (and one GPU context) per file, like the reference's script"}))
    if not args.out:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
