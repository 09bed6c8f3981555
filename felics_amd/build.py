"""Builds the native parts in-tree: felics_amd/_build/libfelics.so (+ cfelics, dfelics).

hipcc cross-compiles gfx950 code without a GPU, so this runs in the build container; the built
files travel to the GPU box with the repository snapshot.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libfelics.so")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def sources():
    inc = os.path.join(os.path.dirname(HERE), "include", "felics.h")
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [inc]


def source_hash():
    """sha256 over the native sources (csrc/* and include/felics.h, by name and content): names a build
    independently of git history, so measurements (profiles/traffic.json) can say which build they belong to."""
    import hashlib

    h = hashlib.sha256()
    for path in sources():
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode() + b"\0")
            h.update(open(path, "rb").read())
    return h.hexdigest()


def build(force=False, quiet=True):
    """make -C csrc all; returns the path of libfelics.so."""
    targets = [LIB, os.path.join(OUT, "cfelics"), os.path.join(OUT, "dfelics"), os.path.join(OUT, "imgconv")]
    if force or any(_stale(t, sources()) for t in targets):
        cmd = ["make", "-C", CSRC, "all"] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return LIB


def ensure_lib():
    """Path of libfelics.so, building it only if it is missing (the GPU box ships it prebuilt).
    FELICS_LIB_PATH names another build of the library (A/B measurements of two builds on one box)."""
    alt = os.environ.get("FELICS_LIB_PATH")
    if alt:
        return alt
    if not os.path.exists(LIB):
        build()
    return LIB
