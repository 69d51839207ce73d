"""Build the gfx950 shared library (libdnaf_hip.so) in-tree with hipcc."""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libdnaf_hip.so")
SOURCES = ["dnaf_api.hip"]
STAMP = LIB + ".srchash"


def dep_files():
    """Every file the library is compiled from: all of csrc/*.hip, csrc/*.h and the public header."""
    import glob

    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    files.append(os.path.normpath(os.path.join(CSRC, "..", "..", "include", "dnaf.h")))
    return files


def source_hash():
    """SHA-256 over the contents of dep_files(): what the built library is stamped with."""
    import hashlib

    h = hashlib.sha256()
    for f in dep_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def is_stale():
    """True when the .so is missing or was built from other sources than the ones in the tree (content hash,
    so a library that travelled to another box with fresh mtimes is still recognised)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build_library(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
