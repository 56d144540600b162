#!/usr/bin/env python3
"""Build id of libmibc.so / libmibc_dbg.so: sha256 over the names and contents of everything the libraries are compiled from
(dorado_amd/csrc/*.hip, *.h, Makefile, mibc.map and include/mibc.h; sorted by name), first 16 hex digits.  The Makefile bakes it
into the library (mibc_build_id()), tests/conftest.py recomputes it from the tree and refuses a library built from other sources
(built artefacts travel with the tree, nothing else ties the mapped .so to the sources)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    c = os.path.join(ROOT, "dorado_amd", "csrc")
    files = sorted(glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.h")))
    return files + [os.path.join(c, "Makefile"), os.path.join(c, "mibc.map"), os.path.join(ROOT, "include", "mibc.h")]


def build_id() -> str:
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build_id())
