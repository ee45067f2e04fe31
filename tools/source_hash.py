#!/usr/bin/env python3
"""sha256 over everything the kernels are built from (device/host sources, generators, robot models, flags):
profiles measured for one build are only quoted by bench.py for that same build."""
import glob
import hashlib
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def source_hash():
    files = sorted(glob.glob(os.path.join(ROOT, "vamp_mvt_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "vamp_mvt_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "vamp_mvt_amd", "csrc", "*.inc")) +
                   glob.glob(os.path.join(ROOT, "vamp_mvt_amd", "robots", "*.json")) +
                   [os.path.join(ROOT, "tools", "gen_hip.py"), os.path.join(ROOT, "tools", "gen_code.py"),
                    os.path.join(ROOT, "include", "vamp_mvt_amd.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
