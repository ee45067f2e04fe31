#!/usr/bin/env python3
"""Developer tool: build the C-ABI library once more with extra hipcc flags / defines into variants/<name>/, for A/B
runs on the GPU box (VMV_LIBRARY=variants/<name>/libvamp_mvt_amd.so python bench.py ...).
usage: tools/build_variant.py NAME [hipcc flags ...]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from __graft_entry__ import HIPFLAGS, _sources  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "variants", name)
os.makedirs(out, exist_ok=True)


def cc(src):
    obj = os.path.join(out, os.path.basename(src) + ".o")
    subprocess.check_call(["hipcc", *HIPFLAGS, *extra, "-c", src, "-o", obj])
    return obj


with ThreadPoolExecutor(7) as ex:
    objs = list(ex.map(cc, _sources()))
lib = os.path.join(out, "libvamp_mvt_amd.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
print(lib)
