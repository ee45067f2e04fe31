#!/usr/bin/env python3
"""rocprofv3 PMC summary (tools/pmc_summary.py --json) -> profiles/r03_pmc.json, the file bench.py quotes in its
`roofline.valu` / `roofline.traffic` fields IF it was measured on the build being benched (source hash).

HBM bytes per launch follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from separate --pmc passes, are in
KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read (x 2).

usage (on the GPU box, after tools/profile.sh TAG):  python3 tools/make_pmc_profile.py gpurun_out/TAG/pmc.json [configs]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import source_hash  # noqa: E402

src = sys.argv[1]
configs = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
raw = json.load(open(src))
kernels = {}
for name, c in raw.items():
    for key in ("validate_env_kernel", "validate_self_kernel", "validate_kernel", "validate_motion_env_kernel",
                "validate_motion_self_kernel"):
        if name.split("<")[0] == key:
            k = dict(c)
            k["name"] = name
            k["hbm_bytes"] = c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 + c.get("WRITE_SIZE", 0.0) * 1024.0
            kernels[key] = k
out = {"source_hash": source_hash(), "configs": configs, "kernels": kernels,
       "source": f"rocprofv3 --pmc passes over `python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline` ({src}); "
                 "SQ_* are per launch (averaged over the launches of the run), *_CYCLES in quad-cycles; "
                 "hbm_bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024"}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r03_pmc.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, "for build", out["source_hash"], "kernels:", sorted(kernels))
