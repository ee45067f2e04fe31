#!/usr/bin/env python3
"""Developer tool: average rocprofv3 --pmc counter values per kernel over one or more output directories.
usage: tools/pmc_summary.py gpurun_out/pmc_x [gpurun_out/pmc_y ...] [--json out.json]"""
import collections
import csv
import glob
import json
import sys


def main():
    args = sys.argv[1:]
    out_json = None
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    out = {}
    waves = {}
    for d in args:
        for f in glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].split("::")[-1]
                agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
                g = int(r.get("Grid_Size", 0) or 0)
                if g:
                    waves[name] = g // 64
            for (kn, c), v in agg.items():
                out.setdefault(kn, {})[c] = sum(v) / len(v)
    for kn, cs in out.items():
        if "kernel" not in kn or "fill" in kn:
            continue
        w = cs.get("SQ_WAVES", waves.get(kn, 0)) or 1
        print(f"{kn}  (waves {w:.0f})")
        for c, v in sorted(cs.items()):
            print(f"    {c:28s} {v:12.4g}   per wave {v / w:10.1f}")
    if out_json:
        json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
