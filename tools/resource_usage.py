#!/usr/bin/env python3
"""Developer tool: per-kernel register / spill / occupancy report of the robot translation units
(hipcc -Rpass-analysis=kernel-resource-usage).  usage: tools/resource_usage.py [robot ...] [-Dflag ...]"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, HIPFLAGS  # noqa: E402


EXTRA = [a for a in sys.argv[1:] if a.startswith("-")]


def report(robot):
    src = os.path.join(CSRC, "gen", f"tu_{robot}.hip")
    r = subprocess.run(["hipcc", *HIPFLAGS, *EXTRA, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                       capture_output=True, text=True)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|"
                      r"ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = {"name": re.sub(r"^_ZN3vmv\d+[a-z0-9]+?\d+|E[PKjmS_\d\w]*$", "", m.group(2))}
            rows.append(cur)
        else:
            cur[m.group(1)] = m.group(2)
    return robot, rows, r.returncode


if __name__ == "__main__":
    robots = [a for a in sys.argv[1:] if not a.startswith("-")] or ["panda", "ur5", "fetch", "baxter"]
    with ThreadPoolExecutor(4) as ex:
        for robot, rows, rc in ex.map(report, robots):
            print(f"== {robot} (hipcc rc {rc})")
            for k in rows:
                print(f"  {k['name']:34s} vgpr {k.get('VGPRs'):>4s} agpr {k.get('AGPRs'):>3s} sspill {k.get('SGPRs Spill'):>4s} "
                      f"vspill {k.get('VGPRs Spill'):>3s} scratch {k.get('ScratchSize [bytes/lane]'):>4s} "
                      f"lds {k.get('LDS Size [bytes/block]'):>6s} occ {k.get('Occupancy [waves/SIMD]')}")
