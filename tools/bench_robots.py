#!/usr/bin/env python3
"""Diagnostic: configuration checks/s of every robot vs the 64-primitive shell environment (1M configs, one MI355X),
environment and self-collision kernels separately.  Not the headline metric (bench.py)."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd._lib import check, lib  # noqa: E402
from vamp_mvt_amd.workloads import environment_from_spec, shell_spec  # noqa: E402


def main():
    vamp.set_device(0)
    env = environment_from_spec(shell_spec(0))
    h = env.handle()
    n = 1 << 20
    s = torch.cuda.current_stream()
    sp = ctypes.c_void_p(s.cuda_stream)
    for name in sys.argv[1:] or ["panda", "ur5", "fetch", "baxter"]:
        mod = getattr(vamp, name)
        q = torch.empty((n, mod.dimension()), device="cuda")
        bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        check(lib.vmv_fill_uniform_configs(mod._id, ctypes.c_void_p(q.data_ptr()), n, 7, sp), "fill")
        qp, bp = ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(bits.data_ptr())
        steps = 30
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        for i in range(steps + 5):
            e = ev[i - 5] if i >= 5 else None
            if e:
                e[0].record(s)
            check(lib.vmv_validate_batch_env(mod._id, h, qp, n, bp, sp), "env")
            if e:
                e[1].record(s)
            check(lib.vmv_validate_batch_self(mod._id, qp, n, bp, sp), "self")
            if e:
                e[2].record(s)
        torch.cuda.synchronize()
        env_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
        self_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
        valid = float(vamp.unpack_bits(bits.cpu().numpy().view("uint64"), n).mean())
        print(json.dumps({"robot": name, "configs": n, "env_ms": env_ms, "self_ms": self_ms,
                          "checks_per_s": n / ((env_ms + self_ms) * 1e-3), "valid_fraction": valid}), flush=True)


if __name__ == "__main__":
    main()
