#!/usr/bin/env python3
"""A/B in one process: vmv_validate_batch with the listed self-collision kernel (VMV_COMPACT=1) vs
the plain two kernels (the default), every robot, the 64-primitive scene.   python tools/experiments/compact_ab.py [n]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import vamp_mvt_amd as vamp  # noqa: E402
from vamp_mvt_amd._lib import check, lib  # noqa: E402
from vamp_mvt_amd.workloads import environment_from_spec, shell_spec  # noqa: E402

vamp.set_device(0)
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = environment_from_spec(shell_spec(0))
h = env.handle()
s0 = torch.cuda.current_stream(dev)
sp = ctypes.c_void_p(s0.cuda_stream)
for name in ("panda", "ur5", "fetch", "baxter"):
    p = getattr(vamp, name)
    q = torch.empty((n, p.dimension()), dtype=torch.float32, device=dev)
    check(lib.vmv_fill_uniform_configs(p._id, ctypes.c_void_p(q.data_ptr()), n, 1234, sp), "fill")
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    out = {}
    for mode in ("plain", "listed", "plain", "listed"):
        if mode == "plain":
            os.environ.pop("VMV_COMPACT", None)
        else:
            os.environ["VMV_COMPACT"] = "1"
        for _ in range(20):
            check(lib.vmv_validate_batch(p._id, h, ctypes.c_void_p(q.data_ptr()), n, ctypes.c_void_p(bits.data_ptr()), sp), "v")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            check(lib.vmv_validate_batch(p._id, h, ctypes.c_void_p(q.data_ptr()), n, ctypes.c_void_p(bits.data_ptr()), sp), "v")
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        out.setdefault(mode, []).append((ms, bits.clone()))
    ws = torch.empty(int(lib.vmv_validate_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    acc = [0.0, 0.0]
    for it in range(220):
        evs[0].record(s0)
        check(lib.vmv_validate_batch_env_ws(p._id, h, ctypes.c_void_p(q.data_ptr()), n, ctypes.c_void_p(bits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), sp), "e")
        evs[1].record(s0)
        check(lib.vmv_validate_batch_self_ws(p._id, ctypes.c_void_p(q.data_ptr()), n, ctypes.c_void_p(bits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), sp), "s")
        evs[2].record(s0)
        torch.cuda.synchronize()
        if it >= 20:
            acc[0] += evs[0].elapsed_time(evs[1]); acc[1] += evs[1].elapsed_time(evs[2])
    t0 = time.perf_counter()
    for _ in range(200):
        check(lib.vmv_validate_batch_env_ws(p._id, h, ctypes.c_void_p(q.data_ptr()), n, ctypes.c_void_p(bits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), sp), "e")
        check(lib.vmv_validate_batch_self_ws(p._id, ctypes.c_void_p(q.data_ptr()), n, ctypes.c_void_p(bits.data_ptr()), ctypes.c_void_p(ws.data_ptr()), sp), "s")
    torch.cuda.synchronize()
    print(f"        caller-provided workspace: {(time.perf_counter() - t0) / 200 * 1e3:.4f} ms per step; env (memset + kernel) {acc[0] / 200:.4f} ms, listed self kernel {acc[1] / 200:.4f} ms")
    same = bool(torch.equal(out["plain"][0][1], out["listed"][0][1]))
    print(f"{name:7s} n={n}: plain {out['plain'][0][0]:.4f} / {out['plain'][1][0]:.4f} ms, listed {out['listed'][0][0]:.4f} / {out['listed'][1][0]:.4f} ms, "
          f"same words {same}, valid {float(vamp.unpack_bits(out['listed'][0][1].cpu().numpy().view('uint64'), n).mean()):.3f}", flush=True)
