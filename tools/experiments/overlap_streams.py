#!/usr/bin/env python3
"""Experiment: do the environment and self-collision kernels of one step gain from running concurrently on two streams
(their tails overlap, the two kernels have different resource profiles)?  Prints ms per step for: sequential on one
stream (what vmv_validate_batch does), and forked onto two streams with an event join + AND of the two word arrays."""
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
p = vamp.panda
q = torch.empty((n, 7), dtype=torch.float32, device=dev)
s0 = torch.cuda.current_stream(dev)
check(lib.vmv_fill_uniform_configs(p._id, ctypes.c_void_p(q.data_ptr()), n, 1234, ctypes.c_void_p(s0.cuda_stream)), "fill")
words = (n + 63) // 64
b_env = torch.zeros(words, dtype=torch.int64, device=dev)
b_self = torch.zeros(words, dtype=torch.int64, device=dev)
h = env.handle()
qp = ctypes.c_void_p(q.data_ptr())
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def sequential():
    sp = ctypes.c_void_p(s1.cuda_stream)
    check(lib.vmv_validate_batch_env(p._id, h, qp, n, ctypes.c_void_p(b_env.data_ptr()), sp), "env")
    check(lib.vmv_validate_batch_self(p._id, qp, n, ctypes.c_void_p(b_env.data_ptr()), sp), "self")


def forked():
    with torch.cuda.stream(s1):
        b_self.fill_(-1)
    fork = s1.record_event()
    s2.wait_event(fork)
    check(lib.vmv_validate_batch_env(p._id, h, qp, n, ctypes.c_void_p(b_env.data_ptr()), ctypes.c_void_p(s1.cuda_stream)), "env")
    check(lib.vmv_validate_batch_self(p._id, qp, n, ctypes.c_void_p(b_self.data_ptr()), ctypes.c_void_p(s2.cuda_stream)), "self")
    s1.wait_event(s2.record_event())
    with torch.cuda.stream(s1):
        b_env.bitwise_and_(b_self)


def timed(fn, iters=200, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


sequential()
torch.cuda.synchronize()
ref = b_env.clone()
forked()
torch.cuda.synchronize()
print("same words:", bool(torch.equal(ref, b_env)))
for name, fn in (("sequential", sequential), ("forked", forked), ("sequential", sequential), ("forked", forked)):
    print(f"{name:10s} {timed(fn):.4f} ms per step", flush=True)
