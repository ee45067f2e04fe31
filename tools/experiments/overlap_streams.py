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


def halves():
    """the batch cut in two, each half's two kernels on its own stream"""
    half = (n // 2) & ~63
    for st, lo, cnt in ((s1, 0, half), (s2, half, n - half)):
        sp = ctypes.c_void_p(st.cuda_stream)
        qq = ctypes.c_void_p(q.data_ptr() + lo * 28)
        bb = ctypes.c_void_p(b_env.data_ptr() + lo // 8)
        check(lib.vmv_validate_batch_env(p._id, h, qq, cnt, bb, sp), "env")
        check(lib.vmv_validate_batch_self(p._id, qq, cnt, bb, sp), "self")


def split(K):
    """the batch cut in K parts, each part's two kernels on its own stream"""
    streams = [torch.cuda.Stream(dev) for _ in range(K)]

    def run():
        step = ((n // K) + 63) & ~63
        lo = 0
        for st in streams:
            cnt = min(step, n - lo)
            if cnt <= 0:
                break
            sp = ctypes.c_void_p(st.cuda_stream)
            qq = ctypes.c_void_p(q.data_ptr() + lo * 28)
            bb = ctypes.c_void_p(b_env.data_ptr() + lo // 8)
            check(lib.vmv_validate_batch_env(p._id, h, qq, cnt, bb, sp), "env")
            check(lib.vmv_validate_batch_self(p._id, qq, cnt, bb, sp), "self")
            lo += cnt
    return run


def forkjoin(K):
    """what a library-internal split costs: the caller's stream forks to K helper streams and joins them again, so
    consecutive steps do not overlap"""
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    inner = None

    def run():
        fork = s1.record_event()
        step = ((n // K) + 63) & ~63
        lo = 0
        for st in streams:
            cnt = min(step, n - lo)
            if cnt <= 0:
                break
            st.wait_event(fork)
            sp = ctypes.c_void_p(st.cuda_stream)
            qq = ctypes.c_void_p(q.data_ptr() + lo * 28)
            bb = ctypes.c_void_p(b_env.data_ptr() + lo // 8)
            check(lib.vmv_validate_batch_env(p._id, h, qq, cnt, bb, sp), "env")
            check(lib.vmv_validate_batch_self(p._id, qq, cnt, bb, sp), "self")
            s1.wait_event(st.record_event())
            lo += cnt
    return run


_alt = [0]
b_alt = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(2)]


def alternate():
    """whole batches, consecutive steps on alternating streams (each step has its own result buffer)"""
    k = _alt[0] % 2
    _alt[0] += 1
    sp = ctypes.c_void_p((s1, s2)[k].cuda_stream)
    check(lib.vmv_validate_batch(p._id, h, qp, n, ctypes.c_void_p(b_alt[k].data_ptr()), sp), "validate")


def pipelined(K):
    """K chunks: the environment kernels queue on one stream, the self-collision kernels on the other, chunk i's
    self-collision kernel waiting for chunk i's environment kernel (so it runs beside chunk i + 1's environment kernel)"""
    def run():
        step = ((n // K) + 63) & ~63
        lo = 0
        while lo < n:
            cnt = min(step, n - lo)
            qq = ctypes.c_void_p(q.data_ptr() + lo * 28)
            bb = ctypes.c_void_p(b_env.data_ptr() + lo // 8)
            check(lib.vmv_validate_batch_env(p._id, h, qq, cnt, bb, ctypes.c_void_p(s1.cuda_stream)), "env")
            s2.wait_event(s1.record_event())
            check(lib.vmv_validate_batch_self(p._id, qq, cnt, bb, ctypes.c_void_p(s2.cuda_stream)), "self")
            lo += cnt
        s1.wait_event(s2.record_event())
    return run


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
for name, fn in (("halves", halves), ("pipelined2", pipelined(2)), ("pipelined4", pipelined(4))):
    b_env.zero_()
    fn()
    torch.cuda.synchronize()
    print(name, "same words:", bool(torch.equal(ref, b_env)))
modes = (("sequential", sequential), ("forked", forked), ("halves", halves), ("split3", split(3)), ("split4", split(4)),
         ("split6", split(6)), ("forkjoin2", forkjoin(2)), ("forkjoin4", forkjoin(4)), ("alternate", alternate), ("sequential", sequential))
if len(sys.argv) > 2:
    modes = [m for m in modes if m[0] in sys.argv[2:]]
for name, fn in modes:
    print(f"{name:10s} {timed(fn):.4f} ms per step", flush=True)
