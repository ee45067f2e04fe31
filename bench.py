#!/usr/bin/env python3
"""bench.py — configuration collision checks/sec (Panda, 64-primitive environment) on N MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (vmv_validate_batch: FK + environment + self-collision, one validity bit
per configuration) over one batch of 1,048,576 synthetic Panda configurations PER GPU, already resident in HBM
(BASELINE.json configs[1]); with N > 1 every step ends with the RCCL all-gather of the packed validity
bitmasks (the path's only exchange step), so every rank holds the whole job's bitmask.  Weak scaling.

Prints ONE JSON line (rank 0).  `roofline` is computed from HIP events recorded on the launch stream around
each kernel launch inside the timed region; `cpu_baseline` times the CPU oracle (a scalar C port of the
reference path; test infrastructure) on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS_PER_GPU = 1 << 20
ALGO_BYTES_PER_CHECK = 4 * 7 + 1.0 / 8.0  # 28 B of joint values read + 1 bit written
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(env_spec, n_sample, threads, target_seconds=12.0):
    """Times the oracle (CPU port) on a bounded sample of the bench workload.  Checker only — not the product."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from envs import build_oracle_env
    from oracle_lib import Oracle

    o = Oracle()
    rid = o.robot("panda")
    lo, span = o.bounds(rid)
    rng = np.random.default_rng(0)
    q = (lo + span * rng.random((n_sample, 7), dtype=np.float32)).astype(np.float32)
    env = build_oracle_env(o, env_spec)
    o.validate_batch(rid, env, q[:2048], threads=threads)  # warm
    t0 = time.perf_counter()
    valid = o.validate_batch(rid, env, q, threads=threads)
    once = time.perf_counter() - t0
    reps = max(1, min(400, int(target_seconds / max(once, 1e-4))))  # bounded: ~target_seconds of CPU wall time
    t0 = time.perf_counter()
    for _ in range(reps):
        o.validate_batch(rid, env, q, threads=threads)
    dt = time.perf_counter() - t0
    one = min(n_sample, 1 << 18)
    t0 = time.perf_counter()
    o.validate_batch(rid, env, q[:one], threads=1)
    dt1 = time.perf_counter() - t0
    return dict(value=n_sample * reps / dt, unit="checks/s", cores=threads, kind="port",
                sample=f"{reps} passes over {n_sample} uniform Panda configs vs the same 64-primitive shell env "
                       f"({dt:.1f} s wall on {threads} threads, {100.0 * float(valid.mean()):.1f}% valid); "
                       f"1 thread on {one} configs: {one / dt1:.3e} checks/s",
                single_thread_value=one / dt1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 200 x 0.33 ms: long enough for the clocks to settle
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--configs", type=int, default=CONFIGS_PER_GPU, help="configurations per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1 << 21)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: run the N-rank control flow with every rank on cuda:0 and the exchange "
                         "over gloo (RCCL refuses two ranks on one device); never used for reported numbers")
    ap.add_argument("--env", default="shell64", choices=["shell64", "empty", "cage"],
                    help="diagnostics only: the metric is defined on shell64 (BASELINE config 2)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL

    import vamp_mvt_amd as vamp
    from vamp_mvt_amd._lib import check, lib
    from vamp_mvt_amd.workloads import environment_from_spec, shell_spec

    vamp.set_device(local_rank)
    spec = shell_spec(seed=0)  # 32 spheres + 32 z-aligned cuboids (BASELINE config 2)
    if args.env == "empty":
        spec = []
    elif args.env == "cage":
        from vamp_mvt_amd.workloads import SPHERE_CAGE
        spec = [("sphere", np.array([*c, 0.2], np.float32)) for c in SPHERE_CAGE]
    env = environment_from_spec(spec)
    panda = vamp.panda
    n = args.configs
    words = (n + 63) // 64

    q = torch.empty((n, 7), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    check(lib.vmv_fill_uniform_configs(panda._id, ctypes.c_void_p(q.data_ptr()), n, 1234 + rank, sptr),
          "vmv_fill_uniform_configs")
    # two result buffers: the all-gather of step i (RCCL's own stream) overlaps the kernels of step i + 1
    bits_buf = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(2)]
    gathered = [torch.zeros(words * world, dtype=torch.int64, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    bits = bits_buf[0]

    h_env = env.handle()
    qp = ctypes.c_void_p(q.data_ptr())
    step_no = [0]

    def step(evs=None):
        # vmv_validate_batch == its two kernels back to back on the launch stream; launched through the two
        # stage entry points here so that HIP events on that stream can bracket each kernel
        k = step_no[0] % 2
        step_no[0] += 1
        if pending[k] is not None:
            pending[k].wait()  # the exchange that still reads this buffer (stream-side wait, the host does not block)
            pending[k] = None
        bp = ctypes.c_void_p(bits_buf[k].data_ptr())
        if evs is not None:
            evs[0].record(stream)
        check(lib.vmv_validate_batch_env(panda._id, h_env, qp, n, bp, sptr), "vmv_validate_batch_env")
        if evs is not None:
            evs[1].record(stream)
        check(lib.vmv_validate_batch_self(panda._id, qp, n, bp, sptr), "vmv_validate_batch_self")
        if evs is not None:
            evs[2].record(stream)
        if world > 1:
            if args.rehearse_on_one_gpu:
                parts = [torch.empty(words, dtype=torch.int64) for _ in range(world)]
                dist.all_gather(parts, bits_buf[k].cpu())
            else:
                pending[k] = dist.all_gather_into_tensor(gathered[k], bits_buf[k], async_op=True)

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e in evs:
        step(e)
    drain()  # every exchange of the timed steps has completed before the clock stops
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    bits = bits_buf[(step_no[0] - 1) % 2]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))  # dominant kernel: environment half
    self_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))

    valid_frac = None
    if rank == 0:
        host_bits = bits.cpu().numpy().view(np.uint64)
        valid_frac = float(vamp.unpack_bits(host_bits, n).mean())

    if rank == 0:
        total_checks = float(n) * world * args.steps
        value = total_checks / elapsed
        achieved = ALGO_BYTES_PER_CHECK * n / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "config collision checks/sec (Panda, 64-prim env)",
            "value": value,
            "unit": "checks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "Panda 7-DoF, 1,048,576 uniform random configs per GPU vs 64 primitives "
                                   "(32 spheres + 32 z-aligned cuboids on a cylindrical shell), configs[1]",
                       "configs_per_gpu": n, "primitives": len(spec), "env": args.env, "valid_fraction": valid_frac,
                       "exchange": "RCCL all_gather of packed validity bitmasks, overlapped with the next step" if world > 1 else "none",
                       "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": lib.vmv_kernel_name(panda._id, b"validate_batch").decode(),
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CHECK * n,
                         "other_kernels_ms": {"validate_self_kernel": self_ms},
                         "note": "path is fp32-VALU bound (~10^4 flop per 28 B); HBM fraction is reported as the "
                                 "metric asks, see DESIGN.md"},
            "kernel_checks_per_s": n / ((kernel_ms + self_ms) * 1e-3),
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(len(os.sched_getaffinity(0)), 16))  # the box's CPU share for one GPU
            out["cpu_baseline"] = cpu_baseline(spec, args.cpu_sample, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
