#!/usr/bin/env python3
"""bench.py — configuration collision checks/sec (Panda, 64-primitive environment) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]        (N > 1: starts its N ranks itself, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (vmv_validate_batch: FK + environment + self-collision, one validity bit per
configuration) over one batch of synthetic Panda configurations already resident in HBM (BASELINE.json configs[1]).

  weak scaling (the line's `value`): 1,048,576 configurations PER GPU per step; with N > 1 every step ends with the
      RCCL all-gather of the packed validity bitmasks (the path's only exchange step), overlapped with the next step.
  strong scaling (the line's `strong` object): ONE 1,048,576-configuration batch per step, cut into 64-aligned
      contiguous shards (vamp_mvt_amd.sharding.shard_range), all-gathered the same way.  At N = 1 both are the same job.

Prints ONE JSON line (rank 0).  `roofline` is computed from HIP events recorded on the launch stream around each
kernel launch inside the timed region; its `valu` / `traffic` parts quote the rocprofv3 PMC summary under profiles/
only if that summary was measured on this very build (source hash); `cpu_baseline` times the CPU oracle (test
infrastructure) on a bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS_PER_GPU = 1 << 20
ALGO_BYTES_PER_CHECK = 4 * 7 + 1.0 / 8.0  # 28 B of joint values read + 1 bit written
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMD = 256 * 4           # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_INST = 2   # MI355X_MICROARCH.md: wave64 v_fma/v_add on the SIMD-32 = 2 cycles
CLOCK_GHZ = 2.4


def _load_by_path(name, path):
    """a module of the package WITHOUT importing the package (which loads the HIP library)"""
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(env_spec, n_sample, threads, target_seconds=10.0):
    """Times the CPU restatement on a bounded sample of the bench workload.  Checker only — not the product.
    Three figures: the scalar C port with one configuration replicated over the rake (what the reference's
    `validate` delivers per configuration), and the AVX2 build with 8 DISTINCT configurations per rake (what the
    reference's SIMD layer can deliver on a batch), single core and all cores."""
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

    def rate(fn, n, budget):
        fn(q[:2048])
        t0 = time.perf_counter()
        fn(q[:n])
        once = time.perf_counter() - t0
        reps = max(1, min(200, int(budget / max(once, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(q[:n])
        return n * reps / (time.perf_counter() - t0)

    one = min(n_sample, 1 << 18)
    scalar_1 = rate(lambda x: o.validate_batch(rid, env, x, threads=1), one, target_seconds * 0.15)
    scalar_all = rate(lambda x: o.validate_batch(rid, env, x, threads=threads), n_sample, target_seconds * 0.25)
    out = dict(value=scalar_all, unit="checks/s", cores=threads, kind="port",
               sample=f"{n_sample} uniform Panda configs vs the same 64-primitive shell env, repeated for about "
                      f"{target_seconds:.0f} s of wall time in all; scalar C port, one configuration per rake "
                      f"(replicated lanes, as the reference's validate): {scalar_1:.3e} checks/s on 1 thread, "
                      f"{scalar_all:.3e} on {threads} threads",
               single_thread_value=scalar_1, cpu_model=_cpu_model())
    if hasattr(o, "validate_batch_avx2") and o.has_avx2():
        want = o.validate_batch(rid, env, q[:one], threads=threads)
        got = o.validate_batch_avx2(rid, env, q[:one], threads=threads)
        assert np.array_equal(want, got), "AVX2 restatement differs from the scalar port"
        avx_1 = rate(lambda x: o.validate_batch_avx2(rid, env, x, threads=1), one, target_seconds * 0.2)
        avx_all = rate(lambda x: o.validate_batch_avx2(rid, env, x, threads=threads), n_sample, target_seconds * 0.4)
        out.update(value=avx_all, kind="port",
                   avx2={"single_thread_value": avx_1, "value": avx_all, "cores": threads,
                         "note": "rake of 8 DISTINCT configurations per AVX2 vector (per-lane validity masks), the "
                                 "reference's vector/avx.hh shape; bit-identical answers to the scalar port"})
        out["sample"] += f"; AVX2 rake-of-8 build: {avx_1:.3e} checks/s on 1 thread, {avx_all:.3e} on {threads} threads"
    return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_profile():
    """profiles/r03_pmc.json (tools/profile.sh + tools/make_pmc_profile.py) if it was measured on THIS build"""
    path = os.path.join(ROOT, "profiles", "r03_pmc.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        current = _load_by_path("source_hash", os.path.join(ROOT, "tools", "source_hash.py")).source_hash()
        if prof.get("source_hash") != current:
            return None, f"profiles/r03_pmc.json was measured on build {prof.get('source_hash')}, this is {current}"
        return prof, None
    except (OSError, ValueError) as e:
        return None, f"no PMC summary: {e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # 200 x 0.3 ms: long enough for the clocks to settle
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--configs", type=int, default=CONFIGS_PER_GPU, help="configurations per GPU per step (weak) and "
                    "per job per step (strong)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-probe", action="store_true", help="skip the timing of the N = 2, 4, 8 shard sizes on this GPU")
    ap.add_argument("--no-two-streams", action="store_true",
                    help="skip the informational two-stream leg (profiling runs: its overlapping launches would mix into the per-kernel averages)")
    ap.add_argument("--cpu-sample", type=int, default=1 << 21)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: run the N-rank control flow with every rank on cuda:0 and the exchange "
                         "over gloo (RCCL refuses two ranks on one device); never used for reported numbers")
    ap.add_argument("--env", default="shell64", choices=["shell64", "empty", "cage"],
                    help="diagnostics only: the metric is defined on shell64 (BASELINE config 2)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way the driver starts the one-GPU bench: become the launcher.  Nothing in this process has touched
        # the GPU yet (no HIP call, no torch.cuda call, the package is not even imported): the ranks are fresh children.
        sharding = _load_by_path("vmv_sharding", os.path.join(ROOT, "vamp_mvt_amd", "sharding.py"))
        sys.exit(sharding.respawn_one_rank_per_gpu(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
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
    from vamp_mvt_amd.sharding import shard_range
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
    stream = torch.cuda.current_stream(dev)
    sptr = ctypes.c_void_p(stream.cuda_stream)
    h_env = env.handle()
    cpu_exchange = args.rehearse_on_one_gpu

    def run(q, n_local, per_rank_words, steps, warmup, events):
        """`steps` timed steps of validate + exchange over q[0 : n_local]; returns (seconds (max over ranks), last words)"""
        words = (n_local + 63) // 64
        # two result buffers: the all-gather of step i (RCCL's own stream) overlaps the kernels of step i + 1
        bits_buf = [torch.zeros(max(per_rank_words, words, 1), dtype=torch.int64, device=dev) for _ in range(2)]
        gathered = [torch.zeros(per_rank_words * world, dtype=torch.int64, device=dev) for _ in range(2)] if world > 1 else None
        pending = [None, None]
        qp = ctypes.c_void_p(q.data_ptr())
        count = [0]

        def step(evs=None):
            k = count[0] % 2
            count[0] += 1
            if pending[k] is not None:
                pending[k].wait()  # the exchange that still reads this buffer (stream-side wait)
                pending[k] = None
            bp = ctypes.c_void_p(bits_buf[k].data_ptr())
            if n_local > 0:
                # vmv_validate_batch == its two kernels back to back on the launch stream; launched through the two stage
                # entry points so that HIP events on that stream can bracket each kernel
                if evs is not None:
                    evs[0].record(stream)
                check(lib.vmv_validate_batch_env(panda._id, h_env, qp, n_local, bp, sptr), "vmv_validate_batch_env")
                if evs is not None:
                    evs[1].record(stream)
                check(lib.vmv_validate_batch_self(panda._id, qp, n_local, bp, sptr), "vmv_validate_batch_self")
                if evs is not None:
                    evs[2].record(stream)
            if world > 1:
                if cpu_exchange:
                    parts = [torch.empty(per_rank_words, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(parts, bits_buf[k][:per_rank_words].cpu())
                else:
                    pending[k] = dist.all_gather_into_tensor(gathered[k], bits_buf[k][:per_rank_words], async_op=True)

        def drain():
            for k in range(2):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None

        for _ in range(warmup):
            step()
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(events[i] if events is not None else None)
        drain()  # every exchange of the timed steps has completed before the clock stops
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if cpu_exchange else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        last = (count[0] - 1) % 2
        full = gathered[last] if world > 1 and not cpu_exchange else bits_buf[last]
        return elapsed, full

    # ---- weak scaling: n configurations per GPU ----------------------------------------------------------------
    q = torch.empty((n, 7), dtype=torch.float32, device=dev)
    check(lib.vmv_fill_uniform_configs(panda._id, ctypes.c_void_p(q.data_ptr()), n, 1234 + rank, sptr),
          "vmv_fill_uniform_configs")
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    elapsed, words_w = run(q, n, (n + 63) // 64, args.steps, args.warmup, evs)
    kernel_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))  # dominant kernel: environment half
    self_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    valid_frac = None
    if rank == 0:
        host_bits = words_w[: (n + 63) // 64].cpu().numpy().view(np.uint64)
        valid_frac = float(vamp.unpack_bits(host_bits, n).mean())

    # ---- strong scaling: ONE n-configuration batch cut into 64-aligned shards --------------------------------------
    if world > 1:
        check(lib.vmv_fill_uniform_configs(panda._id, ctypes.c_void_p(q.data_ptr()), n, 1234, sptr),
              "vmv_fill_uniform_configs")  # the same batch on every rank; each rank validates its own shard of it
        lo, hi = shard_range(n, rank, world)
        per = ((n + 63) // 64 + world - 1) // world
        shard = q[lo:hi] if hi > lo else q[:0]
        elapsed_s, words_s = run(shard, hi - lo, per, args.steps, max(3, args.warmup // 4), None)
        strong_valid = None
        if rank == 0 and not cpu_exchange:
            strong_valid = float(vamp.unpack_bits(words_s[: (n + 63) // 64].cpu().numpy().view(np.uint64), n).mean())
    else:
        elapsed_s, strong_valid = elapsed, valid_frac

    # ---- shard probe (N = 1 only): the shards the strong-scaling leg gives each GPU at N = 2, 4, 8, timed on this one -----
    shard_probe = None
    if world == 1 and not args.no_shard_probe:
        shard_probe = {"note": "pipelined steps of vmv_validate_batch over the first n/N configurations of the batch: what "
                               "ONE rank of the N-GPU strong-scaling leg executes per step (no exchange).  ceiling = the "
                               "1-GPU step time / the shard's step time = the most N GPUs can gain on the 1M job before "
                               "any exchange cost; no N > 1 run has been measured on hardware", "shards": {}}
        for parts in (2, 4, 8):
            lo, hi = shard_range(n, 0, parts)
            m = hi - lo
            dt_shard, _ = run(q[:m], m, (m + 63) // 64, args.steps, max(3, args.warmup // 4), None)
            shard_probe["shards"][str(parts)] = {"configs": m, "ms_per_step": dt_shard / args.steps * 1e3,
                                                 "ceiling": (elapsed / args.steps) / (dt_shard / args.steps)}

    # ---- informational: independent batches on two streams (N = 1 only; never `value`) ------------------------------
    two_streams = None
    if world == 1 and not args.no_two_streams:
        # the same job with consecutive steps submitted on alternating streams (each step = one vmv_validate_batch call
        # over the whole batch, its own result buffer): the next step's kernels fill the tail of the previous step's
        side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        bufs = [torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev) for _ in range(2)]
        qp2 = ctypes.c_void_p(q.data_ptr())

        def step2(k):
            check(lib.vmv_validate_batch(panda._id, h_env, qp2, n, ctypes.c_void_p(bufs[k % 2].data_ptr()),
                                         ctypes.c_void_p(side[k % 2].cuda_stream)), "vmv_validate_batch")
        for st in side:
            st.wait_stream(stream)
        for k in range(args.warmup):
            step2(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step2(k)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        same = bool(torch.equal(bufs[0], words_w[: (n + 63) // 64]) and torch.equal(bufs[1], words_w[: (n + 63) // 64]))
        two_streams = {"value": float(n) * args.steps / dt2, "unit": "checks/s", "ms_per_step": dt2 / args.steps * 1e3,
                       "same_words_as_the_timed_run": same,
                       "note": "informational, not `value`: the same steps submitted on two alternating streams, as a caller "
                               "with independent batches can; kernels of neighbouring steps overlap, so per-launch durations "
                               "(and the roofline above) do not apply to this figure"}

    if rank == 0:
        value = float(n) * world * args.steps / elapsed
        achieved = ALGO_BYTES_PER_CHECK * n / (kernel_ms * 1e-3) / 1e9
        prof, why = pmc_profile()
        traffic, valu, traffic_step = None, None, None
        if prof is not None:
            ke, ks = prof["kernels"]["validate_env_kernel"], prof["kernels"]["validate_self_kernel"]
            traffic = ke["hbm_bytes"]
            traffic_step = {"bytes": ke["hbm_bytes"] + ks["hbm_bytes"],
                            "x_algorithmic": (ke["hbm_bytes"] + ks["hbm_bytes"]) / (ALGO_BYTES_PER_CHECK * prof["configs"]),
                            "note": "both kernels of one step (the self-collision kernel reads the configurations again)"}

            def valu_of(k, ms):
                cycles = k["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / N_SIMD
                return {"wave_instructions_per_launch": k["SQ_INSTS_VALU"], "per_wave": k["SQ_INSTS_VALU"] / k["SQ_WAVES"],
                        "issue_cycles_per_simd": cycles, "issue_time_ms_at_2.4GHz": cycles / (CLOCK_GHZ * 1e6),
                        "frac_of_kernel_time": cycles / (CLOCK_GHZ * 1e6) / ms,
                        "lane_utilisation": k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64.0)}
            valu = {"peak": "1 wave64 VALU instruction per 2 cycles per SIMD, 1,024 SIMDs at 2.4 GHz (157 TFLOP/s fp32 vector)",
                    "validate_env_kernel": valu_of(ke, kernel_ms), "validate_self_kernel": valu_of(ks, self_ms),
                    "source": prof.get("source", "profiles/r03_pmc.json")}
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
            "strong": {"value": float(n) * args.steps / elapsed_s, "unit": "checks/s", "ms_per_step": elapsed_s / args.steps * 1e3,
                       "configs_per_job": n, "valid_fraction": strong_valid,
                       "note": "ONE 1,048,576-configuration batch per step in 64-aligned contiguous shards over the GPUs, "
                               "validity words all-gathered (at 1 GPU: the same job as the weak line)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": lib.vmv_kernel_name(panda._id, b"validate_batch").decode(),
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_CHECK * n,
                         "other_kernels_ms": {"validate_self_kernel": self_ms},
                         "traffic_per_step": traffic_step, "valu": valu, "pmc_note": why,
                         "note": "the path does ~10^4 fp32 operations per 28 B: it is VALU-issue bound, the HBM fraction "
                                 "is reported because the metric asks for it (DESIGN.md §5)"},
            "kernel_checks_per_s": n / ((kernel_ms + self_ms) * 1e-3),
        }
        if shard_probe is not None:
            out["shard_probe"] = shard_probe
        if two_streams is not None:
            out["two_streams"] = two_streams
        if world == 1 and not args.no_cpu_baseline:
            threads = max(1, min(len(os.sched_getaffinity(0)), 16))  # the box's CPU share for one GPU
            out["cpu_baseline"] = cpu_baseline(spec, args.cpu_sample, threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
