#!/usr/bin/env python3
"""The other BASELINE.json configurations (not the headline metric; that is bench.py), on 1..N MI355X.

    python tools/bench_configs.py [--gpus N] [config2 config3 config4 config5 ...]

  config2  Panda, 1M configs vs 64 primitives                            (same as bench.py, for reference)
  config3  Fetch 8-DoF, 1M configs vs a 10k-point CAPT cloud
  config4  UR5 PRM-roadmap shaped: 1M edge validations vs 64 primitives, sharded over the GPUs, RCCL all-gather;
           endpoints = valid Halton samples x valid neighbours at U[0.2, 1.5] rad
  config5  Baxter 14-DoF FCIT*-shaped batch edge check: 262,144 edges, 32 primitives + a 10k-point CAPT cloud;
           edges of the 8-nearest-neighbour graph over valid Halton samples
  config4_uniform_starts / config5_uniform_starts: the generator of rounds 1-2 (uniform, mostly invalid starts), kept as a
           second labelled line

With N > 1 (one process per GPU; started without a launcher the script starts its own ranks) ONE batch is cut into
64-aligned contiguous shards (vamp_mvt_amd.sharding.validate_batch_sharded): every rank builds the same batch from the
same seed, validates its shard against its own copy of the environment and the packed validity words are all-gathered
(strong scaling).  One JSON line per configuration from rank 0: whole-job units/s with the time taken as the MAX over
ranks, between barriers."""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["config2", "config3", "config4", "config5"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the BASELINE batch sizes (smoke runs)")
    ap.add_argument("--shard-probe", action="store_true", help="1 GPU: also time the N = 2, 4, 8 shards of the batch")
    ap.add_argument("--stats", action="store_true", help="edges: also report rakes per edge / rakes walked per edge")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # become the launcher BEFORE anything touches the GPU (the package is not imported yet)
        spec = importlib.util.spec_from_file_location("vmv_sharding", os.path.join(ROOT, "vamp_mvt_amd", "sharding.py"))
        sharding = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(sharding)
        sys.exit(sharding.respawn_one_rank_per_gpu(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import vamp_mvt_amd as vamp
    from vamp_mvt_amd.sharding import shard_range, validate_batch_sharded
    from vamp_mvt_amd.workloads import (POINT_RADIUS, RADII, environment_from_spec, knn_shaped_edges, mixed_spec,
                                        prm_shaped_edges, shell_cloud, shell_spec)

    vamp.set_device(local_rank)

    def uniform(mod, n, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        lo = torch.from_numpy(mod.lower_bounds()).to(dev)
        hi = torch.from_numpy(mod.upper_bounds()).to(dev)
        return (lo + (hi - lo) * torch.rand((n, mod.dimension()), generator=g, device=dev)).contiguous()

    def edges(mod, n, seed, dmin, dmax):
        a = uniform(mod, n, seed)
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        d = torch.randn((n, mod.dimension()), generator=g, device=dev)
        d = d / d.norm(dim=1, keepdim=True)
        length = dmin + (dmax - dmin) * torch.rand((n, 1), generator=g, device=dev)
        return a, (a + d * length).contiguous()

    def rake_stats(mod, env, a, b, sample=65536):
        """rakes per edge of the reference's walk (planning/validate.hh:24-67), on a sample of the batch: n = rakes the
        edge has, walked = rakes evaluated until the first colliding one (all n for a valid edge).  From the per-
        configuration validity of the interpolated rake configurations (vmv_validate_batch), i.e. an estimate for the
        few environments where a rake's answer is not the AND of its lanes (CAPT gating)."""
        m = min(sample, a.shape[0])
        a, b = a[:m], b[:m]
        v = b - a
        n_rakes = torch.clamp(torch.ceil(v.norm(dim=1) / 8.0 * mod.resolution()), min=1).to(torch.int64)
        top = int(n_rakes.max().item())
        walked = n_rakes.clone()
        alive = torch.ones(m, dtype=torch.bool, device=dev)
        pct = (torch.arange(1, 9, device=dev, dtype=torch.float32) / 8.0)[None, :, None]
        for i in range(top):  # rake i of every edge that has one and has not collided yet
            idx = torch.nonzero(alive & (n_rakes > i)).flatten()
            if idx.numel() == 0:
                break
            back = (v[idx] / (8.0 * n_rakes[idx, None].float()))[:, None, :] * float(i)
            cfgs = (a[idx, None, :] + v[idx, None, :] * pct - back).reshape(-1, a.shape[1]).contiguous()
            ok = mod.validate_batch(cfgs, env).reshape(-1, 8).all(dim=1)
            hit = idx[~ok]
            walked[hit] = i + 1
            alive[hit] = False
        return {"sample": m, "rakes_per_edge": float(n_rakes.float().mean()), "rakes_walked_per_edge": float(walked.float().mean()),
                "max_rakes": top}

    def old_edges(mod, n, seed, dmin, dmax):  # rounds 1-2: uniform starts (mostly invalid: edges die on their first rake)
        return edges(mod, n, seed, dmin, dmax)

    for cfg in args.configs:
        shape = None
        if cfg == "config2":
            mod, n, unit = vamp.panda, int((1 << 20) * args.scale), "checks/s"
            env = environment_from_spec(shell_spec(0))
            a, b = uniform(mod, n, 1), None
        elif cfg.startswith("mixed_"):  # diagnostics: 64 primitives of every kind (the five-list kernel variant)
            robot = cfg.split("_", 1)[1]
            mod, n, unit = getattr(vamp, robot), int((1 << 20) * args.scale), "checks/s"
            env = environment_from_spec(mixed_spec(0, *{"panda": (0.45, 0.95), "ur5": (0.45, 0.95), "fetch": (0.6, 1.2),
                                                        "baxter": (0.9, 1.6)}[robot]))
            a, b = uniform(mod, n, 1), None
            shape = "uniform configurations vs 16 spheres, 16 z-cuboids, 16 rotated cuboids, 8 capsules, 8 z-capsules"
        elif cfg.startswith("prm_"):  # diagnostics: config 4's workload for any robot (2^18 roadmap-shaped edges)
            robot = cfg.split("_", 1)[1]
            mod, n, unit = getattr(vamp, robot), int((1 << 18) * args.scale), "edges/s"
            env = environment_from_spec(shell_spec(0, 32, 32, *{"panda": (0.45, 0.95), "ur5": (0.45, 0.95), "fetch": (0.6, 1.2),
                                                              "baxter": (0.9, 1.6)}[robot]))
            a, b = prm_shaped_edges(mod, env, n, 0.2, 1.5, seed=3)
            shape = "PRM-shaped: valid Halton samples x valid neighbours at U[0.2,1.5] rad, 64 primitives"
        elif cfg == "config3":
            mod, n, unit = vamp.fetch, int((1 << 20) * args.scale), "checks/s"
            env = environment_from_spec([("capt", (shell_cloud(10000, 3), *RADII["fetch"], POINT_RADIUS))])
            a, b = uniform(mod, n, 2), None
        elif cfg in ("config4", "config4_uniform_starts"):
            mod, n, unit = vamp.ur5, int((1 << 20) * args.scale), "edges/s"
            env = environment_from_spec(shell_spec(0))
            if cfg == "config4":  # valid Halton samples, each joined to a valid neighbour at U[0.2, 1.5] rad
                a, b = prm_shaped_edges(mod, env, n, 0.2, 1.5, seed=3)
                shape = "PRM-shaped: valid Halton samples x valid neighbours at U[0.2,1.5] rad"
            else:
                a, b = old_edges(mod, n, 3, 0.2, 1.5)
                shape = "uniform starts (mostly invalid), random goals at U[0.2,1.5] rad (the rounds 1-2 generator)"
        elif cfg in ("config5", "config5_uniform_starts"):
            mod, n, unit = vamp.baxter, int((1 << 18) * args.scale), "edges/s"
            env = environment_from_spec(shell_spec(2, 16, 16, 0.9, 1.6) +
                                        [("capt", (shell_cloud(10000, 4, 1.0, 1.8), *RADII["baxter"], POINT_RADIUS))])
            if cfg == "config5":  # FCIT*-shaped: the edges of the 8-nearest-neighbour graph over valid Halton samples
                a, b = knn_shaped_edges(mod, env, n - n % 8, 8)
                n = a.shape[0]
                shape = "FCIT*-shaped: 8 nearest valid Halton samples of each valid Halton sample"
            else:
                a, b = old_edges(mod, n, 5, 0.1, 0.6)
                shape = "uniform starts (mostly invalid), random goals at U[0.1,0.6] rad (the rounds 1-2 generator)"
        else:
            raise SystemExit(cfg)
        stats = rake_stats(mod, env, a, b) if (b is not None and args.stats) else None

        def job():
            return validate_batch_sharded(mod, a, env, goals=b, rank=rank, world=world)

        words = job()  # warm: environment upload, robot part, RCCL channels
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            words = job()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        probe = None
        if world == 1 and args.shard_probe:  # what ONE rank of the N-GPU job executes per step, timed on this GPU
            probe = {}
            for parts in (2, 4, 8):
                lo_p, hi_p = shard_range(n, 0, parts)
                sa, sb = a[lo_p:hi_p].contiguous(), None if b is None else b[lo_p:hi_p].contiguous()
                validate_batch_sharded(mod, sa, env, goals=sb, rank=0, world=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.iters * 4):
                    validate_batch_sharded(mod, sa, env, goals=sb, rank=0, world=1)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / (args.iters * 4) * 1e3
                probe[str(parts)] = {"units": hi_p - lo_p, "ms": ms, "ceiling": (dt / args.iters * 1e3) / ms}
        if rank == 0:
            valid = vamp.unpack_bits(words.cpu().numpy().view(np.uint64), n)
            lo, hi = shard_range(n, 0, world)
            print(json.dumps({"config": cfg, "robot": mod._name, "n": n, "n_gpus": world, "ms": dt / args.iters * 1e3,
                              "value": n * args.iters / dt, "unit": unit, "scaling": "strong",
                              "valid_fraction": float(valid.mean()), "shard0": [lo, hi], "workload": shape, "rakes": stats, "shard_probe": probe,
                              "exchange": "RCCL all_gather of packed validity words" if world > 1 else "none"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
