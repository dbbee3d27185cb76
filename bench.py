#!/usr/bin/env python3
"""bench.py — headline benchmark of the SFF hot path on MI355X.

Workload (BASELINE.json configs[2], the headline): dense_3D.obj map, 6-DoF cylinder robot,
10 roots, SFF solver, 1M-node budget (authored step circum=14 / dtree=18: SURVEY.md §8(d)).
A "step" is one WAVE of the tree-expansion loop: `--wave` frontier slots, each sampled /
neighbour-swept / collision-checked for up to ThresholdMisses rounds, then committed.
`value` = accepted node expansions per second over the K timed waves (whole job), with the
map, robot and node store resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (neighbour-sweep
kernel, HIP-event timed inside the library on its launch stream) and `cpu_baseline` (the CPU
oracle, wave=1 == the reference's sequential loop, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=255)   # 3 + 255 waves of 8192 slots: ~10 -> ~940k nodes of the 1M budget
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--wave", type=int, default=8192)
    ap.add_argument("--budget", type=int, default=1000000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL record exchange even with one rank")
    ap.add_argument("--cpu-iters", type=int, default=120000, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-sweep-micro", action="store_true", help="skip the stand-alone k_sweep roofline measurement")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or args.force_dist
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import common
    import space_filling_forest_star_amd as S

    sc = common.scenario("dense3d")
    ctx = S.Context(local_rank)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(sc["robot"])
    # 10 seeded collision-free roots (identical on every rank): drawn with the GPU collision kernel
    roots = common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 10, seed=1)
    # multi-GPU (BASELINE configs[3]): the SAME forest and the same 1 M-node budget, the wave's sample batch
    # sharded over the ranks (strong scaling: total work fixed).  Rank r evaluates candidates i with
    # i % world == r, the answers are all-gathered over RCCL once per round and every rank replays the identical
    # commit, so the result equals the 1-GPU run bit for bit.
    forest = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                      max_iterations=2**31 - 1, node_budget=args.budget, wave=args.wave, seed=args.seed,
                      rank=rank, world=world)

    def run_waves(k):
        if distributed:
            S.run_distributed(forest, k)
        else:
            forest.run(k)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    run_waves(args.warmup)
    s0 = forest.stats()
    barrier()
    t0 = time.perf_counter()
    run_waves(args.steps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    s1 = forest.stats()
    elapsed = t1 - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps_done = int(s1["waves"] - s0["waves"])
    acc = float(s1["n_nodes"] - s0["n_nodes"])
    checks = float(s1["collide_calls"] - s0["collide_calls"])
    executed = float(s1["poses_executed"] - s0["poses_executed"] + s1["samples_executed"] - s0["samples_executed"])
    if distributed:  # the forest (nodes, reference-equivalent checks) is shared; executed work is per rank
        t = torch.tensor([executed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        executed = float(t.item())

    forest.close()

    out = None
    if rank == 0:
        sweep_ms = s1["sweep_ms"] - s0["sweep_ms"]
        sweeps = s1["sweeps"] - s0["sweeps"]
        sweep_nodes = s1["sweep_nodes"] - s0["sweep_nodes"]
        achieved = (24.0 * sweep_nodes / (sweep_ms * 1e-3)) / 1e9 if sweep_ms > 0 else 0.0
        # HBM-side bytes per neighbour query from the committed PMC summary (separate rocprofv3 --pmc
        # passes of this command; FETCH_SIZE doubled per MI355X_MICROARCH.md, gather pattern uncalibrated)
        traffic = None
        sweep_traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1q_bench_pmc_summary.json")))
            kib = 0.0
            k = "sffk::k_grid_query"
            kib = 2.0 * pm["FETCH_SIZE"][k]["avg_KiB_per_launch"] + pm["WRITE_SIZE"][k]["avg_KiB_per_launch"]
            traffic = kib * 1024.0
            k = "sffk::k_sweep"   # (only launched by the stand-alone sweep measurement below)
            if k in pm["FETCH_SIZE"]:
                sweep_traffic = 1024.0 * (2.0 * pm["FETCH_SIZE"][k]["avg_KiB_per_launch"] + pm["WRITE_SIZE"][k]["avg_KiB_per_launch"])
        except Exception:
            traffic = None
        out = {
            "metric": "accepted node expansions/sec + collision checks/sec, dense_3D 6-DoF",
            "value": acc / elapsed,
            "unit": "accepted nodes/s",
            "n_gpus": world,
            "steps": steps_done,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, steps_done),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "dense_3D.obj (1832 tris) + robot_cylinder_small (124 tris), 6-DoF, 10 seeded roots, SFF, "
                            "circum=14 dtree=18, 1M-node budget; step = one wave of %d frontier slots" % args.wave,
                "wave": args.wave, "wave_per_gpu": args.wave // world, "node_budget": args.budget, "seed": args.seed,
                "nodes_at_start": s0["n_nodes"], "nodes_at_end": s1["n_nodes"],
                "parallelism": "1 GPU" if world == 1 else
                "one forest, wave slots sharded over %d GPUs (i %% world), RCCL all-gather of answer records per round" % world,
            },
            "collision_checks_per_s": checks / elapsed,
            "collision_checks_executed_per_s": executed / elapsed,
            "iterations": s1["iterations"] - s0["iterations"],
            "time_split_ms": {"total": 1e3 * elapsed, "sweep_kernel": sweep_ms,
                              "collide_kernels": s1["collide_ms"] - s0["collide_ms"],
                              "sample_kernel": s1["sample_ms"] - s0["sample_ms"],
                              "host_logic": s1["host_ms"] - s0["host_ms"]},
            "roofline": {
                # neighbour query of one round = k_grid_query over the node grid and over the round's own
                # grid, timed with HIP events on the library's launch stream (every 8th round).  Algorithmic
                # bytes = 24 B x nodes the query has to cover; the grid touches far fewer
                # bytes than that, see `traffic` (rocprofv3 PMC pass of this same command, profiles/).
                "bound": "hbm", "kernel": "sffk::k_grid_query (node grid + the round's own grid)",
                "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic,
                "launches": int(sweeps), "avg_launch_us": 1e3 * sweep_ms / max(1, sweeps),
                "avg_nodes_per_launch": sweep_nodes / max(1, sweeps),
                "avg_queries_per_launch": (s1["sweep_queries"] - s0["sweep_queries"]) / max(1, sweeps),
            },
        }
        if world == 1 and not args.no_sweep_micro:
            # the linear k-NN sweep on its own (SURVEY.md 8(d) micro-benchmark, see profiles/sweep_microbench.py): N
            # uniform nodes, ONE query per pass, radius for ~32 neighbours; kernel time from the library's HIP events
            rs = np.random.RandomState(1)
            lim = np.asarray(sc["limits"], dtype=np.float64)
            Nn = 2000000
            pts = np.empty((Nn, 6))
            for a in range(3):
                pts[:, a] = rs.uniform(lim[2 * a], lim[2 * a + 1], Nn)
            pts[:, 3:] = rs.uniform(-np.pi, np.pi, (Nn, 3))
            ctx.nodes_reset(Nn + 64)
            ctx.nodes_append(pts, np.zeros(Nn, np.int32))
            vol = (lim[1] - lim[0]) * (lim[3] - lim[2]) * (lim[5] - lim[4])
            rad = (32.0 * vol / Nn / 4.19) ** (1.0 / 3.0)
            qq = pts[rs.randint(0, Nn, 1)] + rs.normal(0, 5.0, (1, 6))
            ctx.radius(qq, rad, cap=64)
            ms0, _ = ctx.kernel_times()
            reps = 30
            for _ in range(reps):
                ctx.radius(qq, rad, cap=64)
            ms1, _ = ctx.kernel_times()
            tt = (ms1[0] - ms0[0]) / reps * 1e-3
            out["sweep_kernel_roofline"] = {"kernel": "sffk::k_sweep", "bound": "hbm", "nodes": Nn, "queries_per_pass": 1,
                                            "us_per_pass": tt * 1e6, "achieved": 24.0 * Nn / tt / 1e9, "peak": 8000.0,
                                            "unit": "GB/s", "frac": 24.0 * Nn / tt / 8e12, "traffic": sweep_traffic}
        if args.cpu_iters > 0 and world == 1:
            import oracle_lib as O
            w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
            fo = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                          max_iterations=args.cpu_iters, wave=1, seed=args.seed)
            c0 = time.perf_counter()
            fo.run()
            c1 = time.perf_counter()
            so = fo.stats()
            out["cpu_baseline"] = {
                "value": (so["n_nodes"] - 10) / (c1 - c0), "unit": "accepted nodes/s", "cores": 1, "kind": "port",
                "sample": "first %d iterations of the same workload from the same roots and seed, wave=1 (the "
                          "reference's sequential loop), CPU oracle; reached %d nodes in %.1f s"
                          % (args.cpu_iters, so["n_nodes"], c1 - c0),
                "collision_checks_per_s": so["collide_calls"] / (c1 - c0),
            }
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which would otherwise be flushed AFTER this line at
        # exit: flush it first so that the JSON line is the last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
