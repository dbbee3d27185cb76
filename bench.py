#!/usr/bin/env python3
"""bench.py — headline benchmark of the SFF hot path on MI355X.

Workload (BASELINE.json configs[2], the headline): dense_3D.obj map, 6-DoF cylinder robot, 10 seeded roots,
SFF solver, 1M-node budget (authored step circum=14 / dtree=18: SURVEY.md §8(d)).

A "step" is a fixed block of `--waves-per-step` waves of the tree-expansion loop (a wave = `--wave` frontier
slots, each sampled / neighbour-queried / collision-checked for up to ThresholdMisses rounds, then committed in
order).  With the defaults (and the driver's `--steps 20 --warmup 5`) the 20 timed steps of 9 waves span the WHOLE
job: 10 roots -> the 1M-node budget (166 waves of 16384 slots; the CPU oracle's run of exactly this job is pinned in
tests/golden/full_size_run_w16384.json and the GPU must reproduce it node for node).  The warm-up steps run on a separate, discarded forest of the
same workload so that the timed region starts from the roots with warm kernels, allocations and caches.
`value` = accepted node expansions per second over the timed steps, whole job, with the map, the robot and the
node store resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0): the driver's contract plus
  roofline      neighbour-query kernel of the timed region, HIP-event timed inside the library on its launch
                stream; `traffic` only from a PMC summary collected with exactly these arguments (profiles/)
  cpu_baseline  the CPU oracle (kind "port": the reference binary cannot be built, RAPID is absent from its
                tree), wave = 1 == the reference's sequential loop, on a bounded sample of the same workload
  wave_sweep    the same forest at wave = 1 / 64 / 512 (bounded samples) and a quality block that compares
                the planner's output at small and large waves (the reference loop is wave = 1)
`--gpus N` (N > 1) without a torchrun environment re-launches itself under torch.distributed.run.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PMC_SUMMARY = os.path.join(ROOT, "profiles", "r6_bench_pmc_summary.json")
SWEEP_PMC_SUMMARY = os.path.join(ROOT, "profiles", "r5_sweep_pmc_summary.json")   # profiles/collect_sweep.sh


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--wave", type=int, default=16384)
    ap.add_argument("--waves-per-step", type=int, default=9,
                    help="waves per bench step: 20 steps x 9 waves of 16384 slots reach the 1M-node budget (166 waves)")
    ap.add_argument("--budget", type=int, default=1000000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--force-dist", action="store_true", help="run the RCCL record exchange even with one rank")
    ap.add_argument("--torch-exchange", action="store_true",
                    help="--force-dist with one rank: the caller (Python, torch.distributed) issues the collective of every round "
                         "instead of the library's own RCCL communicator - a host-paced figure, kept for comparison")
    ap.add_argument("--scaled-wave", type=int, default=0,
                    help="--force-dist with one rank: also run the job at this wave size (the wave that grows with the rank count: "
                         "8 ranks x 8192 slots = 65536) and print what ONE OF 8 RANKS would spend per round - measured one-GPU terms")
    ap.add_argument("--native-rccl", action="store_true",
                    help="sharded runs: the library drives the exchange itself (ncclAllGather on its own communicator, "
                         "whole waves enqueued ahead) instead of torch.distributed per round; opt-in until validated on "
                         "a multi-GPU node")
    ap.add_argument("--cpu-iters", type=int, default=120000, help="iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-sweep-micro", action="store_true", help="skip the stand-alone k_sweep roofline measurement")
    ap.add_argument("--no-wave-sweep", action="store_true", help="skip the wave = 1 / 64 / 512 legs and the quality block")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the SFF* (configs[4]) and RRT* legs")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """--gpus N without RANK/WORLD_SIZE: start the N ranks as a child job (before anything touches the GPU)."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    sys.exit(p.returncode)


def forest_quality(S, forest, ctx, sc):
    """Planner-quality figures of a finished forest: acceptance rate, coverage of the free space, tree costs."""
    import numpy as np
    st = forest.stats()
    nd = forest.nodes()
    n = st["n_nodes"]
    q = {"nodes": n, "iterations": st["iterations"], "nodes_per_iteration": n / max(1, st["iterations"]),
         "borders": st["n_borders"], "trees_connected": st["n_connected"]}
    # cost-to-root per unit of straight-line distance to the root (1.0 = straight line): how direct the trees are
    root_of = np.arange(n)
    par = nd["parent"]
    roots = np.where(par < 0)[0]
    root_pos = {int(t): nd["pos"][r] for r, t in zip(roots, nd["tree"][roots])}
    rp = np.array([root_pos[int(t)] for t in nd["tree"]])
    dd = nd["pos"] - rp
    dd[:, 3:] = (dd[:, 3:] + np.pi) % (2 * np.pi) - np.pi
    straight = np.sqrt((dd ** 2).sum(axis=1))
    m = straight > 5 * sc["sampling_dist"]
    if m.any():
        ratio = nd["cost"][m] / straight[m]
        q["cost_over_straight_line"] = {"median": float(np.median(ratio)), "p90": float(np.percentile(ratio, 90)),
                                       "max": float(ratio.max())}
    q["mean_edge_length_over_step"] = float(nd["dpar"][par >= 0].mean() / sc["sampling_dist"]) if n > len(roots) else None
    # coverage: seeded free probe poses that have a node within 2 steps (xyz), among probes inside the hull the
    # forest has reached so far (bounding box of its nodes)
    rs = np.random.RandomState(7)
    lim = sc["limits"]
    lo, hi = nd["pos"][:, :3].min(axis=0), nd["pos"][:, :3].max(axis=0)
    pr = np.zeros((4000, 6))
    for a in range(3):
        pr[:, a] = rs.uniform(max(lim[2 * a], lo[a]), min(lim[2 * a + 1], hi[a]), len(pr))
    free = ctx.collide_poses(pr) == 0
    pr = pr[free]
    if len(pr):
        from scipy.spatial import cKDTree
        d, _ = cKDTree(nd["pos"][:, :3]).query(pr[:, :3])
        q["coverage_within_2_steps"] = float((d < 2 * sc["sampling_dist"]).mean())
        q["coverage_probes"] = int(len(pr))
    return q


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    distributed = world > 1 or args.force_dist
    if args.native_rccl or (args.force_dist and world == 1 and not args.torch_exchange):
        # (one rank: the library's own communicator - whole waves enqueued by the library, no Python between the rounds; with
        # more ranks it is opt-in: ncclAllGather on a communicator of the library's own has never run with more than one rank)
        os.environ["SFFGPU_NATIVE_RCCL"] = "1"
    if args.force_dist and world == 1:
        os.environ["SFFGPU_TEST_EXCHANGE_SELF"] = "1"   # pack -> ncclAllGather (one rank) -> unpack in every round
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
    # sharded over the ranks (strong scaling: total work fixed); the result equals the 1-GPU run bit for bit.
    def make_forest(wave=args.wave, budget=args.budget, max_iterations=2**31 - 1, rk=rank, wd=world):
        return S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                        max_iterations=max_iterations, node_budget=budget, wave=wave, seed=args.seed, rank=rk, world=wd)

    def run_waves(forest, k):
        if distributed:
            S.run_distributed(forest, k)
        else:
            forest.run(k)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    B = max(1, args.waves_per_step)
    # ---- warm-up: W steps of the same workload on a forest that is then thrown away
    if args.warmup > 0:
        fw = make_forest()
        run_waves(fw, args.warmup * B)
        fw.close()

    # ---- timed region: exactly K steps (the last one ends early if the node budget is reached inside it)
    forest = make_forest()
    s0 = forest.stats()
    step_ms = []
    barrier()
    t0 = time.perf_counter()
    steps_done = 0
    for _ in range(args.steps):
        w_before = forest.stats()["waves"]
        ts = time.perf_counter()
        run_waves(forest, B)
        step_ms.append(1e3 * (time.perf_counter() - ts))
        if forest.stats()["waves"] == w_before:
            step_ms.pop()
            break           # the job had already finished: no step was run
        steps_done += 1
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    s1 = forest.stats()
    elapsed = t1 - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    acc = float(s1["n_nodes"] - s0["n_nodes"])
    checks = float(s1["collide_calls"] - s0["collide_calls"])
    executed = float(s1["poses_executed"] - s0["poses_executed"] + s1["samples_executed"] - s0["samples_executed"])
    if distributed:  # the forest (nodes, reference-equivalent checks) is shared; executed work is per rank
        t = torch.tensor([executed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        executed = float(t.item())
    quality_main = forest_quality(S, forest, ctx, sc) if rank == 0 else None
    forest.close()

    out = None
    if rank == 0:
        sweep_ms = s1["sweep_ms"] - s0["sweep_ms"]
        sweeps = s1["sweeps"] - s0["sweeps"]
        sweep_nodes = s1["sweep_nodes"] - s0["sweep_nodes"]
        event_ms = sweep_ms
        # `frac` is priced with the HIP events the contract names (on the library's launch stream, every 32nd wave launched
        # kernel by kernel; ~1.5 us ABOVE the rocprofv3 kernel duration of the same command, profiles/r6_bench_kernel_stats.csv).
        # Beside it: the device-clock bracket of EVERY launch - since round 6 every workgroup reports its first / last clock
        # read into one of 64 shards (round 5 sampled every 16th workgroup and so could miss the last one out) - which is
        # first instruction in .. last instruction out, ~2 us BELOW rocprofv3's dispatch begin .. end.  The rocprofv3 average
        # itself is printed when the committed summary was collected with exactly this run's arguments.
        clock_launches = s1["query_clock_launches"] - s0["query_clock_launches"]
        clock_ms = (s1["query_clock_ms"] - s0["query_clock_ms"]) if (clock_launches > 0 and clock_launches == sweeps) else None
        achieved = (24.0 * sweep_nodes / (sweep_ms * 1e-3)) / 1e9 if sweep_ms > 0 else 0.0
        # HBM-side bytes per launch of the neighbour-query kernel: only from a PMC summary that was collected
        # (separate rocprofv3 --pmc passes, profiles/collect.sh) with exactly the arguments of this run
        traffic, traffic_source, sweep_traffic, rocprof_us = None, None, None, None
        run_key = {"steps": args.steps, "warmup": args.warmup, "wave": args.wave, "waves_per_step": B,
                   "budget": args.budget, "seed": args.seed, "gpus": world}
        try:
            pm = json.load(open(PMC_SUMMARY))
            if pm.get("bench_args") == run_key:
                k = pm.get("query_kernel", "sffk::k_query_block")
                traffic = 1024.0 * (2.0 * pm["FETCH_SIZE"][k]["avg_KiB_per_launch"] + pm["WRITE_SIZE"][k]["avg_KiB_per_launch"])
                traffic_source = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; 2 x FETCH + WRITE (gfx950 correction of MI355X_MICROARCH.md)" % os.path.basename(PMC_SUMMARY)
                rocprof_us = pm.get("kernel_trace_avg_us", {}).get(pm.get("query_kernel", "sffk::k_query_block"))
                k = "sffk::k_sweep"
                if k in pm["FETCH_SIZE"]:
                    sweep_traffic = 1024.0 * (2.0 * pm["FETCH_SIZE"][k]["avg_KiB_per_launch"] + pm["WRITE_SIZE"][k]["avg_KiB_per_launch"])
        except Exception:
            pass
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
                            "circum=14 dtree=18, 1M-node budget; step = %d waves of %d frontier slots; the timed steps "
                            "run the whole job from the roots (warm-up on a separate forest)" % (B, args.wave),
                "wave": args.wave, "waves_per_step": B, "wave_per_gpu": args.wave // world, "node_budget": args.budget,
                "seed": args.seed, "nodes_at_start": s0["n_nodes"], "nodes_at_end": s1["n_nodes"],
                "waves": int(s1["waves"] - s0["waves"]),
                "parallelism": "1 GPU" if world == 1 else
                "one forest, wave slots sharded over %d GPUs (i %% world), RCCL all-gather of answer records per round "
                "(%s)" % (world, "the library's own communicator, whole waves enqueued by the library"
                          if getattr(ctx, "rccl", None) else "torch.distributed on the library's stream"),
            },
            "collision_checks_per_s": checks / elapsed,
            "collision_checks_executed_per_s": executed / elapsed,
            "iterations": s1["iterations"] - s0["iterations"],
            "step_ms": [round(x, 3) for x in step_ms],
            "time_split_ms": {"total": 1e3 * elapsed, "sweep_kernel": sweep_ms,
                              "collide_kernels": s1["collide_ms"] - s0["collide_ms"],
                              "sample_kernel": s1["sample_ms"] - s0["sample_ms"],
                              "host_logic": s1["host_ms"] - s0["host_ms"]},
            "roofline": {
                # neighbour query of one round (grid walk over the node grid and the round's own grid, exact fp64
                # re-test, classification of the hits: one fused kernel), timed live on the library's launch stream:
                # device-clock bracket of every launch (and HIP events around every 8th round beside it).
                # Algorithmic bytes = 24 B x nodes the query has to cover.
                "bound": "hbm", "kernel": "sffk::k_query_block (node grid + the round's own grid + hit classification + the "
                                          "clearance bits of the sample's pose and edge samples; flat work lists of a "
                                          "256-thread workgroup per 8 samples, DESIGN.md 5)",
                "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "traffic_source": traffic_source,
                "launches": int(sweeps), "avg_launch_us": 1e3 * sweep_ms / max(1, sweeps),
                "timing": "HIP events on the launch stream (every 32nd wave is launched kernel by kernel), scaled to all launches",
                "avg_launch_us_by_device_clock_every_workgroup": (1e3 * clock_ms / max(1, sweeps)) if clock_ms is not None else None,
                "avg_launch_us_by_rocprofv3": rocprof_us,
                "own_bytes_GBps": (traffic / (1e-3 * sweep_ms / max(1, sweeps)) / 1e9) if (traffic and sweep_ms > 0) else None,
                "avg_nodes_per_launch": sweep_nodes / max(1, sweeps),
                "avg_queries_per_launch": (s1["sweep_queries"] - s0["sweep_queries"]) / max(1, sweeps),
            },
            "quality": {"wave_%d_full_run" % args.wave: quality_main},
        }
        out["time_split_ms"]["commit_kernels"] = s1["commit_ms"] - s0["commit_ms"]
        out["time_split_ms"]["graph_launched_waves"] = int(s1["graph_launches"] - s0["graph_launches"])
        if distributed:
            # what bounds the strong scaling of ONE forest over N GPUs: per round, the part every rank
            # repeats (sampling + the in-order commit), the part that shards (neighbour query + collision: 1/N each) and
            # the exchange (pack + all-gather of the answer records + unpack).  HIP events of this run (rank 0); the
            # per-wave kernels (frontier picks, closed list: replicated too) are not in it.
            rounds = max(1, sweeps)
            rep = 1e3 * ((s1["sample_ms"] - s0["sample_ms"]) + (s1["commit_ms"] - s0["commit_ms"])) / rounds
            shd = 1e3 * ((s1["sweep_ms"] - s0["sweep_ms"]) + (s1["collide_ms"] - s0["collide_ms"])) / rounds
            exc = 1e3 * (s1["exchange_ms"] - s0["exchange_ms"]) / rounds
            out["dist_budget_us_per_round"] = {
                "replicated": rep, "sharded_over_ranks": shd, "pack_gather_unpack": exc, "ranks": world,
                "amdahl_speedup_bound_at_8_ranks": (rep + shd) / (rep + shd / 8.0 + exc) if rep + shd > 0 else None,
                "exchange_driver": "the library's own RCCL communicator (whole waves enqueued by the library)" if getattr(ctx, "rccl", None)
                else "torch.distributed from Python between the library's round calls (host-paced: the figure includes the host's gaps)",
                "note": "measured with %d rank(s); the all-gather of 8 ranks moves 8 x the bytes over xGMI" % world}
            if world == 1 and args.scaled_wave > 0:
                # The wave that grows with the rank count: strong scaling at a fixed wave is bounded by the part
                # every rank repeats; with wave = ranks x a fixed share the sharded kernels of a rank stay the size they are
                # on one GPU at that share while the job needs fewer rounds.  ONE GPU runs the whole job at the scaled wave
                # here (all of the sharded work itself): rounds, replicated time, sharded time and exchange per round are
                # measured, the time of one of R ranks is composed from them - an Amdahl composition of one-GPU terms,
                # not a multi-GPU measurement.  The forest is the oracle's at that wave
                # (tests/golden/full_size_run_w65536.json).
                R = 8
                fs = make_forest(wave=args.scaled_wave)
                q0 = fs.stats()
                c0 = time.perf_counter()
                run_waves(fs, 0)
                torch.cuda.synchronize()
                dts = time.perf_counter() - c0
                q1 = fs.stats()
                fs.close()
                rs = max(1, q1["sweeps"] - q0["sweeps"])
                rep_s = 1e3 * ((q1["sample_ms"] - q0["sample_ms"]) + (q1["commit_ms"] - q0["commit_ms"])) / rs
                shd_s = 1e3 * ((q1["sweep_ms"] - q0["sweep_ms"]) + (q1["collide_ms"] - q0["collide_ms"])) / rs
                exc_s = 1e3 * (q1["exchange_ms"] - q0["exchange_ms"]) / rs
                timed = rep_s + shd_s + exc_s                       # (the per-round kernels; the per-wave ones are in `other`)
                other = 1e3 * dts / rs * 1e3 - timed if dts > 0 else 0.0
                per_rank = rep_s + shd_s / R + exc_s + max(0.0, other)
                out["wave_scaled_budget"] = {
                    "wave": args.scaled_wave, "ranks_modelled": R, "slots_per_rank": args.scaled_wave // R,
                    "nodes": q1["n_nodes"], "waves": int(q1["waves"] - q0["waves"]), "rounds": int(rs),
                    "one_gpu_seconds_at_this_wave": dts, "one_gpu_nodes_per_s_at_this_wave": (q1["n_nodes"] - q0["n_nodes"]) / dts,
                    "us_per_round": {"replicated": rep_s, "sharded_over_ranks": shd_s, "pack_gather_unpack": exc_s,
                                     "per_wave_kernels_and_gaps": max(0.0, other)},
                    "modelled_seconds_of_one_of_%d_ranks" % R: per_rank * rs * 1e-6,
                    "modelled_speedup_over_the_one_gpu_job_at_wave_%d" % args.wave: elapsed / (per_rank * rs * 1e-6),
                    "kind": "Amdahl composition of measured ONE-GPU terms (no multi-GPU hardware here); the exchange is the "
                            "one-rank figure - 8 ranks move 8 x the bytes"}
        if world == 1 and not args.no_sweep_micro:
            # the linear k-NN sweep on its own (SURVEY.md 8(d) micro-benchmark, see profiles/sweep_microbench.py): N
            # uniform nodes, ONE query per pass, radius for ~32 neighbours; kernel time from the library's HIP events
            rs = np.random.RandomState(1)
            lim = np.asarray(sc["limits"], dtype=np.float64)
            Nn = 16000000   # 384 MB of fp32 columns: beyond the 256 MB Infinity Cache, so the figure is an HBM one
            pts = np.empty((Nn, 6))
            for a in range(3):
                pts[:, a] = rs.uniform(lim[2 * a], lim[2 * a + 1], Nn)
            pts[:, 3:] = rs.uniform(-np.pi, np.pi, (Nn, 3))
            cs = S.Context(local_rank)   # (a context of its own: the forests below size their arrays by their context's store)
            cs.nodes_reset(Nn + 64)
            for a0 in range(0, Nn, 4000000):
                cs.nodes_append(pts[a0:a0 + 4000000], np.zeros(len(pts[a0:a0 + 4000000]), np.int32))
            vol = (lim[1] - lim[0]) * (lim[3] - lim[2]) * (lim[5] - lim[4])
            rad = (32.0 * vol / Nn / 4.19) ** (1.0 / 3.0)
            qq = pts[rs.randint(0, Nn, 1)] + rs.normal(0, 5.0, (1, 6))
            cs.radius(qq, rad, cap=64)
            ms0, _ = cs.kernel_times()
            reps = 30
            for _ in range(reps):
                cs.radius(qq, rad, cap=64)
            ms1, _ = cs.kernel_times()
            cs.close()
            tt = (ms1[0] - ms0[0]) / reps * 1e-3
            sweep_src = None
            try:   # HBM-side bytes per pass from the rocprofv3 --pmc passes of exactly this leg (profiles/collect_sweep.sh)
                sp = json.load(open(SWEEP_PMC_SUMMARY))
                if sp.get("nodes") == Nn:
                    sweep_traffic = 1024.0 * (2.0 * sp["FETCH_SIZE"]["avg_KiB_per_launch"] + sp["WRITE_SIZE"]["avg_KiB_per_launch"])
                    sweep_src = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of profiles/sweep_leg.py (this leg on its "
                                 "own); 2 x FETCH + WRITE; kernel trace average %.1f us" % (os.path.basename(SWEEP_PMC_SUMMARY),
                                                                                             sp.get("kernel_trace", {}).get("avg_us", float("nan"))))
            except Exception:
                pass
            out["sweep_kernel_roofline"] = {"kernel": "sffk::k_sweep", "bound": "hbm", "nodes": Nn, "queries_per_pass": 1,
                                            "us_per_pass": tt * 1e6, "achieved": 24.0 * Nn / tt / 1e9, "peak": 8000.0,
                                            "unit": "GB/s", "frac": 24.0 * Nn / tt / 8e12, "traffic": sweep_traffic,
                                            "traffic_source": sweep_src}
        if world == 1 and not args.no_wave_sweep:
            # the small-wave end: wave = 1 IS the reference's sequential loop (one sample per GPU round trip)
            legs = {}

            def small_wave_leg(wv, iters, optimize=False, spec=None):
                # (the knob is read when the forest is created; SFFGPU_SPEC=0 = the single wavefront of round 5, k_seq_waves)
                old = os.environ.get("SFFGPU_SPEC")
                if spec is not None:
                    os.environ["SFFGPU_SPEC"] = spec
                try:
                    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                                 max_iterations=iters, node_budget=0, wave=wv, seed=args.seed, optimize=optimize)
                finally:
                    if spec is not None:
                        if old is None:
                            os.environ.pop("SFFGPU_SPEC", None)
                        else:
                            os.environ["SFFGPU_SPEC"] = old
                c0 = time.perf_counter()
                f.run()
                dt = time.perf_counter() - c0
                st = f.stats()
                leg = {"accepted_nodes_per_s": (st["n_nodes"] - len(roots)) / dt, "iterations_per_s": st["iterations"] / dt,
                       "iterations": st["iterations"], "nodes": st["n_nodes"], "seconds": dt}
                if wv == 1:
                    # wave = 1 is the reference's own order of operations (src/forest.h:122-202).  k_spec_waves evaluates the
                    # next waves' attempts side by side on many wavefronts and commits the path that really happened
                    leg["kernel"] = "k_spec_waves" if st["spec_steps"] else "k_seq_waves (one wavefront)"
                    if st["spec_steps"]:
                        leg["speculation"] = {"steps": st["spec_steps"], "attempts_evaluated": st["spec_evaluated"],
                                              "attempts_committed": st["spec_committed"],
                                              "evaluated_per_committed": st["spec_evaluated"] / max(1, st["spec_committed"]),
                                              "iterations_per_step": st["spec_committed"] / max(1, st["spec_steps"]),
                                              "host_fallback_waves": st["host_fallback_waves"]}
                f.close()
                return leg

            small_wave_leg(1, 2000)                      # (first launch of the kernel: code object load, LDS attribute)
            legs["wave_1"] = small_wave_leg(1, 8000)
            legs["wave_1_one_wavefront"] = small_wave_leg(1, 8000, spec="0")
            legs["wave_1_100k_iterations"] = small_wave_leg(1, 100000)
            small_wave_leg(1, 2000, optimize=True)
            legs["wave_1_sff_star"] = small_wave_leg(1, 8000, optimize=True)
            legs["wave_1_sff_star_one_wavefront"] = small_wave_leg(1, 8000, optimize=True, spec="0")
            for wv, iters in ((64, 150000), (512, 1000000)):
                legs["wave_%d" % wv] = small_wave_leg(wv, iters)
            out["wave_sweep"] = legs
            # does a large wave degrade the planner?  The same 30 k-node budget at wave 64 (close to the sequential
            # loop: the frontier holds thousands of nodes) and at the bench's wave
            for wv in (64, args.wave):
                f = make_forest(wave=wv, budget=30000, rk=0, wd=1)
                f.run()
                out["quality"]["wave_%d_30k_nodes" % wv] = forest_quality(S, f, ctx, sc)
                f.close()
        if world == 1 and not args.no_extra_legs:
            # the other two solvers north_star names, each with the CPU oracle's sequential loop beside it (bounded samples)
            import oracle_lib as O
            legs = {}
            # ---- SFF* (BASELINE configs[4]: building.obj, 20 seeded roots, optimize = true: choose-parent + rewire on the
            # device engine): the whole job - the forest saturates ("solved") at ~2e5 nodes, far below the 2 M budget
            scb = common.scenario("building")
            cb = S.Context(local_rank)
            cb.upload_env(scb["env"])
            cb.upload_robot(scb["robot"])
            rb = common.free_roots(lambda p: int(cb.collide_poses(p[None, :])[0]), scb["limits"], 20, seed=1)
            kwb = dict(dist_tree=scb["dist_tree"], sampling_dist=scb["sampling_dist"], dim=6, optimize=True, seed=1)
            for rep in range(2):       # (the first run warms the kernels and the allocations up)
                f = S.Forest(cb, rb, scb["limits"], max_iterations=2**31 - 1, node_budget=2000000, wave=8192, **kwb)
                c0 = time.perf_counter()
                f.run()
                dt = time.perf_counter() - c0
                st = f.stats()
                f.close()
            wb = O.World(scb["env"], scb["robot"], O.TRIG_PORTABLE)
            fo = O.Forest(wb, rb, scb["limits"], max_iterations=40000, wave=1, **kwb)
            c0 = time.perf_counter()
            fo.run()
            dto = time.perf_counter() - c0
            so = fo.stats()
            legs["sff_star"] = {
                "workload": "BASELINE configs[4]: building.obj (26908 tris), 6-DoF, 20 seeded roots, SFF* optimize=true, 2M-node "
                            "budget, waves of 8192 slots, run to its end (saturates: solved)",
                "accepted_nodes_per_s": (st["n_nodes"] - 20) / dt, "collision_checks_per_s": st["collide_calls"] / dt,
                "nodes": st["n_nodes"], "iterations": st["iterations"], "solved": st["solved"], "seconds": dt,
                "device_engine": bool(st["star_rounds"] > 0), "host_ms": st["host_ms"], "rounds": st["sweeps"],
                "rewire_fixed_point_passes_per_round": st["star_passes"] / max(1, st["star_rounds"]),
                "rewires": st["star_rewires"], "host_fallback_waves": st["host_fallback_waves"],
                "cpu_oracle_wave_1": {"accepted_nodes_per_s": (so["n_nodes"] - 20) / dto, "iterations": so["iterations"],
                                      "nodes": so["n_nodes"], "seconds": dto, "cores": 1,
                                      "collision_checks_per_s": so["collide_calls"] / dto}}
            cb.close()
            # ---- SFF* in the regime configs[4] names (2 M nodes, k = 34) but building.obj never reaches: the headline map with
            # a step of 11 / dtree 14 has room for it (tests/test_gpu_parity.py::test_sff_star_at_two_million_nodes)
            for rep in range(2):
                f = S.Forest(ctx, roots, sc["limits"], dist_tree=14.0, sampling_dist=11.0, dim=6, optimize=True,
                             max_iterations=2**31 - 1, node_budget=2000000, wave=16384, seed=1)
                c0 = time.perf_counter()
                f.run()
                dt2 = time.perf_counter() - c0
                st2 = f.stats()
                f.close()
            legs["sff_star_2m_nodes"] = {
                "workload": "dense_3D.obj, 6-DoF, 10 seeded roots, SFF* optimize=true, circum 11 / dtree 14, 2M-node budget, "
                            "waves of 16384 slots (k-nearest sets of 33-34 members at the end)",
                "accepted_nodes_per_s": (st2["n_nodes"] - 10) / dt2, "collision_checks_per_s": st2["collide_calls"] / dt2,
                "nodes": st2["n_nodes"], "iterations": st2["iterations"], "seconds": dt2, "rounds": st2["sweeps"],
                "rewire_fixed_point_passes_per_round": st2["star_passes"] / max(1, st2["star_rounds"]),
                "k_nearest_members_per_accepted_node": st2["star_members"] / max(1, st2["n_nodes"] - 10),
                "rewires": st2["star_rewires"], "host_fallback_waves": st2["host_fallback_waves"]}
            # ---- RRT* (src/rrt.h:128-322, rewire :156-201): one tree from the first root of the headline workload, no goal,
            # speculative waves (conflicts cut a wave, the RNG rewinds: the committed sequence is the reference's)
            kwr = dict(dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6, optimize=True, seed=1)
            for rep in range(2):
                r = S.Rrt(ctx, roots[:1], sc["limits"], max_iterations=150000, wave=0, **kwr)
                c0 = time.perf_counter()
                r.run()
                dtr = time.perf_counter() - c0
                sr = r.stats()
                r.close()
            wr = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
            ro = O.Rrt(wr, roots[:1], sc["limits"], max_iterations=5000, **kwr)
            c0 = time.perf_counter()
            ro.run()
            dtro = time.perf_counter() - c0
            sro = ro.stats()
            legs["rrt_star"] = {
                "workload": "dense_3D.obj, 6-DoF, RRT* (optimize=true) from one root, no goal, 150 k iterations, adaptive "
                            "speculative waves",
                "iterations_per_s": sr["iterations"] / dtr, "accepted_nodes_per_s": (sr["n_nodes"] - 1) / dtr,
                "collision_checks_per_s": sr["collide_calls"] / dtr, "nodes": sr["n_nodes"], "iterations": sr["iterations"],
                "waves": sr["waves"], "speculated": sr["speculated"], "committed": sr["committed"], "seconds": dtr,
                "cpu_oracle_sequential": {"iterations_per_s": sro["iterations"] / dtro,
                                          "accepted_nodes_per_s": (sro["n_nodes"] - 1) / dtro, "iterations": sro["iterations"],
                                          "seconds": dtro, "cores": 1, "collision_checks_per_s": sro["collide_calls"] / dtro}}
            # ---- plain RRT (src/rrt.h:128-322 without the rewire) and Multi-T-RRT (5 roots, merging trees, :234-316)
            for key, nroot, opt, iters, oiters in (("rrt", 1, False, 150000, 5000), ("multi_t_rrt", 5, False, 150000, 5000)):
                for rep in range(2):
                    r = S.Rrt(ctx, roots[:nroot], sc["limits"], max_iterations=iters, wave=0, dist_tree=sc["dist_tree"],
                              sampling_dist=sc["sampling_dist"], dim=6, optimize=opt, seed=1)
                    c0 = time.perf_counter()
                    r.run()
                    dtr = time.perf_counter() - c0
                    sr = r.stats()
                    r.close()
                ro = O.Rrt(wr, roots[:nroot], sc["limits"], max_iterations=oiters, dist_tree=sc["dist_tree"],
                           sampling_dist=sc["sampling_dist"], dim=6, optimize=opt, seed=1)
                c0 = time.perf_counter()
                ro.run()
                dtro = time.perf_counter() - c0
                sro = ro.stats()
                legs[key] = {
                    "workload": "dense_3D.obj, 6-DoF, %s from %d root(s), no goal, %d iterations, adaptive speculative waves"
                                % ("RRT" if nroot == 1 else "Multi-T-RRT (merging trees)", nroot, iters),
                    "iterations_per_s": sr["iterations"] / dtr, "accepted_nodes_per_s": (sr["n_nodes"] - nroot) / dtr,
                    "collision_checks_per_s": sr["collide_calls"] / dtr, "nodes": sr["n_nodes"], "iterations": sr["iterations"],
                    "waves": sr["waves"], "speculated": sr["speculated"], "committed": sr["committed"], "seconds": dtr,
                    "cpu_oracle_sequential": {"iterations_per_s": sro["iterations"] / dtro, "iterations": sro["iterations"],
                                              "accepted_nodes_per_s": (sro["n_nodes"] - nroot) / dtro, "seconds": dtro, "cores": 1}}
            # ---- the priority-frontier mode (priorityBias = 0.95: what every XML the reference ships sets; src/forest.h:126-147,
            # 160-181, 360-363, src/heap.h): the headline map, heaps on the device engine (devprio.hip)
            pl = {}
            for wv in (8192, 16384):
                for rep in range(2):
                    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                                 max_iterations=2**31 - 1, node_budget=300000, wave=wv, seed=1, priority_bias=0.95)
                    c0 = time.perf_counter()
                    f.run()
                    dtp = time.perf_counter() - c0
                    sp = f.stats()
                    dev_on = bool(f.device_engine())
                    f.close()
                pl["wave_%d" % wv] = {"accepted_nodes_per_s": (sp["n_nodes"] - 10) / dtp, "nodes": sp["n_nodes"],
                                      "iterations": sp["iterations"], "waves": sp["waves"], "seconds": dtp,
                                      "collision_checks_per_s": sp["collide_calls"] / dtp, "device_engine": dev_on,
                                      "host_fallback_waves": sp["host_fallback_waves"]}
            fo = O.Forest(wr, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=6,
                          max_iterations=30000, wave=1, seed=1, priority_bias=0.95)
            c0 = time.perf_counter()
            fo.run()
            dto = time.perf_counter() - c0
            so = fo.stats()
            pl["workload"] = ("dense_3D.obj, 6-DoF, 10 seeded roots, SFF with priorityBias 0.95 (90 heaps: one per ordered pair of "
                              "trees), 300 k-node budget")
            pl["cpu_oracle_wave_1"] = {"accepted_nodes_per_s": (so["n_nodes"] - 10) / dto, "iterations": so["iterations"],
                                       "nodes": so["n_nodes"], "seconds": dto, "cores": 1}
            legs["sff_priority"] = pl
            out["extra_legs"] = legs
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
                          "reference's sequential loop), CPU oracle (the reference binary itself cannot be built: RAPID "
                          "is absent from its tree); reached %d nodes in %.1f s"
                          % (args.cpu_iters, so["n_nodes"], c1 - c0),
                "collision_checks_per_s": so["collide_calls"] / (c1 - c0),
            }
    ctx.close()   # (also tears the library's own RCCL communicator down before the line is printed)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        # RCCL prints its version banner through C stdio, which would otherwise be flushed AFTER this line at
        # exit: flush it first so that the JSON line is the last thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
