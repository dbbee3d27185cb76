"""One rank of a sharded forest (multi-GPU wave protocol, include/sffgpu.h) in its own process: started by
tests/test_gpu_multiprocess.py, N of them share GPU 0 and exchange their answer records over torch.distributed
(gloo).  Prints one JSON line with the rank's view of the finished forest."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    name, wave, iters, seed, optimize = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    # driver: "caller" = per-round calls around the caller's all-gather; "library" = sffgpu_forest_run enqueues whole waves
    # and calls the all-gather the context was given (sffgpu_ctx_set_allgather) - the path bench.py --native-rccl takes
    # with ncclAllGather.  fault_rank >= 0: that rank alone shrinks its hit lists (SFFGPU_TEST_HITCAP): the overflow flags
    # travel in the answer records, so every rank must take the same fallback decisions.
    driver = sys.argv[9] if len(sys.argv) > 9 else "caller"
    fault_rank = int(sys.argv[10]) if len(sys.argv) > 10 else -1
    if rank == fault_rank:
        os.environ["SFFGPU_TEST_HITCAP"] = "4"
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    import common
    import space_filling_forest_star_amd as S
    sc = common.scenario(name)
    ctx = S.Context(0)
    ctx.upload_env(sc["env"])
    ctx.upload_robot(sc["robot"])
    roots = sc["xml_points"][:5] if sc["xml_points"] is not None else \
        common.free_roots(lambda p: int(ctx.collide_poses(p[None, :])[0]), sc["limits"], 5, seed=seed, dim=sc["dim"])
    f = S.Forest(ctx, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                 max_iterations=iters, wave=wave, seed=seed, optimize=bool(optimize), rank=rank, world=world)
    if driver == "library":
        ctx.set_allgather(S.host_staged_allgather(None), rank, world)
    waves = S.run_distributed(f)
    st = f.stats()
    out = {"rank": rank, "fingerprint": "%016x" % f.fingerprint(), "waves": int(waves), "device_engine": f.device_engine(),
           "stats": {k: int(st[k]) for k in ("iterations", "n_nodes", "n_borders", "collide_calls", "path_free_calls",
                                             "nn_queries", "frontier_size", "closed_size", "solved")},
           "executed": int(st["poses_executed"]), "host_fallback_waves": int(st["host_fallback_waves"]),
           "graph_launches": int(st["graph_launches"])}
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
