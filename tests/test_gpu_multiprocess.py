"""The multi-GPU wave protocol across REAL processes: N fresh interpreters (started before they touch the GPU), one
rank each, all on GPU 0, exchanging their per-round answer records with torch.distributed (gloo) exactly like
`bench.py --gpus N` does over RCCL.  Every rank must end with the forest the single-process CPU oracle builds at the
same wave size, and the work must really have been split."""
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import pytest

import common
import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# (SFF and SFF* run on the device-resident engine at every wave size: answer records all-gathered between device
#  buffers, commit - and for SFF* the rewire fixed point - replicated on every rank's GPU)
# driver "library": sffgpu_forest_run enqueues whole waves, one ahead, and issues the collective itself through the
# all-gather the context was given (sffgpu_ctx_set_allgather; here staged through the host over gloo) - the call
# sequence of bench.py --native-rccl with ncclAllGather, run with 2 and 4 ranks; fault_rank: ONE rank shrinks its hit
# lists, the overflow flags travel in the records and every rank falls back for the same rounds
@pytest.mark.parametrize("name,world,wave,iters,optimize,driver,fault_rank",
                         [("dense3d", 2, 512, 12000, 0, "caller", -1), ("triang", 3, 256, 6000, 1, "caller", -1),
                          ("dense3d", 4, 1024, 40000, 0, "caller", -1), ("dense3d", 2, 64, 3000, 0, "caller", -1),
                          ("dense3d", 2, 512, 12000, 0, "library", -1), ("dense3d", 4, 1024, 30000, 0, "library", -1),
                          ("dense3d_coarse", 2, 512, 12000, 0, "library", 1), ("dense3d_coarse", 2, 512, 12000, 0, "caller", 0),
                          # the wave that grows with the rank count (4 ranks x 16 384 slots: the largest wave the device
                          # engine takes), library-driven
                          ("dense3d", 4, 65536, 35000, 0, "library", -1)])
def test_sharded_forest_across_processes_equals_the_oracle(name, world, wave, iters, optimize, driver, fault_rank):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for attempt in range(3):   # (the rendezvous port is free when it is picked, not necessarily seconds later: retry on EADDRINUSE)
        port = free_port()
        files = [(tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")) for _ in range(world)]
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_forest_worker.py"), str(r), str(world),
                                   str(port), name, str(wave), str(iters), "3", str(optimize), driver, str(fault_rank)],
                                  stdout=files[r][0], stderr=files[r][1], text=True, env=env) for r in range(world)]
        # (a rank that dies leaves the others waiting in the rendezvous: end them with it)
        t_end = time.time() + 600
        while time.time() < t_end and any(p.poll() is None for p in procs) and not any(p.poll() not in (None, 0) for p in procs):
            time.sleep(0.2)
        for p in procs:
            if p.poll() is None:
                p.kill()
        res = []
        for r, p in enumerate(procs):
            p.wait(timeout=60)
            files[r][0].seek(0); files[r][1].seek(0)
            res.append((files[r][0].read(), files[r][1].read(), p.returncode))
            files[r][0].close(); files[r][1].close()
        if attempt < 2 and any(rc != 0 and "EADDRINUSE" in se for _, se, rc in res):
            continue
        break
    outs = []
    for so, se, rc in res:
        assert rc == 0, se[-2000:]
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1]
        outs.append(json.loads(line[7:]))
    sc = common.scenario(name)
    w = O.World(sc["env"], sc["robot"], O.TRIG_PORTABLE)
    roots = sc["xml_points"][:5] if sc["xml_points"] is not None else \
        common.free_roots(w.collide, sc["limits"], 5, seed=3, dim=sc["dim"])
    fo = O.Forest(w, roots, sc["limits"], dist_tree=sc["dist_tree"], sampling_dist=sc["sampling_dist"], dim=sc["dim"],
                  max_iterations=iters, wave=wave, seed=3, optimize=bool(optimize))
    fo.run()
    so = fo.stats()
    assert so["n_nodes"] > 300
    for o in outs:
        assert o["fingerprint"] == "%016x" % fo.fingerprint(), o["rank"]
        for k, v in o["stats"].items():
            assert v == so[k], (o["rank"], k, v, so[k])
    assert all(o["device_engine"] for o in outs)
    # the candidates were sharded: no rank evaluated all poses, together they evaluated each exactly once
    ex = np.array([o["executed"] for o in outs])
    assert ex.max() < 0.75 * ex.sum() and ex.min() > 0
    if fault_rank >= 0:   # the forced overflow really happened, and every rank handed the same waves to the host protocol
        assert outs[0]["host_fallback_waves"] > 0
        assert len({o["host_fallback_waves"] for o in outs}) == 1
