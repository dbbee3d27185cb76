"""world_size-2 gloo test (CPU) of the multi-GPU exchange step: variable-length int32 record
streams are all-gathered in rank order, zero padding never leaks, and every rank receives the
identical concatenation — the property the replicated commit pass relies on."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from space_filling_forest_star_amd import exchange_records
    rs = np.random.RandomState(100 + rank)
    results = []
    for it in range(8):
        # 9000 words exceed the initial capacity: the gather is repeated once with a bigger buffer
        n = [0, 7, 1000, 3, 9000, 2][(it + rank) % 6] if it else (5 if rank == 0 else 0)
        local = rs.randint(-2**31, 2**31 - 1, n).astype(np.int32)
        allw, counts = exchange_records(local)
        assert counts[rank] == n
        off = int(counts[:rank].sum())
        assert np.array_equal(allw[off:off + n], local)
        assert len(allw) == int(counts.sum())
        results.append((allw.copy(), counts.copy()))
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.concatenate([a for a, _ in results]))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_records_gloo(tmp_path):
    world = 2
    for attempt in range(3):   # (the port is free when it is picked, not necessarily when the workers bind it)
        try:
            mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
            break
        except Exception as e:
            if attempt == 2 or "EADDRINUSE" not in str(e):
                raise
    a = np.load(tmp_path / "r0.npy")
    b = np.load(tmp_path / "r1.npy")
    assert np.array_equal(a, b) and len(a) > 0
