"""N>1 path on CPU: two gloo ranks shard the env ids and all-gather episode returns in rank order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from so101_sim_amd.distributed import all_gather_returns, shard_range
    lo, hi = shard_range(n_global, world, rank)
    local = torch.arange(lo, hi, dtype=torch.float32) * 0.5          # stand-in for per-env episode returns
    g = all_gather_returns(local)
    if rank == 0:
        out.put(g.numpy())
    dist.destroy_process_group()


def test_all_gather_returns_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_global = 11                                                     # uneven split: 6 + 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(got, np.arange(n_global, dtype=np.float32) * 0.5)
