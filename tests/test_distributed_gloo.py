"""N>1 path on CPU: two gloo ranks run the functions bench.py uses for its multi-GPU runs
(so101_sim_amd.distributed: rank_info / init / shard_base / barrier / max_over_ranks / all_gather_returns /
run_sharded) with the env step stubbed - sharding by global env id, MAX-over-ranks timing, whole-job value, and the
all-gather of episode returns in rank order."""
import os
import socket
import time

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubEnv:
    """Stands in for BatchedEnvironment: returns are a function of the GLOBAL env id, like the kernels' RNG keying."""

    def __init__(self, env_id_base, n):
        self.ids = torch.arange(env_id_base, env_id_base + n, dtype=torch.float32)
        self.ret = torch.zeros(n)

    def step(self, i, delay):
        time.sleep(delay)
        self.ret += 0.5 * self.ids

    def episode_returns(self):
        return self.ret.clone()


def _worker(rank, world, port, n_global, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from so101_sim_amd import distributed as sd
    assert sd.rank_info() == (rank, rank, world)
    sd.init(torch.device("cpu"))
    # uneven strong-scaling split + gather in rank order
    lo, hi = sd.shard_range(n_global, world, rank)
    g = sd.all_gather_returns(torch.arange(lo, hi, dtype=torch.float32) * 0.5)
    # the weak-scaling timed region of bench.py with a stub step: rank 1 is 3x slower, the job time is ITS time
    per_rank, steps = 4, 3
    delay = 0.02 * (1 + 2 * rank)
    value, elapsed, rets = sd.run_sharded(lambda base: _StubEnv(base, per_rank), lambda env, i: env.step(i, delay), per_rank, steps)
    tmax = sd.max_over_ranks(float(rank + 1))
    # set-up work done on rank 0 only and broadcast (bench.py's pre-grasp pool): every rank ends with rank 0's tensors
    calls = []
    pool = sd.build_on_rank0(lambda: (calls.append(rank), (torch.arange(6, dtype=torch.float32).reshape(2, 3) + 100 * rank, torch.tensor([7 + rank], dtype=torch.int32)))[1])
    assert calls == ([0] if rank == 0 else []) and pool[0].tolist() == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]] and pool[1].tolist() == [7] and pool[1].dtype == torch.int32
    # the periodic logging exchange of configs[4]: every 2 steps here; the gather of step 3 holds both ranks' returns of THAT step
    log = sd.PeriodicReturnsGather(interval=2, device=torch.device("cpu"))
    env = _StubEnv(sd.shard_base(rank, per_rank), per_rank)
    fired = []
    for i in range(5):
        env.step(i, 0.0)
        fired.append(log.maybe(i, env.episode_returns))
    assert fired == [False, True, False, True, False] and log.count == 2
    lstep, lret = log.latest()
    assert lstep == 3 and lret.tolist() == [0.5 * 4 * k for k in range(world * per_rank)]
    if rank == 0:
        out.put((g.numpy(), value, elapsed, rets.numpy(), tmax))
    sd.finalize()


def test_bench_sharding_path_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_global = 11                                                     # uneven split: 6 + 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, q)) for r in range(2)]
    for p in procs:
        p.start()
    g, value, elapsed, rets, tmax = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(g, np.arange(n_global, dtype=np.float32) * 0.5)
    # global env ids 0..7 over two ranks of 4, three steps of 0.5 * id each, gathered in rank order
    np.testing.assert_allclose(rets, 1.5 * np.arange(8, dtype=np.float32))
    assert elapsed >= 3 * 0.06 * 0.9                                  # the slow rank's time, not rank 0's
    np.testing.assert_allclose(value, 2 * 4 * 3 / elapsed)            # whole-job aggregate
    assert tmax == 2.0


def test_single_process_is_a_no_op():
    from so101_sim_amd import distributed as sd
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert sd.rank_info() == (0, 0, 1)
    sd.init(None)
    sd.barrier()
    assert sd.max_over_ranks(1.25) == 1.25
    r = torch.arange(4, dtype=torch.float32)
    assert torch.equal(sd.all_gather_returns(r), r)
    assert sd.shard_base(3, 4096) == 12288
