"""Multi-GPU sharding of independent environments: contiguous global env-id ranges per rank, no
data-path collective; one all-gather of episode returns (RCCL over xGMI on GPUs, gloo in CPU tests)
for logging only.  bench.py drives its N > 1 runs through exactly these functions, and
tests/test_distributed_gloo.py runs the same functions with world size 2 on CPU."""
from __future__ import annotations

import os
import time


def rank_info() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(device=None, backend: str | None = None):
    """Initialise the process group when WORLD_SIZE > 1 (nccl = RCCL on ROCm for a GPU device, gloo otherwise)."""
    import torch.distributed as dist
    _, _, world = rank_info()
    if world == 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if device is not None and getattr(device, "type", "cpu") == "cuda" else "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def _active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def barrier():
    if _active():
        import torch.distributed as dist
        dist.barrier()


def shard_base(rank: int, envs_per_rank: int) -> int:
    """Weak scaling: every rank owns `envs_per_rank` envs; global id of its env 0 (keys the per-env RNG, so a shard
    reproduces the same envs of the unsharded run)."""
    return int(rank) * int(envs_per_rank)


def shard_range(n_global: int, world_size: int, rank: int) -> tuple[int, int]:
    """[start, stop) of the global env ids owned by `rank` (strong scaling); remainders go to the lowest ranks."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_global), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def build_on_rank0(build, src: int = 0):
    """`build()` -> tuple of tensors, run on rank `src` only; the other ranks receive the result (shapes and dtypes first, then
    the data: RCCL broadcast on GPUs).  For set-up work that is the same on every rank - the pre-grasp pool of bench.py's
    pick-and-place workload - so that eight ranks do not each repeat it."""
    import torch
    import torch.distributed as dist
    rank, _, _ = rank_info()
    if not _active():
        return build()
    out = build() if rank == src else None
    meta = [[(tuple(t.shape), str(t.dtype).replace("torch.", ""), str(t.device.type)) for t in out]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    if rank != src:
        dev = "cuda" if meta[0][0][2] == "cuda" else "cpu"
        out = tuple(torch.empty(shape, dtype=getattr(torch, dt), device=dev) for shape, dt, _ in meta[0])
    for t in out:
        dist.broadcast(t, src=src)
    return tuple(out)


def max_over_ranks(value: float, device=None) -> float:
    """Slowest rank's time: what the whole-job throughput is computed from."""
    if not _active():
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def evidence(per_rank_value: float = 0.0, device=None) -> dict:
    """What the initialised process group itself says about the run, for the bench line: backend, world size as the
    GROUP reports it (not the environment), the rank ids that answered an all-gather, and one float per rank (bench.py: each rank's
    own ms per step, before the MAX).  A plain single-process run reports backend None, world size 1."""
    import torch
    import torch.distributed as dist
    if not _active():
        return {"backend": None, "world_size": 1, "ranks_reporting": 1, "ranks": [0], "per_rank": [float(per_rank_value)]}
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    mine = torch.tensor([float(dist.get_rank()), float(per_rank_value)], dtype=torch.float64, device=dev)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    ranks = sorted({int(g[0].item()) for g in got})
    return {"backend": dist.get_backend(), "world_size": world, "ranks_reporting": len(ranks), "ranks": ranks,
            "per_rank": [float(g[1].item()) for g in got]}


def all_gather_returns(local_returns):
    """Concatenate per-rank episode-return vectors in rank order (ranks may own different counts)."""
    import torch
    import torch.distributed as dist
    if not _active():
        return local_returns.clone()
    world = dist.get_world_size()
    n = torch.tensor([local_returns.numel()], device=local_returns.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(nmax, device=local_returns.device, dtype=local_returns.dtype)
    pad[: local_returns.numel()] = local_returns
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[: int(s.item())] for o, s in zip(out, sizes)])


class PeriodicReturnsGather:
    """The path's one exchange step (BASELINE configs[4]): every `interval` control steps the ranks all-gather their envs' episode
    returns - for LOGGING only, so it must stay off the critical path of the steps: the returns are snapshotted on the stepping stream (one
    small copy), and the collective is enqueued on a SIDE stream behind that snapshot's event (RCCL collectives order themselves after the
    stream that is current when they are called, so calling it under the side stream keeps the stepping stream out of it) and is never waited
    for until somebody reads `latest()`.  128 KiB per rank at 32 768 envs: ~1 us of wire time over xGMI, tens of us of launch latency.
    On CPU (gloo, the tests) the same calls run without streams.  World size 1: the snapshot is the result."""

    def __init__(self, interval: int = 100, device=None):
        import torch
        self.interval = max(int(interval), 1)
        self.count = 0                      # collectives enqueued so far
        self._pending = None                # (step, work handle or None, gathered tensor, event or None)
        self._latest = None                 # (step, tensor of the whole job's returns in rank order)
        self._cuda = device is not None and torch.device(device).type == "cuda"
        self._side = torch.cuda.Stream(device) if self._cuda else None
        self._device = device

    def maybe(self, step: int, returns_fn):
        """Call after control step number `step` (0-based) on the stepping stream; `returns_fn()` gives this rank's [N_local] returns tensor.
        Enqueues the gather when (step + 1) is a multiple of the interval; returns True when it did."""
        if (step + 1) % self.interval:
            return False
        import torch
        import torch.distributed as dist
        self._drain()
        snap = returns_fn().detach().clone()            # on the stepping stream: the values of exactly this step
        if not _active():
            self._latest = (step, snap)
            self.count += 1
            return True
        world = dist.get_world_size()
        out = torch.empty(world * snap.numel(), dtype=snap.dtype, device=snap.device)
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._device))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                work = dist.all_gather_into_tensor(out, snap, async_op=True)
            snap.record_stream(self._side); out.record_stream(self._side)
        else:
            out = [torch.empty_like(snap) for _ in range(world)]        # (gloo: the list form)
            work = dist.all_gather(out, snap, async_op=True)
        self._pending = (step, work, out)
        self.count += 1
        return True

    def _drain(self):
        if self._pending is None:
            return
        import torch
        step, work, out = self._pending
        if work is not None:
            work.wait()                 # (NCCL: makes the current stream wait for the collective; gloo: blocks until it is done)
        self._latest = (step, torch.cat(list(out)) if isinstance(out, (list, tuple)) else out)
        self._pending = None

    def latest(self):
        """(step, returns of every env of the job in rank order) of the last gather, or None; joins the pending collective."""
        self._drain()
        return self._latest


def run_sharded(make_env, step_fn, envs_per_rank: int, steps: int, device=None, sync=None):
    """The timed region of a weak-scaling run, as bench.py does it: every rank builds its shard with
    `make_env(env_id_base)`, a barrier brackets `steps` calls of `step_fn(env, i)`, the time is the MAX over ranks and
    the value is world * envs_per_rank * steps / time.  Returns (value, elapsed, all episode returns in rank order).
    `sync()` drains the device before the clocks are read (torch.cuda.synchronize on a GPU)."""
    rank, _, world = rank_info()
    env = make_env(shard_base(rank, envs_per_rank))
    if sync:
        sync()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(env, i)
    if sync:
        sync()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, device)
    returns = all_gather_returns(env.episode_returns())
    return world * envs_per_rank * steps / elapsed, elapsed, returns
