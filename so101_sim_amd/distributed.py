"""Multi-GPU sharding of independent environments: contiguous global env-id ranges per rank, no
data-path collective; one all-gather of episode returns (RCCL over xGMI on GPUs, gloo in CPU tests)
for logging only."""
from __future__ import annotations


def shard_range(n_global: int, world_size: int, rank: int) -> tuple[int, int]:
    """[start, stop) of the global env ids owned by `rank`; remainders go to the lowest ranks."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_global), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_returns(local_returns):
    """Concatenate per-rank episode-return vectors in rank order (ranks may own different counts)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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
