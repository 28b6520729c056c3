"""Data-parallel plumbing: the denoising path shards by independent clips (scenes) and has NO collective inside a step
(SURVEY.md §8(e)).  One process per GPU; torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU
tests) is used only to start together, to reduce wall-clock time with MAX and to gather per-rank clip counts."""
import os

import torch


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, world, local_rank, dist or None).  A
    single process needs no process group (dist = None) unless `force` asks for one — a one-rank RCCL group, which is how the
    collective code paths get exercised on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and not force:
        return rank, world, local, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local, dist


def shard_clips(n_clips, rank, world):
    """Clip indices owned by `rank`: round-robin, so any prefix of the clip list is balanced."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_clips, world))


def clip_seed(base_seed, clip_index):
    """Per-clip RNG seed: a clip's noise depends on WHICH clip it is, never on which rank or how many ranks run —
    so outputs are identical under any sharding."""
    return (int(base_seed) * 1_000_003 + int(clip_index)) % (2 ** 63 - 1)


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(value, dist, device="cpu"):
    """Wall-clock of the slowest rank."""
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device="cpu"):
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
