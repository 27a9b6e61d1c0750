"""Multi-GPU plumbing of bench.py: ranks are independent replicas of the fused cycle batch.

No data-path collective exists on this path (SURVEY.md §8e "replicas"): torch.distributed is used only to line the ranks
up (barrier) and to take the max elapsed time over ranks.  Backend "nccl" (= RCCL on ROCm) on the GPU box, "gloo" in
the CPU tests."""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, local_rank=0):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def barrier(dist, cuda):
    if cuda:
        import torch
        torch.cuda.synchronize()
    dist.barrier()
    if cuda:
        import torch
        torch.cuda.synchronize()


def max_over_ranks(dist, value, cuda):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device="cuda" if cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cycle_seeds(rank, batch):
    """disjoint synthetic-input streams per rank: global cycle ids [rank*batch, (rank+1)*batch)"""
    return list(range(rank * batch, (rank + 1) * batch))


def aggregate_throughput(cycles_per_rank, steps, world, elapsed_max):
    """whole-job value: every rank processed cycles_per_rank * steps cycles in the (max) elapsed time"""
    return cycles_per_rank * steps * world / elapsed_max
