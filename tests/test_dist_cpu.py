"""CPU, world_size 2, gloo: the N > 1 path of bench.py (barrier, max-over-ranks timing, disjoint cycle shards)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from trackiellm_amd import dist as D
    dist = D.init("gloo")
    D.barrier(dist, cuda=False)
    elapsed = 1.0 + 0.5 * rank                      # rank 1 is the slow one
    mx = D.max_over_ranks(dist, elapsed, cuda=False)
    seeds = D.cycle_seeds(rank, 16)
    value = D.aggregate_throughput(16, 3, world, mx)
    D.barrier(dist, cuda=False)
    q.put((rank, mx, seeds, value))
    dist.destroy_process_group()


def test_two_rank_gloo_aggregation():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] == out[1][1] == 1.5                          # MAX over ranks, seen by both
    assert set(out[0][2]).isdisjoint(out[1][2]) and len(out[0][2]) + len(out[1][2]) == 32
    assert out[0][3] == pytest.approx(16 * 3 * 2 / 1.5)           # whole-job aggregate, not per GPU


def test_model_per_gpu_roles_cover_every_stream():
    """bench.py --placement model-per-gpu: rank 0 is the LLM, every other rank has a perception role, both
    perception streams are served at every size"""
    import bench
    assert bench.model_per_gpu_roles(2) == ([1], [1])
    assert bench.model_per_gpu_roles(4) == ([1, 3], [2])
    vis, aud = bench.model_per_gpu_roles(8)
    assert vis == [1, 3, 5, 7] and aud == [2, 4, 6] and 0 not in vis + aud and sorted(vis + aud) == list(range(1, 8))
