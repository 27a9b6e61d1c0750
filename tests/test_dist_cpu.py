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


def test_combined_roles_and_stage_bounds():
    """BASELINE configs[4] / SURVEY.md 8e row "8": k pipeline ranks + a detector rank + a VAD / ASR rank; every rank has exactly one role
    above two GPUs, the stages cover the layers without gaps"""
    from trackiellm_amd import dist as D
    assert D.combined_roles(1) == {"llm": [0], "vision": [0], "audio": [0]}
    assert D.combined_roles(2) == {"llm": [0], "vision": [1], "audio": [1]}
    assert D.combined_roles(3) == {"llm": [0, 1], "vision": [2], "audio": [2]}
    assert D.combined_roles(4) == {"llm": [0, 1], "vision": [2], "audio": [3]}
    r8 = D.combined_roles(8)
    assert r8 == {"llm": [0, 1, 2, 3, 4, 5], "vision": [6], "audio": [7]}
    for w in (4, 5, 6, 7, 8):
        r = D.combined_roles(w)
        assert sorted(r["llm"] + r["vision"] + r["audio"]) == list(range(w))
    b = D.stage_bounds(32, 6)
    assert b[0] == 0 and b[-1] == 32 and all(y > x for x, y in zip(b, b[1:])) and max(y - x for x, y in zip(b, b[1:])) - min(y - x for x, y in zip(b, b[1:])) <= 1
    assert D.stage_bounds(32, 8) == list(range(0, 33, 4)) and D.stage_bounds(32, 1) == [0, 32]
    with pytest.raises(ValueError):
        D.stage_bounds(4, 5)


class _RecorderPipe:
    """what LibPipeline drives, without a GPU: records the calls"""

    def __init__(self, rank, g, stage, n_stages, l0, l1):
        self.rank, self.g = rank, g
        self.handle_bytes = ("mailbox r%d g%d" % (rank, g)).encode().ljust(80, b".")
        self.meta = (stage, n_stages, l0, l1)
        self.calls = []
        self.links = None

    def connect(self, nxt, prv):
        self.links = (nxt, prv)

    def new_unique_id(self):  # the collective transport: stage 0 names a communicator per row group
        return ("rccl id of r%d g%d" % (self.rank, self.g)).encode().ljust(128, b".")

    def connect_rccl(self, unique_id):
        self.links = ("rccl", unique_id)

    def enqueue(self, seq, pos, tok, head):
        self.calls.append(("pass", list(map(int, seq)), list(map(int, pos)), None if tok is None else list(map(int, tok)), bool(head)))

    def decode(self, nrows, n_steps):
        self.calls.append(("decode", nrows, n_steps))

    def sync(self, nrows, n_steps):
        import numpy as np
        self.calls.append(("sync", nrows, n_steps))
        return np.zeros((n_steps, nrows), np.int32) if n_steps else None


def _combined_worker(rank, world, port, q, rccl=False, groups=2):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from trackiellm_amd import dist as D
    dist = D.init("gloo")
    roles = D.combined_roles(world)
    made = []

    def make(g, stage, n_stages, l0, l1):
        made.append(_RecorderPipe(rank, g, stage, n_stages, l0, l1))
        return made[-1]

    pipe = D.LibPipeline(dist, roles["llm"], 32, groups, make if rank in roles["llm"] else None, rccl=rccl)
    rng = np.random.default_rng(1)
    prompts = [rng.integers(3, 100, (3, 6)).astype(np.int32) for _ in range(groups)]
    out = pipe.generate(prompts, 4, rows_per_pass=8)
    D.barrier(dist, cuda=False)
    q.put((rank, pipe.stage, [(p.meta, p.links, p.calls) for p in made], None if out is None else [o.shape for o in out]))
    dist.destroy_process_group()


def test_world_size_3_combined_job_two_stages_plus_perception_rank():
    """3 ranks over gloo: ranks 0 and 1 are the two LLM stages, rank 2 the perception rank.  The mailbox handles reach the right
    neighbours, both stages enqueue the SAME pass list (tokens only on stage 0), the perception rank owns no stage but takes part in the
    exchange"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + (os.getpid() % 150)
    procs = [ctx.Process(target=_combined_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in range(3))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[2][1] is None and res[2][2] == [] and res[2][3] is None          # the perception rank
    s0, s1 = res[0][2], res[1][2]
    assert [m for m, _, _ in s0] == [(0, 2, 0, 16)] * 2 and [m for m, _, _ in s1] == [(1, 2, 16, 32)] * 2
    for g in range(2):                                                           # two stages: the other one is both next and previous
        assert s0[g][1] == (("mailbox r1 g%d" % g).encode().ljust(80, b"."),) * 2
        assert s1[g][1] == (("mailbox r0 g%d" % g).encode().ljust(80, b"."),) * 2
    strip = lambda calls: [(c[0], c[1], c[2], c[4]) if c[0] == "pass" else c for c in calls]
    for g in range(2):
        assert strip(s0[g][2]) == strip(s1[g][2])                               # the same passes in the same order
        passes0 = [c for c in s0[g][2] if c[0] == "pass"]
        passes1 = [c for c in s1[g][2] if c[0] == "pass"]
        assert all(c[3] is not None for c in passes0) and all(c[3] is None for c in passes1)   # tokens enter at stage 0 only
        assert [c[4] for c in passes0] == [False, False, True]                  # 15 prompt rows in chunks of 8, then the sampling pass
        assert ("decode", 3, 4) in s0[g][2] and s0[g][2][-1] == ("sync", 3, 4)
    assert res[0][3] == [(4, 3), (4, 3)] and res[1][3] == [(4, 3), (4, 3)]


def test_world_size_3_combined_job_over_the_rccl_transport():
    """bench.py --pipe-rccl: stage 0 names the RCCL communicator of the (one) row group, the 128-byte id reaches every stage through the
    same exchange and each stage joins it; the pass lists are the mailbox transport's.  More than one row group is refused: several
    communicators on one device with eager send / recv on separate streams is RCCL's documented deadlock pattern"""
    from trackiellm_amd import dist as D
    with pytest.raises(ValueError, match="one row group"):
        D.LibPipeline(None, [0, 1], 32, 2, lambda *a: _RecorderPipe(0, *a), rccl=True)
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 150)
    procs = [ctx.Process(target=_combined_worker, args=(r, 3, port, q, True, 1)) for r in range(3)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in range(3))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = res[0][2], res[1][2]
    for g in range(1):
        want = ("rccl", ("rccl id of r0 g%d" % g).encode().ljust(128, b"."))
        assert s0[g][1] == want and s1[g][1] == want
        assert [c[0] for c in s0[g][2]] == [c[0] for c in s1[g][2]]
    assert res[2][2] == []


def _handover_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    import numpy as np
    from trackiellm_amd import dist as D
    dist = D.init("gloo")
    ex = D.PerceptionExchange(dist, dst=0, plan=D.perception_plan([1], [2], 4, 16))
    log = []
    for step in range(4):
        if rank == 0:                                   # the LLM's first rank: needs the batch's results before it generates
            try:
                got = ex.require(4, 4)
                log.append((len(got), ex.bytes_last, ex.checksum, [sorted(b.keys()) for b in got], int(got[0]["dets"][0, 0, 2]), int(got[1]["tokens"][0, 0])))
            except RuntimeError:
                log.append("missing before the first hand-over")
            time.sleep(0.3)                             # a slow LLM step: the perception ranks must not be held up by it
            t = time.time()
            assert ex.hand_over(None) is None
            log.append(("post_s", time.time() - t))
        elif rank == 1:                                 # detector rank: 4 frames, the detector wrapper's tuples, one frame over the cap
            dets = [[(7, b"person", 0.9, (1 + step, 2, 30, 40))], [], [(k % 80, b"x", 0.5, (k, k, 5, 5)) for k in range(25)], [(1, b"a", 0.6, (0, 0, 1, 1))] * 2]
            t = time.time()
            ex.hand_over(D.pack_perception(dets, None))
            log.append(time.time() - t)
        else:                                           # VAD + ASR rank: 4 utterances x 16 token ids
            t = time.time()
            ex.hand_over(D.pack_perception(None, np.arange(64, dtype=np.int32).reshape(4, 16) + step))
            log.append(time.time() - t)
    if rank == 1:
        with pytest.raises(ValueError):                 # a message of another size than planned is refused on the sender
            ex.hand_over(D.pack_perception([[]], None))
    ex.finish()
    D.barrier(dist, cuda=False)
    q.put((rank, log))
    dist.destroy_process_group()


def test_perception_results_reach_the_llm_rank_every_step():
    """bench.py --placement model-per-gpu / combined: the detections (<= 20 per frame) and ASR token ids of every batch travel to the LLM's
    first rank as fixed-size point-to-point messages (isend / irecv), it refuses to generate for a batch whose results are not there, and
    neither side's hand_over blocks on the other (world size 3 over gloo)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29250 + (os.getpid() % 150)
    procs = [ctx.Process(target=_handover_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = {r[0]: r[1] for r in (q.get(timeout=120) for _ in range(3))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    log = res[0]
    assert log[0] == "missing before the first hand-over"
    steps = [e for e in log[1:] if e[0] != "post_s"]
    assert len(steps) == 3 and all(s[0] == 2 for s in steps)                     # two perception ranks contributed each time
    assert all(s[3] == [["dets", "n_dets"], ["tokens"]] for s in steps)
    assert steps[0][1] == 4 * 20 * 6 * 4 + 4 * 4 + 4 * 16 * 4                    # frames x 20 x 6 floats + counts + token ids
    assert len({s[2] for s in steps}) == 3                                       # the payload changed every step and was read
    assert [s[4] for s in steps] == [1, 2, 3] and [s[5] for s in steps] == [0, 1, 2]   # batch k's results, in order
    assert all(e[1] < 0.1 for e in log if e[0] == "post_s")                      # posting the receives does not wait for the senders
    # the senders' first two hand-overs (two buffers) return without waiting for the LLM rank's 0.3 s steps
    assert max(res[1][:2]) < 0.15 and max(res[2][:2]) < 0.15


def test_world_size_8_combined_schedule_six_stages_detector_asr():
    """BASELINE configs[4] as SURVEY.md 8e row "8" places it: ranks 0 .. 5 are six LLM stages of 5 - 6 layers, rank 6 the detector, rank 7
    VAD + ASR.  Over gloo with the recording pipe: every stage is linked to its ring neighbours, all six enqueue the same pass list (tokens
    on stage 0 only), the perception ranks own no stage, and their results reach stage 0 through the point-to-point exchange"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150)
    procs = [ctx.Process(target=_combined8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=240) for _ in range(8))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from trackiellm_amd import dist as D
    bounds = D.stage_bounds(32, 6)
    strip = lambda calls: [(c[0], c[1], c[2], c[4]) if c[0] == "pass" else c for c in calls]
    for r in range(6):
        stage, made = res[r][1], res[r][2]
        assert stage == r and [m for m, _, _ in made] == [(r, 6, bounds[r], bounds[r + 1])] * 2
        for g in range(2):
            assert made[g][1] == (("mailbox r%d g%d" % ((r + 1) % 6, g)).encode().ljust(80, b"."), ("mailbox r%d g%d" % ((r - 1) % 6, g)).encode().ljust(80, b"."))
            assert strip(made[g][2]) == strip(res[0][2][g][2])
            assert all((c[3] is not None) == (r == 0) for c in made[g][2] if c[0] == "pass")
    assert res[6][1] is None and res[7][1] is None and res[6][2] == [] and res[7][2] == []
    assert res[0][3] == [(2, 2, [["dets", "n_dets"], ["tokens"]])] * 2            # two steps: detector's 6 frames + the ASR rank's 6 utterances


def _combined8_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    from trackiellm_amd import dist as D
    dist = D.init("gloo")
    roles = D.combined_roles(world)
    made = []

    def make(g, stage, n_stages, l0, l1):
        made.append(_RecorderPipe(rank, g, stage, n_stages, l0, l1))
        return made[-1]

    pipe = D.LibPipeline(dist, roles["llm"], 32, 2, make if rank in roles["llm"] else None)
    cycles = 6
    ex = D.PerceptionExchange(dist, dst=roles["llm"][0], plan=D.perception_plan(roles["vision"], roles["audio"], cycles, 16))
    rng = np.random.default_rng(1)
    prompts = [rng.integers(3, 100, (3, 6)).astype(np.int32) for _ in range(2)]
    seen = []
    for step in range(3):                               # bench.py run_pipeline's step(), with the recording pipe
        res = None
        if rank in roles["llm"]:
            if rank == roles["llm"][0] and step > 0:
                got = ex.require(cycles, cycles)
                seen.append((len(got), sum(len(b.get("n_dets", [])) for b in got) // 3, [sorted(b.keys()) for b in got]))
            pipe.generate(prompts, 4, rows_per_pass=8)
        elif rank in roles["vision"]:
            res = D.pack_perception([[(3, b"car", 0.7, (step, 1, 2, 3))]] * cycles, None)
        else:
            res = D.pack_perception(None, np.full((cycles, 16), step, np.int32))
        ex.hand_over(res)
    ex.finish()
    D.barrier(dist, cuda=False)
    q.put((rank, pipe.stage, [(p.meta, p.links, p.calls[:4]) for p in made], seen))
    dist.destroy_process_group()


