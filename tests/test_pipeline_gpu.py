"""The LLM layer-sharded over ranks: stage execution is bit-identical to the unsplit pass, and a 2-rank pipeline
(two processes sharing the one GPU of the test box, gloo transport through host memory — on an 8-GPU node the same code moves the
stream with RCCL send / recv) generates the ids of a single session."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stage_split_is_bit_identical_to_one_pass(gpu):
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    hp = model.hparams
    whole = gpu.LlmSession(model, 3, 32)
    split = gpu.LlmSession(model, 3, 32)
    rng = np.random.default_rng(0)
    for n, seq, pos in ((7, [0] * 4 + [1] * 3, [0, 1, 2, 3, 0, 1, 2]), (3, [0, 1, 2], [4, 3, 0]), (40, [2] * 20 + [0] * 10 + [1] * 10,
                        list(range(1, 21)) + list(range(5, 15)) + list(range(4, 14)))):
        tok = rng.integers(3, hp.vocab, n).astype(np.int32)
        _, want = whole.forward(seq, pos, tok, want_logits=False)
        x1 = np.empty((n, hp.d_model), np.float32)
        x2 = np.empty((n, hp.d_model), np.float32)
        split.forward_stage(seq, pos, 0, 1, tok=tok, x_out=x1)            # stage 0: embed + layer 0
        split.forward_stage(seq, pos, 1, 3, x_in=x1, x_out=x2)            # stage 1: layers 1, 2
        got = split.forward_stage(seq, pos, 3, 4, x_in=x2, head=True)     # stage 2: layer 3 + head
        assert np.array_equal(got, want)
    with pytest.raises(gpu.TkError):
        split.forward_stage([0], [9], 0, 2, tok=[5], head=True)           # the head belongs to the last layer
    with pytest.raises(gpu.TkError):
        split.forward_stage([0], [9], 0, 2, tok=[5])                      # no head and nowhere to put the stream


def _pipeline_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import trackiellm_amd as tk
    from trackiellm_amd import dist as D
    dist = D.init("gloo")
    hp = tk.TINY()
    hp.n_layer = 4
    model = tk.LlmModel(hp).fill_synthetic(31)
    hp = model.hparams
    sess = tk.LlmSession(model, 6, 32)
    pipe = D.LlmPipeline(dist, sess, hp.n_layer, hp.d_model, cuda_tensors=False)
    rng = np.random.default_rng(5)
    prompts = [rng.integers(3, hp.vocab, (2, 5)).astype(np.int32) for _ in range(3)]
    out = pipe.generate(prompts, 6, rows_per_pass=3)
    q.put((rank, (pipe.l0, pipe.l1), [o.tolist() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pipeline_matches_single_session(gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 150)
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] == (0, 2) and res[1][1] == (2, 4)
    assert res[0][2] == res[1][2]                     # stage 0 got every id back from the last stage
    # single session, same model / prompts: prefill + greedy decode per group
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    hp = model.hparams
    rng = np.random.default_rng(5)
    prompts = [rng.integers(3, hp.vocab, (2, 5)).astype(np.int32) for _ in range(3)]
    for gi, pr in enumerate(prompts):
        sess = gpu.LlmSession(model, 2, 32)
        first = sess.prefill(pr)
        toks, _ = sess.decode(2, 5)
        want = np.concatenate([first[None, :], toks])
        assert np.array_equal(np.array(res[0][2][gi]), want), gi
        sess.close()


# ---- the stage hand-off inside the library (csrc/llm/tk_llm_pipe.h): device mailboxes, no host synchronisation per pass, graph replays ----

def _single_session_reference(gpu, model, prompts, n_dec):
    out = []
    for pr in prompts:
        sess = gpu.LlmSession(model, pr.shape[0], 48)
        first = sess.prefill(pr)
        toks, _ = sess.decode(pr.shape[0], n_dec)
        out.append(np.concatenate([first[None, :], toks]))
        sess.close()
    return out


def _drive_stage(pipe, prompts_of_group, n_dec, rows_per_pass, part="all"):
    """what every stage enqueues for ONE row group, in the order all stages share: prompt chunks (no sampling), the sampling pass, the decode loop"""
    g = prompts_of_group
    nseq, n_prompt = g.shape
    if part in ("all", "prompt"):
        seq = np.repeat(np.arange(nseq, dtype=np.int32), n_prompt - 1)
        pos = np.tile(np.arange(n_prompt - 1, dtype=np.int32), nseq)
        tok = g[:, :-1].reshape(-1)
        for i in range(0, len(seq), rows_per_pass):
            pipe.enqueue(seq[i:i + rows_per_pass], pos[i:i + rows_per_pass], tok[i:i + rows_per_pass])
        pipe.enqueue(np.arange(nseq, dtype=np.int32), np.full(nseq, n_prompt - 1, np.int32), g[:, -1], head=True)
    if part in ("all", "decode"):
        pipe.decode(nseq, n_dec)


@pytest.mark.parametrize("f16_payload", [False, True])
def test_in_library_handoff_three_stages_in_one_process(gpu, f16_payload):
    """three stages (layers 0-0, 1-2, 3-3 of a 4-layer model) as three sessions + pipes of one process on the one GPU, linked by pointer:
    prompt chunks that run several messages ahead (the credit path), a sampling pass, a 9-step decode loop of graph replays — with NO host
    synchronisation between enqueue and the final sync.  fp32 payload: ids identical to a single session (bit-exact stream);
    f16 payload: the stream is rounded at each boundary, the run completes and the ids are valid tokens."""
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    hp = model.hparams
    rng = np.random.default_rng(7)
    prompts = rng.integers(3, hp.vocab, (5, 12)).astype(np.int32)     # 5 sequences x 11 prompt rows = 55 rows in chunks of 4: 14 messages > 8 slots
    n_dec = 9
    bounds = [0, 1, 3, 4]
    sess = [gpu.LlmSession(model, 5, 48) for _ in range(3)]
    pipes = [gpu.LlmPipe(sess[s], s, 3, bounds[s], bounds[s + 1], payload_f16=f16_payload) for s in range(3)]
    for s in range(3):
        pipes[s].connect_local(pipes[(s + 1) % 3], pipes[(s + 2) % 3])
    for part in ("prompt", "decode"):                                 # every call returns at once: the GPU does the waiting
        for s in range(3):
            _drive_stage(pipes[s], prompts, n_dec, rows_per_pass=4, part=part)
    fed = pipes[0].sync(5, n_dec)                                     # stage 0: the ids it fed at each step
    pipes[1].sync()
    sampled = pipes[2].sync(5, n_dec)                                 # last stage: the ids sampled at each step
    want = _single_session_reference(gpu, model, [prompts], n_dec)[0]  # [1 + n_dec][5]: first token, then the decode steps
    if not f16_payload:
        assert np.array_equal(sampled, want[1:]), "last stage's samples differ from a single session's"
        assert np.array_equal(fed, want[:n_dec]), "stage 0 fed other ids than the last stage sampled"
    else:
        assert sampled.min() >= 0 and sampled.max() < hp.vocab and np.array_equal(fed[1:], sampled[:-1])
    for p in pipes:
        p.close()
    for s in sess:
        s.close()


def _ipc_stage_worker(stage, conn, f16_payload):
    sys.path.insert(0, ROOT)
    import trackiellm_amd as tk
    hp = tk.TINY()
    hp.n_layer = 4
    model = tk.LlmModel(hp).fill_synthetic(31)
    hp = model.hparams
    rng = np.random.default_rng(7)
    prompts = rng.integers(3, hp.vocab, (5, 12)).astype(np.int32)
    sess = tk.LlmSession(model, 5, 48)
    bounds = [0, 2, 4]
    pipe = tk.LlmPipe(sess, stage, 2, bounds[stage], bounds[stage + 1], payload_f16=f16_payload)
    conn.send(pipe.handle.to_bytes())                                 # 80 plain bytes: any channel will do (torch.distributed on a real node)
    other = tk.PipeHandle.from_bytes(conn.recv())
    pipe.connect(other, other)                                        # two stages: the one neighbour is both next and previous
    _drive_stage(pipe, prompts, 9, rows_per_pass=4)
    toks = pipe.sync(5, 9)
    conn.send(toks.tolist())
    conn.recv()                                                       # keep the mailbox mapped until the peer is done with it
    pipe.close()
    sess.close()


def test_in_library_handoff_two_processes_ipc_mapped_mailboxes(gpu):
    """one process per stage, mailboxes exchanged as hipIpc handles and mapped into the peer (the path a node with one process per GPU
    takes; here both processes share the test box's one GPU): ids equal a single session's, bit for bit"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    ends = [ctx.Pipe() for _ in range(2)]
    procs = [ctx.Process(target=_ipc_stage_worker, args=(s, ends[s][1], False)) for s in range(2)]
    for p in procs:
        p.start()
    handles = [ends[s][0].recv() for s in range(2)]
    ends[0][0].send(handles[1])
    ends[1][0].send(handles[0])
    toks = []
    for s in range(2):
        assert ends[s][0].poll(240), "stage %d did not finish" % s
        toks.append(np.array(ends[s][0].recv()))
    for s in range(2):
        ends[s][0].send("bye")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    prompts = np.random.default_rng(7).integers(3, model.hparams.vocab, (5, 12)).astype(np.int32)
    want = _single_session_reference(gpu, model, [prompts], 9)[0]
    assert np.array_equal(toks[1], want[1:]) and np.array_equal(toks[0], want[:9])


def test_pipe_wait_is_bounded_and_reports_a_missing_peer(gpu):
    """a stage whose producer never publishes must not hang: the device-side wait gives up (TK_PIPE_TIMEOUT_S) and sync() returns
    TK_ERROR_TIMEOUT — exercised through argument errors only here (the 20 s wait itself is not spent in the suite)"""
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    sess = gpu.LlmSession(model, 2, 16)
    with pytest.raises(gpu.TkError):
        gpu.LlmPipe(sess, 0, 2, 1, 2)                                 # stage 0 must start at layer 0
    with pytest.raises(gpu.TkError):
        gpu.LlmPipe(sess, 1, 2, 2, 3)                                 # the last stage must end at the last layer
    p = gpu.LlmPipe(sess, 0, 2, 0, 2)
    with pytest.raises(gpu.TkError):
        p.enqueue([0], [0], [5])                                      # not connected
    with pytest.raises(gpu.TkError):
        p.connect(p.handle, p.handle)                                 # a handle of this very process: connect_local is the way
    p.close()
    sess.close()
