"""The LLM layer-sharded over ranks: stage execution is bit-identical to the unsplit pass, and the in-library hand-off
(device mailboxes, csrc/llm/tk_llm_pipe.h) — three stages in one process, two processes over hipIpc on the one GPU of the test box —
generates the ids of a single session, generation after generation on the same pipes; a missing peer fails fast."""
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


def test_second_generation_on_the_same_pipes_takes_fresh_ids(gpu):
    """a generation sends one id message more than stage 0 takes (the last step's sample); the next generation's decode loop
    must start from ITS prompt's sample, not from that leftover.  Three generations with different prompts on one set of pipes, each
    compared with a single session; the third continues decoding with a second decode() call (the FIFO hands the last sample on)."""
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    hp = model.hparams
    bounds = [0, 2, 4]
    sess = [gpu.LlmSession(model, 3, 48) for _ in range(2)]
    pipes = [gpu.LlmPipe(sess[s], s, 2, bounds[s], bounds[s + 1]) for s in range(2)]
    for s in range(2):
        pipes[s].connect_local(pipes[1 - s], pipes[1 - s])
    rng = np.random.default_rng(11)
    for gen, (n_prompt, n_dec) in enumerate(((6, 5), (9, 7), (4, 6))):
        prompts = rng.integers(3, hp.vocab, (3, n_prompt)).astype(np.int32)
        for s in range(2):
            _drive_stage(pipes[s], prompts, n_dec, rows_per_pass=4)
        fed = pipes[0].sync(3, n_dec)
        sampled = pipes[1].sync(3, n_dec)
        want = _single_session_reference(gpu, model, [prompts], n_dec + 3)[0]
        assert np.array_equal(sampled, want[1:n_dec + 1]), "generation %d: samples differ from a single session's" % gen
        assert np.array_equal(fed, want[:n_dec]), "generation %d: stage 0 fed stale ids" % gen
    for s in range(2):                                                # continuation: three more steps of the last generation
        pipes[s].decode(3, 3)
    fed = pipes[0].sync(3, 3)
    sampled = pipes[1].sync(3, 3)
    assert np.array_equal(sampled, want[n_dec + 1:n_dec + 4]) and np.array_equal(fed, want[n_dec:n_dec + 3])
    for p in pipes:
        p.close()
    for s in sess:
        s.close()


def test_ten_generations_on_one_set_of_pipes(gpu):
    """the id-mailbox bookkeeping is stage 0's alone — on the other stages the count of outstanding sampled ids only
    ever grew, and the 8th prompt's sampling pass of a long-lived pipe was refused there while stage 0 carried on and then timed out.  Ten
    generations on one connect_local set of three stages, each compared with a single session."""
    hp = gpu.TINY()
    hp.n_layer = 3
    model = gpu.LlmModel(hp).fill_synthetic(37)
    hp = model.hparams
    sess = [gpu.LlmSession(model, 2, 40) for _ in range(3)]
    pipes = [gpu.LlmPipe(sess[s], s, 3, s, s + 1) for s in range(3)]
    for s in range(3):
        pipes[s].connect_local(pipes[(s + 1) % 3], pipes[(s - 1) % 3])
    rng = np.random.default_rng(12)
    for gen in range(10):
        n_prompt, n_dec = 3 + gen % 4, 2 + gen % 3
        prompts = rng.integers(3, hp.vocab, (2, n_prompt)).astype(np.int32)
        for s in range(3):
            _drive_stage(pipes[s], prompts, n_dec, rows_per_pass=4)
        fed = pipes[0].sync(2, n_dec)
        pipes[1].sync(2, 0)
        sampled = pipes[2].sync(2, n_dec)
        want = _single_session_reference(gpu, model, [prompts], n_dec + 1)[0]
        assert np.array_equal(sampled, want[1:n_dec + 1]), "generation %d: samples differ from a single session's" % gen
        assert np.array_equal(fed, want[:n_dec]), "generation %d: stage 0 fed stale ids" % gen
    for p in pipes:
        p.close()
    for s in sess:
        s.close()


def test_pipelined_decode_takes_the_long_context_form_past_512_positions(gpu):
    """a pipe stage cached ONE decode graph per row count, captured with whatever attention form the first capture met, so a pipe
    first captured on a short prompt never took the long-context form (scores / PV chains spread over the chip) past 512 positions.  The stage now counts
    the position on the host (last pass + steps) and keeps a graph per (form, row count).  Two stages of one Mistral-shaped layer each, one sequence:
    a short generation first (captures the fused form), then a 500-token prompt decoded across position 512 — ids equal a single session's, whose
    decode loop switches forms itself; both forms are bit-identical, so equality does not depend on which one ran: the plan at the positions the loop
    crosses is asserted from the launcher's own answer."""
    hp = gpu.MISTRAL_7B()
    hp.n_layer = 2
    model = gpu.LlmModel(hp).fill_synthetic(4)
    hp = model.hparams
    CTX = 560
    assert gpu.attention_plan(1, hp.n_head, hp.n_kv_head, hp.head_dim, CTX, True, top_position=500)[0] != 3
    assert gpu.attention_plan(1, hp.n_head, hp.n_kv_head, hp.head_dim, CTX, True, top_position=520)[0] == 3
    sess = [gpu.LlmSession(model, 1, CTX) for _ in range(2)]
    pipes = [gpu.LlmPipe(sess[s], s, 2, s, s + 1) for s in range(2)]
    for s in range(2):
        pipes[s].connect_local(pipes[1 - s], pipes[1 - s])
    rng = np.random.default_rng(13)
    ref = gpu.LlmSession(model, 1, CTX)
    for n_prompt, n_dec in ((6, 4), (500, 30)):
        prompts = rng.integers(3, hp.vocab, (1, n_prompt)).astype(np.int32)
        for s in range(2):
            _drive_stage(pipes[s], prompts, n_dec, rows_per_pass=256)
        fed = pipes[0].sync(1, n_dec)
        sampled = pipes[1].sync(1, n_dec)
        first = ref.prefill(prompts)
        toks, _ = ref.decode(1, n_dec)
        want = np.concatenate([first[None, :], toks])
        assert np.array_equal(sampled, want[1:n_dec + 1]) and np.array_equal(fed, want[:n_dec]), (n_prompt, n_dec)
    for p in pipes:
        p.close()
    for s_ in sess + [ref]:
        s_.close()
    model.close()


@pytest.mark.timeout(120)
def test_pipe_wait_is_bounded_fails_fast_and_stays_failed(gpu, monkeypatch):
    """a stage whose producer never publishes must not hang: the first device-side wait gives up after the pipe's timeout
    ($TK_MI355X_PIPE_TIMEOUT_S = 1 s here), every wait still enqueued behind it — the rest of the prompt, 40 graph replays — returns at
    once, sync() reports TK_ERROR_TIMEOUT (1005) within seconds, and the pipe stays failed."""
    import time
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    sess = [gpu.LlmSession(model, 2, 64) for _ in range(2)]
    with pytest.raises(gpu.TkError):
        gpu.LlmPipe(sess[0], 0, 2, 1, 2)                              # stage 0 must start at layer 0
    with pytest.raises(gpu.TkError):
        gpu.LlmPipe(sess[0], 1, 2, 2, 3)                              # the last stage must end at the last layer
    monkeypatch.setenv("TK_MI355X_PIPE_TIMEOUT_S", "1")
    p0 = gpu.LlmPipe(sess[0], 0, 2, 0, 2)
    p1 = gpu.LlmPipe(sess[1], 1, 2, 2, 4)
    with pytest.raises(gpu.TkError):
        p0.enqueue([0], [0], [5])                                     # not connected
    with pytest.raises(gpu.TkError):
        p0.connect(p0.handle, p0.handle)                              # a handle of this very process: connect_local is the way
    p0.connect_local(p1, p1)
    p1.connect_local(p0, p0)
    prompts = np.random.default_rng(3).integers(3, model.hparams.vocab, (2, 6)).astype(np.int32)
    t0 = time.time()
    _drive_stage(p1, prompts, 40, rows_per_pass=4)                    # stage 1 alone: stage 0 never enqueues anything
    with pytest.raises(gpu.TkError) as ei:
        p1.sync(2, 40)
    took = time.time() - t0
    assert ei.value.code == 1005 and took < 20.0, (ei.value.code, took)   # one timeout, not one per wait (44 waits x 1 s)
    with pytest.raises(gpu.TkError):
        p1.decode(2, 1)                                               # failed for good
    with pytest.raises(gpu.TkError):
        p1.sync()
    for p in (p0, p1):
        p.close()
    for s in sess:
        s.close()


def _rccl_stage_worker(stage, conn, device):
    sys.path.insert(0, ROOT)
    import trackiellm_amd as tk
    tk.lib().tk_mi355x_set_default_device(device)
    hp = tk.TINY()
    hp.n_layer = 4
    model = tk.LlmModel(hp, device=device).fill_synthetic(31)
    hp = model.hparams
    prompts = np.random.default_rng(7).integers(3, hp.vocab, (5, 12)).astype(np.int32)
    sess = tk.LlmSession(model, 5, 48)
    bounds = [0, 2, 4]
    pipe = tk.LlmPipe(sess, stage, 2, bounds[stage], bounds[stage + 1])
    if stage == 0:
        conn.send(tk.LlmPipe.rccl_unique_id())
    uid = conn.recv()
    pipe.connect_rccl(uid)                                            # both ranks enter ncclCommInitRank together
    for gen in range(2):                                              # two generations on one communicator: the id messages stay paired
        _drive_stage(pipe, prompts, 9, rows_per_pass=4)
        toks = pipe.sync(5, 9)
    conn.send(toks.tolist())
    conn.recv()
    pipe.close()
    sess.close()


def test_rccl_transport_needs_one_gpu_per_stage_and_says_so(gpu):
    """the collective form of the stage hand-off (ncclSend / ncclRecv; north_star: "RCCL over xGMI only for the LLM shard";
    tk_mi355x_pipe_connect_rccl).  With >= 2 visible devices: two processes, one stage per device, two generations — the last stage's ids
    equal a single session's (the mailbox transport's cross-check on a real link).  With fewer — the GPU boxes of this build — selecting the
    transport must FAIL loudly, never fall back to the mailboxes: that refusal is what is asserted there."""
    import ctypes as C
    ndev = gpu.lib().tk_mi355x_device_count()
    hp = gpu.TINY()
    hp.n_layer = 4
    model = gpu.LlmModel(hp).fill_synthetic(31)
    if ndev < 2:
        sess = gpu.LlmSession(model, 5, 48)
        pipe = gpu.LlmPipe(sess, 0, 2, 0, 2)
        uid = gpu.LlmPipe.rccl_unique_id()                            # librccl itself loads and answers
        assert len(uid) == 128 and any(uid)
        with pytest.raises(gpu.TkError) as e:
            pipe.connect_rccl(uid)
        assert "one GPU per stage" in e.value.detail and "1 device(s) visible" in e.value.detail, e.value.detail
        with pytest.raises(gpu.TkError):                              # and the pipe stays unconnected: nothing silently took another path
            pipe.enqueue(np.zeros(1, np.int32), np.zeros(1, np.int32), np.ones(1, np.int32))
        pipe.close()
        sess.close()
        model.close()
        return
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    ends = [ctx.Pipe() for _ in range(2)]
    procs = [ctx.Process(target=_rccl_stage_worker, args=(s, ends[s][1], s)) for s in range(2)]
    for p in procs:
        p.start()
    uid = ends[0][0].recv()
    for s in range(2):
        ends[s][0].send(uid)
    toks = []
    for s in range(2):
        assert ends[s][0].poll(300), "stage %d did not finish" % s
        toks.append(np.array(ends[s][0].recv()))
    for s in range(2):
        ends[s][0].send("bye")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    prompts = np.random.default_rng(7).integers(3, model.hparams.vocab, (5, 12)).astype(np.int32)
    want = _single_session_reference(gpu, model, [prompts], 9)[0]
    assert np.array_equal(toks[1], want[1:]) and np.array_equal(toks[0], want[:9])
