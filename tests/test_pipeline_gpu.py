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
