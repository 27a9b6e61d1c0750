"""GPU parity: the HIP LLM path (through the C-ABI) against the oracle — bit-exact logits and token ids."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def oracle_cfg_from(hp, max_ctx, max_seq):
    return O.LlmConfig(n_layer=hp.n_layer, d_model=hp.d_model, n_head=hp.n_head, n_kv_head=hp.n_kv_head, head_dim=hp.head_dim,
                       d_ff=hp.d_ff, vocab=hp.vocab, max_ctx=max_ctx, max_seq=max_seq, rms_eps=hp.rms_eps, rope_theta=hp.rope_theta,
                       ks_qkv=hp.ks_qkv, ks_o=hp.ks_o, ks_gateup=hp.ks_gateup, ks_down=hp.ks_down, ks_out=hp.ks_out)


def copy_oracle_weights(orc, model, n_layer):
    for which in (O.T_TOKEN_EMBD, O.T_OUT_NORM, O.T_OUTPUT):
        t, buf = orc.get_tensor(-1, which)
        model.set_tensor(-1, which, t, buf)
    for l in range(n_layer):
        for which in range(9):
            t, buf = orc.get_tensor(l, which)
            model.set_tensor(l, which, t, buf)


@pytest.fixture(scope="module")
def tiny(gpu):
    hp = gpu.TINY()
    model = gpu.LlmModel(hp)
    hp = model.hparams  # with the K-split plan filled in
    orc = O.OracleLlm(oracle_cfg_from(hp, 64, 4), seed=4)
    copy_oracle_weights(orc, model, hp.n_layer)
    sess = gpu.LlmSession(model, 4, 64)
    return gpu, model, sess, orc, hp


def test_tiny_prefill_rows_bit_exact(tiny):
    gpu, model, sess, orc, hp = tiny
    rng = np.random.default_rng(3)
    toks = rng.integers(3, hp.vocab, 16).astype(np.int32)
    seq = np.zeros(16, np.int32)
    pos = np.arange(16, dtype=np.int32)
    orc.reset()
    want, want_am = orc.forward(seq, pos, toks)
    got, got_am = sess.forward(seq, pos, toks)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(got_am, want_am)


def test_tiny_ragged_multi_sequence(tiny):
    """rows from different sequences at different positions, fewer than 16 rows, nrows = 1"""
    gpu, model, sess, orc, hp = tiny
    orc.reset()
    rng = np.random.default_rng(5)
    # build three sequences of lengths 5, 1, 9 incrementally in mixed passes
    plan = [([0, 1, 2], [0, 0, 0]), ([0, 2, 2, 2], [1, 1, 2, 3]), ([0], [2]), ([0, 0, 2, 2, 2, 2, 2], [3, 4, 4, 5, 6, 7, 8])]
    for seq, pos in plan:
        tok = rng.integers(3, hp.vocab, len(seq)).astype(np.int32)
        want, wam = orc.forward(seq, pos, tok)
        got, gam = sess.forward(seq, pos, tok)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(gam, wam)


def test_tiny_two_mtiles_24_rows(tiny):
    """more than 16 rows per pass: second MFMA M-tile (rows 16..31), ragged over three sequences"""
    gpu, model, sess, orc, hp = tiny
    orc.reset()
    rng = np.random.default_rng(17)
    seq = np.array([0] * 10 + [1] * 9 + [3] * 5, np.int32)
    pos = np.concatenate([np.arange(10), np.arange(9), np.arange(5)]).astype(np.int32)
    tok = rng.integers(3, hp.vocab, 24).astype(np.int32)
    want, wam = orc.forward(seq, pos, tok)
    got, gam = sess.forward(seq, pos, tok)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(gam, wam)
    # then 32 decode rows would need 32 sequences: continue the three sequences one step each + 17th..: mixed 20-row pass
    seq2 = np.array([0, 1, 3], np.int32)
    pos2 = np.array([10, 9, 5], np.int32)
    tok2 = rng.integers(3, hp.vocab, 3).astype(np.int32)
    w2, _ = orc.forward(seq2, pos2, tok2)
    g2, _ = sess.forward(seq2, pos2, tok2)
    assert np.array_equal(g2.view(np.uint32), w2.view(np.uint32))


def test_tiny_greedy_decode_ids(tiny):
    gpu, model, sess, orc, hp = tiny
    orc.reset()
    rng = np.random.default_rng(11)
    B, P, N = 3, 6, 24
    prompts = rng.integers(3, hp.vocab, (B, P)).astype(np.int32)
    first = sess.prefill(prompts)
    toks, ms = sess.decode(B, N)
    # oracle: same schedule
    for s in range(B):
        orc.forward(np.full(P - 1, s, np.int32), np.arange(P - 1, dtype=np.int32), prompts[s, :P - 1], want_logits=False)
    _, cur = orc.forward(np.arange(B, dtype=np.int32), np.full(B, P - 1, np.int32), prompts[:, P - 1], want_logits=False)
    assert np.array_equal(first, cur)
    for i in range(N):
        _, cur = orc.forward(np.arange(B, dtype=np.int32), np.full(B, P + i, np.int32), cur, want_logits=False)
        assert np.array_equal(toks[i], cur), f"step {i}"


def test_synthetic_fill_matches_oracle_generator(gpu):
    """weights generated + quantised on the GPU are the oracle's weights: logits bit-exact"""
    model = gpu.LlmModel(gpu.TINY()).fill_synthetic(9)
    hp = model.hparams
    sess = gpu.LlmSession(model, 2, 32)
    orc = O.OracleLlm(oracle_cfg_from(hp, 32, 2), seed=9)
    tok = np.array([7, 300, 12, 44, 259], np.int32)
    want, _ = orc.forward(np.zeros(5, np.int32), np.arange(5, dtype=np.int32), tok)
    got, _ = sess.forward(np.zeros(5, np.int32), np.arange(5, dtype=np.int32), tok)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_mistral_shaped_layer_bit_exact(gpu):
    """one Mistral-7B-shaped layer (full 4096/14336/32000 geometry, production K-split plan), 16 rows"""
    hp = gpu.MISTRAL_7B()
    hp.n_layer = 1
    model = gpu.LlmModel(hp).fill_synthetic(4)
    hp = model.hparams
    assert (hp.ks_qkv, hp.ks_o, hp.ks_gateup, hp.ks_down) == (4, 4, 1, 7)
    sess = gpu.LlmSession(model, 16, 32)
    orc = O.OracleLlm(oracle_cfg_from(hp, 32, 16), seed=4)
    rng = np.random.default_rng(1)
    seq = np.arange(16, dtype=np.int32)
    for p in range(3):
        tok = rng.integers(3, hp.vocab, 16).astype(np.int32)
        want, wam = orc.forward(seq, np.full(16, p, np.int32), tok)
        got, gam = sess.forward(seq, np.full(16, p, np.int32), tok)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
        assert np.array_equal(gam, wam)


@pytest.mark.parametrize("nrows", [33, 48, 64, 65, 100, 128, 129, 161, 200, 255, 256])  # 129..256: k_gemm32_w4a8, the second row half partly or barely filled
def test_tiny_batched_pass_rows(gpu, nrows):
    """passes of more than 32 rows take the K-streamed batched kernel (4, 8 or 16 M-tiles per weight tile, folded Q6_K):
    ragged rows over several sequences, then one decode row per sequence — bit-exact logits"""
    hp = gpu.TINY()
    model = gpu.LlmModel(hp).fill_synthetic(21)
    hp = model.hparams
    nseq = 7
    sess = gpu.LlmSession(model, nseq, 64)  # 256 rows over 7 sequences: up to 37 positions each
    orc = O.OracleLlm(oracle_cfg_from(hp, 64, nseq), seed=21)
    rng = np.random.default_rng(nrows)
    lens = np.full(nseq, nrows // nseq)
    lens[: nrows - lens.sum()] += 1
    seq = np.concatenate([np.full(n, s) for s, n in enumerate(lens)]).astype(np.int32)
    pos = np.concatenate([np.arange(n) for n in lens]).astype(np.int32)
    tok = rng.integers(3, hp.vocab, nrows).astype(np.int32)
    want, wam = orc.forward(seq, pos, tok)
    got, gam = sess.forward(seq, pos, tok)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(gam, wam)
    seq2 = np.arange(nseq, dtype=np.int32)
    tok2 = rng.integers(3, hp.vocab, nseq).astype(np.int32)
    w2, _ = orc.forward(seq2, lens.astype(np.int32), tok2)
    g2, _ = sess.forward(seq2, lens.astype(np.int32), tok2)
    assert np.array_equal(g2.view(np.uint32), w2.view(np.uint32))


def test_mistral_shaped_layer_batched_bit_exact(gpu):
    """the Mistral-7B-shaped layer again through the batched kernel: 40 rows (4 M-tiles), 128 rows (8) and 256 rows (16 M-tiles: the
    mixed q / k / v matrix goes as one launch per tensor type), Q4_K and Q6_K tensors"""
    hp = gpu.MISTRAL_7B()
    hp.n_layer = 1
    model = gpu.LlmModel(hp).fill_synthetic(4)
    hp = model.hparams
    rng = np.random.default_rng(2)
    for nrows in (40, 128, 256):
        sess = gpu.LlmSession(model, nrows, 8)
        orc = O.OracleLlm(oracle_cfg_from(hp, 8, nrows), seed=4)
        seq = np.arange(nrows, dtype=np.int32)
        for p in range(2):
            tok = rng.integers(3, hp.vocab, nrows).astype(np.int32)
            want, wam = orc.forward(seq, np.full(nrows, p, np.int32), tok)
            got, gam = sess.forward(seq, np.full(nrows, p, np.int32), tok)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
            assert np.array_equal(gam, wam)
        sess.close()


@pytest.mark.parametrize("nrows", [1, 2])
def test_mistral_shaped_one_and_two_row_passes_form_their_inputs_in_the_matvec_launch(gpu, nrows, monkeypatch):
    """passes of one or two rows (the reference's own use: one runner, one token per step) run the norm / SwiGLU producers INSIDE the
    q|k|v, gate|up, down and logits launches (TkGemvArgs::fuse, csrc/llm/tk_llm_engine.hip enqueue_range): two Mistral-7B-shaped layers
    (so the second layer folds the first one's seven down-projection slabs), three positions through the KV cache, logits bit-exact
    against the oracle — and against the same passes with the producers as launches of their own (TK_MI355X_NO_FUSE=1)."""
    hp = gpu.MISTRAL_7B()
    hp.n_layer = 2
    model = gpu.LlmModel(hp).fill_synthetic(4)
    hp = model.hparams
    orc = O.OracleLlm(oracle_cfg_from(hp, 8, nrows), seed=4)
    rng = np.random.default_rng(3)
    seq = np.arange(nrows, dtype=np.int32)
    toks = [rng.integers(3, hp.vocab, nrows).astype(np.int32) for _ in range(3)]
    runs = []
    for no_fuse in ("0", "1"):
        monkeypatch.setenv("TK_MI355X_NO_FUSE", no_fuse)
        sess = gpu.LlmSession(model, nrows, 8)
        runs.append([sess.forward(seq, np.full(nrows, p, np.int32), toks[p]) for p in range(3)])
        sess.close()
    wams = []
    for p in range(3):
        want, wam = orc.forward(seq, np.full(nrows, p, np.int32), toks[p])
        wams.append(wam)
        for got, gam in (runs[0][p], runs[1][p]):
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
            assert np.array_equal(gam, wam)
    # layer-sharded with the producers inside the launches: stage 0 = layer 0 (its stream leaves folded), stage 1 = layer 1 + head
    monkeypatch.setenv("TK_MI355X_NO_FUSE", "0")
    a, b = gpu.LlmSession(model, nrows, 8), gpu.LlmSession(model, nrows, 8)
    for p in range(3):
        x01 = np.empty((nrows, hp.d_model), np.float32)
        a.forward_stage(seq, np.full(nrows, p, np.int32), 0, 1, tok=toks[p], x_out=x01)
        am = b.forward_stage(seq, np.full(nrows, p, np.int32), 1, 2, x_in=x01, head=True)
        assert np.array_equal(am, wams[p])
    a.close()
    b.close()


def test_full_7b_pass_width_invariance(gpu):
    """size-independent property at BASELINE size (full 32-layer Mistral-7B Q4_K_M, synthetic weights): a row's logits do not depend on
    how many other rows share its pass — 256 rows in one pass (16 M-tiles, batched kernel), the same rows as 2 x 128 and 16 x 16 passes
    (8 M-tiles; one M-tile of the GEMV kernel) give bit-identical logits and ids.  (The 16-row path is the one pinned to the oracle.)"""
    model = gpu.LlmModel(gpu.MISTRAL_7B()).fill_synthetic(4)
    hp = model.hparams
    rng = np.random.default_rng(8)
    n = 256
    seq = np.arange(n, dtype=np.int32)
    pos = np.zeros(n, np.int32)
    tok = rng.integers(3, hp.vocab, n).astype(np.int32)
    tok2 = rng.integers(3, hp.vocab, n).astype(np.int32)
    outs = []
    for width in (256, 128, 16):
        sess = gpu.LlmSession(model, n, 4)
        lg = np.empty((2, n, hp.vocab), np.float32)
        for i in range(0, n, width):
            lg[0, i:i + width], _ = sess.forward(seq[i:i + width], pos[i:i + width], tok[i:i + width])
        for i in range(0, n, width):                       # second position: the KV cache written by the first pass is read back
            lg[1, i:i + width], _ = sess.forward(seq[i:i + width], pos[i:i + width] + 1, tok2[i:i + width])
        outs.append(lg)
        sess.close()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))
    assert np.isfinite(outs[0]).all() and len(np.unique(outs[0].argmax(-1))) > 8


def test_full_7b_f16_pass_width_invariance(gpu):
    """the same property for BASELINE configs[4]'s weights (full 32-layer Mistral-7B as an fp16 checkpoint, 14.2 GB, synthetic): 64 rows in one
    pass (4 M-tiles of the tiled GEMM), as 2 x 32 (2 M-tiles) and as 4 x 16 (1 M-tile) give bit-identical logits over two positions"""
    model = gpu.LlmModel(gpu.MISTRAL_7B()).fill_synthetic(4, f16=True)
    hp = model.hparams
    rng = np.random.default_rng(9)
    n = 64
    seq = np.arange(n, dtype=np.int32)
    pos = np.zeros(n, np.int32)
    tok = rng.integers(3, hp.vocab, n).astype(np.int32)
    tok2 = rng.integers(3, hp.vocab, n).astype(np.int32)
    outs = []
    for width in (64, 32, 16):
        sess = gpu.LlmSession(model, n, 4)
        lg = np.empty((2, n, hp.vocab), np.float32)
        for i in range(0, n, width):
            lg[0, i:i + width], _ = sess.forward(seq[i:i + width], pos[i:i + width], tok[i:i + width])
        for i in range(0, n, width):
            lg[1, i:i + width], _ = sess.forward(seq[i:i + width], pos[i:i + width] + 1, tok2[i:i + width])
        outs.append(lg)
        sess.close()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))
    assert np.isfinite(outs[0]).all() and len(np.unique(outs[0].argmax(-1))) > 4
    model.close()


def test_reference_runner_surface(gpu):
    """tk_model_loader_* + tk_llm_runner_* as the reference's Rust GgufRunner drives them"""
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=4")
    h2 = loader.load("synthetic://tiny?seed=4")
    assert h.value == h2.value  # find-or-load by path
    runner = gpu.LlmRunner(h, context_size=64)
    runner.prepare("hi")
    pieces = []
    for _ in range(8):
        p = runner.next_token()
        if p is None:
            break
        pieces.append(p)
    assert len(pieces) == 8
    # same ids as the oracle: BOS + bytes, greedy
    orc = O.OracleLlm(O.tiny_config(ks_qkv=1, ks_o=1, ks_gateup=1, ks_down=1), seed=4)
    hp = gpu.LlmHParams()
    gpu.lib().tk_mi355x_llm_model_get_hparams(h, __import__("ctypes").byref(hp))
    orc = O.OracleLlm(oracle_cfg_from(hp, 64, 1), seed=4)
    ids = [1, 3 + ord("h"), 3 + ord("i")]
    _, am = orc.forward([0, 0, 0], [0, 1, 2], ids, want_logits=False)
    cur, want = int(am[-1]), []
    for i in range(8):
        want.append(cur)
        _, am = orc.forward([0], [3 + i], [cur], want_logits=False)
        cur = int(am[0])
    tokz = [3 + p[0] if len(p) == 1 else int(p.decode().strip()[1:]) for p in pieces]
    assert tokz == want
    runner.reset()
    assert runner.next_token() is None
    runner.close()
    loader.unload(h)
    loader.unload(h2)
    loader.close()


def _piece_of(tok):
    """display piece of a token of the synthetic byte vocabulary (csrc/llm/tk_tokenizer.h)"""
    if 3 <= tok < 259:
        return bytes([tok - 3])
    return b"" if tok < 3 else (" t%d" % tok).encode()


@pytest.mark.parametrize("seed", [4, 6, 8, 13])  # 4, 8: the grammar completes ("{ ... }"); 6, 13: long strings with multi-byte pieces, context end
def test_tool_grammar_constrained_generation(gpu, seed):
    """use_tool_grammar = true: greedy sampling runs over the tokens the GBNF grammar allows (mask applied by k_argmax on the
    device), the completed grammar returns the sentinel (const char*)1 without decoding (tk_runner_streaming.c:69-75).
    Checked against the oracle's logits with the constraint restated here token by token."""
    import ctypes as C
    L = gpu.lib()
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=%d" % seed)
    n_ctx = 96
    runner = gpu.LlmRunner(h, context_size=n_ctx)
    runner.prepare("call", use_tool_grammar=True)
    got, ended = [], None
    for _ in range(n_ctx):
        p = runner.next_token()
        if p is None or p == "<tool_call>":
            ended = p
            break
        got.append(p)
    text = b"".join(got)

    def accepted(bs):
        n, c = C.c_int32(), C.c_int32()
        assert L.tk_mi355x_grammar_check(None, bs, C.byref(n), C.byref(c)) == 0
        return n.value == len(bs), bool(c.value)

    # the oracle with the same constraint: arg max over the tokens whose piece keeps the text inside the grammar
    hp = gpu.LlmHParams()
    L.tk_mi355x_llm_model_get_hparams(h, C.byref(hp))
    orc = O.OracleLlm(oracle_cfg_from(hp, n_ctx, 1), seed=seed)
    ids = [1] + [3 + b for b in b"call"]
    logits, _ = orc.forward([0] * len(ids), list(range(len(ids))), ids)
    lg, pos, want, sofar, want_end = logits[-1], len(ids), [], b"", None
    while True:
        best = None
        for t in np.argsort(-lg, kind="stable"):
            pc = _piece_of(int(t))
            if int(t) == 2:
                ok = accepted(sofar)[1]
            else:
                ok = bool(pc) and accepted(sofar + pc)[0]
            if ok:
                best = int(t)
                break
        assert best is not None
        if best == 2:
            want_end = None
            break
        sofar += _piece_of(best)
        if accepted(sofar)[1]:
            want_end = "<tool_call>"
            break
        if pos + 1 >= n_ctx:
            break
        want.append(_piece_of(best))
        lg = orc.forward([0], [pos], [best])[0][0]
        pos += 1
    assert got == want
    assert ended == want_end
    ok, done = accepted(text)
    assert ok
    if ended == "<tool_call>":
        L.tk_mi355x_llm_runner_tool_call_text.restype = C.c_char_p
        call = L.tk_mi355x_llm_runner_tool_call_text(runner.h)
        assert accepted(call) == (True, True) and call.startswith(text)
        import json
        assert isinstance(json.loads(call.decode("utf-8", "replace")), dict)
    runner.close()
    loader.unload(h)
    loader.close()


def test_error_paths(gpu):
    import ctypes as C
    hp = gpu.TINY()
    hp.d_model = 250
    with pytest.raises(gpu.TkError) as e:
        gpu.LlmModel(hp)
    assert e.value.code == 4000 and "256" in e.value.detail
    model = gpu.LlmModel(gpu.TINY())
    with pytest.raises(gpu.TkError):  # tensors missing
        gpu.LlmSession(model, 1, 16)
    with pytest.raises(gpu.TkError):  # wrong size
        model.set_tensor(0, 1, 12, np.zeros(10, np.uint8))
    model.fill_synthetic(1)
    sess = gpu.LlmSession(model, 1, 16)
    with pytest.raises(gpu.TkError):  # position beyond context
        sess.forward([0], [16], [5])
    n_over = gpu.lib().tk_mi355x_llm_max_rows() + 1
    with pytest.raises(gpu.TkError):  # one row more than a pass holds
        sess.forward(np.zeros(n_over, np.int32), np.arange(n_over, dtype=np.int32) % 16, np.full(n_over, 5, np.int32))
    assert gpu.lib().tk_mi355x_llm_forward(None, 1, None, None, None, None, None) == 1001


def test_gguf_checkpoint_end_to_end(gpu, tmp_path):
    """a GGUF file (llama arch, Q4_K/Q6_K/F32 tensors, SentencePiece vocab) through tk_model_loader + tk_llm_runner"""
    import gguf_util
    cfg = O.tiny_config()
    orc = O.OracleLlm(cfg, seed=4)
    path = str(tmp_path / "tiny.gguf")
    gguf_util.write_llama_gguf(path, orc, cfg)
    loader = gpu.ModelLoader()
    h = loader.load(path)
    hp = gpu.LlmHParams()
    gpu.lib().tk_mi355x_llm_model_get_hparams(h, __import__("ctypes").byref(hp))
    orc2 = O.OracleLlm(oracle_cfg_from(hp, 64, 1), seed=4)
    runner = gpu.LlmRunner(h, context_size=64)
    runner.prepare("hello world")
    ids = [1, 263, 273]  # llama SPM: " hello world" -> [bos, "▁hello", "▁world"] (asserted on CPU in test_gguf_cpu.py)
    _, am = orc2.forward([0, 0, 0], [0, 1, 2], ids, want_logits=False)
    cur = int(am[-1])
    for i in range(6):
        piece = runner.next_token()
        if cur == 2:
            assert piece is None
            break
        assert piece == gguf_util.expected_piece(cfg.vocab, cur), (i, cur, piece)
        _, am = orc2.forward([0], [3 + i], [cur], want_logits=False)
        cur = int(am[0])
    runner.close()
    loader.unload(h)
    loader.close()
    with pytest.raises(gpu.TkError):
        gpu.ModelLoader().load(str(tmp_path / "nope.gguf"))


@pytest.mark.timeout(120)
def test_gguf_with_a_missing_tensor_fails_the_load_instead_of_hanging(gpu, tmp_path):
    """the loader holds its registry lock across the GGUF load; the failure path must not take it again (a file with a missing
    tensor hung tk_model_loader_load_model).  TK_ERROR_MODEL_LOAD_FAILED = 4000, and the loader stays usable afterwards."""
    import gguf_util
    cfg = O.tiny_config()
    orc = O.OracleLlm(cfg, seed=4)
    bad = str(tmp_path / "missing.gguf")
    gguf_util.write_llama_gguf(bad, orc, cfg, drop=("blk.1.ffn_up.weight",))
    loader = gpu.ModelLoader()
    with pytest.raises(gpu.TkError) as ei:
        loader.load(bad)
    assert ei.value.code == 4000 and "blk.1.ffn_up.weight" in str(ei.value)
    good = str(tmp_path / "whole.gguf")
    gguf_util.write_llama_gguf(good, orc, cfg)
    h = loader.load(good)                                             # the registry lock was released
    loader.unload(h)
    loader.close()


class _BorrowedModel:
    """a tk_mi355x_llm_model_t* owned by a tk_model_loader (the handle tk_model_loader_load_model returns) seen as a LlmModel"""

    def __init__(self, gpu, handle):
        import ctypes as C
        self.h = handle
        self.hparams = gpu.LlmHParams()
        gpu.lib().tk_mi355x_llm_model_get_hparams(handle, C.byref(self.hparams))


def _ids_of(pieces):
    """token ids of the synthetic byte vocabulary's display pieces (csrc/llm/tk_tokenizer.h)"""
    return [3 + p[0] if len(p) == 1 else int(p.decode().strip()[1:]) for p in pieces]


def test_7b_through_reference_runner_32_tokens(gpu):
    """BASELINE configs[0] (tests/tk_cortex_test.cpp's model + token count) through the reference entry points only:
    tk_model_loader_load_model -> tk_llm_runner_prepare_generation -> 32 x tk_llm_runner_generate_next_token on the full 32-layer
    Mistral-7B Q4_K_M geometry; every id equal to the ORACLE's on the same weights and token stream (the prompt's rows in one oracle pass
    + 31 single-row passes, ~10 s of CPU)."""
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://mistral-7b?seed=4")
    runner = gpu.LlmRunner(h, context_size=128)
    prompt = "The user is in a room."
    runner.prepare(prompt)
    pieces = []
    for _ in range(32):
        p = runner.next_token()
        assert p is not None and p != "<tool_call>"
        pieces.append(p)
    got = _ids_of(pieces)
    hp = _BorrowedModel(gpu, h).hparams
    assert hp.n_layer == 32 and hp.d_model == 4096
    runner.close()
    loader.unload(h)
    loader.close()
    ids = np.array([1] + [3 + b for b in prompt.encode()], np.int32)   # BOS + the byte vocabulary's ids (csrc/llm/tk_tokenizer.h)
    n = len(ids)
    orc = O.OracleLlm(oracle_cfg_from(hp, 128, 1), seed=4)
    _, am = orc.forward(np.zeros(n, np.int32), np.arange(n, dtype=np.int32), ids, want_logits=False)
    cur, want = int(am[-1]), []
    for i in range(32):
        want.append(cur)
        if i + 1 < 32:
            _, am = orc.forward([0], [n + i], [cur], want_logits=False)
            cur = int(am[0])
    orc.close()
    assert got == want
    assert len(set(got)) > 1


def test_runner_add_tool_response_continues_the_context(gpu):
    """tk_llm_runner_add_tool_response (src/ai_models/tk_runner_helpers.c:78-133): the formatted tool result is tokenised WITHOUT a
    BOS and decoded into the KV cache at n_past; generation resumes from its last position.  Ids against the oracle fed the same
    token stream."""
    import ctypes as C
    L = gpu.lib()
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=4")
    model = _BorrowedModel(gpu, h)
    runner = gpu.LlmRunner(h, context_size=192)
    runner.prepare("go")
    first = [runner.next_token() for _ in range(3)]
    assert all(p is not None for p in first)
    assert L.tk_llm_runner_add_tool_response(runner.h, b"get_time", b"{\"time\": \"12:00\"}") == 0
    after = [runner.next_token() for _ in range(5)]
    assert all(p is not None for p in after)
    # oracle: prompt, 3 generated tokens, then the tool text, then 5 more
    orc = O.OracleLlm(oracle_cfg_from(model.hparams, 192, 1), seed=4)
    ids = [1, 3 + ord("g"), 3 + ord("o")]
    _, am = orc.forward([0] * 3, [0, 1, 2], ids, want_logits=False)
    cur, pos, want_first = int(am[-1]), 3, []
    for _ in range(3):
        want_first.append(cur)
        _, am = orc.forward([0], [pos], [cur], want_logits=False)
        cur, pos = int(am[0]), pos + 1
    assert _ids_of(first) == want_first
    # the token sampled after the third decode is still pending inside the runner: the reference discards it too (the tool text
    # is decoded at n_past, sampling restarts from the logits of its last token)
    tool = b"[TOOL_RESULT] name: \"get_time\", output: {\"time\": \"12:00\"} [/TOOL_RESULT]"
    tids = [3 + b for b in tool]
    _, am = orc.forward([0] * len(tids), list(range(pos, pos + len(tids))), tids, want_logits=False)
    cur, pos, want_after = int(am[-1]), pos + len(tids), []
    for _ in range(5):
        want_after.append(cur)
        _, am = orc.forward([0], [pos], [cur], want_logits=False)
        cur, pos = int(am[0]), pos + 1
    assert _ids_of(after) == want_after
    # argument and state errors
    assert L.tk_llm_runner_add_tool_response(runner.h, None, b"x") == 1001
    assert L.tk_llm_runner_add_tool_response(None, b"a", b"x") == 1001
    long_out = b"y" * 400  # does not fit what is left of the 192-token context
    assert L.tk_llm_runner_add_tool_response(runner.h, b"a", long_out) != 0
    runner.close()
    loader.unload(h)
    loader.close()


def test_decode_with_kv_cache_matches_hf_fixture(gpu):
    """the HIP path against the HF KV-cache decode fixture (llm_tiny_decode.npz): prefill 16 tokens in one pass, then 8 single-row passes
    teacher-forced with HF's ids — step logits bit-identical to the committed oracle logits, within quantisation noise of HF's."""
    import os
    hp = gpu.TINY()
    hp.ks_qkv = hp.ks_o = hp.ks_gateup = hp.ks_down = 1          # the fixture's K-split plan (oracle_lib.tiny_config defaults)
    model = gpu.LlmModel(hp).fill_synthetic(4)
    sess = gpu.LlmSession(model, 1, 64)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "llm_tiny_decode.npz"))
    assert int(g["seed"]) == 4
    toks, ids, hf = g["tokens"], g["hf_ids"], g["hf_step_logits"]
    P = len(toks)
    _, am = sess.forward(np.zeros(P, np.int32), np.arange(P, dtype=np.int32), toks)
    assert int(am[-1]) == int(g["oracle_first_id"])
    got = np.stack([sess.forward([0], [P + i], [ids[i]])[0][0] for i in range(len(ids))])
    assert np.array_equal(got.view(np.uint32), g["oracle_step_logits"].view(np.uint32))
    assert np.abs(got - hf).max() < 0.05 * np.abs(hf).max()


@pytest.mark.parametrize("order", ["arrival", "sampled_rows_first"])
def test_runners_on_one_model_share_decode_passes(gpu, monkeypatch, order):
    """continuous batching behind the reference ABI (csrc/llm/tk_llm_batcher.h): K tk_llm_runner_t handles driven from K host threads
    through tk_llm_runner_* only.  Every runner gets exactly the tokens it gets when it runs alone (and the oracle's), while the
    scheduler's counters show that their rows shared passes.  Both pass orders of the scheduler (arrival order, the default; sampled rows
    before prompt rows, TK_MI355X_BATCHER_DECODE_FIRST=1, read when the model's scheduler starts) give the same tokens."""
    import threading
    monkeypatch.setenv("TK_MI355X_BATCHER_DECODE_FIRST", "1" if order == "sampled_rows_first" else "0")
    K, NTOK = 8, 12
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=4")
    gpu.ModelLoader.set_runner_slots(h, K)
    model = _BorrowedModel(gpu, h)
    prompts = ["p%d %s" % (i, "ab" * i) for i in range(K)]          # different lengths: prompt chunks and decode rows mix in passes

    def generate(runner, prompt):
        runner.prepare(prompt)
        out = []
        for _ in range(NTOK):
            p = runner.next_token()
            if p is None:
                break
            out.append(p)
        return out

    # alone, one after the other
    solo = []
    r0 = gpu.LlmRunner(h, context_size=96)
    for pr in prompts:
        solo.append(generate(r0, pr))
    r0.close()
    p_before, rows_before, _ = gpu.ModelLoader.batch_stats(h)
    assert 0 < p_before < rows_before                               # prompt chunks are multi-row passes
    # the oracle agrees with the first one
    orc = O.OracleLlm(oracle_cfg_from(model.hparams, 96, 1), seed=4)
    ids = [1] + [3 + b for b in prompts[0].encode()]
    _, am = orc.forward([0] * len(ids), list(range(len(ids))), ids, want_logits=False)
    cur, want = int(am[-1]), []
    for i in range(NTOK):
        want.append(cur)
        _, am = orc.forward([0], [len(ids) + i], [cur], want_logits=False)
        cur = int(am[0])
    assert _ids_of(solo[0]) == want[:len(solo[0])]
    # together: K runners, K threads, twice (slots are reused after prepare_generation restarts a sequence)
    runners = [gpu.LlmRunner(h, context_size=96) for _ in range(K)]
    for rnd in range(2):
        got = [None] * K
        th = [threading.Thread(target=lambda i=i: got.__setitem__(i, generate(runners[i], prompts[i]))) for i in range(K)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert got == solo, f"round {rnd}: a runner's tokens changed when other runners shared its passes"
    p_after, rows_after, widest = gpu.ModelLoader.batch_stats(h)
    assert rows_after - rows_before == 2 * (rows_before)            # the same rows were fed, twice
    assert widest >= 2 and (p_after - p_before) < 2 * K * NTOK, (p_after - p_before, widest)   # decode rows of several runners per pass
    # a ninth runner on a full session gets a session of its own (still correct, not batched with the others)
    extra = gpu.LlmRunner(h, context_size=96)
    assert generate(extra, prompts[3]) == solo[3]
    extra.close()
    for r in runners:
        r.close()
    loader.unload(h)
    loader.close()


def test_run_ahead_rows_are_invisible(gpu):
    """the scheduler feeds a sequence's sampled id one position AHEAD of its owner's next generate_next_token (csrc/llm/tk_llm_batcher.h).
    Whatever the owner does next, it sees the tokens of a runner nobody ran ahead of: (a) it comes back late (the row is done and waiting),
    (b) it comes back at once (the row is in flight), (c) it stops and starts a new prompt (the row is wasted, its cache row overwritten).  A tool response fed at the position a wasted
    row wrote is test_runner_add_tool_response_continues_the_context.  Reference: the oracle's greedy ids."""
    import time
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=4")
    model = _BorrowedModel(gpu, h)
    orc = O.OracleLlm(oracle_cfg_from(model.hparams, 96, 1), seed=4)

    def oracle_ids(text_ids, n, then=()):
        orc.reset()
        _, am = orc.forward([0] * len(text_ids), list(range(len(text_ids))), text_ids, want_logits=False)
        cur, out, pos = int(am[-1]), [], len(text_ids)
        for _ in range(n):
            out.append(cur)
            _, am = orc.forward([0], [pos], [cur], want_logits=False)
            cur, pos = int(am[0]), pos + 1
        return out, pos, cur

    def ids_of_prompt(p):
        return [1] + [3 + b for b in p.encode()]

    r = gpu.LlmRunner(h, context_size=96)
    w0 = gpu.ModelLoader.run_ahead_wasted(h)
    # (a) a slow owner: every row is finished before it is asked for
    r.prepare("late owner")
    got = []
    for _ in range(6):
        time.sleep(0.02)
        got.append(r.next_token())
    assert None not in got
    want, _, _ = oracle_ids(ids_of_prompt("late owner"), 6)
    assert _ids_of(got) == want
    # (b) + (c): a fast owner, then a new prompt in the middle of a generation — the row that ran ahead of the abandoned one is wasted
    r.prepare("second prompt, longer than the first")
    got = [r.next_token() for _ in range(5)]
    assert None not in got
    want, _, _ = oracle_ids(ids_of_prompt("second prompt, longer than the first"), 5)
    assert _ids_of(got) == want
    r.prepare("x")   # shorter than what the cache holds: stale rows beyond it, one of them written by a run-ahead row
    got = [r.next_token() for _ in range(8)]
    assert None not in got
    want, _, _ = oracle_ids(ids_of_prompt("x"), 8)
    assert _ids_of(got) == want
    assert gpu.ModelLoader.run_ahead_wasted(h) >= w0 + 2
    # rows an owner asked for: 3 prompts + 6 + 5 + 8 decode rows — the run-ahead rows that were taken count once, the wasted ones never
    _, rows, _ = gpu.ModelLoader.batch_stats(h)
    assert rows == len(ids_of_prompt("late owner")) + len(ids_of_prompt("second prompt, longer than the first")) + len(ids_of_prompt("x")) + 6 + 5 + 8
    r.close()
    loader.unload(h)
    loader.close()


def test_f16_weights_bit_exact_and_pipeline_split(gpu):
    """fp16 checkpoints (BASELINE configs[4]): every matrix IEEE f16, run on the exact fp32 MFMA GEMM over f16-rounded activations.
    Logits bit-identical to the oracle (one fma chain per output over k ascending); a decode loop through the captured graphs; the
    layer-sharded split of configs[4] (forward_stage over two stages) bit-identical to the unsplit pass."""
    model = gpu.LlmModel(gpu.TINY()).fill_synthetic(12, f16=True)
    hp = model.hparams
    sess = gpu.LlmSession(model, 4, 64)
    orc = O.OracleLlm(oracle_cfg_from(hp, 64, 4), seed=12, f16=True)
    assert orc.get_tensor(0, 1)[0] == 1 and orc.get_tensor(-1, 0)[0] == 1          # ggml type F16 for matrices and the embedding
    rng = np.random.default_rng(31)
    seq = np.array([0] * 9 + [2] * 7 + [3] * 4, np.int32)
    pos = np.concatenate([np.arange(9), np.arange(7), np.arange(4)]).astype(np.int32)
    tok = rng.integers(3, hp.vocab, 20).astype(np.int32)
    want, wam = orc.forward(seq, pos, tok)
    got, gam = sess.forward(seq, pos, tok)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(gam, wam)
    # greedy decode, graph replay: ids equal the oracle's
    s2, p2, cur = np.array([0, 2, 3], np.int32), np.array([9, 7, 4], np.int32), np.array(gam[[8, 15, 19]], np.int32)
    _, gcur = sess.forward(s2, p2, cur, want_logits=False)
    _, wcur = orc.forward(s2, p2, cur, want_logits=False)
    assert np.array_equal(gcur, wcur)
    # the same weights loaded tensor by tensor (the GGUF path's entry) give the same bits
    m2 = gpu.LlmModel(gpu.TINY())
    copy_oracle_weights(orc, m2, hp.n_layer)
    se2 = gpu.LlmSession(m2, 4, 64)
    g2, _ = se2.forward(seq, pos, tok)
    assert np.array_equal(g2.view(np.uint32), want.view(np.uint32))
    # ... and so does a GGUF file whose tensors are F16 (what a real fp16 checkpoint is)
    import gguf_util, tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "tiny-f16.gguf")
        gguf_util.write_llama_gguf(path, orc, O.tiny_config())
        m3 = gpu.LlmModel(gguf=path)
        se3 = gpu.LlmSession(m3, 4, 64)
        g3, _ = se3.forward(seq, pos, tok)
        assert np.array_equal(g3.view(np.uint32), want.view(np.uint32))
        se3.close()
    # layer-sharded: stage 0 = layer 0, stage 1 = layer 1 + head, the residual stream crosses as fp32
    a, b = gpu.LlmSession(model, 4, 64), gpu.LlmSession(model, 4, 64)
    x01 = np.empty((20, hp.d_model), np.float32)
    a.forward_stage(seq, pos, 0, 1, tok=tok, x_out=x01)
    am = b.forward_stage(seq, pos, 1, 2, x_in=x01, head=True)
    assert np.array_equal(am, wam)
    for s in (sess, se2, a, b):
        s.close()


@pytest.mark.parametrize("nseq,npos", [(5, 14), (8, 25)])  # 70 rows (8 M-tiles) and 200 rows (16 M-tiles)
def test_f16_weights_wide_passes_bit_exact(gpu, nseq, npos):
    """fp16 checkpoint, prefill-shaped passes of 70 / 200 rows: the tiled GEMM's 8- and 16-M-tile variants against the oracle, bit for bit,
    and the rows of a wide pass equal the same rows fed through narrow passes (the K-split plan does not depend on the pass width)"""
    model = gpu.LlmModel(gpu.TINY()).fill_synthetic(13, f16=True)
    hp = model.hparams
    sess = gpu.LlmSession(model, nseq, 32)
    orc = O.OracleLlm(oracle_cfg_from(hp, 32, nseq), seed=13, f16=True)
    rng = np.random.default_rng(nseq)
    seq = np.repeat(np.arange(nseq, dtype=np.int32), npos)
    pos = np.tile(np.arange(npos, dtype=np.int32), nseq)
    tok = rng.integers(3, hp.vocab, nseq * npos).astype(np.int32)
    want, wam = orc.forward(seq, pos, tok)
    got, gam = sess.forward(seq, pos, tok)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(gam, wam)
    narrow = gpu.LlmSession(model, nseq, 32)
    for s in range(nseq):  # one sequence per pass: 14 / 25 rows
        sel = seq == s
        g1, _ = narrow.forward(seq[sel], pos[sel], tok[sel])
        assert np.array_equal(g1.view(np.uint32), got[sel].view(np.uint32))
    sess.close(); narrow.close(); model.close()


SAMPLING_CASES = [  # (temperature, top_k, top_p, min_p): llama.cpp's defaults first, then each filter alone / switched off
    (0.8, 40, 0.95, 0.05), (1.0, 0, 1.0, 0.0), (0.3, 8, 1.0, 0.0), (1.7, 64, 0.5, 0.0), (1.0, 40, 1.0, 0.3), (0.0, 40, 0.95, 0.05),
]


@pytest.mark.parametrize("seed", [0, 7, 0xDEADBEEF, (1 << 63) + 12345])
def test_stochastic_sampler_ids_equal_the_oracle(gpu, seed):
    """the reference samples through llama.cpp's default chain (/root/reference/src/ai_models/tk_runner_lifecycle.c:76-77,
    tk_runner_streaming.c:60-61); here it runs on the device (k_argmax / sample_row: radix select + one canonical arithmetic order) and the
    oracle restates it (orc_sample_row): for a fixed (seed, counter) every id is equal — 12 rows per pass with different parameter sets and
    counters, greedy rows mixed in, over three steps of a Mistral-sized vocabulary (32000 logits per row) and on the tiny model."""
    for hp0, nl in ((gpu.MISTRAL_7B(), 1), (gpu.TINY(), 2)):
        hp0.n_layer = nl
        model = gpu.LlmModel(hp0).fill_synthetic(4)
        hp = model.hparams
        n = 12
        sess = gpu.LlmSession(model, n, 16)
        orc = O.OracleLlm(oracle_cfg_from(hp, 16, n), seed=4)
        rng = np.random.default_rng(seed & 0xFFFF)
        seq = np.arange(n, dtype=np.int32)
        tok = rng.integers(3, hp.vocab, n).astype(np.int32)
        for step in range(3):
            pos = np.full(n, step, np.int32)
            samp = [SAMPLING_CASES[(r + step) % len(SAMPLING_CASES)] + (seed, 100 * r + step) for r in range(n)]
            want_logits, wam = orc.forward(seq, pos, tok)
            got_logits, ids = sess.forward_sampled(seq, pos, tok, samp)
            assert np.array_equal(got_logits.view(np.uint32), want_logits.view(np.uint32))
            want = [O.sample_row(want_logits[r], *samp[r]) for r in range(n)]
            assert ids.tolist() == want, (step, ids.tolist(), want)
            greedy = [r for r in range(n) if samp[r][0] <= 0]
            assert greedy and all(ids[r] == wam[r] for r in greedy)
            tok = ids.astype(np.int32)
        assert len(set(ids.tolist())) > 3
        # a pass without sampling states afterwards is greedy again (the table of the previous pass is cleared)
        _, am = sess.forward(seq, np.full(n, 3, np.int32), tok, want_logits=False)
        _, wam = orc.forward(seq, np.full(n, 3, np.int32), tok, want_logits=False)
        assert np.array_equal(am, wam)
        sess.close()
        model.close()
        orc.close()


def test_runner_with_the_reference_sampler_is_seeded_and_batch_invariant(gpu):
    """tk_llm_config_t.random_seed (src/ai_models/tk_model_runner.h:45-49) + tk_mi355x_llm_runner_set_sampling: a runner's stochastic ids
    equal the oracle's for its seed, do not change when other runners share its passes, and differ for another seed"""
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=4")
    hp = gpu.LlmHParams()
    gpu.lib().tk_mi355x_llm_model_get_hparams(h, __import__("ctypes").byref(hp))

    def oracle_ids(seed, n):
        orc = O.OracleLlm(oracle_cfg_from(hp, 64, 1), seed=4)
        ids = [1, 3 + ord("h"), 3 + ord("i")]
        lg, _ = orc.forward([0, 0, 0], [0, 1, 2], ids)
        out, counter = [], 0
        cur = O.sample_row(lg[-1], 0.8, 40, 0.95, 0.05, seed, counter)
        for i in range(n):
            out.append(cur)
            counter += 1
            lg, _ = orc.forward([0], [3 + i], [cur])
            cur = O.sample_row(lg[0], 0.8, 40, 0.95, 0.05, seed, counter)
        orc.close()
        return out

    def run(runner, n):
        runner.prepare("hi")
        out = []
        for _ in range(n):
            p = runner.next_token()
            if p is None:
                break
            out.append(3 + p[0] if len(p) == 1 else int(p.decode().strip()[1:]))
        return out

    want = oracle_ids(21, 10)
    eos = 2
    if eos in want:
        want = want[: want.index(eos)]
    r1 = gpu.LlmRunner(h, context_size=64, random_seed=21)
    r1.set_sampling(0.8)
    assert run(r1, 10) == want
    # a second generation continues the generator (the counter is the runner's, not the prompt's): different ids, same as the oracle's continuation
    # and two more runners with other seeds decoding at the same time do not change runner 1's stream
    import threading
    r2 = gpu.LlmRunner(h, context_size=64, random_seed=21)
    r2.set_sampling(0.8)
    r3 = gpu.LlmRunner(h, context_size=64, random_seed=22)
    r3.set_sampling(0.8)
    res = {}
    ths = [threading.Thread(target=lambda k, r: res.__setitem__(k, run(r, 10)), args=(k, r)) for k, r in (("a", r2), ("b", r3))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert res["a"] == want and res["b"] != want
    r1.set_sampling(0.0)  # back to greedy
    g = run(r1, 6)
    r4 = gpu.LlmRunner(h, context_size=64)
    assert g == run(r4, 6)
    with pytest.raises(gpu.TkError):
        r1.set_sampling(-1.0)
    for r in (r1, r2, r3, r4):
        r.close()
    loader.unload(h)
    loader.close()


@pytest.mark.parametrize("seed", [4, 8])
def test_stochastic_sampling_under_the_tool_grammar_stays_inside_it(gpu, seed):
    """the reference arms the grammar AND samples through llama.cpp's default chain (tk_runner_streaming.c:44-48, 60-61): with
    tk_mi355x_llm_runner_set_sampling on, the draw is taken among the tokens the grammar allows (the mask enters the on-device top-k
    selection), so every sampled piece keeps the text inside the grammar, the ids equal the oracle's sampler under the same masks, and two
    runners with the same seed produce the same text"""
    import ctypes as C
    L = gpu.lib()
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=%d" % seed)
    hp = gpu.LlmHParams()
    L.tk_mi355x_llm_model_get_hparams(h, C.byref(hp))
    n_ctx = 96

    def accepted(bs):
        n, c = C.c_int32(), C.c_int32()
        assert L.tk_mi355x_grammar_check(None, bs, C.byref(n), C.byref(c)) == 0
        return n.value == len(bs), bool(c.value)

    def generate(rng_seed):
        runner = gpu.LlmRunner(h, context_size=n_ctx, random_seed=rng_seed)
        runner.set_sampling(0.9, 40, 0.95, 0.05)
        runner.prepare("call", use_tool_grammar=True)
        got, ended = [], None
        for _ in range(n_ctx):
            p = runner.next_token()
            if p is None or p == "<tool_call>":
                ended = p
                break
            got.append(p)
        runner.close()
        return got, ended

    got, ended = generate(77)
    assert generate(77) == (got, ended)                    # seeded
    text = b"".join(got)
    assert accepted(text)[0]                                # never left the grammar
    # the oracle's sampler under the same masks: allowed = tokens whose piece keeps the text inside the grammar (EOS: only when complete)
    orc = O.OracleLlm(oracle_cfg_from(hp, n_ctx, 1), seed=seed)
    ids = [1] + [3 + b for b in b"call"]
    logits, _ = orc.forward([0] * len(ids), list(range(len(ids))), ids)
    lg, pos, sofar, want, counter = logits[-1], len(ids), b"", [], 0
    while True:
        allow = np.zeros((hp.vocab + 31) // 32, np.uint32)
        for t in range(hp.vocab):
            pc = _piece_of(t)
            ok = accepted(sofar)[1] if t == 2 else (bool(pc) and accepted(sofar + pc)[0])
            if ok:
                allow[t >> 5] |= np.uint32(1 << (t & 31))
        tok = O.sample_row(lg, 0.9, 40, 0.95, 0.05, 77, counter, allow)
        counter += 1
        if tok == 2:
            break
        sofar += _piece_of(tok)
        if accepted(sofar)[1] or pos + 1 >= n_ctx:
            break
        want.append(_piece_of(tok))
        lg = orc.forward([0], [pos], [tok])[0][0]
        pos += 1
    assert got == want
    orc.close()
    loader.unload(h)
    loader.close()


def _lora_factors(cfg, rng, r, which=((0, 1), (1, 3), (1, 8), (0, 6), (0, 4), (-1, O.T_OUTPUT))):
    D, QD, KVD, FF, V = cfg.d_model, cfg.n_head * cfg.head_dim, cfg.n_kv_head * cfg.head_dim, cfg.d_ff, cfg.vocab
    shape = {1: (QD, D), 2: (KVD, D), 3: (KVD, D), 4: (D, QD), 6: (FF, D), 7: (FF, D), 8: (D, FF)}
    out = {}
    for layer, w in which:
        n, k = (V, D) if layer < 0 else shape[w]
        out[(layer, w)] = (rng.normal(0, 0.05, (r, k)).astype(np.float32), rng.normal(0, 0.05, (n, r)).astype(np.float32))
    return out


@pytest.mark.parametrize("f16", [False, True])
def test_lora_adapter_merged_at_load_bit_exact(gpu, tmp_path, f16):
    """the reference applies a LoRA adapter once, in place, right after the load (tk_model_loader.c:259-270): W' = W + (alpha / r) B A, quantised back
    to W's own type.  The GPU merge (k_lora_merge, while a matrix is installed) gives the oracle's merged model bit for bit — Q4_K, Q6_K and f16
    matrices, both adapter containers, f32 and f16 factors"""
    import gguf_util
    cfg = O.tiny_config(max_ctx=32, max_seq=2)
    rng = np.random.default_rng(31)
    fs = _lora_factors(cfg, rng, 8)
    # f16 factors: both sides get the values the file holds
    fs16 = {k: (a.astype(np.float16).astype(np.float32), b.astype(np.float16).astype(np.float32)) for k, (a, b) in fs.items()}
    files = []
    for name, writer, fac in (("a.ggla", lambda p: gguf_util.write_lora_ggla(p, 8, 16, fs), fs), ("b.gguf", lambda p: gguf_util.write_lora_gguf(p, 16.0, fs), fs),
                              ("c.gguf", lambda p: gguf_util.write_lora_gguf(p, 16.0, fs, f16=True), fs16),
                              ("d.ggla", lambda p: gguf_util.write_lora_ggla(p, 8, 16, fs, f16=True), fs16)):
        path = str(tmp_path / name)
        writer(path)
        files.append((path, fac))
    base = O.OracleLlm(cfg, seed=4, f16=f16)
    seq, pos, tok = [0, 0, 0, 1], [0, 1, 2, 0], [5, 9, 300, 7]
    base_logits, _ = base.forward(seq, pos, tok)
    hp = gpu.LlmHParams(cfg.n_layer, cfg.d_model, cfg.n_head, cfg.n_kv_head, cfg.head_dim, cfg.d_ff, cfg.vocab, cfg.rms_eps, cfg.rope_theta,
                        cfg.ks_qkv, cfg.ks_o, cfg.ks_gateup, cfg.ks_down, cfg.ks_out)
    seen = []
    for path, fac in files:
        orc = O.OracleLlm(cfg, seed=4, f16=f16)
        for (layer, w), (A, B) in fac.items():
            orc.apply_lora(layer, w, A, B, 16.0 / 8)
        want, _ = orc.forward(seq, pos, tok)
        model = gpu.LlmModel(hp).set_lora(path)
        copy_oracle_weights(base, model, cfg.n_layer)                   # the BASE weights go in: the library merges while it installs them
        assert model.lora_merged == len(fac)
        sess = gpu.LlmSession(model, 2, 32)
        got, _ = sess.forward(seq, pos, tok)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (path, np.abs(got - want).max())
        assert np.abs(got - base_logits).max() > 1e-3                   # and the adapter is what moved the logits
        seen.append(got)
        sess.close(); model.close()
    assert np.array_equal(seen[0], seen[1]) and np.array_equal(seen[2], seen[3])   # the container does not matter, the factor values do
    # an adapter that does not fit: refused with the reference's code, nothing half-applied
    bad = str(tmp_path / "bad.gguf")
    gguf_util.write_lora_gguf(bad, 16.0, {(0, 1): (fs[(0, 1)][0][:, :128], fs[(0, 1)][1])})
    model = gpu.LlmModel(hp)
    with pytest.raises(gpu.TkError) as ei:
        model.set_lora(bad)
    assert ei.value.code == 4000
    far = str(tmp_path / "far.gguf")
    gguf_util.write_lora_gguf(far, 16.0, {(7, 1): fs[(0, 1)]})          # layer 7 of a 2-layer model
    with pytest.raises(gpu.TkError):
        model.set_lora(far)
    model.close()


def test_lora_adapter_through_the_reference_loader(gpu, tmp_path):
    """tk_model_load_params_t.lora_adapter, the reference's way in: a GGUF checkpoint + a ggla adapter through tk_model_loader_load_model and
    tk_llm_runner_*; the registry keeps the adapted and the plain model apart"""
    import gguf_util
    cfg = O.tiny_config()
    orc = O.OracleLlm(cfg, seed=4)
    path = str(tmp_path / "tiny.gguf")
    gguf_util.write_llama_gguf(path, orc, cfg)
    rng = np.random.default_rng(32)
    fs = _lora_factors(cfg, rng, 4)
    adapter = str(tmp_path / "tiny.ggla")
    gguf_util.write_lora_ggla(adapter, 4, 32, fs)
    loader = gpu.ModelLoader()
    h_plain = loader.load(path)
    h_lora = loader.load(path, lora_adapter=adapter)
    h_again = loader.load(path, lora_adapter=adapter)
    assert h_plain.value != h_lora.value and h_again.value == h_lora.value
    assert gpu.lib().tk_mi355x_llm_model_lora_merged(h_lora) == len(fs) and gpu.lib().tk_mi355x_llm_model_lora_merged(h_plain) == 0
    hp = gpu.LlmHParams()
    gpu.lib().tk_mi355x_llm_model_get_hparams(h_lora, __import__("ctypes").byref(hp))
    ids = [1, 263, 273]                                                  # " hello world" (test_gguf_cpu.py)
    out = {}
    for name, h, adapted in (("plain", h_plain, False), ("lora", h_lora, True)):
        o = O.OracleLlm(oracle_cfg_from(hp, 64, 1), seed=4)
        if adapted:
            for (layer, w), (A, B) in fs.items():
                o.apply_lora(layer, w, A, B, 32.0 / 4)
        runner = gpu.LlmRunner(h, context_size=64)
        runner.prepare("hello world")
        _, am = o.forward([0, 0, 0], [0, 1, 2], ids, want_logits=False)
        cur, got = int(am[-1]), []
        for i in range(8):
            piece = runner.next_token()
            got.append(cur)
            if cur == 2:
                assert piece is None
                break
            assert piece == gguf_util.expected_piece(cfg.vocab, cur), (name, i, cur, piece)
            _, am = o.forward([0], [3 + i], [cur], want_logits=False)
            cur = int(am[0])
        out[name] = got
        runner.close()
    assert out["plain"] != out["lora"]                                   # alpha / r = 8: the adapter changes what is said
    with pytest.raises(gpu.TkError) as ei:
        loader.load(path, lora_adapter=str(tmp_path / "absent.ggla"))
    assert ei.value.code == 4000                                         # TK_ERROR_MODEL_LOAD_FAILED, as the reference returns
    for h in (h_again, h_lora, h_plain):
        loader.unload(h)
    loader.close()


def test_tiny_long_prompt_chunks_head_dim_64(gpu):
    """k_attention_prefill at head_dim 64 (four query heads per workgroup, 1 024 threads): a 300-token prompt of the tiny geometry in passes of
    200 + 100 rows plus a ragged pass over three sequences, all reaching past position 128 (below it the per-row form runs) — logits = oracle"""
    hp = gpu.TINY()
    model = gpu.LlmModel(hp)
    hp = model.hparams
    assert gpu.attention_plan(200, hp.n_head, hp.n_kv_head, hp.head_dim, 320, False)[0] == 2
    orc = O.OracleLlm(oracle_cfg_from(hp, 320, 4), seed=4)
    copy_oracle_weights(orc, model, hp.n_layer)
    sess = gpu.LlmSession(model, 4, 320)
    rng = np.random.default_rng(41)
    tok = rng.integers(3, hp.vocab, 300).astype(np.int32)
    for lo, hi in ((0, 200), (200, 300)):
        seq, pos = np.zeros(hi - lo, np.int32), np.arange(lo, hi, dtype=np.int32)
        want, wam = orc.forward(seq, pos, tok[lo:hi])
        got, gam = sess.forward(seq, pos, tok[lo:hi])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (lo, np.abs(got - want).max())
        assert np.array_equal(gam, wam)
    # ragged: sequence 1 gets 150 rows from position 0, sequence 2 gets 21, sequence 0 continues with 7 rows at 300
    seq = np.array([1] * 150 + [2] * 21 + [0] * 7, np.int32)
    pos = np.concatenate([np.arange(150), np.arange(21), np.arange(300, 307)]).astype(np.int32)
    tk = rng.integers(3, hp.vocab, seq.size).astype(np.int32)
    want, wam = orc.forward(seq, pos, tk)
    got, gam = sess.forward(seq, pos, tk)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(gam, wam)
    sess.close()
    model.close()
    orc.close()


def test_lora_merge_at_mistral_geometry_bit_exact(gpu, tmp_path):
    """the merge kernel at the real matrix sizes: one Mistral-7B-shaped layer, rank-8 adapter on attn_q (Q4_K, 4096 x 4096), ffn_down (Q6_K,
    4096 x 14336: 56 blocks per row) and output.weight (Q6_K, 32000 x 4096: 512 000 blocks) — merged blocks and logits equal to the oracle's"""
    import gguf_util
    hp = gpu.MISTRAL_7B()
    hp.n_layer = 1
    model = gpu.LlmModel(hp)
    hp = model.hparams
    cfg = oracle_cfg_from(hp, 16, 2)
    base = O.OracleLlm(cfg, seed=4)
    rng = np.random.default_rng(51)
    shapes = {(0, 1): (hp.n_head * hp.head_dim, hp.d_model), (0, 8): (hp.d_model, hp.d_ff), (-1, O.T_OUTPUT): (hp.vocab, hp.d_model)}
    fs = {k: (rng.normal(0, 0.02, (8, kk)).astype(np.float32), rng.normal(0, 0.02, (n, 8)).astype(np.float32)) for k, (n, kk) in shapes.items()}
    path = str(tmp_path / "m.gguf")
    gguf_util.write_lora_gguf(path, 16.0, fs)
    model.set_lora(path)
    copy_oracle_weights(base, model, 1)
    assert model.lora_merged == 3
    for (layer, w), (A, B) in fs.items():
        base.apply_lora(layer, w, A, B, 2.0)
    sess = gpu.LlmSession(model, 2, 16)
    seq, pos, tok = [0, 0, 1], [0, 1, 0], [1, 777, 31999]
    want, wam = base.forward(seq, pos, tok)
    got, gam = sess.forward(seq, pos, tok)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    assert np.array_equal(gam, wam)
    sess.close()
    model.close()
    base.close()


@pytest.mark.parametrize("f16", [False, True])
def test_tiny_long_context_decode_head_dim_64(gpu, f16):
    """the long-context decode form at head_dim 64 (scores per 64-position block on the matrix pipe, one PV chain per (row, head, class) wave):
    three sequences prefilled to 530 / 601 / 550 positions through prompt chunks, then decode rows at those positions, alone and together (one to
    four rows: the form from position 512), logits = oracle; and a short device-side decode loop that crosses the switch position
    (508 .. 515) gives the oracle's ids"""
    hp = gpu.TINY()
    model = gpu.LlmModel(hp)
    hp = model.hparams
    assert gpu.attention_plan(3, hp.n_head, hp.n_kv_head, hp.head_dim, 768, True, top_position=601)[0] == 3
    assert gpu.attention_plan(3, hp.n_head, hp.n_kv_head, hp.head_dim, 768, True, top_position=500)[0] != 3
    assert gpu.attention_plan(1, hp.n_head, hp.n_kv_head, hp.head_dim, 768, True, top_position=512)[0] == 3
    orc = O.OracleLlm(oracle_cfg_from(hp, 768, 4), seed=4, f16=f16)   # f16: the fp16 checkpoint recipe (the attention output also feeds the tiled GEMM's f16-rounded image)
    copy_oracle_weights(orc, model, hp.n_layer)
    sess = gpu.LlmSession(model, 4, 768)
    rng = np.random.default_rng(43)
    lens = {0: 530, 1: 601, 2: 550, 3: 508}
    for sq, n in lens.items():
        tok = rng.integers(3, hp.vocab, n).astype(np.int32)
        for lo in range(0, n, 256):
            hi = min(n, lo + 256)
            seq, pos = np.full(hi - lo, sq, np.int32), np.arange(lo, hi, dtype=np.int32)
            orc.forward(seq, pos, tok[lo:hi], want_logits=False)
            sess.forward(seq, pos, tok[lo:hi], want_logits=False)
    for rows in ([1], [0, 1, 2], [2]):
        seq = np.array(rows, np.int32)
        pos = np.array([lens[r] for r in rows], np.int32)
        tok = rng.integers(3, hp.vocab, len(rows)).astype(np.int32)
        want, wam = orc.forward(seq, pos, tok)
        got, gam = sess.forward(seq, pos, tok)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rows, np.abs(got - want).max())
        assert np.array_equal(gam, wam)
        for r in rows:
            lens[r] += 1
    # sequence 3 alone through the device-side loop: positions 508 .. 515 cross the 512 switch inside one decode() call
    tok = rng.integers(3, hp.vocab, 1).astype(np.int32)
    _, am = orc.forward([3], [508], tok, want_logits=False)
    _, gam = sess.forward([3], [508], tok, want_logits=False)
    assert np.array_equal(am, gam)
    want = []
    cur = int(am[0])
    for i in range(7):
        _, am = orc.forward([3], [509 + i], [cur], want_logits=False)
        cur = int(am[0])
        want.append(cur)
    toks, _ = sess.decode(1, 7)
    assert [int(t) for t in toks[:7, 0]] == want
    sess.close()
    model.close()
    orc.close()


def test_runner_long_prompt_through_the_reference_surface(gpu):
    """tk_llm_runner_prepare_generation with a 700-byte prompt (701 tokens: prompt chunks of 256 rows through k_attention_prefill) and
    tk_llm_runner_generate_next_token past position 701 (one-row passes in the long-context decode form), behind the batcher: every id the
    oracle's"""
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=4")
    hp = gpu.LlmHParams()
    gpu.lib().tk_mi355x_llm_model_get_hparams(h, __import__("ctypes").byref(hp))
    assert gpu.attention_plan(1, hp.n_head, hp.n_kv_head, hp.head_dim, 1024, True, top_position=701)[0] == 3
    runner = gpu.LlmRunner(h, context_size=1024)
    text = "".join(chr(97 + (i * 7) % 26) for i in range(700))
    runner.prepare(text)
    pieces = []
    for _ in range(10):
        p = runner.next_token()
        if p is None:
            break
        pieces.append(p)
    orc = O.OracleLlm(oracle_cfg_from(hp, 1024, 1), seed=4)
    ids = [1] + [3 + ord(ch) for ch in text]
    am = None
    for lo in range(0, len(ids), 256):
        hi = min(len(ids), lo + 256)
        _, am = orc.forward(np.zeros(hi - lo, np.int32), np.arange(lo, hi, dtype=np.int32), ids[lo:hi], want_logits=False)
    cur, want = int(am[-1]), []
    for i in range(len(pieces)):
        want.append(cur)
        _, am = orc.forward([0], [len(ids) + i], [cur], want_logits=False)
        cur = int(am[0])
    got = [3 + p[0] if len(p) == 1 else int(p.decode().strip()[1:]) for p in pieces]
    assert len(pieces) >= 4 and got == want
    runner.close()
    loader.unload(h)
    loader.close()
    orc.close()
