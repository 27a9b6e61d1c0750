"""GPU: the reference's ROCm HAL surface and the minimal cortex loop (mirrors tests/tk_cortex_test.cpp of the reference)."""
import ctypes as C
import time

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def test_hal_dispatcher_preprocess_and_depth(gpu):
    d = gpu.RocmDispatcher(0)
    rng = np.random.default_rng(6)
    frame = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    src = d.malloc(frame.nbytes)
    dst = d.malloc(3 * 32 * 32 * 4)
    d.upload(src, frame)
    p = gpu.PreprocessParams(d.ptr(src), 64, 48, 64 * 3, d.ptr(dst), 32, 32, gpu.Float3(0.485, 0.456, 0.406), gpu.Float3(0.229, 0.224, 0.225),
                             np.float32(1.0) / np.float32(255.0))
    assert gpu.lib().tk_rocm_dispatch_preprocess_image(d.h, C.byref(p)) == 0
    out = d.download(dst, (3, 32, 32), np.float32)
    assert np.array_equal(out.view(np.uint32), O.preprocess(frame, 32, 32).view(np.uint32))      # canonical CPU formula
    # kernel launcher with an explicit stream (tk_kernels_preprocess_image(params, stream))
    assert gpu.lib().tk_kernels_preprocess_image(C.byref(p), d.stream()) == 0
    assert np.array_equal(d.download(dst, (3, 32, 32), np.float32), out)
    # depth post-process + unprojection
    raw = rng.random((24, 40), dtype=np.float32)
    raw[3, 5] = -1.0
    a, b, pc = d.malloc(raw.nbytes), d.malloc(raw.nbytes), d.malloc(raw.size * 12)
    d.upload(a, raw)
    pp = gpu.DepthPostParams(d.ptr(a), 40, 24, d.ptr(b), 2.5, 0.25)
    assert gpu.lib().tk_kernels_postprocess_depth_map(C.byref(pp), d.stream()) == 0
    metric = d.download(b, (24, 40), np.float32)
    assert np.array_equal(metric, raw * np.float32(2.5) + np.float32(0.25))
    dp = gpu.DepthToPointsParams(d.ptr(b), 40, 24, d.ptr(pc), 50.0, 60.0, 20.0, 12.0)
    assert gpu.lib().tk_rocm_dispatch_depth_to_point_cloud(d.h, C.byref(dp)) == 0
    pts = d.download(pc, (24, 40, 3), np.float32)
    u, v = np.meshgrid(np.arange(40, dtype=np.float32), np.arange(24, dtype=np.float32))
    want = np.stack([(u - 20) * metric / np.float32(50), (v - 12) * metric / np.float32(60), metric], -1)
    want[metric <= 0] = 0
    assert np.array_equal(pts, want.astype(np.float32))
    assert gpu.lib().tk_rocm_dispatch_upload_async(d.h, src, frame.ctypes.data_as(C.c_void_p), C.c_size_t(frame.nbytes + 1)) == 1001
    for buf in (src, dst, a, b, pc):
        d.free(buf)
    d.close()


def test_softmax_known_answer_of_the_reference(gpu):
    """replays tests/tk_gpu_softmax_test.cpp:24-69 — the only numeric known-answer test the reference holds: 4 x 256, input i % 256,
    CPU softmax, |gpu - cpu| <= 1e-6 — through the same entry (tk_kernels_softmax(&params, stream)); the expected values are the
    committed fixture tests/golden/softmax_4x256.npz (make_softmax_golden.py)."""
    import os

    class SoftmaxParams(C.Structure):
        _fields_ = [("d_input_tensor", C.c_void_p), ("d_output_tensor", C.c_void_p), ("num_rows", C.c_uint32), ("num_cols", C.c_uint32)]

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "softmax_4x256.npz"))
    x, want, tol = g["input"], g["expected"], float(g["tolerance"])
    assert x.shape == (4, 256) and np.array_equal(x[2], np.arange(256, dtype=np.float32)) and tol == np.float32(1e-6)
    d = gpu.RocmDispatcher(0)
    a, b = d.malloc(x.nbytes), d.malloc(x.nbytes)
    d.upload(a, x)
    p = SoftmaxParams(d.ptr(a), d.ptr(b), 4, 256)
    assert gpu.lib().tk_kernels_softmax(C.byref(p), d.stream()) == 0
    got = d.download(b, (4, 256), np.float32)
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()
    assert np.array_equal(d.download(a, (4, 256), np.float32), x)            # out of place: the input is untouched
    # a width the reference's kernel cannot take (not a power of two, > 1024), in place: the Whisper score rows are 1500 wide
    rng = np.random.default_rng(12)
    y = rng.normal(0, 4, (3, 1500)).astype(np.float32)
    c = d.malloc(y.nbytes)
    d.upload(c, y)
    q = SoftmaxParams(d.ptr(c), d.ptr(c), 3, 1500)
    assert gpu.lib().tk_kernels_softmax(C.byref(q), d.stream()) == 0
    e = np.exp(y.astype(np.float64) - y.max(1, keepdims=True))
    assert np.abs(d.download(c, (3, 1500), np.float32) - e / e.sum(1, keepdims=True)).max() <= tol
    assert gpu.lib().tk_kernels_softmax(None, None) == 1001
    bad = SoftmaxParams(d.ptr(a), d.ptr(b), 0, 256)
    assert gpu.lib().tk_kernels_softmax(C.byref(bad), d.stream()) == 1001
    for buf in (a, b, c):
        d.free(buf)
    d.close()


def test_cortex_cycle_like_reference_test(gpu):
    """reference tests/tk_cortex_test.cpp: one frame + 2 s of PCM, state-change callback must fire > 2 times"""
    cx = gpu.Cortex(llm="synthetic://tiny?seed=4", detector="synthetic://yolov8n?seed=5&cls_bias=0.5")
    cx.set_max_tokens(8)
    assert cx.state() == 2                                       # IDLE after INITIALIZING
    cx.start()
    frame = np.full((480, 640, 3), 128, np.uint8)                # tk_cortex_test.cpp:79-84
    assert cx.inject_frame(frame) == 0
    rng = np.random.default_rng(2)
    loud = np.clip(rng.normal(0, 9000, 16000), -32768, 32767).astype(np.int16)
    quiet = np.zeros(16000, np.int16)                            # tk_cortex_test.cpp:90 (silence)
    for chunk in np.split(np.concatenate([loud, quiet]), 20):    # 100 ms chunks like the reference's mock microphone
        assert cx.inject_audio(chunk) == 0
    deadline = time.time() + 60
    # expected speech segments from the oracle VAD on the same signal (cortex thresholds 0.8 / 500 ms)
    f = np.concatenate([loud, quiet]).astype(np.float32) / np.float32(32768.0)
    nwin = (len(f) - 480) // 160 + 1
    probs = O.vad_probabilities(7, np.stack([f[k * 160:k * 160 + 480] for k in range(nwin)]))
    ev, _ = O.vad_run(probs, threshold=0.8, min_silence_ms=500.0)
    want_segments = sum(1 for _, e in ev if e == 1)
    while time.time() < deadline:
        s = cx.stats()
        if s.frames_processed >= 1 and s.llm_responses >= 1 + want_segments:
            break
        time.sleep(0.05)
    cx.stop()
    s = cx.stats()
    assert s.frames_processed == 1 and s.frames_with_objects == 1
    assert s.speech_segments == want_segments
    assert s.llm_responses == 1 + want_segments and s.llm_tokens == 8 * s.llm_responses
    assert len(cx.states) > 2                                    # the reference's pass criterion (tk_cortex_test.cpp:111-116)
    assert 4 in cx.states and 5 in cx.states                     # PROCESSING and RESPONDING were reported
    assert len(cx.last_response()) > 0
    # the prompt is the contextual reasoner's context string (tk_contextual_reasoner.c:681-743): objects, navigation, conversation
    pr = cx.last_prompt().decode("utf-8", "replace")   # a System turn carries raw bytes of the random model's tokens
    assert "(0.0m, " in pr and "% confidence)" in pr and "No clear path. 0 hazards detected." in pr
    assert pr.endswith("No recent conversation") or 'User: "' in pr or 'System: "' in pr
    assert s.responses_parsed == 0 and s.actions_parsed == 0   # a random-weight model does not speak the decision engine's JSON
    assert gpu.lib().tk_cortex_inject_video_frame(cx.h, None) == 1001
    cx.close()


def test_cortex_with_depth_model_reports_distances(gpu, tmp_path):
    """tk_cortex_config_t.model_paths.depth_estimation_model set: the cortex's vision pipeline runs the ENVIRONMENT_AWARENESS preset
    (tk_cortex_main.c:1188) and the fused distances reach the prompt instead of 0.0 m"""
    import re

    import onnx_util as OX
    p = tmp_path / "depth.onnx"
    p.write_bytes(OX.depth_model(OX.depth_weights(11)))
    cx = gpu.Cortex(llm="synthetic://tiny?seed=4", detector="synthetic://yolov8n?seed=5&cls_bias=-0.45", depth=str(p))
    cx.set_max_tokens(4)
    cx.start()
    frame = np.random.default_rng(4).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    assert cx.inject_frame(frame) == 0
    deadline = time.time() + 60
    while time.time() < deadline and cx.stats().llm_responses < 1:
        time.sleep(0.05)
    cx.stop()
    assert cx.stats().frames_with_objects == 1 and cx.stats().llm_responses >= 1
    pr = cx.last_prompt().decode("utf-8", "replace")
    dist = [float(m) for m in re.findall(r"\((\d+\.\d)m, ", pr)]
    assert dist and any(0.1 <= d <= 10.0 for d in dist), pr[:400]
    cx.close()


def test_module_executors_run_the_three_streams(gpu):
    """tk_mi355x_module_executor (include/tk/tk_module_exec.h): "detect" / "transcribe" / "generate" through the reference's plugin
    signature give what the wrapped tk_* entry points give when called directly"""
    import ctypes as C
    from trackiellm_amd import vision as V, audio as A
    L = gpu.lib()
    L.tk_mi355x_module_executor.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_void_p]

    class CmdDetect(C.Structure):
        _fields_ = [("detector", C.c_void_p), ("frame", C.c_void_p), ("results", C.POINTER(V.DetectionResult)), ("count", C.c_size_t), ("error", C.c_int)]

    class CmdTranscribe(C.Structure):
        _fields_ = [("asr", C.c_void_p), ("pcm", C.c_void_p), ("frame_count", C.c_size_t), ("is_final", C.c_bool), ("result", C.POINTER(A.AsrResult)), ("error", C.c_int)]

    class CmdGenerate(C.Structure):
        _fields_ = [("runner", C.c_void_p), ("prompt", C.c_char_p), ("use_tool_grammar", C.c_bool), ("max_tokens", C.c_int32), ("out_text", C.c_void_p),
                    ("out_cap", C.c_size_t), ("out_len", C.c_size_t), ("n_tokens", C.c_int32), ("tool_call", C.c_bool), ("error", C.c_int)]

    rng = np.random.default_rng(3)
    # vision
    det = gpu.ObjectDetector(model="synthetic://yolov8n?seed=5&cls_bias=-0.45", width=640, height=640, conf=0.5, iou=0.5)
    frame = rng.integers(0, 256, (640, 640, 3), dtype=np.uint8)
    want = det.detect(frame)
    f, keep = V.make_frame(frame)
    cd = CmdDetect(det.h, C.cast(C.pointer(f), C.c_void_p), None, 0, 0)
    assert L.tk_mi355x_module_executor(None, 10, b"detect", C.byref(cd)) == 0 and cd.error == 0
    got = [(cd.results[i].class_id, cd.results[i].label, cd.results[i].confidence, (cd.results[i].bbox.x, cd.results[i].bbox.y, cd.results[i].bbox.w, cd.results[i].bbox.h))
           for i in range(cd.count)]
    L.tk_object_detector_free_results(C.byref(cd.results))
    assert got == want and len(got) > 0
    det.close()
    # audio
    asr = gpu.Asr()
    asr.set_decode_steps(6)
    pcm = np.clip(rng.normal(0, 3000, 16000), -32768, 32767).astype(np.int16)
    want_a = asr.process_audio(pcm, True)
    ct = CmdTranscribe(asr.h, pcm.ctypes.data_as(C.c_void_p), pcm.size, True, None, 0)
    assert L.tk_mi355x_module_executor(None, 20, b"transcribe", C.byref(ct)) == 0 and ct.error == 0 and bool(ct.result)
    got_a = (ct.result.contents.text.decode() if ct.result.contents.text else None, ct.result.contents.text_length, ct.result.contents.confidence, ct.result.contents.is_partial)
    L.tk_asr_whisper_free_result(C.byref(ct.result))
    assert got_a == want_a
    asr.close()
    # cortex / LLM
    loader = gpu.ModelLoader()
    h = loader.load("synthetic://tiny?seed=4")
    runner = gpu.LlmRunner(h, context_size=64)
    runner.prepare("hi there")
    pieces = []
    for _ in range(5):
        p = runner.next_token()
        if p is None or p == "<tool_call>":
            break
        pieces.append(p)
    runner.reset()
    out = C.create_string_buffer(256)
    cg = CmdGenerate(runner.h, b"hi there", False, 5, C.cast(out, C.c_void_p), 256, 0, 0, False, 0)
    assert L.tk_mi355x_module_executor(None, 0, b"generate", C.byref(cg)) == 0 and cg.error == 0
    assert out.raw[:cg.out_len] == b"".join(pieces) and cg.n_tokens == len(pieces) and not cg.tool_call
    small = C.create_string_buffer(1)
    cg2 = CmdGenerate(runner.h, b"hi there", False, 5, C.cast(small, C.c_void_p), 1, 0, 0, False, 0)
    if len(b"".join(pieces)) > 0:
        assert L.tk_mi355x_module_executor(None, 0, b"generate", C.byref(cg2)) == -2 and cg2.error == 1004   # TK_ERROR_BUFFER_TOO_SMALL
    runner.close()
    loader.unload(h)
    loader.close()


def test_cortices_on_one_model_give_the_tokens_each_gives_alone(gpu):
    """the data-dependent cycle at more than one cortex (reference loop: /root/reference/src/cortex/tk_cortex_main.c:1149-1237, 1323-1379):
    three tk_cortex_t handles share one LLM model file; each builds its prompts from ITS frame's detections and ITS audio's transcript.  Driven
    at the same time from three threads (their LLM rows share decode passes behind the runner API) each returns exactly the response — and has
    built exactly the prompt — it produces when it runs alone."""
    import threading

    def cycle(cx, seed, out):
        rng = np.random.default_rng(seed)
        frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
        pcm = np.concatenate([np.clip(rng.normal(0, 9000, 16000), -32768, 32767), np.zeros(9600)]).astype(np.int16)
        base = cx.stats().llm_responses
        for chunk in np.split(pcm, 16):
            assert cx.inject_audio(chunk) == 0
        out["segments"] = int(cx.stats().speech_segments)  # 0 or 1: the synthetic VAD may or may not call this noise speech
        want = base + out["segments"]
        deadline = time.time() + 120
        while cx.stats().llm_responses < want and time.time() < deadline:
            time.sleep(0.01)
        first = (cx.last_prompt(), cx.last_response())
        frames0 = cx.stats().frames_processed
        assert cx.inject_frame(frame) == 0
        while cx.stats().frames_processed == frames0 and time.time() < deadline:
            time.sleep(0.01)
        want += 1  # cls_bias 0.5: every frame has detections
        while cx.stats().llm_responses < want and time.time() < deadline:
            time.sleep(0.01)
        assert cx.stats().llm_responses == want
        out["result"] = (out["segments"], first, (cx.last_prompt(), cx.last_response()))

    def make():
        cx = gpu.Cortex(llm="synthetic://tiny?seed=4", detector="synthetic://yolov8n?seed=5&cls_bias=0.5")
        cx.set_max_tokens(12)
        cx.start()
        return cx

    alone = {}
    for seed in (31, 32, 33):  # one cortex at a time
        cx = make()
        o = {}
        cycle(cx, seed, o)
        alone[seed] = o["result"]
        cx.stop()
        cx.close()
    assert alone[31] != alone[32]                      # the cycles differ: prompts are built from each cortex's own perception
    assert any(alone[sd][0] for sd in (31, 32, 33)), "no seed produced a speech segment: pick others"
    for sd in (31, 32, 33):
        nseg, (p1, r1), (p2, r2) = alone[sd]
        if nseg:                                        # the transcript became a conversation turn of the first prompt
            assert b'User: "' in p1 and len(r1) > 0
        assert b"% confidence)" in p2 and p2 != p1      # the frame's detections entered the second prompt
    cxs = [make(), make(), make()]                      # all at once, one thread each
    outs = [{}, {}, {}]
    th = [threading.Thread(target=cycle, args=(cxs[i], 31 + i, outs[i])) for i in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(3):
        assert outs[i]["result"] == alone[31 + i]
    for cx in cxs:
        cx.stop()
        cx.close()
