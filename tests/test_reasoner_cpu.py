"""CPU: prompt assembly (tk_contextual_reasoner_*) and LLM-response parsing (tk_decision_engine_parse_llm_response_text) — host code either
side of the runner.  The reference TUs do not compile here (include/tk/tk_reasoner.h), so the expected strings are derived by hand from
the format strings and limits in src/cortex/tk_contextual_reasoner.c:681-743,1015-1093 and tk_decision_engine.c:1632-1810."""
import ctypes as C

import pytest

import trackiellm_amd as tk
from trackiellm_amd.vision import Rect, VisionObject, VisionResult


class ContextConfig(C.Structure):
    _fields_ = [("max_context_history_items", C.c_size_t), ("max_conversation_history_turns", C.c_size_t), ("context_relevance_threshold", C.c_float),
                ("memory_decay_rate", C.c_float), ("context_update_interval_ms", C.c_uint32)]


class TraversabilityMap(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("resolution_m_per_cell", C.c_float), ("grid", C.c_void_p)]


class FreeSpace(C.Structure):
    _fields_ = [("sectors", C.c_void_p), ("sector_count", C.c_size_t), ("is_any_path_clear", C.c_bool), ("clearest_path_angle_deg", C.c_float),
                ("clearest_path_distance_m", C.c_float)]


class Vec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Obstacle(C.Structure):
    _fields_ = [("id", C.c_uint32), ("status", C.c_int), ("position_m", Vec2), ("velocity_mps", Vec2), ("dimensions_m", Vec2), ("age_frames", C.c_uint32),
                ("unseen_frames", C.c_uint32)]


class WorldState(C.Structure):
    _fields_ = [("last_update_timestamp_ns", C.c_uint64), ("orientation", C.c_float * 4), ("motion_state", C.c_int), ("is_speech_detected", C.c_bool)]


class Turn(C.Structure):
    _fields_ = [("timestamp_ns", C.c_uint64), ("is_user_input", C.c_bool), ("content", C.c_char_p), ("confidence", C.c_float)]


class Summary(C.Structure):
    _fields_ = [("visible_object_count", C.c_size_t), ("visible_objects", C.POINTER(VisionObject)), ("has_clear_path", C.c_bool),
                ("clear_path_direction_deg", C.c_float), ("clear_path_distance_m", C.c_float), ("hazard_count", C.c_size_t), ("hazards", C.c_void_p),
                ("conversation_turn_count", C.c_size_t), ("recent_conversation", C.POINTER(Turn)), ("recent_events_summary", C.c_char_p),
                ("is_navigation_active", C.c_bool), ("is_listening_for_commands", C.c_bool), ("system_confidence", C.c_float),
                ("user_motion_state", C.c_int), ("detected_sound_type", C.c_int), ("detected_navigation_cue", C.c_int)]


FIRE_ALARM, CAR_HORN, SIREN, BABY, DOORBELL = 1, 2, 3, 4, 5          # tk_ambient_sound_type_e
STEP_UP, STEP_DOWN, DOORWAY, STAIRS_UP, STAIRS_DOWN = 1, 2, 3, 4, 5  # tk_navigation_cue_type_e
STATIONARY, WALKING, RUNNING, FALLING = 1, 2, 3, 4                   # tk_motion_state_e


class Reasoner:
    def __init__(self, items=100, turns=20, threshold=0.3, decay=0.1):
        self.h = C.c_void_p()
        cfg = ContextConfig(items, turns, threshold, decay, 100)
        assert tk.lib().tk_contextual_reasoner_create(C.byref(self.h), C.byref(cfg)) == 0

    def vision(self, objs):
        arr = (VisionObject * max(len(objs), 1))()
        for i, (label, dist, conf) in enumerate(objs):
            arr[i] = VisionObject(i, label.encode(), conf, Rect(0, 0, 1, 1), dist, 0, 0, False, None, None)
        res = VisionResult(0, 1, len(objs), arr, 0, None, None, None)
        assert tk.lib().tk_contextual_reasoner_update_vision_context(self.h, C.byref(res)) == 0

    def say(self, user, text, conf=0.9):
        assert tk.lib().tk_contextual_reasoner_add_conversation_turn(self.h, user, text.encode(), C.c_float(conf)) == 0

    def nav(self, clear, d, m, obstacles=()):
        """tk_contextual_reasoner_update_navigation_context with a free-space analysis (and tracked obstacles) of the navigation engine's types"""
        tmap = TraversabilityMap(4, 4, 0.1, None)
        fs = FreeSpace(None, 0, clear, d, m)
        arr = (Obstacle * max(len(obstacles), 1))()
        for i, (x, y, w, dpt) in enumerate(obstacles):
            arr[i] = Obstacle(i, 1, Vec2(x, y), Vec2(0, 0), Vec2(w, dpt), 3, 0)
        return tk.lib().tk_contextual_reasoner_update_navigation_context(self.h, C.byref(tmap), C.byref(fs), arr, C.c_size_t(len(obstacles)))

    def summary(self):
        s = Summary()
        assert tk.lib().tk_contextual_reasoner_get_context_summary(self.h, C.byref(s)) == 0
        return s

    def context(self, budget=2048):
        p = C.c_char_p()
        assert tk.lib().tk_contextual_reasoner_generate_context_string(self.h, C.byref(p), C.c_size_t(budget)) == 0
        s = p.value.decode()
        tk.lib().tk_contextual_reasoner_free_context_string.argtypes = [C.c_void_p]
        tk.lib().tk_contextual_reasoner_free_context_string(C.cast(p, C.c_void_p))
        return s

    def stats(self):
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        assert tk.lib().tk_contextual_reasoner_get_memory_stats(self.h, C.byref(a), C.byref(b), C.byref(c)) == 0
        return a.value, b.value, c.value

    def close(self):
        tk.lib().tk_contextual_reasoner_destroy(C.byref(self.h))


def test_context_string_formats_and_limits():
    r = Reasoner()
    assert r.context() == "No visible objects No clear path. 0 hazards detected. No recent conversation"
    r.vision([("person", 1.5, 0.91), ("chair", 3.25, 0.65), ("dog", 0.0, 0.995), ("cup", 2.0, 0.8)])
    # three objects at most, "%s (%.1fm, %.0f%% confidence); " joined, trailing "; " stripped
    env = "person (1.5m, 91% confidence); chair (3.2m, 65% confidence); dog (0.0m, 100% confidence)"
    assert r.context().startswith(env + " No clear path.")
    assert r.stats()[0] == 3                                   # context items only for confidence >= 0.7: person, dog, cup
    assert r.nav(True, 15.4, 2.26) == 0
    nav = "Clear path ahead at 15°, 2.3m away. 0 hazards detected."    # the reference resets the hazard list per update and never refills it (.c:463)
    r.say(True, "where is the door")
    r.say(False, "it is ahead of you")
    r.say(True, "thanks")
    r.say(True, "and the chair?")
    conv = 'User: "and the chair?"; User: "thanks"; System: "it is ahead of you"'      # newest first, three turns
    assert r.context() == env + " " + nav + " " + conv
    assert r.stats()[2] == 4 and r.stats()[0] == 4           # + the "Clear path at 15°, distance 2.3m" navigation item
    # budget: a part that does not fit max_token_budget * 4 characters is skipped whole, later parts may still fit
    assert r.context(budget=(len(env) + 1) // 4 + 1) == env
    r.close()


def test_context_string_budget_skips_parts_independently():
    r = Reasoner()
    r.say(True, "hi")
    # 10 tokens = 40 chars: environment (18 + 1) fits, navigation (36 + 1) does not, conversation (10 + 1) fits
    assert r.context(budget=10) == 'No visible objects User: "hi"'
    assert r.context(budget=0) == ""
    r.close()


def test_context_fixed_buffers_truncate_like_the_reference():
    r = Reasoner()
    long_label = "x" * 120
    r.vision([(long_label, 1.0, 0.9), (long_label, 2.0, 0.9), ("cat", 1.0, 0.9)])
    # 256-byte buffer: the second entry (120 + 24 chars) no longer fits after the first and stops the list
    assert r.context().startswith(long_label + " (1.0m, 90% confidence) No clear path")
    r.say(True, "a" * 300)
    r.say(False, "b" * 300)
    # 512-byte buffer: 'System: "bbb..."; ' (313 chars) fits, the user turn after it does not
    assert r.context().endswith('System: "' + "b" * 300 + '"')
    assert tk.lib().tk_contextual_reasoner_clear_context(r.h) == 0
    # items, conversation and bookkeeping are cleared; the vision snapshot is not (tk_contextual_reasoner.c:756-793 does not touch it)
    assert r.context() == long_label + " (1.0m, 90% confidence) No clear path. 0 hazards detected. No recent conversation" and r.stats() == (0, 0, 0)
    r.close()


def test_conversation_ring_overwrites_oldest():
    r = Reasoner(turns=2)
    for i in range(5):
        r.say(i % 2 == 0, "t%d" % i)
    assert r.context().endswith('User: "t4"; System: "t3"') and r.stats()[2] == 2
    r.close()
    assert tk.lib().tk_contextual_reasoner_create(None, None) == 1001
    assert tk.lib().tk_contextual_reasoner_generate_context_string(None, None, 1) == 1001


class _Speak(C.Structure):
    _fields_ = [("text", C.c_char_p), ("priority", C.c_int), ("volume", C.c_float)]


class _Guide(C.Structure):
    _fields_ = [("direction_deg", C.c_float), ("distance_m", C.c_float), ("instruction", C.c_char_p)]


class _Warn(C.Structure):
    _fields_ = [("urgency", C.c_int), ("warning_text", C.c_char_p), ("hazard_distance_m", C.c_float), ("hazard_direction_deg", C.c_float), ("obstacle_id", C.c_uint32)]


class _DescObj(C.Structure):
    _fields_ = [("object_id", C.c_uint32), ("object_label", C.c_char_p), ("distance_m", C.c_float)]


class _Setting(C.Structure):
    _fields_ = [("setting_name", C.c_char_p), ("setting_value", C.c_char_p)]


class _Alert(C.Structure):
    _fields_ = [("alert_message", C.c_char_p), ("repeat_alert", C.c_bool), ("repeat_interval_ms", C.c_uint32)]


class _DescEnv(C.Structure):
    _fields_ = [("a", C.c_bool), ("b", C.c_bool), ("c", C.c_bool), ("detail", C.c_float)]


class _Read(C.Structure):
    _fields_ = [("text_content", C.c_char_p), ("reading_speed", C.c_float)]


class _Query(C.Structure):
    _fields_ = [("response_text", C.c_char_p), ("requires_context", C.c_bool)]


class _Params(C.Union):
    _fields_ = [("speak", _Speak), ("navigate_guide", _Guide), ("navigate_warn", _Warn), ("describe_environment", _DescEnv), ("describe_object", _DescObj),
                ("read_text", _Read), ("system_setting", _Setting), ("user_query_response", _Query), ("emergency_alert", _Alert)]


class Action(C.Structure):
    _fields_ = [("type", C.c_int), ("confidence", C.c_float), ("timeout_ms", C.c_uint32), ("params", _Params)]


class LlmResponse(C.Structure):
    _fields_ = [("response_text", C.c_char_p), ("priority", C.c_int), ("action_count", C.c_size_t), ("actions", C.POINTER(Action))]


def parse(text):
    out = C.POINTER(LlmResponse)()
    rc = tk.lib().tk_decision_engine_parse_llm_response_text(text.encode() if text is not None else None, C.byref(out))
    return rc, out


def test_parse_llm_response_schema():
    rc, r = parse('{"response_text": "Door ahead \\u00e9\\n", "priority": "high", "actions": ['
                  '{"type": "SPEAK", "confidence": 0.9, "params": {"text": "hello"}},'
                  '{"type": "NAVIGATE_GUIDE", "confidence": 0.75, "params": {"instruction": "turn left", "direction_deg": -30.5}},'
                  '{"type": "NAVIGATE_WARN", "params": {"warning_text": "step", "obstacle_id": 7}},'
                  '{"type": "DESCRIBE_OBJECT", "confidence": 1, "params": {"object_id": 3.9, "object_label": "cup"}},'
                  '{"type": "EMERGENCY_ALERT", "confidence": 1e0, "params": {"alert_message": "fire", "repeat_alert": true, "repeat_interval_ms": 5000}},'
                  '{"type": "SYSTEM_SETTING", "confidence": 0.5, "params": {"setting_name": "volume"}},'
                  '{"type": "SYSTEM_MODE_CHANGE", "confidence": 0.5, "params": {}}]} trailing text is ignored')
    assert rc == 0
    v = r.contents
    assert v.response_text == "Door ahead é\n".encode() and v.priority == 1 and v.action_count == 7
    a = v.actions
    assert a[0].type == 0 and abs(a[0].confidence - 0.9) < 1e-7 and a[0].params.speak.text == b"hello"
    assert a[1].type == 1 and a[1].params.navigate_guide.instruction == b"turn left" and a[1].params.navigate_guide.direction_deg == -30.5
    assert a[2].type == 2 and a[2].confidence == 0.0 and a[2].params.navigate_warn.warning_text == b"step" and a[2].params.navigate_warn.obstacle_id == 7
    assert a[3].type == 4 and a[3].params.describe_object.object_id == 3 and a[3].params.describe_object.object_label == b"cup"
    assert a[4].type == 9 and a[4].params.emergency_alert.alert_message == b"fire" and a[4].params.emergency_alert.repeat_alert and \
        a[4].params.emergency_alert.repeat_interval_ms == 5000
    assert a[5].type == 7 and a[5].params.system_setting.setting_name == b"volume" and a[5].params.system_setting.setting_value == b""
    assert a[6].type == 6
    tk.lib().tk_decision_engine_free_response(C.byref(r))
    assert not r
    # defaults: no actions array is not an error; unknown priority strings are "normal"; "critical" is the emergency level
    rc, r = parse('{"priority": "critical"}')
    assert rc == 0 and r.contents.response_text == b"" and r.contents.priority == 0 and r.contents.action_count == 0 and not r.contents.actions
    tk.lib().tk_decision_engine_free_response(C.byref(r))
    rc, r = parse('{"response_text": 5, "priority": "whatever", "actions": []}')
    assert rc == 0 and r.contents.response_text == b"" and r.contents.priority == 2
    tk.lib().tk_decision_engine_free_response(C.byref(r))


@pytest.mark.parametrize("text", ["not json", "", "{", '{"actions": [5]}', '{"actions": [{"params": {}}]}', '{"actions": [{"type": "FLY", "params": {}}]}',
                                  '{"actions": [{"type": "SPEAK"}]}', '{"actions": [{"type": "SPEAK", "params": []}]}', '{"a": "\\ud800"}'])
def test_parse_llm_response_rejections(text):
    rc, r = parse(text)
    assert rc == 3005 and not r                                   # TK_ERROR_INVALID_FORMAT (= TK_ERROR_CONFIG_PARSE_FAILED)


def test_parse_argument_errors():
    assert parse(None)[0] == 1001
    assert tk.lib().tk_decision_engine_parse_llm_response_text(b"{}", None) == 1001
    tk.lib().tk_decision_engine_free_response(None)


# ---- the rest of the reasoner surface (tk_contextual_reasoner.h:198-389) -------------------------------------------------------------

def test_ambient_sound_navigation_cue_and_motion_updates():
    """formats and priorities of src/cortex/tk_contextual_reasoner.c:243-350,1121-1160, read back through the summary (.c:626-677)"""
    L = tk.lib()
    r = Reasoner()
    s = r.summary()
    assert (s.detected_sound_type, s.detected_navigation_cue, s.user_motion_state) == (0, 0, 0)
    assert abs(s.system_confidence - 0.8) < 1e-6 and not s.is_listening_for_commands and s.recent_events_summary is None
    assert L.tk_contextual_reasoner_update_ambient_sound(r.h, SIREN, C.c_float(0.87)) == 0
    assert L.tk_contextual_reasoner_update_navigation_cues(r.h, DOORWAY, C.c_float(2.46)) == 0
    ws = WorldState(1, (C.c_float * 4)(1, 0, 0, 0), WALKING, False)
    assert L.tk_contextual_reasoner_update_motion_context(r.h, C.byref(ws)) == 0
    assert L.tk_contextual_reasoner_update_motion_context(r.h, C.byref(ws)) == 0        # unchanged state: no second item
    assert r.stats()[0] == 3                                                            # siren, doorway, "User started walking"
    s = r.summary()
    assert (s.detected_sound_type, s.detected_navigation_cue, s.user_motion_state) == (SIREN, DOORWAY, WALKING)
    st = C.c_int(-1)
    assert L.tk_contextual_reasoner_get_motion_state(r.h, C.byref(st)) == 0 and st.value == WALKING
    assert L.tk_contextual_reasoner_update_ambient_sound(r.h, 0, C.c_float(0.0)) == 0   # NONE: forgets the sound, files nothing
    assert r.summary().detected_sound_type == 0 and r.stats()[0] == 3
    # argument checks
    assert L.tk_contextual_reasoner_update_ambient_sound(None, SIREN, C.c_float(1)) == 1001
    assert L.tk_contextual_reasoner_update_navigation_cues(None, DOORWAY, C.c_float(1)) == 1001
    assert L.tk_contextual_reasoner_update_motion_context(r.h, None) == 1001
    assert L.tk_contextual_reasoner_get_motion_state(r.h, None) == 1001
    assert L.tk_contextual_reasoner_get_context_summary(r.h, None) == 1001
    r.close()


def test_navigation_context_and_summary_views():
    L = tk.lib()
    r = Reasoner()
    assert r.nav(False, 0.0, 0.0, obstacles=[(0.5, 1.0, 0.4, 0.3), (3.0, 4.0, 1.0, 1.0)] + [(9, 9, 1, 1)] * 5) == 0
    # "No clear navigation path detected" + the first five obstacles only
    assert r.stats()[0] == 6
    s = r.summary()
    assert not s.has_clear_path and not s.is_navigation_active and s.hazard_count == 0
    assert r.nav(True, -20.0, 3.5) == 0
    s = r.summary()
    assert s.has_clear_path and s.is_navigation_active and abs(s.clear_path_direction_deg + 20.0) < 1e-6 and abs(s.clear_path_distance_m - 3.5) < 1e-6
    assert L.tk_contextual_reasoner_update_navigation_context(r.h, None, None, None, 0) == 1001
    # the summary's views of the vision snapshot and of the conversation ring
    r.vision([("person", 1.5, 0.91), ("door", 2.5, 0.8)])
    r.say(True, "hello")
    r.say(False, "hi there", 0.5)
    s = r.summary()
    assert s.visible_object_count == 2 and s.visible_objects[0].label == b"person" and abs(s.visible_objects[1].distance_meters - 2.5) < 1e-6
    assert s.conversation_turn_count == 2
    assert s.recent_conversation[0].content == b"hello" and s.recent_conversation[0].is_user_input
    assert s.recent_conversation[1].content == b"hi there" and not s.recent_conversation[1].is_user_input and abs(s.recent_conversation[1].confidence - 0.5) < 1e-6
    r.close()


def test_process_context_decays_and_prunes():
    """.c:604-622, 965-1012: relevance *= exp(-rate * age_s); items below the threshold go, order kept, the ring continues after the survivors"""
    import math, time
    L = tk.lib()
    r = Reasoner(items=4, threshold=0.5, decay=1.0)
    r.vision([("cat", 1.0, 0.72)])                     # relevance = confidence 0.72
    assert L.tk_contextual_reasoner_add_context_item(r.h, 0, 2, b"custom", None, 0) == 0   # relevance 1.0
    now = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    assert L.tk_contextual_reasoner_process_context(r.h, C.c_uint64(now)) == 0             # age ~ 0: both stay
    assert r.stats()[0] == 2
    # 0.5 s later: 0.72 * exp(-0.5) = 0.437 < 0.5 goes; 1.0 * exp(-0.5) = 0.607 stays
    assert L.tk_contextual_reasoner_process_context(r.h, C.c_uint64(now + 500_000_000)) == 0
    assert r.stats()[0] == 1
    assert 0.72 * math.exp(-0.5) < 0.5 < math.exp(-0.5)
    # decay compounds on the stored score: one more second takes 0.607 * exp(-1.5) = 0.135 below the threshold
    assert L.tk_contextual_reasoner_process_context(r.h, C.c_uint64(now + 1_500_000_000)) == 0
    assert r.stats()[0] == 0
    for i in range(6):                                  # the ring is usable again and wraps at its capacity
        assert L.tk_contextual_reasoner_add_context_item(r.h, 0, 2, b"item%d" % i, None, 0) == 0
    assert r.stats()[0] == 4
    assert L.tk_contextual_reasoner_process_context(None, 0) == 1001
    r.close()


# ---- the reference's own (enabled) test of this row: tests/tk_cortex_full_test.c:36-70 ------------------------------------------------

class _ModelPaths(C.Structure):
    _fields_ = [(n, C.c_char_p) for n in ("llm_model", "object_detection_model", "depth_estimation_model", "asr_model", "tts_model_dir", "vad_model", "tesseract_data_dir")]


class _CortexConfig(C.Structure):
    _fields_ = [("model_paths", _ModelPaths), ("gpu_device_id", C.c_int), ("main_loop_frequency_hz", C.c_float), ("user_language", C.c_char_p), ("user_data", C.c_void_p)]


class _Callbacks(C.Structure):
    _fields_ = [("on_state_change", C.c_void_p), ("on_tts_audio_ready", C.c_void_p)]


def test_reference_stress_scenario_prioritization_replayed():
    """tests/tk_cortex_full_test.c (enabled at tests/CMakeLists.txt:15-17), step for step: a cortex with every model path NULL on device -1,
    its reasoner bound to the prompt generator, a fire alarm (0.9) and a step down (1.0 m), the user's question — and the test's five
    assertions on the prompt.  Runs without a GPU."""
    L = tk.lib()
    L.tk_cortex_get_contextual_reasoner.restype = C.c_void_p
    L.tk_cortex_get_contextual_reasoner.argtypes = [C.c_void_p]
    L.tk_cortex_generate_prompt.restype = C.c_bool
    cfg = _CortexConfig(_ModelPaths(), -1, 10.0, b"pt-BR", None)
    cx = C.c_void_p()
    assert L.tk_cortex_create(C.byref(cx), C.byref(cfg), _Callbacks()) == 0 and cx.value
    reasoner = L.tk_cortex_get_contextual_reasoner(cx)
    assert reasoner
    L.tk_cortex_rust_init_reasoner(C.c_void_p(reasoner))
    assert L.tk_contextual_reasoner_update_ambient_sound(C.c_void_p(reasoner), FIRE_ALARM, C.c_float(0.9)) == 0
    assert L.tk_contextual_reasoner_update_navigation_cues(C.c_void_p(reasoner), STEP_DOWN, C.c_float(1.0)) == 0
    query = "Onde está minha garrafa de água?"
    buf = C.create_string_buffer(2048)
    assert L.tk_cortex_generate_prompt(buf, C.c_size_t(2048), query.encode())
    prompt = buf.value.decode()
    assert "URGENTE" in prompt
    assert "ALARME DE INCÊNDIO DETECTADO" in prompt
    assert "degrau para baixo" in prompt
    assert "garrafa de água" in prompt
    assert prompt.index("URGENTE") < prompt.index("garrafa de água")
    # the whole string, from src/cortex/src/reasoning.rs:452-493
    assert prompt == ("URGENTE: ALARME DE INCÊNDIO DETECTADO. Há um degrau para baixo à frente. O usuário está parado. "
                      "O usuário perguntou: 'Onde está minha garrafa de água?'. Com base em tudo isso, qual a ação mais segura e útil?")
    # a cortex without models refuses data and needs no GPU
    assert L.tk_cortex_inject_audio_frame(cx, (C.c_int16 * 4)(), C.c_size_t(4)) == 1002
    # the other branches of the generator: falling, stairs, running, the user_name fact, truncation, NULL query
    ws = WorldState(1, (C.c_float * 4)(1, 0, 0, 0), FALLING, False)
    assert L.tk_contextual_reasoner_update_motion_context(C.c_void_p(reasoner), C.byref(ws)) == 0
    assert L.tk_contextual_reasoner_update_navigation_cues(C.c_void_p(reasoner), STAIRS_UP, C.c_float(2.0)) == 0
    assert L.tk_contextual_reasoner_update_ambient_sound(C.c_void_p(reasoner), DOORBELL, C.c_float(0.5)) == 0
    L.tk_cortex_rust_set_fact(b"user_name", "João".encode())
    assert L.tk_cortex_generate_prompt(buf, C.c_size_t(2048), None)
    assert buf.value.decode() == ("URGENTE: QUEDA DO USUÁRIO DETECTADA. Há escadas para cima à frente. O usuário está parado. O nome do usuário é João. "
                                  "O usuário perguntou: ''. Com base em tudo isso, qual a ação mais segura e útil?")
    ws.motion_state = RUNNING
    assert L.tk_contextual_reasoner_update_motion_context(C.c_void_p(reasoner), C.byref(ws)) == 0
    small = C.create_string_buffer(24)
    assert L.tk_cortex_generate_prompt(small, C.c_size_t(24), b"x")
    assert small.raw[23] == 0 and small.value == "Há escadas para cima à frente. O usuário está correndo.".encode()[:23]
    assert not L.tk_cortex_generate_prompt(None, C.c_size_t(10), b"x") and not L.tk_cortex_generate_prompt(small, C.c_size_t(0), b"x")
    L.tk_cortex_destroy(C.byref(cx))
    # the generator was bound to the destroyed cortex's reasoner: unbound now, it writes the reference's fallback prompt (ffi.rs:399-403)
    assert L.tk_cortex_generate_prompt(buf, C.c_size_t(2048), b"x") and buf.value == b"An error occurred. Please describe the general situation."


def test_reference_cortex_full_test_as_a_c_host(tmp_path):
    """the same scenario as a strict-C11 program against the public headers, linked with the library like trackie-core's test target would be
    (tests/CMakeLists.txt:15-17 links `cortex`): exit status 0 = the test's five assertions hold"""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "full_test.c"
    src.write_text('''
#include <stdio.h>
#include <string.h>
#include "tk/tk_cortex.h"
#include "tk/tk_reasoner.h"
int main(void) {
    tk_cortex_config_t config;
    tk_cortex_callbacks_t callbacks;
    tk_cortex_t* cortex = NULL;
    tk_contextual_reasoner_t* reasoner;
    char prompt[2048];
    const char *urgent, *query;
    memset(&config, 0, sizeof config);          /* every model path NULL */
    memset(&callbacks, 0, sizeof callbacks);
    config.gpu_device_id = -1;
    config.main_loop_frequency_hz = 10.0f;
    config.user_language = "pt-BR";
    if (tk_cortex_create(&cortex, &config, callbacks) != TK_SUCCESS || !cortex) return 1;
    reasoner = tk_cortex_get_contextual_reasoner(cortex);
    if (!reasoner) return 2;
    tk_cortex_rust_init_reasoner(reasoner);
    if (tk_contextual_reasoner_update_ambient_sound(reasoner, TK_AMBIENT_SOUND_FIRE_ALARM, 0.9f) != TK_SUCCESS) return 3;
    if (tk_contextual_reasoner_update_navigation_cues(reasoner, TK_NAVIGATION_CUE_STEP_DOWN, 1.0f) != TK_SUCCESS) return 4;
    if (!tk_cortex_generate_prompt(prompt, sizeof prompt, "Onde est\\xC3\\xA1 minha garrafa de \\xC3\\xA1gua?")) return 5;
    puts(prompt);
    urgent = strstr(prompt, "URGENTE");
    query = strstr(prompt, "garrafa de \\xC3\\xA1gua");
    if (!urgent) return 6;
    if (!strstr(prompt, "ALARME DE INC\\xC3\\x8ANDIO DETECTADO")) return 7;
    if (!strstr(prompt, "degrau para baixo")) return 8;
    if (!query) return 9;
    if (!(urgent < query)) return 10;
    tk_cortex_destroy(&cortex);
    return cortex ? 11 : 0;
}
''')
    exe = tmp_path / "full_test"
    libdir = os.path.join(root, "trackiellm_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-I" + os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L" + libdir, "-ltrackie_mi355x", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "URGENTE" in out.stdout
