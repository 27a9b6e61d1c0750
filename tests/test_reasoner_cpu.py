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


class Reasoner:
    def __init__(self, items=100, turns=20):
        self.h = C.c_void_p()
        cfg = ContextConfig(items, turns, 0.3, 0.1, 100)
        assert tk.lib().tk_contextual_reasoner_create(C.byref(self.h), C.byref(cfg)) == 0

    def vision(self, objs):
        arr = (VisionObject * max(len(objs), 1))()
        for i, (label, dist, conf) in enumerate(objs):
            arr[i] = VisionObject(i, label.encode(), conf, Rect(0, 0, 1, 1), dist, 0, 0, False, None, None)
        res = VisionResult(0, 1, len(objs), arr, 0, None, None, None)
        assert tk.lib().tk_contextual_reasoner_update_vision_context(self.h, C.byref(res)) == 0

    def say(self, user, text, conf=0.9):
        assert tk.lib().tk_contextual_reasoner_add_conversation_turn(self.h, user, text.encode(), C.c_float(conf)) == 0

    def nav(self, clear, d, m, hz):
        assert tk.lib().tk_mi355x_reasoner_set_navigation(self.h, clear, C.c_float(d), C.c_float(m), C.c_size_t(hz)) == 0

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
    r.nav(True, 15.4, 2.26, 2)
    nav = "Clear path ahead at 15°, 2.3m away. 2 hazards detected."
    r.say(True, "where is the door")
    r.say(False, "it is ahead of you")
    r.say(True, "thanks")
    r.say(True, "and the chair?")
    conv = 'User: "and the chair?"; User: "thanks"; System: "it is ahead of you"'      # newest first, three turns
    assert r.context() == env + " " + nav + " " + conv
    assert r.stats()[2] == 4
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
    assert r.context() == "No visible objects No clear path. 0 hazards detected. No recent conversation" and r.stats() == (0, 0, 0)
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
