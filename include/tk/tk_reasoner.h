/*
 * tk_reasoner.h — the two host-side steps either side of the LLM runner on the hot path:
 *   prompt assembly      tk_contextual_reasoner_generate_context_string   src/cortex/tk_contextual_reasoner.c:681-743
 *                        (+ the state it reads: _update_vision_context :379-441, _add_conversation_turn :521-557, the three
 *                        description generators :1015-1093, _clear_context :756-793)
 *   response parsing     parse_llm_response_text                          src/cortex/tk_decision_engine.c:1632-1810
 *                        (static in the reference, reached through tk_decision_engine_process_llm_response :218; exported here under
 *                        its own name) and tk_decision_engine_free_response (:432 of the header)
 * Same names, argument meaning and error codes as the reference.  Both reference TUs are unbuildable here (undeclared
 * tk_critical_event_cb_t / TK_ERROR_SYSTEM_ERROR, cJSON absent): the formats are restated from the sources cited above and
 * pinned by hand-derived strings in tests/test_reasoner_cpu.py — "parity unpinned" against a compiled reference.
 * Out of scope (navigation / sensor fusion): _update_navigation_context and _update_motion_context take types of those
 * subsystems; the navigation sentence of the context string is driven through tk_mi355x_reasoner_set_navigation instead.
 * Pure host code: no GPU is involved.
 */
#ifndef TK_MI355X_REASONER_H
#define TK_MI355X_REASONER_H

#include "tk_types.h"
#include "tk_vision.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tk_contextual_reasoner_s tk_contextual_reasoner_t;

/* src/cortex/tk_contextual_reasoner.h: tk_context_config_t */
typedef struct {
    size_t max_context_history_items;
    size_t max_conversation_history_turns;
    float context_relevance_threshold;
    float memory_decay_rate;
    uint32_t context_update_interval_ms;
} tk_context_config_t;

typedef enum { TK_CONTEXT_PRIORITY_CRITICAL = 0, TK_CONTEXT_PRIORITY_HIGH = 1, TK_CONTEXT_PRIORITY_MEDIUM = 2, TK_CONTEXT_PRIORITY_LOW = 3, TK_CONTEXT_PRIORITY_COUNT } tk_context_priority_e;
typedef enum {
    TK_CONTEXT_TYPE_ENVIRONMENTAL, TK_CONTEXT_TYPE_NAVIGATIONAL, TK_CONTEXT_TYPE_CONVERSATIONAL, TK_CONTEXT_TYPE_TEMPORAL, TK_CONTEXT_TYPE_USER_STATE,
    TK_CONTEXT_TYPE_SYSTEM_STATE
} tk_context_type_e;

TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_create(tk_contextual_reasoner_t** out_reasoner, const tk_context_config_t* config);
TK_API void tk_contextual_reasoner_destroy(tk_contextual_reasoner_t** reasoner);
/* copies up to 64 objects (labels are copied too: the result may be destroyed afterwards) and files one context item
 * "Detected <label> at <d>m (confidence <c>%)" per object with confidence >= 0.7 */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_update_vision_context(tk_contextual_reasoner_t* reasoner, const tk_vision_result_t* vision_result);
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_add_conversation_turn(tk_contextual_reasoner_t* reasoner, bool is_user_input, const char* content,
                                                                                 float confidence);
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_add_context_item(tk_contextual_reasoner_t* reasoner, tk_context_type_e type,
                                                                            tk_context_priority_e priority, const char* description, const void* data,
                                                                            size_t data_size);
/* "<environment> <navigation> <conversation>", each part dropped when it does not fit max_token_budget * 4 characters; caller frees
 * with tk_contextual_reasoner_free_context_string */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_generate_context_string(tk_contextual_reasoner_t* reasoner, char** out_context_string,
                                                                                   size_t max_token_budget);
TK_API tk_error_code_t tk_contextual_reasoner_free_context_string(char* ptr);
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_clear_context(tk_contextual_reasoner_t* reasoner);
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_get_memory_stats(tk_contextual_reasoner_t* reasoner, size_t* out_total_items,
                                                                            size_t* out_total_memory_bytes, size_t* out_conversation_turns);
/* extension: the navigation state the reference fills from its path planner (out of scope) */
TK_API TK_NODISCARD tk_error_code_t tk_mi355x_reasoner_set_navigation(tk_contextual_reasoner_t* reasoner, bool has_clear_path, float direction_deg,
                                                                      float distance_m, size_t hazard_count);

/* ---- LLM response parsing: src/cortex/tk_decision_engine.h:61-190 ---- */
typedef enum {
    TK_ACTION_TYPE_SPEAK, TK_ACTION_TYPE_NAVIGATE_GUIDE, TK_ACTION_TYPE_NAVIGATE_WARN, TK_ACTION_TYPE_DESCRIBE_ENVIRONMENT, TK_ACTION_TYPE_DESCRIBE_OBJECT,
    TK_ACTION_TYPE_READ_TEXT, TK_ACTION_TYPE_SYSTEM_MODE_CHANGE, TK_ACTION_TYPE_SYSTEM_SETTING, TK_ACTION_TYPE_USER_QUERY_RESPONSE, TK_ACTION_TYPE_EMERGENCY_ALERT
} tk_action_type_e;
typedef enum { TK_RESPONSE_PRIORITY_EMERGENCY = 0, TK_RESPONSE_PRIORITY_HIGH = 1, TK_RESPONSE_PRIORITY_NORMAL = 2, TK_RESPONSE_PRIORITY_LOW = 3, TK_RESPONSE_PRIORITY_COUNT } tk_response_priority_e;
/* the parser compares the "priority" string with "critical" and assigns TK_RESPONSE_PRIORITY_CRITICAL (tk_decision_engine.c:1674), an
 * enumerator the header does not declare: it is the header's EMERGENCY level */
#define TK_RESPONSE_PRIORITY_CRITICAL TK_RESPONSE_PRIORITY_EMERGENCY

typedef struct {
    tk_action_type_e type;
    float confidence;
    uint32_t timeout_ms;
    union {
        struct { char* text; tk_response_priority_e priority; float volume_multiplier; } speak;
        struct { float direction_deg; float distance_m; char* instruction; } navigate_guide;
        /* the parser fills `obstacle_id` (tk_decision_engine.c:1768), a member the reference header lacks: added at the end */
        struct { tk_context_priority_e urgency; char* warning_text; float hazard_distance_m; float hazard_direction_deg; uint32_t obstacle_id; } navigate_warn;
        struct { bool include_objects; bool include_hazards; bool include_navigation; float detail_level; } describe_environment;
        struct { uint32_t object_id; char* object_label; float distance_m; } describe_object;
        struct { char* text_content; float reading_speed; } read_text;
        struct { char* setting_name; char* setting_value; } system_setting;
        struct { char* response_text; bool requires_context; } user_query_response;
        struct { char* alert_message; bool repeat_alert; uint32_t repeat_interval_ms; } emergency_alert;
    } params;
} tk_action_params_t;

typedef struct {
    char* response_text;
    tk_response_priority_e priority;
    size_t action_count;
    tk_action_params_t* actions;
} tk_llm_response_t;

/* {"response_text": str, "priority": "normal" | "high" | "critical", "actions": [{"type": "SPEAK" | ..., "confidence": num, "params": {...}}]}
 * -> tk_llm_response_t.  TK_ERROR_INVALID_FORMAT (= TK_ERROR_CONFIG_PARSE_FAILED, see tk_error_handling.h) when the text is
 * not JSON, an action is not an object, lacks a string "type" / an object "params", or names an unknown type; a missing "actions"
 * array is not an error. */
TK_API TK_NODISCARD tk_error_code_t tk_decision_engine_parse_llm_response_text(const char* text, tk_llm_response_t** out_response);
TK_API void tk_decision_engine_free_response(tk_llm_response_t** response);

#ifdef __cplusplus
}
#endif
#endif
