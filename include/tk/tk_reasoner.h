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
 * The rest of the reasoner's public surface (src/cortex/tk_contextual_reasoner.h:198-389) is here too: ambient sound / navigation cue /
 * navigation / motion updates (.c:243-352, 443-519, 1121-1160), relevance decay + pruning (_process_context .c:604-622, 965-1012), the
 * structured summary (_get_context_summary .c:626-677) and _get_motion_state (.c:226-239).  The navigation and sensor-fusion ENGINES that
 * produce their inputs stay out of scope: only the plain structs those entry points read are declared below, field for
 * field (src/navigation/tk_path_planner.h:71-85, tk_free_space_detector.h:50-68, tk_obstacle_avoider.h:46-77, src/sensors/tk_sensors_fusion.h:65-93).
 * The prompt generator the reference keeps on its Rust side — tk_cortex_rust_init_reasoner / tk_cortex_generate_prompt / tk_cortex_rust_set_fact
 * (src/cortex/src/ffi.rs:262,370,427; body src/cortex/src/reasoning.rs:436-480) — is restated in C++ (csrc/cortex/tk_prompt.cpp) and pinned by
 * the reference's own enabled test, tests/tk_cortex_full_test.c:36-70, replayed in tests/test_reasoner_cpu.py and as a C host.
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

/* src/cortex/tk_contextual_reasoner.h:88-112 */
typedef enum { TK_AMBIENT_SOUND_NONE, TK_AMBIENT_SOUND_FIRE_ALARM, TK_AMBIENT_SOUND_CAR_HORN, TK_AMBIENT_SOUND_SIREN, TK_AMBIENT_SOUND_BABY_CRYING, TK_AMBIENT_SOUND_DOORBELL } tk_ambient_sound_type_e;
typedef enum { TK_NAVIGATION_CUE_NONE, TK_NAVIGATION_CUE_STEP_UP, TK_NAVIGATION_CUE_STEP_DOWN, TK_NAVIGATION_CUE_DOORWAY, TK_NAVIGATION_CUE_STAIRS_UP, TK_NAVIGATION_CUE_STAIRS_DOWN } tk_navigation_cue_type_e;

/* inputs of the navigation / motion updates: plain structs of the (out of scope) engines that fill them */
typedef enum { TK_TRAVERSABILITY_UNKNOWN, TK_TRAVERSABILITY_TRAVERSABLE, TK_TRAVERSABILITY_OBSTACLE, TK_TRAVERSABILITY_HAZARD_STEP_UP, TK_TRAVERSABILITY_HAZARD_STEP_DOWN, TK_TRAVERSABILITY_HAZARD_HOLE } tk_traversability_type_e;
typedef struct { uint32_t width; uint32_t height; float resolution_m_per_cell; tk_traversability_type_e* grid; } tk_traversability_map_t;   /* tk_path_planner.h:71-75 */
typedef struct { tk_traversability_type_e type; float distance_m; float direction_deg; } tk_navigation_hazard_t;                          /* tk_path_planner.h:81-85 */
typedef struct { float center_angle_deg; float max_clear_distance_m; bool is_clear; } tk_space_sector_t;                                  /* tk_free_space_detector.h:50-54 */
typedef struct { const tk_space_sector_t* sectors; size_t sector_count; bool is_any_path_clear; float clearest_path_angle_deg; float clearest_path_distance_m; } tk_free_space_analysis_t; /* :60-68 */
typedef enum { TK_OBSTACLE_STATUS_NEW, TK_OBSTACLE_STATUS_TRACKED, TK_OBSTACLE_STATUS_COASTED } tk_obstacle_status_e;
typedef struct { float x; float y; } tk_vector2d_t;
typedef struct { uint32_t id; tk_obstacle_status_e status; tk_vector2d_t position_m; tk_vector2d_t velocity_mps; tk_vector2d_t dimensions_m; uint32_t age_frames; uint32_t unseen_frames; } tk_obstacle_t; /* tk_obstacle_avoider.h:69-77 */
typedef enum { TK_MOTION_STATE_UNKNOWN, TK_MOTION_STATE_STATIONARY, TK_MOTION_STATE_WALKING, TK_MOTION_STATE_RUNNING, TK_MOTION_STATE_FALLING } tk_motion_state_e; /* tk_sensors_fusion.h:65-71 */
typedef struct { float w, x, y, z; } tk_quaternion_t;
typedef struct { uint64_t last_update_timestamp_ns; tk_quaternion_t orientation; tk_motion_state_e motion_state; bool is_speech_detected; } tk_world_state_t; /* tk_sensors_fusion.h:88-93 */

/* src/cortex/tk_contextual_reasoner.h:132-176 */
typedef struct { uint64_t timestamp_ns; bool is_user_input; char* content; float confidence; } tk_conversation_turn_t;
typedef struct {
    size_t visible_object_count;
    const tk_vision_object_t* visible_objects;
    bool has_clear_path;
    float clear_path_direction_deg;
    float clear_path_distance_m;
    size_t hazard_count;
    const tk_navigation_hazard_t* hazards;
    size_t conversation_turn_count;
    const tk_conversation_turn_t* recent_conversation; /* the circular buffer itself, max_conversation_history_turns entries (read-only) */
    char* recent_events_summary;                       /* always NULL ("left for future use", .c:667) */
    bool is_navigation_active;
    bool is_listening_for_commands;
    float system_confidence;
    tk_motion_state_e user_motion_state;
    tk_ambient_sound_type_e detected_sound_type;
    tk_navigation_cue_type_e detected_navigation_cue;
} tk_context_summary_t;

TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_create(tk_contextual_reasoner_t** out_reasoner, const tk_context_config_t* config);
/* remembers the sound and files "<Sound> detected (confidence: <c>%)" as an ENVIRONMENTAL item (fire alarm CRITICAL, siren / horn HIGH, baby
 * MEDIUM, doorbell LOW); NONE only clears the remembered sound (.c:243-297) */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_update_ambient_sound(tk_contextual_reasoner_t* reasoner, tk_ambient_sound_type_e sound_type, float confidence);
/* remembers the cue and files "<Cue> detected at <d>m" as a NAVIGATIONAL item (HIGH; doorway MEDIUM) (.c:301-350) */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_update_navigation_cues(tk_contextual_reasoner_t* reasoner, tk_navigation_cue_type_e cue_type, float distance_m);
/* clear-path snapshot from the free-space analysis (the hazard list is reset and, as in the reference, never refilled), one item for the
 * path and one per obstacle for the first five (HIGH below 1.5 m) (.c:443-519) */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_update_navigation_context(tk_contextual_reasoner_t* reasoner, const tk_traversability_map_t* traversability_map,
                                                                                     const tk_free_space_analysis_t* free_space_analysis, const tk_obstacle_t* obstacles,
                                                                                     size_t obstacle_count);
/* motion state from sensor fusion; a change files a USER_STATE item ("User started walking", "Fall detected!", ...) (.c:1121-1160) */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_update_motion_context(tk_contextual_reasoner_t* reasoner, const tk_world_state_t* world_state);
/* relevance *= exp(-memory_decay_rate * age_s) for every item, then items below context_relevance_threshold are dropped (.c:604-622, 965-1012) */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_process_context(tk_contextual_reasoner_t* reasoner, uint64_t current_time_ns);
/* pointers stay owned by the reasoner, valid until the next update (.c:626-677) */
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_get_context_summary(tk_contextual_reasoner_t* reasoner, tk_context_summary_t* out_summary);
TK_API TK_NODISCARD tk_error_code_t tk_contextual_reasoner_get_motion_state(tk_contextual_reasoner_t* reasoner, tk_motion_state_e* out_state);
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

/* ---- prompt generator (the reference's Rust side: src/cortex/src/ffi.rs:262,370,427; src/cortex/src/reasoning.rs:436-480) ----
 * One process-wide generator bound to a reasoner, as the reference's `static REASONER`.  tk_cortex_generate_prompt reads the reasoner's
 * summary and writes, in this order: "URGENTE: ALARME DE INCÊNDIO DETECTADO. " (fire alarm) / "URGENTE: QUEDA DO USUÁRIO DETECTADA. "
 * (falling), the navigation cue sentence ("Há um degrau para baixo à frente. " ...), the motion sentence ("O usuário está andando. " /
 * "... correndo. " / "... parado. "), "O nome do usuário é <name>. " when the fact user_name is set, "O usuário perguntou: '<query>'. " and
 * "Com base em tudo isso, qual a ação mais segura e útil?"; copied with strncpy semantics into prompt_buffer (always NUL terminated).
 * Returns false for a NULL / empty buffer; an unbound generator writes the reference's fallback prompt and returns true. */
TK_API void tk_cortex_rust_init_reasoner(tk_contextual_reasoner_t* reasoner_ptr);
TK_API bool tk_cortex_generate_prompt(char* prompt_buffer, size_t buffer_size, const char* user_query);
TK_API void tk_cortex_rust_set_fact(const char* key, const char* value);

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
