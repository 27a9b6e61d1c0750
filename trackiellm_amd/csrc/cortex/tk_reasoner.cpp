/*
 * tk_reasoner.cpp — prompt assembly and LLM-response parsing, the host-side steps either side of the LLM runner
 * (include/tk/tk_reasoner.h lists the reference lines each entry restates).  Pure host code.
 */
#include "tk/tk_reasoner.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

/* ------------------------------------------------------------------ contextual reasoner ---- */

namespace {
struct ContextItem { uint64_t ts; tk_context_type_e type; tk_context_priority_e prio; float relevance; std::string text; size_t data_size; };
struct Turn { uint64_t ts; bool user; std::string content; float confidence; bool used = false; };
struct VisibleObject { std::string label, attributes; tk_vision_object_t c; };

uint64_t now_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ULL + (uint64_t)ts.tv_nsec;
}
}  // namespace

struct tk_contextual_reasoner_s {
    tk_context_config_t config;
    std::mutex mu;
    std::vector<ContextItem> items;          /* circular once max_context_history_items is reached */
    size_t item_next = 0;
    std::vector<Turn> turns;                 /* circular buffer of max_conversation_history_turns */
    size_t turn_next = 0, turn_count = 0;
    std::vector<VisibleObject> visible;      /* capacity 64 (tk_contextual_reasoner.c:181) */
    /* navigation_state (.c:87-97) */
    bool has_clear_path = false;
    float path_dir = 0.0f, path_dist = 0.0f;
    size_t hazards = 0;                      /* reset by every navigation update and never refilled (.c:463, nothing writes the list) */
    tk_motion_state_e motion = TK_MOTION_STATE_UNKNOWN;
    tk_navigation_cue_type_e last_cue = TK_NAVIGATION_CUE_NONE;
    /* audio_state (.c:79-82) */
    tk_ambient_sound_type_e last_sound = TK_AMBIENT_SOUND_NONE;
    uint64_t last_sound_ns = 0;
    /* system_state (.c:102-107): defaults of _create (.c:209-211) */
    uint64_t last_process_ns = 0;
    bool listening = false;
    float system_confidence = 0.8f;
    /* C views handed out by get_context_summary: rebuilt there, owned here, valid until the next update */
    std::vector<tk_vision_object_t> visible_c;
    std::vector<tk_conversation_turn_t> turns_c;
};

static void add_item(tk_contextual_reasoner_s* r, tk_context_type_e type, tk_context_priority_e prio, float relevance, const char* text, size_t data_size) {
    ContextItem it{now_ns(), type, prio, relevance, text ? text : "", data_size};
    const size_t cap = r->config.max_context_history_items ? r->config.max_context_history_items : 1;
    if (r->items.size() < cap) r->items.push_back(std::move(it));
    else { r->items[r->item_next % cap] = std::move(it); }
    r->item_next = (r->item_next + 1) % cap;
}

/* tk_contextual_reasoner_add_context_item (.c:561-600): relevance starts at 1 */
static void add_item_locked(tk_contextual_reasoner_s* r, tk_context_type_e type, tk_context_priority_e prio, const char* text) {
    std::lock_guard<std::mutex> lk(r->mu);
    add_item(r, type, prio, 1.0f, text, 0);
}

void tk_prompt_forget_reasoner(tk_contextual_reasoner_t* r); /* tk_prompt.cpp */

extern "C" {

tk_error_code_t tk_contextual_reasoner_create(tk_contextual_reasoner_t** out, const tk_context_config_t* cfg) {
    if (!out || !cfg) return TK_ERROR_INVALID_ARGUMENT;
    if (cfg->max_context_history_items == 0 || cfg->max_conversation_history_turns == 0) return TK_ERROR_INVALID_ARGUMENT; /* the reference's calloc(0) containers cannot hold anything either */
    tk_contextual_reasoner_s* r = new (std::nothrow) tk_contextual_reasoner_s();
    if (!r) return TK_ERROR_OUT_OF_MEMORY;
    r->config = *cfg;
    r->turns.resize(cfg->max_conversation_history_turns);
    *out = r;
    return TK_SUCCESS;
}

void tk_contextual_reasoner_destroy(tk_contextual_reasoner_t** reasoner) {
    if (!reasoner || !*reasoner) return;
    tk_prompt_forget_reasoner(*reasoner);
    delete *reasoner;
    *reasoner = nullptr;
}

tk_error_code_t tk_contextual_reasoner_get_motion_state(tk_contextual_reasoner_t* r, tk_motion_state_e* out_state) { /* .c:226-239 */
    if (!r || !out_state) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(r->mu);
    *out_state = r->motion;
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_update_ambient_sound(tk_contextual_reasoner_t* r, tk_ambient_sound_type_e sound_type, float confidence) { /* .c:243-297 */
    if (!r) return TK_ERROR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->last_sound = sound_type;
        r->last_sound_ns = now_ns();
    }
    if (sound_type != TK_AMBIENT_SOUND_NONE) {
        const char* what = "Unknown sound";
        tk_context_priority_e prio = TK_CONTEXT_PRIORITY_MEDIUM;
        switch (sound_type) {
            case TK_AMBIENT_SOUND_FIRE_ALARM: what = "Fire alarm detected"; prio = TK_CONTEXT_PRIORITY_CRITICAL; break;
            case TK_AMBIENT_SOUND_SIREN: what = "Siren detected"; prio = TK_CONTEXT_PRIORITY_HIGH; break;
            case TK_AMBIENT_SOUND_CAR_HORN: what = "Car horn detected"; prio = TK_CONTEXT_PRIORITY_HIGH; break;
            case TK_AMBIENT_SOUND_BABY_CRYING: what = "Baby crying detected"; prio = TK_CONTEXT_PRIORITY_MEDIUM; break;
            case TK_AMBIENT_SOUND_DOORBELL: what = "Doorbell detected"; prio = TK_CONTEXT_PRIORITY_LOW; break;
            default: break;
        }
        char desc[128];
        snprintf(desc, sizeof desc, "%s (confidence: %.0f%%)", what, confidence * 100.0f);
        add_item_locked(r, TK_CONTEXT_TYPE_ENVIRONMENTAL, prio, desc);
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_update_navigation_cues(tk_contextual_reasoner_t* r, tk_navigation_cue_type_e cue_type, float distance_m) { /* .c:301-350 */
    if (!r) return TK_ERROR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->last_cue = cue_type;
    }
    if (cue_type != TK_NAVIGATION_CUE_NONE) {
        const char* what = "Unknown navigation cue";
        tk_context_priority_e prio = TK_CONTEXT_PRIORITY_HIGH;
        switch (cue_type) {
            case TK_NAVIGATION_CUE_STEP_UP: what = "Step up detected"; break;
            case TK_NAVIGATION_CUE_STEP_DOWN: what = "Step down detected"; break;
            case TK_NAVIGATION_CUE_DOORWAY: what = "Doorway detected"; prio = TK_CONTEXT_PRIORITY_MEDIUM; break;
            case TK_NAVIGATION_CUE_STAIRS_UP: what = "Stairs up detected"; break;
            case TK_NAVIGATION_CUE_STAIRS_DOWN: what = "Stairs down detected"; break;
            default: break;
        }
        char desc[128];
        snprintf(desc, sizeof desc, "%s at %.1fm", what, distance_m);
        add_item_locked(r, TK_CONTEXT_TYPE_NAVIGATIONAL, prio, desc);
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_update_vision_context(tk_contextual_reasoner_t* r, const tk_vision_result_t* vr) {
    if (!r || !vr) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(r->mu);
    size_t n = vr->object_count;
    if (n > 64) n = 64;
    r->visible.clear();
    for (size_t i = 0; i < n; ++i) { /* the reference memcpy's the structs, label pointers included (.c:394-397); here the strings are copied so the result may be destroyed */
        const tk_vision_object_t& o = vr->objects[i];
        VisibleObject v;
        v.label = o.label ? o.label : "object";
        v.attributes = o.attributes ? o.attributes : "";
        v.c = o;
        v.c.recognized_text = nullptr;
        r->visible.push_back(std::move(v));
    }
    for (size_t i = 0; i < n; ++i) {
        const tk_vision_object_t& o = vr->objects[i];
        if (o.confidence < 0.7f) continue; /* ignore low confidence (tk_contextual_reasoner.c:411) */
        char desc[256];
        snprintf(desc, sizeof desc, "Detected %s at %.1fm (confidence %.0f%%)", o.label ? o.label : "object", o.distance_meters, o.confidence * 100.0f);
        add_item(r, TK_CONTEXT_TYPE_ENVIRONMENTAL, o.distance_meters < 2.0f ? TK_CONTEXT_PRIORITY_HIGH : TK_CONTEXT_PRIORITY_MEDIUM, o.confidence, desc, 0);
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_update_navigation_context(tk_contextual_reasoner_t* r, const tk_traversability_map_t* map, const tk_free_space_analysis_t* fs,
                                                                 const tk_obstacle_t* obstacles, size_t obstacle_count) { /* .c:443-519 */
    if (!r || !map || !fs) return TK_ERROR_INVALID_ARGUMENT;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        r->has_clear_path = fs->is_any_path_clear;
        r->path_dir = fs->clearest_path_angle_deg;
        r->path_dist = fs->clearest_path_distance_m;
        r->hazards = 0; /* "will be filled later if needed" — nothing does */
    }
    if (fs->is_any_path_clear) {
        char buf[256];
        snprintf(buf, sizeof buf, "Clear path at %.0f\xC2\xB0, distance %.1fm", fs->clearest_path_angle_deg, fs->clearest_path_distance_m);
        add_item_locked(r, TK_CONTEXT_TYPE_NAVIGATIONAL, TK_CONTEXT_PRIORITY_HIGH, buf);
    } else {
        add_item_locked(r, TK_CONTEXT_TYPE_NAVIGATIONAL, TK_CONTEXT_PRIORITY_CRITICAL, "No clear navigation path detected");
    }
    const size_t n = obstacles ? (obstacle_count < 5 ? obstacle_count : 5) : 0; /* the reference dereferences `obstacles` unchecked; a NULL list is read as empty here */
    for (size_t i = 0; i < n; ++i) {
        const tk_obstacle_t& o = obstacles[i];
        char buf[256];
        snprintf(buf, sizeof buf, "Obstacle at (%.1f, %.1f)m size %.1fx%.1fm", o.position_m.x, o.position_m.y, o.dimensions_m.x, o.dimensions_m.y);
        const float dist = sqrtf(o.position_m.x * o.position_m.x + o.position_m.y * o.position_m.y);
        add_item_locked(r, TK_CONTEXT_TYPE_NAVIGATIONAL, dist < 1.5f ? TK_CONTEXT_PRIORITY_HIGH : TK_CONTEXT_PRIORITY_MEDIUM, buf);
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_update_motion_context(tk_contextual_reasoner_t* r, const tk_world_state_t* ws) { /* .c:1121-1160 */
    if (!r || !ws) return TK_ERROR_INVALID_ARGUMENT;
    tk_motion_state_e old;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        old = r->motion;
        r->motion = ws->motion_state;
    }
    if (old != ws->motion_state) {
        const char* what = "UNKNOWN";
        switch (ws->motion_state) {
            case TK_MOTION_STATE_STATIONARY: what = "User is now stationary"; break;
            case TK_MOTION_STATE_WALKING: what = "User started walking"; break;
            case TK_MOTION_STATE_RUNNING: what = "User started running"; break;
            case TK_MOTION_STATE_FALLING: what = "Fall detected!"; break;
            default: break;
        }
        add_item_locked(r, TK_CONTEXT_TYPE_USER_STATE, TK_CONTEXT_PRIORITY_MEDIUM, what);
    }
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_add_conversation_turn(tk_contextual_reasoner_t* r, bool is_user_input, const char* content, float confidence) {
    if (!r || !content) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(r->mu);
    Turn& t = r->turns[r->turn_next];
    t.ts = now_ns(); t.user = is_user_input; t.content = content; t.confidence = confidence; t.used = true;
    r->turn_next = (r->turn_next + 1) % r->turns.size();
    if (r->turn_count < r->turns.size()) r->turn_count++;
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_add_context_item(tk_contextual_reasoner_t* r, tk_context_type_e type, tk_context_priority_e priority,
                                                        const char* description, const void* data, size_t data_size) {
    if (!r || !description) return TK_ERROR_INVALID_ARGUMENT;
    (void)data;
    std::lock_guard<std::mutex> lk(r->mu);
    add_item(r, type, priority, 1.0f, description, data_size);
    return TK_SUCCESS;
}

/* .c:604-622: decay every item's relevance by exp(-rate * age) (.c:980-989; applied to the CURRENT score, so repeated calls compound, as in
 * the reference), drop what falls below the threshold keeping the order (.c:991-1012) */
tk_error_code_t tk_contextual_reasoner_process_context(tk_contextual_reasoner_t* r, uint64_t current_time_ns) {
    if (!r) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(r->mu);
    for (ContextItem& it : r->items) {
        const double age_s = (double)(current_time_ns - it.ts) / 1e9; /* unsigned difference, as the reference computes it */
        it.relevance = it.relevance * expf(-r->config.memory_decay_rate * (float)age_s);
    }
    size_t w = 0;
    for (size_t i = 0; i < r->items.size(); ++i)
        if (r->items[i].relevance >= r->config.context_relevance_threshold) { if (w != i) r->items[w] = std::move(r->items[i]); ++w; }
    r->items.resize(w);
    const size_t cap = r->config.max_context_history_items ? r->config.max_context_history_items : 1;
    r->item_next = w % cap;
    r->last_process_ns = current_time_ns;
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_get_context_summary(tk_contextual_reasoner_t* r, tk_context_summary_t* out) { /* .c:626-677 */
    if (!r || !out) return TK_ERROR_INVALID_ARGUMENT;
    memset(out, 0, sizeof *out);
    std::lock_guard<std::mutex> lk(r->mu);
    r->visible_c.clear();
    for (VisibleObject& v : r->visible) {
        tk_vision_object_t o = v.c;
        o.label = v.label.c_str();
        o.attributes = v.attributes.empty() ? nullptr : &v.attributes[0];
        r->visible_c.push_back(o);
    }
    out->visible_object_count = r->visible_c.size();
    out->visible_objects = r->visible_c.empty() ? nullptr : r->visible_c.data();
    out->has_clear_path = r->has_clear_path;
    out->clear_path_direction_deg = r->path_dir;
    out->clear_path_distance_m = r->path_dist;
    out->hazard_count = r->hazards;
    out->hazards = nullptr;
    r->turns_c.assign(r->turns.size(), tk_conversation_turn_t{0, false, nullptr, 0.0f});
    for (size_t i = 0; i < r->turns.size(); ++i)
        if (r->turns[i].used) r->turns_c[i] = tk_conversation_turn_t{r->turns[i].ts, r->turns[i].user, &r->turns[i].content[0], r->turns[i].confidence};
    out->conversation_turn_count = r->turn_count;
    out->recent_conversation = r->turns_c.data();
    out->recent_events_summary = nullptr;
    out->is_navigation_active = r->has_clear_path;
    out->is_listening_for_commands = r->listening;
    out->system_confidence = r->system_confidence;
    out->user_motion_state = r->motion;
    out->detected_sound_type = r->last_sound;
    out->detected_navigation_cue = r->last_cue;
    return TK_SUCCESS;
}

/* the three generators write into fixed stack buffers in the reference (256 / 256 / 512 bytes) and stop at the first entry that does not
 * fit; restated with the same limits so long labels / turns truncate at the same entry */
static std::string environment(const tk_contextual_reasoner_s* r) {
    if (r->visible.empty()) return "No visible objects";
    const size_t limit = r->visible.size() < 3 ? r->visible.size() : 3; /* "we limit to 3 objects for brevity" (:1025) */
    char buf[256];
    size_t pos = 0;
    for (size_t i = 0; i < limit; ++i) {
        const VisibleObject& o = r->visible[i];
        int n = snprintf(buf + pos, sizeof buf - pos, "%s (%.1fm, %.0f%% confidence); ", o.label.c_str(), o.c.distance_meters, o.c.confidence * 100.0f);
        if (n < 0 || (size_t)n >= sizeof buf - pos) { buf[pos] = 0; break; }
        pos += (size_t)n;
    }
    if (pos > 2) buf[pos - 2] = '\0'; /* strip trailing "; " */
    else buf[pos] = '\0';
    return buf;
}

static std::string navigation(const tk_contextual_reasoner_s* r) {
    char buf[256];
    if (r->has_clear_path) snprintf(buf, sizeof buf, "Clear path ahead at %.0f\xC2\xB0, %.1fm away. %zu hazards detected.", r->path_dir, r->path_dist, r->hazards);
    else snprintf(buf, sizeof buf, "No clear path. %zu hazards detected.", r->hazards);
    return buf;
}

static std::string conversation(const tk_contextual_reasoner_s* r, size_t max_turns) {
    if (r->turn_count == 0) return "No recent conversation";
    const size_t to_show = max_turns < r->turn_count ? max_turns : r->turn_count;
    const size_t cap = r->turns.size();
    char buf[512];
    size_t pos = 0;
    for (size_t i = 0; i < to_show; ++i) { /* newest first (:1082-1084) */
        const Turn& t = r->turns[(r->turn_next + cap - 1 - i) % cap];
        int n = snprintf(buf + pos, sizeof buf - pos, "%s: \"%s\"; ", t.user ? "User" : "System", t.content.c_str());
        if (n < 0 || (size_t)n >= sizeof buf - pos) { buf[pos] = 0; break; }
        pos += (size_t)n;
    }
    if (pos > 2) buf[pos - 2] = '\0';
    else buf[pos] = '\0';
    return buf;
}

tk_error_code_t tk_contextual_reasoner_generate_context_string(tk_contextual_reasoner_t* r, char** out, size_t max_token_budget) {
    if (!r || !out) return TK_ERROR_INVALID_ARGUMENT;
    *out = NULL;
    const size_t max_chars = max_token_budget * 4; /* "4 is a safe average" (:689-690) */
    char* buf = (char*)calloc(1, max_chars + 1);
    if (!buf) return TK_ERROR_OUT_OF_MEMORY;
    size_t used = 0;
    std::string parts[3];
    {
        std::lock_guard<std::mutex> lk(r->mu);
        parts[0] = environment(r);
        parts[1] = navigation(r);
        parts[2] = conversation(r, 3);
    }
    for (const std::string& p : parts) { /* a part that does not fit is skipped whole, later (shorter) parts may still fit (:699-733) */
        if (used + p.size() + 1 <= max_chars) {
            memcpy(buf + used, p.data(), p.size());
            used += p.size();
            buf[used++] = ' ';
        }
    }
    if (used && buf[used - 1] == ' ') buf[--used] = '\0';
    else buf[used] = '\0';
    *out = buf;
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_free_context_string(char* ptr) {
    if (!ptr) return TK_ERROR_INVALID_ARGUMENT;
    free(ptr);
    return TK_SUCCESS;
}

/* .c:756-793: the generic items, the conversation and the bookkeeping go; the environment / navigation / audio SNAPSHOTS stay (the
 * reference does not touch them either: the next frame or update replaces them) */
tk_error_code_t tk_contextual_reasoner_clear_context(tk_contextual_reasoner_t* r) {
    if (!r) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(r->mu);
    r->items.clear(); r->item_next = 0;
    for (Turn& t : r->turns) t = Turn{};
    r->turn_next = r->turn_count = 0;
    r->last_process_ns = 0;
    return TK_SUCCESS;
}

tk_error_code_t tk_contextual_reasoner_get_memory_stats(tk_contextual_reasoner_t* r, size_t* total_items, size_t* total_bytes, size_t* turns) {
    if (!r) return TK_ERROR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(r->mu);
    size_t bytes = 0;
    for (const ContextItem& it : r->items) bytes += sizeof(ContextItem) + it.text.size() + 1 + it.data_size;
    if (total_items) *total_items = r->items.size();
    if (total_bytes) *total_bytes = bytes;
    if (turns) *turns = r->turn_count;
    return TK_SUCCESS;
}

}  /* extern "C" */

/* ------------------------------------------------------------------ JSON (what the parser needs of cJSON) ---- */

namespace {
struct JVal {
    enum Kind { JNULL, JBOOL, JNUM, JSTR, JARR, JOBJ } kind = JNULL;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* key) const { /* cJSON_GetObjectItemCaseSensitive: first member of that name */
        if (kind != JOBJ) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    int depth = 0;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    static void utf8(std::string& s, unsigned cp) {
        if (cp < 0x80) s += (char)cp;
        else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 63)); }
        else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 63)); s += (char)(0x80 | (cp & 63)); }
        else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 63)); s += (char)(0x80 | ((cp >> 6) & 63)); s += (char)(0x80 | (cp & 63)); }
    }
    bool hex4(unsigned* v) {
        if (end - p < 4) return false;
        unsigned x = 0;
        for (int i = 0; i < 4; ++i) {
            char c = p[i];
            x <<= 4;
            if (c >= '0' && c <= '9') x |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') x |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') x |= (unsigned)(c - 'A' + 10);
            else return false;
        }
        p += 4;
        *v = x;
        return true;
    }
    bool string(std::string* out) {
        if (p >= end || *p != '"') return false;
        ++p;
        out->clear();
        while (p < end && *p != '"') {
            if ((unsigned char)*p < 0x20) return false;
            if (*p != '\\') { *out += *p++; continue; }
            if (++p >= end) return false;
            switch (*p++) {
                case '"': *out += '"'; break;
                case '\\': *out += '\\'; break;
                case '/': *out += '/'; break;
                case 'b': *out += '\b'; break;
                case 'f': *out += '\f'; break;
                case 'n': *out += '\n'; break;
                case 'r': *out += '\r'; break;
                case 't': *out += '\t'; break;
                case 'u': {
                    unsigned cp;
                    if (!hex4(&cp)) return false;
                    if (cp >= 0xD800 && cp <= 0xDBFF) { /* surrogate pair */
                        unsigned lo;
                        if (end - p < 6 || p[0] != '\\' || p[1] != 'u') return false;
                        p += 2;
                        if (!hex4(&lo) || lo < 0xDC00 || lo > 0xDFFF) return false;
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) return false;
                    utf8(*out, cp);
                    break;
                }
                default: return false;
            }
        }
        if (p >= end) return false;
        ++p;
        return true;
    }
    bool value(JVal* v) {
        if (++depth > 64) return false;
        ws();
        if (p >= end) return false;
        bool ok = false;
        if (*p == '{') {
            ++p;
            v->kind = JVal::JOBJ;
            ws();
            if (p < end && *p == '}') { ++p; ok = true; }
            else
                for (;;) {
                    ws();
                    std::string key;
                    if (!string(&key)) break;
                    ws();
                    if (p >= end || *p != ':') break;
                    ++p;
                    JVal child;
                    if (!value(&child)) break;
                    v->obj.emplace_back(std::move(key), std::move(child));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == '}') { ++p; ok = true; }
                    break;
                }
        } else if (*p == '[') {
            ++p;
            v->kind = JVal::JARR;
            ws();
            if (p < end && *p == ']') { ++p; ok = true; }
            else
                for (;;) {
                    JVal child;
                    if (!value(&child)) break;
                    v->arr.push_back(std::move(child));
                    ws();
                    if (p < end && *p == ',') { ++p; continue; }
                    if (p < end && *p == ']') { ++p; ok = true; }
                    break;
                }
        } else if (*p == '"') {
            v->kind = JVal::JSTR;
            ok = string(&v->str);
        } else if (end - p >= 4 && !strncmp(p, "true", 4)) { v->kind = JVal::JBOOL; v->b = true; p += 4; ok = true; }
        else if (end - p >= 5 && !strncmp(p, "false", 5)) { v->kind = JVal::JBOOL; v->b = false; p += 5; ok = true; }
        else if (end - p >= 4 && !strncmp(p, "null", 4)) { v->kind = JVal::JNULL; p += 4; ok = true; }
        else if (*p == '-' || (*p >= '0' && *p <= '9')) {
            char tmp[64];
            size_t n = 0;
            while (p + n < end && n < sizeof tmp - 1 && (strchr("+-0123456789.eE", p[n]) != nullptr) && p[n] != 0) ++n;
            memcpy(tmp, p, n);
            tmp[n] = 0;
            char* e = nullptr;
            v->num = strtod(tmp, &e);
            if (e != tmp) { v->kind = JVal::JNUM; p += (e - tmp); ok = true; }
        }
        --depth;
        return ok;
    }
};

/* cJSON's valueint: the double saturated to int */
int valueint(double d) {
    if (d >= 2147483647.0) return 2147483647;
    if (d <= -2147483648.0) return (-2147483647 - 1);
    return (int)d;
}

char* dup_field(const JVal* obj, const char* key) { /* PARSE_STRING_FIELD: the string or "" */
    const JVal* it = obj->get(key);
    return strdup(it && it->kind == JVal::JSTR ? it->str.c_str() : "");
}
}  // namespace

extern "C" {

void tk_decision_engine_free_response(tk_llm_response_t** response) {
    if (!response || !*response) return;
    tk_llm_response_t* r = *response;
    free(r->response_text);
    for (size_t i = 0; i < r->action_count && r->actions; ++i) {
        tk_action_params_t* a = &r->actions[i];
        switch (a->type) {
            case TK_ACTION_TYPE_SPEAK: free(a->params.speak.text); break;
            case TK_ACTION_TYPE_NAVIGATE_GUIDE: free(a->params.navigate_guide.instruction); break;
            case TK_ACTION_TYPE_NAVIGATE_WARN: free(a->params.navigate_warn.warning_text); break;
            case TK_ACTION_TYPE_DESCRIBE_OBJECT: free(a->params.describe_object.object_label); break;
            case TK_ACTION_TYPE_READ_TEXT: free(a->params.read_text.text_content); break;
            case TK_ACTION_TYPE_SYSTEM_SETTING: free(a->params.system_setting.setting_name); free(a->params.system_setting.setting_value); break;
            case TK_ACTION_TYPE_USER_QUERY_RESPONSE: free(a->params.user_query_response.response_text); break;
            case TK_ACTION_TYPE_EMERGENCY_ALERT: free(a->params.emergency_alert.alert_message); break;
            default: break;
        }
    }
    free(r->actions);
    free(r);
    *response = NULL;
}

tk_error_code_t tk_decision_engine_parse_llm_response_text(const char* text, tk_llm_response_t** out_response) {
    if (!text || !out_response) return TK_ERROR_INVALID_ARGUMENT;
    *out_response = NULL;
    JVal root;
    JParser ps{text, text + strlen(text)};
    if (!ps.value(&root)) { tk_error_set_detail("Failed to parse JSON response from LLM"); return TK_ERROR_INVALID_FORMAT; }
    /* cJSON_Parse does not require the end of the input after the value: trailing text is ignored */
    tk_llm_response_t* resp = (tk_llm_response_t*)calloc(1, sizeof(tk_llm_response_t));
    if (!resp) return TK_ERROR_OUT_OF_MEMORY;
    tk_error_code_t rc = TK_SUCCESS;
    auto bail = [&](tk_error_code_t e) { tk_decision_engine_free_response(&resp); return e; };

    const JVal* rt = root.get("response_text");
    resp->response_text = strdup(rt && rt->kind == JVal::JSTR ? rt->str.c_str() : "");
    if (!resp->response_text) return bail(TK_ERROR_OUT_OF_MEMORY);
    const JVal* pr = root.get("priority");
    resp->priority = TK_RESPONSE_PRIORITY_NORMAL;
    if (pr && pr->kind == JVal::JSTR) {
        if (pr->str == "high") resp->priority = TK_RESPONSE_PRIORITY_HIGH;
        else if (pr->str == "critical") resp->priority = TK_RESPONSE_PRIORITY_CRITICAL;
    }
    const JVal* acts = root.get("actions");
    if (!acts || acts->kind != JVal::JARR || acts->arr.empty()) { /* "a response can have no actions" */
        *out_response = resp;
        return TK_SUCCESS;
    }
    resp->actions = (tk_action_params_t*)calloc(acts->arr.size(), sizeof(tk_action_params_t));
    if (!resp->actions) return bail(TK_ERROR_OUT_OF_MEMORY);
    resp->action_count = acts->arr.size();
    static const struct { const char* name; tk_action_type_e type; } kTypes[] = {
        {"SPEAK", TK_ACTION_TYPE_SPEAK}, {"NAVIGATE_GUIDE", TK_ACTION_TYPE_NAVIGATE_GUIDE}, {"NAVIGATE_WARN", TK_ACTION_TYPE_NAVIGATE_WARN},
        {"DESCRIBE_OBJECT", TK_ACTION_TYPE_DESCRIBE_OBJECT}, {"READ_TEXT", TK_ACTION_TYPE_READ_TEXT}, {"SYSTEM_MODE_CHANGE", TK_ACTION_TYPE_SYSTEM_MODE_CHANGE},
        {"SYSTEM_SETTING", TK_ACTION_TYPE_SYSTEM_SETTING}, {"USER_QUERY_RESPONSE", TK_ACTION_TYPE_USER_QUERY_RESPONSE},
        {"EMERGENCY_ALERT", TK_ACTION_TYPE_EMERGENCY_ALERT}}; /* DESCRIBE_ENVIRONMENT is not in the reference's parser either (:1718-1731) */
    for (size_t i = 0; i < acts->arr.size() && rc == TK_SUCCESS; ++i) {
        const JVal& aj = acts->arr[i];
        tk_action_params_t* a = &resp->actions[i];
        if (aj.kind != JVal::JOBJ) { rc = TK_ERROR_INVALID_FORMAT; break; }
        const JVal* tj = aj.get("type");
        if (!tj || tj->kind != JVal::JSTR) { rc = TK_ERROR_INVALID_FORMAT; break; }
        bool known = false;
        for (const auto& kt : kTypes)
            if (tj->str == kt.name) { a->type = kt.type; known = true; break; }
        if (!known) { tk_error_set_detail("Unknown action type '%s'", tj->str.c_str()); rc = TK_ERROR_INVALID_FORMAT; break; }
        const JVal* cj = aj.get("confidence");
        a->confidence = cj && cj->kind == JVal::JNUM ? (float)cj->num : 0.0f;
        const JVal* pj = aj.get("params");
        if (!pj || pj->kind != JVal::JOBJ) { rc = TK_ERROR_INVALID_FORMAT; break; }
        auto num = [&](const char* key, double dflt) { const JVal* v = pj->get(key); return v && v->kind == JVal::JNUM ? v->num : dflt; };
        auto isnum = [&](const char* key) { const JVal* v = pj->get(key); return v && v->kind == JVal::JNUM; };
        bool oom = false;
        switch (a->type) {
            case TK_ACTION_TYPE_SPEAK: oom = !(a->params.speak.text = dup_field(pj, "text")); break;
            case TK_ACTION_TYPE_NAVIGATE_GUIDE:
                oom = !(a->params.navigate_guide.instruction = dup_field(pj, "instruction"));
                a->params.navigate_guide.direction_deg = (float)num("direction_deg", 0.0);
                break;
            case TK_ACTION_TYPE_NAVIGATE_WARN:
                oom = !(a->params.navigate_warn.warning_text = dup_field(pj, "warning_text"));
                a->params.navigate_warn.obstacle_id = isnum("obstacle_id") ? (uint32_t)valueint(num("obstacle_id", 0.0)) : 0;
                break;
            case TK_ACTION_TYPE_DESCRIBE_OBJECT: {
                a->params.describe_object.object_id = isnum("object_id") ? (uint32_t)valueint(num("object_id", 0.0)) : 0;
                const JVal* lj = pj->get("object_label");
                if (lj && lj->kind == JVal::JSTR) oom = !(a->params.describe_object.object_label = strdup(lj->str.c_str())); /* stays NULL when absent (:1777) */
                break;
            }
            case TK_ACTION_TYPE_READ_TEXT: oom = !(a->params.read_text.text_content = dup_field(pj, "text_content")); break;
            case TK_ACTION_TYPE_SYSTEM_SETTING:
                oom = !(a->params.system_setting.setting_name = dup_field(pj, "setting_name")) || !(a->params.system_setting.setting_value = dup_field(pj, "setting_value"));
                break;
            case TK_ACTION_TYPE_USER_QUERY_RESPONSE: oom = !(a->params.user_query_response.response_text = dup_field(pj, "response_text")); break;
            case TK_ACTION_TYPE_EMERGENCY_ALERT: {
                oom = !(a->params.emergency_alert.alert_message = dup_field(pj, "alert_message"));
                const JVal* rj = pj->get("repeat_alert");
                a->params.emergency_alert.repeat_alert = rj && rj->kind == JVal::JBOOL ? rj->b : false;
                a->params.emergency_alert.repeat_interval_ms = isnum("repeat_interval_ms") ? (uint32_t)valueint(num("repeat_interval_ms", 0.0)) : 0;
                break;
            }
            default: break; /* SYSTEM_MODE_CHANGE carries no parsed parameters (:1748-1798) */
        }
        if (oom) rc = TK_ERROR_OUT_OF_MEMORY;
    }
    if (rc != TK_SUCCESS) return bail(rc);
    *out_response = resp;
    return TK_SUCCESS;
}

}  /* extern "C" */
