/*
 * tk_prompt.cpp — the prompt generator the reference keeps on its Rust side, restated in C++ (no Rust toolchain in this image; the callers
 * are C: tests/tk_cortex_full_test.c:12-13,41,55).
 *   tk_cortex_rust_init_reasoner   src/cortex/src/ffi.rs:262-269   binds the process-wide generator to a C reasoner (`static REASONER`)
 *   tk_cortex_generate_prompt      src/cortex/src/ffi.rs:370-419   -> ContextualReasoner::generate_prompt_for_llm, src/cortex/src/reasoning.rs:436-493
 *   tk_cortex_rust_set_fact        src/cortex/src/ffi.rs:427-449   long-term-memory facts (src/cortex/src/memory_manager.rs:293-300); only
 *                                                                  "user_name" is read by the prompt
 * Not restated: the world-model rules engine (tk_cortex_reasoner_run_rules, tk_cortex_rust_process_event) — the Rust workers' layer, out
 * of scope.  Pure host code.
 */
#include <string.h>

#include <map>
#include <mutex>
#include <string>

#include "tk/tk_reasoner.h"

namespace {
std::mutex g_mu;
tk_contextual_reasoner_t* g_reasoner = nullptr;
std::map<std::string, std::string> g_facts;
}  // namespace

/* called by tk_contextual_reasoner_destroy: the reference requires the bound reasoner to outlive the program; here a destroyed one is
 * simply unbound (the generator then writes its fallback prompt) */
void tk_prompt_forget_reasoner(tk_contextual_reasoner_t* r) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_reasoner == r) g_reasoner = nullptr;
}

extern "C" {

void tk_cortex_rust_init_reasoner(tk_contextual_reasoner_t* reasoner_ptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_reasoner = reasoner_ptr; /* ContextualReasoner::new(ptr): a fresh generator — its memory manager starts empty */
    g_facts.clear();
}

void tk_cortex_rust_set_fact(const char* key, const char* value) {
    if (!key || !value) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_facts[key] = value;
}

bool tk_cortex_generate_prompt(char* prompt_buffer, size_t buffer_size, const char* user_query) {
    if (!prompt_buffer || buffer_size == 0) return false;
    std::string prompt;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        tk_context_summary_t s;
        if (!g_reasoner || tk_contextual_reasoner_get_context_summary(g_reasoner, &s) != TK_SUCCESS) {
            prompt = "An error occurred. Please describe the general situation."; /* ffi.rs:399-403 */
        } else {
            /* critical alerts first (reasoning.rs:458-464) */
            if (s.detected_sound_type == TK_AMBIENT_SOUND_FIRE_ALARM) prompt += "URGENTE: ALARME DE INC\xC3\x8ANDIO DETECTADO. ";
            if (s.user_motion_state == TK_MOTION_STATE_FALLING) prompt += "URGENTE: QUEDA DO USU\xC3\x81RIO DETECTADA. ";
            /* navigation cue (:467-474) */
            switch (s.detected_navigation_cue) {
                case TK_NAVIGATION_CUE_STEP_DOWN: prompt += "H\xC3\xA1 um degrau para baixo \xC3\xA0 frente. "; break;
                case TK_NAVIGATION_CUE_STEP_UP: prompt += "H\xC3\xA1 um degrau para cima \xC3\xA0 frente. "; break;
                case TK_NAVIGATION_CUE_STAIRS_DOWN: prompt += "H\xC3\xA1 escadas para baixo \xC3\xA0 frente. "; break;
                case TK_NAVIGATION_CUE_STAIRS_UP: prompt += "H\xC3\xA1 escadas para cima \xC3\xA0 frente. "; break;
                default: break;
            }
            /* motion state (:477-482) */
            switch (s.user_motion_state) {
                case TK_MOTION_STATE_WALKING: prompt += "O usu\xC3\xA1rio est\xC3\xA1 andando. "; break;
                case TK_MOTION_STATE_RUNNING: prompt += "O usu\xC3\xA1rio est\xC3\xA1 correndo. "; break;
                default: prompt += "O usu\xC3\xA1rio est\xC3\xA1 parado. "; break;
            }
            /* long-term memory (:485-487) */
            auto it = g_facts.find("user_name");
            if (it != g_facts.end()) prompt += "O nome do usu\xC3\xA1rio \xC3\xA9 " + it->second + ". ";
            /* the question and the instruction (:490-493) */
            prompt += std::string("O usu\xC3\xA1rio perguntou: '") + (user_query ? user_query : "") + "'. ";
            prompt += "Com base em tudo isso, qual a a\xC3\xA7\xC3\xA3o mais segura e \xC3\xBAtil?";
        }
    }
    if (prompt.find('\0') != std::string::npos) return false; /* CString::new fails on an interior NUL (ffi.rs:406-412) */
    strncpy(prompt_buffer, prompt.c_str(), buffer_size - 1);
    prompt_buffer[buffer_size - 1] = 0;
    return true;
}

} /* extern "C" */
