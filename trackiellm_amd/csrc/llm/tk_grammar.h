/*
 * tk_grammar.h — GBNF grammars for constrained sampling.
 *
 * The reference hands `src/ai_models/grammars/tool_call.gbnf` to llama.cpp's grammar sampler
 * (src/ai_models/tk_runner_lifecycle.c:59, tk_runner_streaming.c:44-48) and treats "the grammar just completed" as the
 * tool-call signal (tk_runner_streaming.c:69-75).  llama.cpp is not in the reference tree, so this is a restatement of the
 * published GBNF semantics: rules of alternatives of sequences over literals, character classes, rule references, groups
 * and the repetition operators * + ? {n} {n,} {n,m}; the matcher keeps the set of parse stacks that are still alive
 * (llama.cpp's llama_grammar stacks) and advances it one input unit at a time.
 *
 * Units are BYTES: literals contribute their UTF-8 bytes, a negated class accepts every byte >= 0x80 (so any UTF-8 text
 * passes through `[^"\\]`), a positive class may only name ASCII.  That is exact for grammars whose structure is ASCII —
 * the tool-call JSON grammar is — and lets byte-fallback tokens (<0xXX>) be filtered without partial-code-point state.
 *
 * The sampling side: TkTokenTrie indexes the vocabulary's pieces; TkGrammarState::mask() walks trie and stacks together
 * and produces the allowed-token bit set that k_argmax applies on the device (greedy sampling = arg max over the allowed
 * set; the end-of-sequence token is allowed exactly when the grammar is complete).
 */
#ifndef TK_GRAMMAR_H
#define TK_GRAMMAR_H

#include <stdint.h>

#include <array>
#include <map>
#include <string>
#include <vector>

class TkGrammar {
public:
    enum { END = 0, ALT = 1, RULE = 2, SET = 3 };
    struct Elem { uint8_t type; uint32_t value; }; /* RULE: rule id; SET: index into sets */
    typedef std::array<uint32_t, 8> ByteSet;        /* 256 bits */

    bool parse(const std::string& text, std::string* err);
    int root() const { return root_; }
    const std::vector<Elem>& rule(uint32_t id) const { return rules_[id]; }
    bool in_set(uint32_t set, uint8_t b) const { return (sets_[set][b >> 5] >> (b & 31)) & 1u; }
    size_t n_rules() const { return rules_.size(); }

private:
    std::vector<std::vector<Elem>> rules_;
    std::vector<ByteSet> sets_;
    std::map<std::string, uint32_t> names_;
    int root_ = -1;

    uint32_t rule_id(const std::string& name);
    uint32_t new_rule(const std::string& base);
    uint32_t add_set(const ByteSet& s);
    struct Parser;
};

class TkTokenTrie {
public:
    void build(const std::vector<std::string>& pieces); /* pieces[id]; empty piece = never allowed by a grammar */
    struct Node { int32_t child[256]; std::vector<int32_t> tokens; Node() { for (int i = 0; i < 256; ++i) child[i] = -1; } };
    const std::vector<Node>& nodes() const { return nodes_; }
    int vocab() const { return vocab_; }

private:
    std::vector<Node> nodes_;
    int vocab_ = 0;
};

class TkGrammarState {
public:
    typedef std::pair<uint32_t, uint32_t> Pos; /* (rule, element index) still to be matched */
    typedef std::vector<Pos> Stack;            /* top of the parse stack at the back */

    void init(const TkGrammar* g);
    bool accept(uint8_t byte);                  /* false (state unchanged) when no stack accepts the byte */
    bool accept(const std::string& bytes);      /* all or nothing */
    bool complete() const;                      /* some stack is empty: the root rule has been matched */
    bool dead() const { return stacks_.empty(); }
    /* bit i of bits[] = token i may be sampled next; eos_id (>= 0) is allowed iff complete() */
    void mask(const TkTokenTrie& trie, int eos_id, std::vector<uint32_t>* bits) const;

private:
    const TkGrammar* g_ = nullptr;
    std::vector<Stack> stacks_;
    void advance(const Stack& st, std::vector<Stack>* out) const;
    void step(const std::vector<Stack>& in, uint8_t byte, std::vector<Stack>* out) const;
    void walk(const TkTokenTrie& trie, int node, const std::vector<Stack>& stacks, std::vector<uint32_t>* bits) const;
};

/* the tool-call grammar used when the reference's file is not found next to the process (same language as
 * src/ai_models/grammars/tool_call.gbnf: one JSON object, empty or {"tool_call": {"name": string, "arguments": object}}) */
extern const char* const TK_DEFAULT_TOOL_CALL_GBNF;

#endif
