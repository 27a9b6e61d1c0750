/* tk_grammar.cpp — GBNF parser, byte-level stack matcher and token masks (see tk_grammar.h) */
#include "tk_grammar.h"

#include <string.h>

#include <algorithm>

const char* const TK_DEFAULT_TOOL_CALL_GBNF =
    "root ::= \"{\" ws ( \"\\\"tool_call\\\":\" ws call )? ws \"}\"\n"
    "call ::= \"{\" ws \"\\\"name\\\":\" ws string \",\" ws \"\\\"arguments\\\":\" ws args ws \"}\"\n"
    "args ::= \"{\" ws ( string \":\" ws value ( \",\" ws string \":\" ws value )* )? ws \"}\"\n"
    "value ::= object | array | string | number | \"true\" | \"false\" | \"null\"\n"
    "object ::= \"{\" ws ( \"\\\"tool_call\\\":\" ws call )? ws \"}\"\n"
    "array ::= \"[\" ws ( value ( \",\" ws value )* )? ws \"]\"\n"
    "string ::= \"\\\"\" ( [^\"\\\\] | \"\\\\\" ( [\"\\\\/bfnrt] | \"u\" [0-9a-fA-F]{4} ) )* \"\\\"\" ws\n"
    "number ::= \"-\"? ( [0-9] | [1-9] [0-9]* ) ( \".\" [0-9]+ )? ( [eE] [-+]? [0-9]+ )? ws\n"
    "ws ::= [ \\t\\n]*\n";

/* ------------------------------------------------------------------ parser ------ */

uint32_t TkGrammar::rule_id(const std::string& name) {
    auto it = names_.find(name);
    if (it != names_.end()) return it->second;
    const uint32_t id = (uint32_t)rules_.size();
    names_[name] = id;
    rules_.emplace_back();
    return id;
}

uint32_t TkGrammar::new_rule(const std::string& base) {
    return rule_id(base + "_" + std::to_string(rules_.size())); /* '_<n>' cannot collide: n grows with every rule */
}

uint32_t TkGrammar::add_set(const ByteSet& s) {
    for (size_t i = 0; i < sets_.size(); ++i)
        if (sets_[i] == s) return (uint32_t)i;
    sets_.push_back(s);
    return (uint32_t)sets_.size() - 1;
}

struct TkGrammar::Parser {
    TkGrammar& g;
    const char* p;
    const char* end;
    std::string err;

    bool fail(const std::string& m) { if (err.empty()) err = m + " near '" + std::string(p, std::min<size_t>(16, (size_t)(end - p))) + "'"; return false; }
    static bool word(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-'; }

    void space(bool newline_ok) {
        while (p < end) {
            if (*p == ' ' || *p == '\t' || ((*p == '\n' || *p == '\r') && newline_ok)) ++p;
            else if (*p == '#') { while (p < end && *p != '\n') ++p; }
            else break;
        }
    }
    bool name(std::string* out) {
        const char* s = p;
        while (p < end && word(*p)) ++p;
        if (p == s) return fail("expected a rule name");
        out->assign(s, p);
        return true;
    }
    static int hexv(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
    /* one (possibly escaped) character -> code point */
    bool chr(uint32_t* cp) {
        if (p >= end) return fail("unexpected end");
        if (*p != '\\') {
            const unsigned char c = (unsigned char)*p;
            int n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
            if (!n || p + n > end) return fail("bad UTF-8 in grammar");
            uint32_t v = n == 1 ? c : c & (0xff >> (n + 1));
            for (int i = 1; i < n; ++i) v = (v << 6) | ((unsigned char)p[i] & 0x3f);
            *cp = v;
            p += n;
            return true;
        }
        if (p + 1 >= end) return fail("dangling escape");
        const char e = p[1];
        p += 2;
        int digits = e == 'x' ? 2 : e == 'u' ? 4 : e == 'U' ? 8 : 0;
        if (digits) {
            uint32_t v = 0;
            for (int i = 0; i < digits; ++i) {
                if (p >= end || hexv(*p) < 0) return fail("bad hex escape");
                v = v * 16 + (uint32_t)hexv(*p++);
            }
            *cp = v;
            return true;
        }
        switch (e) {
            case 'n': *cp = '\n'; break;
            case 't': *cp = '\t'; break;
            case 'r': *cp = '\r'; break;
            case '\\': case '"': case '[': case ']': *cp = (uint32_t)e; break;
            default: return fail("unknown escape");
        }
        return true;
    }
    static void utf8(uint32_t cp, std::string* out) {
        if (cp < 0x80) *out += (char)cp;
        else if (cp < 0x800) { *out += (char)(0xC0 | (cp >> 6)); *out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { *out += (char)(0xE0 | (cp >> 12)); *out += (char)(0x80 | ((cp >> 6) & 0x3F)); *out += (char)(0x80 | (cp & 0x3F)); }
        else { *out += (char)(0xF0 | (cp >> 18)); *out += (char)(0x80 | ((cp >> 12) & 0x3F)); *out += (char)(0x80 | ((cp >> 6) & 0x3F)); *out += (char)(0x80 | (cp & 0x3F)); }
    }
    void single(uint8_t b, std::vector<Elem>* out) {
        ByteSet s{};
        s[b >> 5] |= 1u << (b & 31);
        out->push_back(Elem{SET, g.add_set(s)});
    }

    bool repeat(const std::string& rname, std::vector<Elem>* out, size_t sym, long lo, long hi /* -1 = unbounded */) {
        const std::vector<Elem> S(out->begin() + (long)sym, out->end());
        if (S.empty()) return fail("repetition without a symbol");
        out->resize(sym);
        for (long i = 0; i < lo; ++i) out->insert(out->end(), S.begin(), S.end());
        if (hi < 0) { /* S* : star ::= S star | */
            const uint32_t star = g.new_rule(rname);
            std::vector<Elem> body(S);
            body.push_back(Elem{RULE, star});
            body.push_back(Elem{ALT, 0});
            body.push_back(Elem{END, 0});
            g.rules_[star] = body;
            out->push_back(Elem{RULE, star});
        } else { /* (hi - lo) nested optionals: opt_k ::= S opt_(k-1) | */
            uint32_t last = 0;
            bool have = false;
            for (long i = 0; i < hi - lo; ++i) {
                const uint32_t r = g.new_rule(rname);
                std::vector<Elem> body(S);
                if (have) body.push_back(Elem{RULE, last});
                body.push_back(Elem{ALT, 0});
                body.push_back(Elem{END, 0});
                g.rules_[r] = body;
                last = r;
                have = true;
            }
            if (have) out->push_back(Elem{RULE, last});
        }
        return true;
    }

    bool sequence(const std::string& rname, std::vector<Elem>* out, bool nested) {
        size_t sym = out->size();
        while (p < end) {
            if (*p == '"') {
                ++p;
                sym = out->size();
                while (p < end && *p != '"') {
                    uint32_t cp;
                    if (!chr(&cp)) return false;
                    std::string b;
                    utf8(cp, &b);
                    for (unsigned char c : b) single(c, out);
                }
                if (p >= end) return fail("unterminated literal");
                ++p;
                space(nested);
            } else if (*p == '[') {
                ++p;
                bool neg = false;
                if (p < end && *p == '^') { neg = true; ++p; }
                ByteSet s{};
                while (p < end && *p != ']') {
                    uint32_t a, b;
                    if (!chr(&a)) return false;
                    b = a;
                    if (p + 1 < end && *p == '-' && p[1] != ']') { ++p; if (!chr(&b)) return false; }
                    if (a >= 0x80 || b >= 0x80) return fail("non-ASCII characters in a class are not supported by the byte-level matcher");
                    for (uint32_t c = a; c <= b; ++c) s[c >> 5] |= 1u << (c & 31);
                }
                if (p >= end) return fail("unterminated character class");
                ++p;
                if (neg) for (auto& w : s) w = ~w; /* every byte >= 0x80 passes a negated class */
                sym = out->size();
                out->push_back(Elem{SET, g.add_set(s)});
                space(nested);
            } else if (*p == '.') {
                ++p;
                ByteSet s;
                s.fill(0xffffffffu);
                sym = out->size();
                out->push_back(Elem{SET, g.add_set(s)});
                space(nested);
            } else if (word(*p)) {
                std::string n;
                if (!name(&n)) return false;
                space(nested);
                sym = out->size();
                out->push_back(Elem{RULE, g.rule_id(n)});
            } else if (*p == '(') {
                ++p;
                space(true);
                const uint32_t sub = g.new_rule(rname);
                if (!alternates(rname, sub, true)) return false;
                if (p >= end || *p != ')') return fail("expected ')'");
                ++p;
                space(nested);
                sym = out->size();
                out->push_back(Elem{RULE, sub});
            } else if (*p == '*' || *p == '+' || *p == '?') {
                const char op = *p++;
                space(nested);
                if (!repeat(rname, out, sym, op == '+' ? 1 : 0, op == '?' ? 1 : -1)) return false;
                sym = out->size();
            } else if (*p == '{') {
                ++p;
                long lo = 0, hi;
                if (p >= end || *p < '0' || *p > '9') return fail("expected a repetition count");
                while (p < end && *p >= '0' && *p <= '9') lo = lo * 10 + (*p++ - '0');
                hi = lo;
                if (p < end && *p == ',') {
                    ++p;
                    if (p < end && *p >= '0' && *p <= '9') { hi = 0; while (p < end && *p >= '0' && *p <= '9') hi = hi * 10 + (*p++ - '0'); }
                    else hi = -1;
                }
                if (p >= end || *p != '}') return fail("expected '}'");
                ++p;
                space(nested);
                if ((hi >= 0 && hi < lo) || lo > 1024 || hi > 1024) return fail("bad repetition bounds");
                if (!repeat(rname, out, sym, lo, hi)) return false;
                sym = out->size();
            } else {
                break;
            }
        }
        return true;
    }

    bool alternates(const std::string& rname, uint32_t rule, bool nested) {
        std::vector<Elem> body;
        if (!sequence(rname, &body, nested)) return false;
        while (p < end && *p == '|') {
            ++p;
            space(true);
            body.push_back(Elem{ALT, 0});
            if (!sequence(rname, &body, nested)) return false;
        }
        body.push_back(Elem{END, 0});
        g.rules_[rule] = body;
        return true;
    }

    bool rule() {
        std::string n;
        if (!name(&n)) return false;
        space(false);
        if (p + 2 >= end || p[0] != ':' || p[1] != ':' || p[2] != '=') return fail("expected ::=");
        p += 3;
        space(true);
        const uint32_t id = g.rule_id(n);
        if (!alternates(n, id, false)) return false;
        if (p < end && *p == '\r') ++p;
        if (p < end && *p != '\n') return fail("expected end of rule");
        space(true);
        return true;
    }
};

bool TkGrammar::parse(const std::string& text, std::string* err) {
    rules_.clear();
    sets_.clear();
    names_.clear();
    root_ = -1;
    Parser ps{*this, text.data(), text.data() + text.size(), std::string()};
    ps.space(true);
    while (ps.p < ps.end)
        if (!ps.rule()) { if (err) *err = ps.err; return false; }
    for (const auto& kv : names_)
        if (rules_[kv.second].empty()) { if (err) *err = "undefined rule: " + kv.first; return false; }
    auto it = names_.find("root");
    if (it == names_.end()) { if (err) *err = "grammar has no root rule"; return false; }
    root_ = (int)it->second;
    return true;
}

/* ------------------------------------------------------------------ vocabulary trie ------ */

void TkTokenTrie::build(const std::vector<std::string>& pieces) {
    nodes_.clear();
    nodes_.emplace_back();
    vocab_ = (int)pieces.size();
    for (int id = 0; id < vocab_; ++id) {
        const std::string& s = pieces[id];
        if (s.empty()) continue;
        int n = 0;
        for (unsigned char c : s) {
            if (nodes_[n].child[c] < 0) {
                const int nn = (int)nodes_.size();
                nodes_[n].child[c] = nn; /* index first: emplace_back may move the nodes */
                nodes_.emplace_back();
            }
            n = nodes_[n].child[c];
        }
        nodes_[n].tokens.push_back(id);
    }
}

/* ------------------------------------------------------------------ matcher ------ */

void TkGrammarState::advance(const Stack& st, std::vector<Stack>* out) const {
    if (st.empty()) { out->push_back(st); return; }
    const Pos top = st.back();
    const std::vector<TkGrammar::Elem>& r = g_->rule(top.first);
    const TkGrammar::Elem e = r[top.second];
    if (e.type == TkGrammar::SET) { out->push_back(st); return; }
    if (e.type != TkGrammar::RULE) return; /* END / ALT never sit on a stack */
    Stack base(st.begin(), st.end() - 1);
    const uint8_t nt = r[top.second + 1].type;
    if (nt != TkGrammar::END && nt != TkGrammar::ALT) base.push_back(Pos(top.first, top.second + 1));
    if (base.size() > 4096) return; /* runaway recursion guard */
    const std::vector<TkGrammar::Elem>& sub = g_->rule(e.value);
    uint32_t i = 0;
    for (;;) {
        Stack ns(base);
        if (sub[i].type != TkGrammar::END && sub[i].type != TkGrammar::ALT) ns.push_back(Pos(e.value, i));
        advance(ns, out);
        while (sub[i].type != TkGrammar::END && sub[i].type != TkGrammar::ALT) ++i;
        if (sub[i].type == TkGrammar::END) break;
        ++i;
    }
}

static void dedupe(std::vector<TkGrammarState::Stack>* v) {
    std::sort(v->begin(), v->end());
    v->erase(std::unique(v->begin(), v->end()), v->end());
}

void TkGrammarState::init(const TkGrammar* g) {
    g_ = g;
    stacks_.clear();
    const std::vector<TkGrammar::Elem>& r = g->rule((uint32_t)g->root());
    uint32_t i = 0;
    for (;;) {
        Stack s;
        if (r[i].type != TkGrammar::END && r[i].type != TkGrammar::ALT) s.push_back(Pos((uint32_t)g->root(), i));
        advance(s, &stacks_);
        while (r[i].type != TkGrammar::END && r[i].type != TkGrammar::ALT) ++i;
        if (r[i].type == TkGrammar::END) break;
        ++i;
    }
    dedupe(&stacks_);
}

void TkGrammarState::step(const std::vector<Stack>& in, uint8_t byte, std::vector<Stack>* out) const {
    out->clear();
    for (const Stack& st : in) {
        if (st.empty()) continue;
        const Pos top = st.back();
        const std::vector<TkGrammar::Elem>& r = g_->rule(top.first);
        if (r[top.second].type != TkGrammar::SET || !g_->in_set(r[top.second].value, byte)) continue;
        Stack ns(st.begin(), st.end() - 1);
        const uint8_t nt = r[top.second + 1].type;
        if (nt != TkGrammar::END && nt != TkGrammar::ALT) ns.push_back(Pos(top.first, top.second + 1));
        advance(ns, out);
    }
    dedupe(out);
}

bool TkGrammarState::accept(uint8_t byte) {
    std::vector<Stack> next;
    step(stacks_, byte, &next);
    if (next.empty()) return false;
    stacks_.swap(next);
    return true;
}

bool TkGrammarState::accept(const std::string& bytes) {
    std::vector<Stack> cur(stacks_), next;
    for (unsigned char c : bytes) {
        step(cur, c, &next);
        if (next.empty()) return false;
        cur.swap(next);
    }
    stacks_.swap(cur);
    return true;
}

bool TkGrammarState::complete() const {
    for (const Stack& s : stacks_)
        if (s.empty()) return true;
    return false;
}

void TkGrammarState::walk(const TkTokenTrie& trie, int node, const std::vector<Stack>& stacks, std::vector<uint32_t>* bits) const {
    const TkTokenTrie::Node& n = trie.nodes()[node];
    std::vector<Stack> next;
    for (int b = 0; b < 256; ++b) {
        const int c = n.child[b];
        if (c < 0) continue;
        step(stacks, (uint8_t)b, &next);
        if (next.empty()) continue;
        for (int32_t t : trie.nodes()[c].tokens) (*bits)[t >> 5] |= 1u << (t & 31);
        walk(trie, c, next, bits);
    }
}

void TkGrammarState::mask(const TkTokenTrie& trie, int eos_id, std::vector<uint32_t>* bits) const {
    bits->assign(((size_t)trie.vocab() + 31) / 32, 0u);
    if (!trie.nodes().empty()) walk(trie, 0, stacks_, bits);
    if (eos_id >= 0 && eos_id < trie.vocab() && complete()) (*bits)[eos_id >> 5] |= 1u << (eos_id & 31);
}
