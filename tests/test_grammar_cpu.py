"""GBNF engine behind use_tool_grammar (no GPU): parser, byte-level matcher, completion flag — through the C-ABI.

The reference arms llama.cpp's grammar sampler with src/ai_models/grammars/tool_call.gbnf (tk_runner_streaming.c:44-48) and
reads "grammar completed" as the tool-call signal (:69-75).  llama.cpp is not in the reference tree, so these tests pin the
language itself: documents built by an independent generator must be accepted and complete, mutated ones must stop where a
hand-derived expectation says, and the reference's own grammar file (when the reference tree is present) must agree with the
built-in text on every sample.
"""
import ctypes as C
import json
import os
import random

import pytest

import trackiellm_amd as tk

REF_GBNF = "/root/reference/src/ai_models/grammars/tool_call.gbnf"


def check(text, gbnf=None):
    n, c = C.c_int32(), C.c_int32()
    rc = tk.lib().tk_mi355x_grammar_check(None if gbnf is None else gbnf.encode(), text.encode(), C.byref(n), C.byref(c))
    assert rc == 0, tk.lib().tk_error_get_detail()
    return n.value, bool(c.value)


def next_bytes(prefix, gbnf=None):
    allowed = (C.c_uint8 * 256)()
    c = C.c_int32()
    rc = tk.lib().tk_mi355x_grammar_next_bytes(None if gbnf is None else gbnf.encode(), prefix.encode(), allowed, C.byref(c))
    assert rc == 0
    return {b for b in range(256) if allowed[b]}, bool(c.value)


def gen_value(rng, depth):
    k = rng.randrange(7 if depth < 3 else 5)
    ws = lambda: rng.choice(["", " ", "\n", "\t ", "  "])  # noqa: E731
    if k == 0:
        return json.dumps(rng.choice(["", "a", "tool", "é ü", 'q"uote', "back\\slash", "tab\tnl\n", "é中"])) + ws()
    if k == 1:
        return rng.choice(["0", "-0", "7", "-12", "3.25", "1e5", "-2.5E-3", "10.0e+2"]) + ws()
    if k == 2:
        return rng.choice(["true", "false", "null"])
    if k == 3:
        return "{" + ws() + ws() + "}"
    if k == 4:
        return "{" + ws() + gen_call(rng, depth + 1) + ws() + "}"
    if k == 5:
        n = rng.randrange(4)
        return "[" + ws() + ("," + ws()).join(gen_value(rng, depth + 1) for _ in range(n)) + ws() + "]"
    return "[" + ws() + "]"


def gen_call(rng, depth):
    ws = lambda: rng.choice(["", " ", "\n"])  # noqa: E731
    n = rng.randrange(3)
    args = ("," + ws()).join(json.dumps(rng.choice(["a", "key", "x y"])) + ws() + ":" + ws() + gen_value(rng, depth + 1) for _ in range(n))
    return ('"tool_call":' + ws() + "{" + ws() + '"name":' + ws() + json.dumps(rng.choice(["navigate", "speak", "f"])) + ws() + "," + ws() +
            '"arguments":' + ws() + "{" + ws() + args + ws() + "}" + ws() + "}")


def test_generated_tool_calls_are_accepted_and_complete():
    rng = random.Random(7)
    for _ in range(300):
        doc = "{" + rng.choice(["", " ", "\n"]) + gen_call(rng, 0) + rng.choice(["", " "]) + "}"
        n, done = check(doc)
        assert (n, done) == (len(doc.encode()), True), doc
        # json agrees on the structure (whitespace and escapes included)
        parsed = json.loads(doc)
        assert set(parsed) == {"tool_call"} and set(parsed["tool_call"]) == {"name", "arguments"}
        # nothing may follow the closing brace, and no proper prefix is complete
        assert check(doc + " ")[0] == len(doc.encode())
        cut = rng.randrange(1, len(doc))
        assert not check(doc[:cut])[1] or doc[:cut].rstrip() != doc[:cut]


@pytest.mark.parametrize("text,n,done", [
    ("{}", 2, True), ("{ \n\t}", 5, True), ("", 0, False), ("{", 1, False), ("[1]", 0, False),
    ('{"x":1}', 2, False),                                   # only the tool_call key is allowed at the top
    ('{"tool_call" :{}}', 12, False),                        # no space between key and colon
    ('{"tool_call":{"name":5}}', 21, False),                 # name must be a string
    ('{"tool_call":{"arguments":{},"name":"a"}}', 15, False),  # fixed key order
    ('{"tool_call":{"name":"a","arguments":{"k":01}}}', 43, False),  # no leading zeros
    ('{"tool_call":{"name":"\\u12G4","arguments":{}}}', 26, False),  # exactly four hex digits
    ('{"tool_call":{"name":"\\x","arguments":{}}}', 23, False),      # unknown escape
    ('{"tool_call":{"name":"a","arguments":{"k":tru}}}', 45, False),
])
def test_hand_derived_stop_points(text, n, done):
    assert check(text) == (n, done)


def test_next_bytes_at_key_points():
    ws = {0x20, 0x09, 0x0A}
    assert next_bytes("") == ({ord("{")}, False)
    assert next_bytes("{") == (ws | {ord('"'), ord("}")}, False)
    assert next_bytes('{"') == ({ord("t")}, False)
    assert next_bytes("{}") == (set(), True)
    inside, _ = next_bytes('{"tool_call":{"name":"ab')
    assert inside == set(range(256)) - set()  # every byte: text, closing quote, backslash escape ...
    esc, _ = next_bytes('{"tool_call":{"name":"ab\\')
    assert esc == {ord(c) for c in '"\\/bfnrtu'}
    num, _ = next_bytes('{"tool_call":{"name":"a","arguments":{"k":-')
    assert num == {ord(c) for c in "0123456789"}
    after0, _ = next_bytes('{"tool_call":{"name":"a","arguments":{"k":0')
    assert after0 == ws | {ord(c) for c in ".eE,}"}


def test_operators_groups_comments_and_errors():
    g = 'root ::= item+ tail? # trailing comment\nitem ::= "ab" | [x-z]{2,3} | (\n  "(" root ")"\n)\ntail ::= "!"*\n'
    assert check("ab", g) == (2, True)
    assert check("abxy", g) == (4, True)
    assert check("x", g) == (1, False)
    assert check("xyzx", g) == (4, True)       # two items of two
    assert check("abx", g) == (3, False)       # a lone x needs one more
    assert check("xyzxy!!", g) == (7, True)
    assert check("(ab(zz))!", g) == (9, True)
    assert check("(ab", g) == (3, False)
    assert check("ab?", g) == (2, True)        # '?' is not in the language; the accepted prefix is complete
    assert check("é", 'root ::= [^a]') == (1, True)   # a negated class takes one BYTE (the matcher is byte-level): the second byte is left over
    assert check("é", 'root ::= [^a] [^a]') == (2, True)
    assert check("é", 'root ::= "é"') == (2, True)
    for bad in ('root ::= missing', 'other ::= "a"', 'root ::= "abc', 'root ::= [a-', 'root = "a"', 'root ::= "a"{3,2}', 'root ::= [à]'):
        n, c = C.c_int32(), C.c_int32()
        assert tk.lib().tk_mi355x_grammar_check(bad.encode(), b"a", C.byref(n), C.byref(c)) != 0, bad


@pytest.mark.skipif(not os.path.exists(REF_GBNF), reason="reference tree not present (GPU box)")
def test_reference_grammar_file_defines_the_same_language_on_samples():
    ref = open(REF_GBNF).read()
    rng = random.Random(11)
    docs = ["{}", "{ }", '{"x":1}', '{"tool_call":{"name":5}}', "", "{", '{"tool_call":{"name":"a","arguments":{"k":01}}}']
    docs += ["{" + gen_call(rng, 0) + "}" for _ in range(100)]
    for d in list(docs):
        if len(d) > 4:
            docs.append(d[: rng.randrange(1, len(d))] + rng.choice(['"', "x", "}", " ", ":"]) + d[rng.randrange(1, len(d)):])
    for d in docs:
        assert check(d) == check(d, ref), d
