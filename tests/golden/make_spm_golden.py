"""Fixture generator (run in the BUILD container, where `sentencepiece` is importable; the GPU box and the test suite only read the
JSON): an independent pin of the library's SentencePiece-BPE tokenizer (csrc/llm/tk_tokenizer.cpp; the reference tokenises with
llama.cpp's `llama_tokenize`, src/ai_models/tk_runner_streaming.c:24).

A small BPE model is trained with the settings of the Llama / Mistral `tokenizer.model` (model_type bpe, byte_fallback, identity
normalisation, add_dummy_prefix, remove_extra_whitespaces off, split_digits, whitespace-only pieces allowed), its vocabulary is written
out the way llama.cpp's converter stores it in a GGUF (piece text, score, token type: 1 normal, 2 unknown, 3 control, 6 byte), and the
ids the sentencepiece LIBRARY itself produces for a list of strings are the expected outputs.

    python tests/golden/make_spm_golden.py        ->  tests/golden/spm_bpe_tiny.json
"""
import io
import json
import os
import random

import sentencepiece as spm

HERE = os.path.dirname(os.path.abspath(__file__))

WORDS = ["the", "quick", "brown", "fox", "jumps", "over", "lazy", "dog", "hello", "world", "person", "chair", "table", "door", "left",
         "right", "ahead", "meters", "detected", "user", "asked", "what", "is", "in", "front", "of", "me", "there", "a", "an", "obstacle",
         "stairs", "walk", "stop", "careful", "café", "naïve", "straße", "東京", "こんにちは", "привет", "мир", "日本語", "emoji", "😀",
         "tool_call", "{\"name\":", "arguments", "}", "[", "]", ":", ",", "0", "1", "2", "3", "42", "3.14", "USER", "ASSISTANT", "System",
         "The", "I", "you", "can", "see", "2 persons", "and", "1 chair", "near", "far", "it's", "don't", "(ok)", "a-b", "x=y", "50%"]

STRINGS = [
    "", " ", "  ", "hello world", "hello", " hello", "hello ", "  hello   world  ", "Hello World", "HELLO", "hé", "café naïve straße",
    "東京 tower", "こんにちは世界", "привет мир", "日本語 and English", "😀", "😀😀 emoji 😀", "a\tb", "line one\nline two", "\n", "\r\n\t",
    "What is in front of me?", "There is a chair 2 meters ahead.", "The user asked: what is there", "I can see 2 persons and 1 chair",
    "{\"name\": \"speak\", \"arguments\": {\"text\": \"stop\"}}", "[1, 2, 3]", "3.14159", "42", "1234567890", "x=y+z", "a-b-c", "50% off",
    "it's don't (ok)", "USER: hello\nASSISTANT:", "System: you are a helpful assistant.", "tool_call", "über", "ß", "é", " ",
    "▁", "a▁b", "tab\tseparated\tvalues", "trailing space ", " leading", "dog.", "dog,cat", "zzzzqqqq", "Ω≈ç√∫", "\x01\x02",
    "the quick brown fox jumps over the lazy dog", "obstacle ahead: stairs, walk left", "careful", "person detected 3 meters right",
]


def main():
    random.seed(1)
    lines = [" ".join(random.choice(WORDS) for _ in range(random.randint(3, 14))) for _ in range(4000)]
    model = io.BytesIO()
    spm.SentencePieceTrainer.train(sentence_iterator=iter(lines), model_writer=model, vocab_size=600, model_type="bpe", byte_fallback=True,
                                   normalization_rule_name="identity", add_dummy_prefix=True, remove_extra_whitespaces=False, split_digits=True,
                                   allow_whitespace_only_pieces=True, character_coverage=0.995, unk_id=0, bos_id=1, eos_id=2, pad_id=-1,
                                   num_threads=1, minloglevel=2)
    sp = spm.SentencePieceProcessor(model_proto=model.getvalue())
    n = sp.get_piece_size()
    tokens, scores, types = [], [], []
    for i in range(n):
        tokens.append(sp.id_to_piece(i))
        scores.append(float(sp.get_score(i)))
        types.append(2 if sp.is_unknown(i) else 3 if sp.is_control(i) else 6 if sp.is_byte(i) else 5 if sp.is_unused(i) else 1)
    cases = [{"text": t, "ids": [int(x) for x in sp.encode(t)]} for t in STRINGS]
    out = {"source": "sentencepiece %s, BPE, byte_fallback, identity normalisation, add_dummy_prefix" % spm.__version__,
           "bos_id": 1, "eos_id": 2, "tokens": tokens, "scores": scores, "types": types, "cases": cases}
    with open(os.path.join(HERE, "spm_bpe_tiny.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("pieces", n, "cases", len(cases))


if __name__ == "__main__":
    main()
