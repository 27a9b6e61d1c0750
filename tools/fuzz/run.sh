#!/bin/bash
# Builds tools/fuzz/fuzz_readers.cpp with g++ -fsanitize=address,undefined (host sources only) and runs it over samples written by the
# tests' own writers.   tools/fuzz/run.sh [iterations per sample, default 1500]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=${TMPDIR:-/tmp}/tk_fuzz; mkdir -p $W
cd $ROOT/tests
python - "$W" <<'PY'
import sys
import numpy as np
import gguf_util, onnx_util as X, oracle_lib as O
W = sys.argv[1]
cfg = O.tiny_config()
rng = np.random.default_rng(1)
D, QD, FF = cfg.d_model, cfg.n_head * cfg.head_dim, cfg.d_ff
fs = {(0, 1): (rng.normal(0, .05, (4, D)).astype(np.float32), rng.normal(0, .05, (QD, 4)).astype(np.float32)),
      (1, 8): (rng.normal(0, .05, (4, FF)).astype(np.float32), rng.normal(0, .05, (D, 4)).astype(np.float32))}
gguf_util.write_lora_ggla(W + "/a.ggla", 4, 8, fs)
gguf_util.write_lora_gguf(W + "/b.gguf", 8.0, fs, f16=True)
gguf_util.write_llama_gguf(W + "/model.gguf", O.OracleLlm(cfg, seed=4), cfg)
convs = [{"name": "c%d" % i, "w": rng.normal(0, .1, s).astype(np.float32), "b": rng.normal(0, .1, s[0]).astype(np.float32)} for i, s in enumerate([(8, 3, 3, 3), (8, 8, 1, 1)])]
open(W + "/y.onnx", "wb").write(X.yolo_model(convs, with_dfl=False))
open(W + "/loop.onnx", "wb").write(X.loopnet_model(X.loopnet_weights(3), X.loopnet_spec("cond")))  # nested graphs: the bodies of a Loop and a Scan
open(W + "/g.gbnf", "w").write('root ::= "{" ws item ("," ws item)* "}"\nitem ::= [a-z]+ ":" [0-9]+\nws ::= [ \\t\\n]*\n')
PY
cd $ROOT/trackiellm_amd/csrc
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I. -I$ROOT/include \
    $ROOT/tools/fuzz/fuzz_readers.cpp llm/tk_lora.cpp llm/tk_gguf.cpp llm/tk_grammar.cpp llm/tk_tokenizer.cpp audio/tk_whisper_ggml.cpp nn/tk_onnx_graph.cpp vision/tk_onnx_weights.cpp -o $W/fuzz_readers
$W/fuzz_readers $W/scratch.bin ${1:-1500} $(ls $W/a.ggla $W/b.gguf $W/model.gguf $W/g.gbnf $W/y.onnx $W/loop.onnx 2>/dev/null)
