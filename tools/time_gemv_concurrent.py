#!/usr/bin/env python3
"""Do k_gemv_w4a8 launches of concurrent streams overlap as plain reads do (tools/concurrent_read_probe.hip)?  S sessions of one model,
each on its own stream, time the SAME launch shape at the same moment (tk_mi355x_llm_time_gemv: a graph of launches cycling through the
layers, HIP events on the session's stream); prints each stream's average launch time and the aggregate weight rate.
    python tools/time_gemv_concurrent.py [rows] [streams ...]"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import trackiellm_amd as tk  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
streams = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
model = tk.LlmModel(tk.MISTRAL_7B(), device=0).fill_synthetic(4)
hp = model.hparams
wb = bench.weight_bytes_per_shape(hp)
q6 = bench.q6_layers(hp.n_layer)
q4 = [l for l in range(hp.n_layer) if l not in q6]
sess = [tk.LlmSession(model, rows, 64) for _ in range(max(streams))]
for name, which, layer in (("gate_up", 0, 0), ("down_q6", 1, q6[0]), ("down_q4", 1, q4[0]), ("qkv_q4", 2, q4[0]), ("o", 4, 0), ("lm_head", 3, 0)):
    for S in streams:
        res = [None] * S
        bar = threading.Barrier(S)

        def run(i):
            sess[i].time_gemv(layer, which, rows, 20)
            bar.wait()
            res[i] = sess[i].time_gemv(layer, which, rows, 200)[0]

        th = [threading.Thread(target=run, args=(i,)) for i in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        worst = max(res)
        print(f"{name:8s} {rows} rows, {S} stream(s): {' '.join('%.2f' % (1e3 * r) for r in res)} us per launch; aggregate {S * wb[name] / worst / 1e9:.2f} TB/s "
              f"({S * wb[name] / worst / 8e9:.3f} of 8)", flush=True)
