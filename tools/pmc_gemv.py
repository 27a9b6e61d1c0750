#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) to HBM bytes per launch of the
W4A8 kernels, per GEMV shape.  Usage: tools/pmc_gemv.py <fetch counter_collection.csv> <write counter_collection.csv> <rows> profiles/rNN_pmc_{gemm|gemv}_b<rows>.json
   [family [cached_positions]]   (family k_attention: the decode attention launches of tools/time_attention.py at that context length)
The summary is stamped with the sha of the kernel source it was collected from (bench.py reports `traffic` only on a match).

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128 B request for 16 B/lane streams
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section); the shapes are told apart by (grid size, workgroup size, LDS size).
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha  # noqa: E402  (bench.py refuses a summary whose stamp differs from the source it runs)


def load(path, counter, family):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or family not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        acc[key].append(float(r["Counter_Value"]))
    return acc


def main():
    rows = int(sys.argv[3])
    family = "k_gemm32_w4a8" if rows > 128 else "k_gemm_w4a8" if rows > 32 else "k_gemv_w4a8"  # passes of more than 32 rows run the batched variants
    if len(sys.argv) > 5:
        family = sys.argv[5]
    fetch, write = load(sys.argv[1], "FETCH_SIZE", family), load(sys.argv[2], "WRITE_SIZE", family)
    per, tot_b, tot_n = {}, 0.0, 0
    for key in sorted(fetch):
        f, w = fetch[key], write.get(key, [0.0])
        fm, wm = sum(f) / len(f), sum(w) / len(w)
        b = (2.0 * fm + wm) * 1024.0
        per["%s grid=%d wg=%d" % key] = {"FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm, "dispatches": len(f), "hbm_bytes_per_launch": round(b)}
        tot_b += b * len(f)
        tot_n += len(f)
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the same eager bench command; "
                   "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction for 16 B/lane streams)",
           "rows_per_pass": rows, "kernel": family, **({"cached_positions": int(sys.argv[6])} if len(sys.argv) > 6 else {}), "kernel_source_sha": kernel_source_sha(), "per_variant": per, "hbm_bytes_per_average_launch": round(tot_b / max(tot_n, 1)), "launches": tot_n}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in per.items()}, indent=1))
    print("average per launch:", out["hbm_bytes_per_average_launch"])


if __name__ == "__main__":
    main()
