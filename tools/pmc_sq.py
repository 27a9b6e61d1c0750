#!/usr/bin/env python3
"""Reduce one rocprofv3 --pmc pass of SQ counters over `tools/time_gemv.py` to per-launch-shape fractions of SQ_WAVE_CYCLES.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU \\
        --output-format csv -d D -o p -- python3 tools/time_gemv.py --rows 256       (TK_MI355X_NO_GRAPH=1)
    tools/pmc_sq.py D/.../p_counter_collection.csv

Reading (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAIT_ANY = wave parked at s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall,
SQ_ACTIVE_INST_ANY = issuing; the three add up to ~1 of SQ_WAVE_CYCLES.  SQ_WAVE_CYCLES etc. count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
counts cycles: the MFMA pipe's busy share of a SIMD's time is MFMA_BUSY / (4 * SQ_BUSY_CYCLES-per-SIMD) and is printed as reported.
"""
import csv
import sys
from collections import defaultdict


FILTER = sys.argv[2:] or ["w4a8", "k_attention", "k_gemm_tiled"]  # kernel-name substrings (further arguments replace the default)


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(sys.argv[1])):
        if not any(f in r["Kernel_Name"] for f in FILTER):
            continue
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        n = len(next(iter(acc[key].values())))
        frac = {k: round(v / wc, 3) for k, v in c.items() if k != "SQ_WAVE_CYCLES"}
        print(f"{key[0]} grid={key[1]} wg={key[2]} launches={n} SQ_WAVE_CYCLES={wc:.4g} " + " ".join(f"{k}={v}" for k, v in sorted(frac.items())))


if __name__ == "__main__":
    main()
