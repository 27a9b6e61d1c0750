#!/usr/bin/env python3
"""Compare bench.py's HIP-event launch times of the W4A8 kernels with rocprofv3's kernel durations for the same command.

    rocprofv3 --kernel-trace -d D -o rl --output-format csv -- python3 bench.py --roofline-only > run.log   (TK_MI355X_NO_GRAPH=1)
    tools/roofline_check.py D/rl_kernel_trace.csv run.log

The HIP-event figure is wall time per launch of back-to-back launches (it includes the inter-kernel gap); rocprofv3 reports
begin-to-end of each dispatch.  Shapes are matched by (kernel instantiation, grid size, workgroup size); at 256 rows gate|up and the Q4_K
down projection are the same launch shape (k_gemm32_w4a8<1>, 448 workgroups): their rocprofv3 line is the mixture of the two.
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    trace, log = sys.argv[1], sys.argv[2]
    line = [l for l in open(log) if l.startswith("{")][-1]
    rl = json.loads(line)["roofline"]
    dur = defaultdict(list)
    for r in csv.DictReader(open(trace)):
        name = r["Kernel_Name"]
        if "w4a8" not in name:
            continue
        fam = name.split("(")[0].replace("void ", "")  # with the template arguments: k_gemm32_w4a8<1> (Q4_K), <2> (Q6_K), <3> (mixed q / k / v)
        dur[(fam, int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]), int(r["Workgroup_Size_X"] if "Workgroup_Size_X" in r else r["Workgroup_Size"]))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("rocprofv3 kernel durations by launch shape (us):")
    for k in sorted(dur):
        v = dur[k]
        print(f"  {k[0]:24s} grid={k[1]:7d} wg={k[2]:4d}  n={len(v):5d}  mean={sum(v) / len(v):8.2f}  min={min(v):8.2f}")
    print("bench.py HIP-event per-launch times (us):")
    for name, d in rl["per_shape"].items():
        print(f"  {name:8s} {d['ms'] * 1e3:8.2f}")
    print(f"  weighted average per launch: {rl['avg_launch_ms'] * 1e3:.2f} us, frac of HBM peak {rl['frac']}")


if __name__ == "__main__":
    main()
