#!/bin/bash
# Collects the round's judged measurements on an MI355X box (run through gpurun from the repo root):
#   1. PMC traffic of the W4A8 kernels (FETCH_SIZE and WRITE_SIZE in separate passes, eager launches) at 256 and 16 rows per pass
#   2. rocprofv3 --kernel-trace --stats of the headline bench command and of --roofline-only (eager: see DESIGN.md "Profiling")
#   3. the default bench.py line (hipGraph replay, the shipped path)
# Everything lands under gpurun_out/; copy what should be judged into profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${ROUND:-r02}
O=gpurun_out
export TK_MI355X_NO_GRAPH=1
for B in 256 16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "/tmp/pmc_${B}_${C}"
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${B}_$C -o p -- python3 bench.py --llm-only --no-extras --batch $B --sessions 1 --steps 1 --warmup 0 --prompt 4 --decode 8 > $O/${R}_pmc_${B}_$C.log 2>&1 || { echo "pmc $B $C failed"; tail -5 $O/${R}_pmc_${B}_$C.log; exit 1; }
    echo "pmc $B $C done"
  done
  fam=gemm; [ $B -le 32 ] && fam=gemv
  python3 tools/pmc_gemv.py $(find /tmp/pmc_${B}_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_${B}_WRITE_SIZE -name '*counter_collection.csv' | head -1) $B $O/${R}_pmc_${fam}_b$B.json > $O/${R}_pmc_${fam}_b$B.txt || exit 1
  tail -1 $O/${R}_pmc_${fam}_b$B.txt
done
rm -rf /tmp/kt_rl
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_rl -o rl -- python3 bench.py --roofline-only > $O/${R}_roofline_only_run.json 2> $O/${R}_roofline_only.err || { echo "roofline-only trace failed"; exit 1; }
cp $(find /tmp/kt_rl -name '*kernel_stats.csv' | head -1) $O/${R}_roofline_only_kernel_stats.csv
python3 tools/roofline_check.py $(find /tmp/kt_rl -name '*kernel_trace.csv' | head -1) $O/${R}_roofline_only_run.json > $O/${R}_roofline_check.txt 2>&1; tail -12 $O/${R}_roofline_check.txt
rm -rf /tmp/kt_f
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-extras > $O/${R}_fused_3x256_rocprof_run.json 2> $O/${R}_fused_rocprof.err || { echo "fused trace failed"; exit 1; }
cp $(find /tmp/kt_f -name '*kernel_stats.csv' | head -1) $O/${R}_fused_3x256_kernel_stats.csv
echo "traces done"
unset TK_MI355X_NO_GRAPH
timeout -k 10 900 python3 bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err; echo "bench rc=$?"; tail -c 1500 $O/${R}_bench_default.json
