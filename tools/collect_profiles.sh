#!/bin/bash
# Collects everything under profiles/rNN_* in one GPU call (run through gpurun from the repo root):
#   1. PMC traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of the W4A8 launch set at 256 and 16 rows and of k_attention at 256 rows x 128 positions
#   2. kernel traces of the roofline-only run and of the fused default run — of the GRAPH path (the shipped one): rocprofv3's queue
#      interceptor walks a multi-packet submission linearly and runs off the end of the 16384-packet AQL ring when a hipGraphLaunch
#      batch straddles it (profiles/r03_rocprofv3_hipgraph_crash_symbolised.txt), so the ring is made large enough never to wrap inside a
#      profiled run (ROC_AQL_QUEUE_SIZE)
#   3. the default bench.py line (unprofiled)
# Everything lands under gpurun_out/; copy what should be judged into profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=${ROUND:-r05}
O=gpurun_out
one_csv() { # exactly one counter file per pass, or stop: a stale directory must never feed a freshly stamped summary
  local n; n=$(find "$1" -name '*counter_collection.csv' | wc -l)
  [ "$n" = 1 ] || { echo "expected one counter_collection.csv under $1, found $n" >&2; exit 1; }
  find "$1" -name '*counter_collection.csv'
}
# PMC passes run the GRAPH path too (the shipped one): a counter pass yields one dispatch record per kernel node of a replayed graph.
# ROC_AQL_QUEUE_SIZE: rocprofv3's queue interceptor walks a multi-packet submission linearly and runs off the end of the AQL ring when a
# hipGraphLaunch batch straddles it (profiles/r03_rocprofv3_hipgraph_crash_symbolised.txt); the ring must therefore never wrap inside a
# profiled process.  PACKET BUDGET: every profiled command below must submit fewer packets per queue than the ring holds — kernels + a
# barrier packet per graph launch; bench.py --steps 1 --warmup 1 at 3 x 256 is the largest here: 359 k dispatches over its streams' queues
# (measured, round 4; three LLM groups x (64 prefill passes + 128 decode steps) x ~260 kernels, twice, + perception).  More steps, warm-up or decode length need a larger ring FIRST, or the abort of round 2 comes back.
export ROC_AQL_QUEUE_SIZE=524288
check_budget() { # kernel-trace rows of a finished run against the ring
  local n; n=$(($(cat "$1" | wc -l) - 1))
  echo "packet budget: $n dispatches in $1 (all queues together; a HIP stream has a queue of its own) against a ring of $ROC_AQL_QUEUE_SIZE packets per queue"
  # the count is over ALL queues, so it over-estimates every single queue's share: a run that passes this test cannot have wrapped any ring
  [ "$n" -lt $((ROC_AQL_QUEUE_SIZE * 9 / 10)) ] || { echo "ERROR: $n dispatches are within 10 % of the ring of $ROC_AQL_QUEUE_SIZE packets: enlarge ROC_AQL_QUEUE_SIZE before profiling this run" >&2; exit 1; }
}
for B in 256 16; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf "/tmp/pmc_${B}_${C}"
    timeout -k 10 400 rocprofv3 --pmc $C --output-format csv -d "/tmp/pmc_${B}_${C}" -o p -- python3 tools/time_gemv.py --rows $B --iters 10 > $O/${R}_pmc_${B}_$C.log 2>&1 || { echo "pmc $B $C failed"; tail -5 $O/${R}_pmc_${B}_$C.log; exit 1; }
    echo "pmc $B $C done"
  done
  fam=gemm; [ $B -le 32 ] && fam=gemv
  python3 tools/pmc_gemv.py "$(one_csv /tmp/pmc_${B}_FETCH_SIZE)" "$(one_csv /tmp/pmc_${B}_WRITE_SIZE)" $B $O/${R}_pmc_${fam}_b$B.json > $O/${R}_pmc_${fam}_b$B.txt || exit 1
  tail -1 $O/${R}_pmc_${fam}_b$B.txt
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf "/tmp/pmc_att_${C}"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "/tmp/pmc_att_${C}" -o p -- python3 tools/time_attention.py 256 128 > $O/${R}_pmc_att_$C.log 2>&1 || { echo "pmc attention $C failed"; tail -5 $O/${R}_pmc_att_$C.log; exit 1; }
done
python3 tools/pmc_gemv.py "$(one_csv /tmp/pmc_att_FETCH_SIZE)" "$(one_csv /tmp/pmc_att_WRITE_SIZE)" 256 $O/${R}_pmc_attention_b256.json k_attention 128 > $O/${R}_pmc_attention_b256.txt || exit 1
tail -1 $O/${R}_pmc_attention_b256.txt
# kernel traces of the graph path
rm -rf /tmp/kt_rl
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_rl -o rl -- python3 bench.py --roofline-only > $O/${R}_roofline_only_run.json 2> $O/${R}_roofline_only.err || { echo "roofline-only trace failed"; exit 1; }
cp "$(find /tmp/kt_rl -name '*kernel_stats.csv' | head -1)" $O/${R}_roofline_only_kernel_stats.csv
python3 tools/roofline_check.py "$(find /tmp/kt_rl -name '*kernel_trace.csv' | head -1)" $O/${R}_roofline_only_run.json > $O/${R}_roofline_check.txt 2>&1; tail -12 $O/${R}_roofline_check.txt
rm -rf /tmp/kt_f
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-extras > $O/${R}_fused_3x256_rocprof_run.json 2> $O/${R}_fused_rocprof.err || { echo "fused trace failed"; exit 1; }
cp "$(find /tmp/kt_f -name '*kernel_stats.csv' | head -1)" $O/${R}_fused_3x256_kernel_stats.csv
check_budget "$(find /tmp/kt_f -name '*kernel_trace.csv' | head -1)"
for B in 256 16; do
  rm -rf /tmp/kt_b
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b -o b -- python3 bench.py --llm-only --batch $B --sessions 1 --steps 1 --warmup 1 --no-extras > $O/${R}_llm_b${B}_solo_run.json 2> $O/${R}_llm_b${B}_solo.err || { echo "solo trace $B failed"; exit 1; }
  cp "$(find /tmp/kt_b -name '*kernel_stats.csv' | head -1)" $O/${R}_llm_b${B}_solo_kernel_stats.csv
done
# one runner through the reference entry points (tools/time_b1.py), the norm / SwiGLU producers inside the mat-vec launches (shipped) and as launches of their own
: > $O/${R}_b1_runner.txt
for NF in 0 1; do
  name=fused; [ $NF = 1 ] && name=unfused
  rm -rf /tmp/kt_b1
  TK_MI355X_NO_FUSE=$NF timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b1 -o b -- python3 tools/time_b1.py > $O/${R}_b1_${name}.log 2>&1 || { echo "b1 trace $name failed"; exit 1; }
  cp "$(find /tmp/kt_b1 -name '*kernel_stats.csv' | head -1)" $O/${R}_b1_runner_${name}_kernel_stats.csv
done
echo "traces done"
unset ROC_AQL_QUEUE_SIZE
for NF in 0 1 0 1; do TK_MI355X_NO_FUSE=$NF python3 tools/time_b1.py >> $O/${R}_b1_runner.txt 2>&1; done; cat $O/${R}_b1_runner.txt
timeout -k 10 900 python3 bench.py > $O/${R}_bench_default.json 2> $O/${R}_bench_default.err; echo "bench rc=$?"; tail -c 1500 $O/${R}_bench_default.json
