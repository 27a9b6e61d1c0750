#!/bin/bash
# Rehearsal of every N > 1 bench.py mode on a ONE-GPU box under the driver's launch line (TK_BENCH_SHARE_GPU=1: every rank drives device 0,
# gloo carries the barrier and the hand-overs).  Launch contract, handle exchange, in-library stage hand-off across processes (hipIpc),
# the per-step perception hand-over.  NOT scaling measurements: every line is marked `rehearsal`.
#   gpurun -- 'bash tools/rehearse_multirank.sh > gpurun_out/r04_multirank_rehearsal.txt 2>&1'
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TK_BENCH_SHARE_GPU=1
P=29700
run() { # name, ranks, bench args...
  local name=$1 n=$2; shift 2
  P=$((P + 1))
  timeout -k 10 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $P bench.py --gpus $n "$@" > /tmp/rh.out 2> /tmp/rh.err
  local rc=$?
  echo "== $name (rc=$rc): bench.py --gpus $n $*"
  grep '^{' /tmp/rh.out | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print({k: d.get(k) for k in ('value', 'n_gpus', 'ms_per_step', 'scaling', 'rehearsal', 'perception_handover')}, d['config'].get('parallelism'))"
  [ $rc = 0 ] || tail -5 /tmp/rh.err
}
run "replicas x2" 2 --steps 1 --warmup 1 --batch 64 --sessions 1 --no-extras
run "pipeline x2, Q4_K_M, two generations on the same pipes" 2 --pipeline --steps 2 --warmup 1 --batch 64 --sessions 2
run "pipeline x3, f16 payload" 3 --pipeline --pipe-f16 --steps 1 --warmup 1 --batch 32 --sessions 3
run "combined x4 (2 stages + detector + ASR), perception hand-over every step" 4 --placement combined --steps 2 --warmup 1 --batch 32 --sessions 2
run "model-per-gpu x3, perception hand-over every step" 3 --placement model-per-gpu --steps 2 --warmup 1 --batch 32 --sessions 1 --no-extras
# the RCCL transport refuses below two devices (it never falls back to the mailboxes): rc must be non-zero here, with the reason on stderr
run "pipeline x2 over RCCL on ONE device: must refuse" 2 --pipeline --pipe-rccl --steps 1 --warmup 0 --batch 32 --sessions 1
grep -h "one GPU per stage" /tmp/rh.err | head -1
