#!/bin/bash
# Developer aid: run a command; if it is still running after $1 seconds, dump every thread's backtrace with rocgdb (host frames) into $2 and kill it.
#   tools/stall_backtrace.sh 25 gpurun_out/bt.txt python tools/time_batched_cortex.py 64 128 --progress
W=$1; OUT=$2; shift 2
"$@" &
PID=$!
for i in $(seq 1 "$W"); do
  sleep 1
  kill -0 $PID 2>/dev/null || { wait $PID; echo "finished on its own (exit $?) after ${i}s"; exit 0; }
done
echo "still running after ${W}s: dumping backtraces of $PID into $OUT"
GDB=$(command -v rocgdb || command -v gdb || echo /opt/rocm/bin/rocgdb)
$GDB -p $PID -batch -ex "set pagination off" -ex "thread apply all bt 18" > "$OUT" 2>&1
kill $PID; sleep 2; kill -9 $PID 2>/dev/null
grep -c "^Thread" "$OUT"
