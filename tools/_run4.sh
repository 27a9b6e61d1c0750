cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest4.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r02_pytest4.log
if [ $rc -ne 0 ]; then exit 1; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_d.json 2> gpurun_out/r02_bench_d.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r02_bench_d.err
export TK_MI355X_NO_GRAPH=1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b16 -o b16 --output-format csv -- python3 bench.py --batch 16 --sessions 1 --llm-only --steps 2 --warmup 1 --no-extras > gpurun_out/r02_prof_b16.log 2>&1; echo "prof b16 rc=$?"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b256 -o b256 --output-format csv -- python3 bench.py --batch 256 --sessions 1 --llm-only --steps 1 --warmup 1 --no-extras > gpurun_out/r02_prof_b256.log 2>&1; echo "prof b256 rc=$?"
