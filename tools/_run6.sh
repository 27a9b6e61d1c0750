cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -X faulthandler -m pytest tests -m gpu -x -q -o faulthandler_timeout=150 > gpurun_out/r02_pytest8.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 gpurun_out/r02_pytest8.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python bench.py --weights f16 --batch 16 --sessions 1 --steps 2 --warmup 1 > gpurun_out/r02_bench_f16_b16.json 2> gpurun_out/r02_bench_f16_b16.err; echo "f16 b16 rc=$?"; tail -c 900 gpurun_out/r02_bench_f16_b16.json
timeout -k 10 300 python bench.py --weights f16 --batch 256 --sessions 1 --steps 1 --warmup 1 > gpurun_out/r02_bench_f16_b256.json 2> gpurun_out/r02_bench_f16_b256.err; echo "f16 b256 rc=$?"; tail -c 500 gpurun_out/r02_bench_f16_b256.json
