cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_llm_gpu.py tests/test_fused_gpu.py -m gpu -x -q > gpurun_out/r02_pytest5.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/r02_pytest5.log
if [ $rc -ne 0 ]; then exit 1; fi
for r in 256 128 64; do echo "== rows $r"; python tools/time_gemv.py --rows $r; done > gpurun_out/r02_gemm_stagger.txt 2>&1
cat gpurun_out/r02_gemm_stagger.txt
