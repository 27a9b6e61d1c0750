cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2 0 1; do
  export TK_EXP_MSPLIT=$v
  timeout -k 10 200 python bench.py --roofline-only --batch 256 > gpurun_out/r02_ms_$v.json 2> gpurun_out/r02_ms_$v.err || exit 1
  python - <<P
import json
d=json.loads(open("gpurun_out/r02_ms_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("msplit=$v", "256:", r["avg_launch_ms"], r["frac"], {k:v2["ms"] for k,v2 in r["per_shape"].items()})
P
done
export TK_EXP_MSPLIT=1
timeout -k 10 300 python -X faulthandler -m pytest tests/test_llm_gpu.py -m gpu -x -q -k "batched or width_invariance" -o faulthandler_timeout=120 > gpurun_out/r02_pytest_ms1.log 2>&1; echo "pytest ms1 rc=$?"; tail -3 gpurun_out/r02_pytest_ms1.log
export TK_EXP_MSPLIT=2
timeout -k 10 300 python -X faulthandler -m pytest tests/test_llm_gpu.py -m gpu -x -q -k "batched or width_invariance" -o faulthandler_timeout=120 > gpurun_out/r02_pytest_ms2.log 2>&1; echo "pytest ms2 rc=$?"; tail -3 gpurun_out/r02_pytest_ms2.log
