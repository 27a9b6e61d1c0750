cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $VARIANTS; do
  export TK_MI355X_LIB=$GRAFT_REPO_ROOT/build/variants/lib_$v.so
  timeout -k 10 200 python bench.py --roofline-only --batch 256 > gpurun_out/r02_v_$v.json 2> gpurun_out/r02_v_$v.err || exit 1
  python - <<P
import json
d=json.loads(open("gpurun_out/r02_v_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$v", "256:", r["avg_launch_ms"], r["frac"], {k:v2["ms"] for k,v2 in r["per_shape"].items()})
P
done
