cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default slpoff default slpoff; do
  if [ $v = default ]; then unset TK_MI355X_LIB; else export TK_MI355X_LIB=$GRAFT_REPO_ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 200 python bench.py --roofline-only --batch 256 > gpurun_out/r02_var_$v.json 2> gpurun_out/r02_var_$v.err || exit 1
  python - <<P
import json
d=json.loads(open("gpurun_out/r02_var_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]; print("$v", "256:", r["avg_launch_ms"], r["frac"], [ (k,v2) for k,v2 in r.get("per_shape",{}).items()] if isinstance(r.get("per_shape"),dict) else r.get("per_shape"))
r=d["roofline_16_rows"]; print("$v", "16:", r["avg_launch_ms"], r["frac"], r.get("per_shape"))
print(d["roofline_attention"])
P
done
