cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TK_MI355X_NO_GRAPH=1
for B in 256 16; do
rm -rf /tmp/prof_$B
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$B -o r -- python3 bench.py --llm-only --no-extras --batch $B --sessions 1 --steps 1 --warmup 1 > gpurun_out/r02_prof_b$B.json 2> gpurun_out/r02_prof_b$B.err || exit 1
f=$(find /tmp/prof_$B -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r02_llm_b${B}_solo_kernel_stats.csv
done
echo done
