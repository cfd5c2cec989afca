set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-split-math --no-roofline --no-other-configs --steps 8 --warmup 4 > $GRAFT_REPO_ROOT/gpurun_out/tl.json 2> $GRAFT_REPO_ROOT/gpurun_out/tl.err
F=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $F 2 20 > $GRAFT_REPO_ROOT/gpurun_out/r04_timeline_gaps.txt
