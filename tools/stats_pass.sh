#!/bin/bash
# One kernel-trace pass over the training step on ONE stream (every dispatch timed in isolation): gpurun_out/<tag>_kernel_stats.csv
#   tools/stats_pass.sh <tag> [steps]
set -eo pipefail
TAG=${1:-tmp}
STEPS=${2:-10}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp GLOWTTS_SIDE_STREAM=0
cd /tmp
D=/tmp/stats_${TAG}
rm -rf "$D"
rocprofv3 --kernel-trace --stats -d "$D" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-split-math --no-roofline --no-other-configs --steps $STEPS --warmup 3 > "$ROOT/gpurun_out/${TAG}_trace.bench.json" 2> "$ROOT/gpurun_out/${TAG}_trace.err"
DB=$(find "$D" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_summary.py" kernels "$DB" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
