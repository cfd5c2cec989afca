#!/bin/bash
# One rocprofv3 counter pass over a micro-benchmark (GPU box only; counters with --kernel-trace only, never with other trace domains).
#   tools/pmc_micro.sh <tag> "<COUNTER ...>" <kernel-name substring> <python script> [args...]   ->  gpurun_out/<tag>.csv
set -eo pipefail
TAG=$1; GROUP=$2; SUB=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
D=/tmp/pmcm_$TAG
rm -rf "$D"
cd /tmp
rocprofv3 --kernel-trace --pmc $GROUP -d "$D" -o run -- python3 "$ROOT/$1" "${@:2}" > "$OUT/${TAG}.stdout" 2> "$OUT/${TAG}.stderr"
DB=$(find "$D" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_summary.py" counters "$DB" "$OUT/${TAG}.csv" "$SUB"
