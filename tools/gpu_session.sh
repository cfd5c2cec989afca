#!/bin/bash
# One GPU-box session: a list of steps, each under its own timeout, output under gpurun_out/<tag>_*.  A step that TIMES OUT (or is
# killed) ends the session — nothing further is started on a GPU that may be wedged; a step that merely fails (a test assertion)
# does not.   tools/gpu_session.sh <tag> "<name>|<seconds>|<command>" ...
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
for step in "$@"; do
    name=${step%%|*}; rest=${step#*|}; secs=${rest%%|*}; cmd=${rest#*|}
    echo "[session $TAG] $name: $cmd" | tee -a "$OUT/${TAG}_session.log"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/${TAG}_${name}.out" 2> "$OUT/${TAG}_${name}.err"
    rc=$?
    echo "[session $TAG] $name: exit $rc after $(( $(date +%s) - start )) s" | tee -a "$OUT/${TAG}_session.log"
    tail -n 3 "$OUT/${TAG}_${name}.out"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "[session $TAG] $name timed out: stopping the session" | tee -a "$OUT/${TAG}_session.log"
        exit $rc
    fi
done
exit 0
