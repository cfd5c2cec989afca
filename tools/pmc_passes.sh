#!/bin/bash
# Counter passes over the whole training step (GPU box only).  One counter group per run, kernel trace only — never
# combined with sys/hip/hsa traces (gpurun refuses that) — one stream so that every dispatch is counted in isolation.
#   tools/pmc_passes.sh <tag>      ->  gpurun_out/<tag>_pmc_<group>.csv  (+ the kernel-trace stats of a plain pass)
set -eo pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp GLOWTTS_SIDE_STREAM=0
cd /tmp
ARGS="--no-cpu-baseline --no-split-math --no-roofline --no-other-configs --steps 2 --warmup 1 ${BENCH_EXTRA:-}"
for GROUP in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS"; do
    NAME=$(echo "$GROUP" | cut -d' ' -f1)
    D=/tmp/pmc_${TAG}_${NAME}
    rm -rf "$D"
    echo "[pmc] pass $GROUP" >&2
    rocprofv3 --kernel-trace --pmc $GROUP -d "$D" -o run -- python3 "$ROOT/bench.py" $ARGS > "$OUT/${TAG}_pmc_${NAME}.bench.json" 2> "$OUT/${TAG}_pmc_${NAME}.err"
    DB=$(find "$D" -name '*.db' | head -1)
    python3 "$ROOT/tools/rocpd_summary.py" counters "$DB" "$OUT/${TAG}_pmc_${NAME}.csv"
done
D=/tmp/pmc_${TAG}_trace
rm -rf "$D"
echo "[pmc] kernel-trace pass" >&2
rocprofv3 --kernel-trace --stats -d "$D" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-split-math --no-roofline --no-other-configs --steps 10 --warmup 3 ${BENCH_EXTRA:-} > "$OUT/${TAG}_trace.bench.json" 2> "$OUT/${TAG}_trace.err"
DB=$(find "$D" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_summary.py" kernels "$DB" "$OUT/${TAG}_kernel_stats.csv"
# a second, shorter kernel-trace pass: tools/launch_census.py takes the difference of the two (launches per step)
D=/tmp/pmc_${TAG}_trace5
rm -rf "$D"
echo "[pmc] kernel-trace pass (5 steps)" >&2
rocprofv3 --kernel-trace --stats -d "$D" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-split-math --no-roofline --no-other-configs --steps 5 --warmup 3 ${BENCH_EXTRA:-} > "$OUT/${TAG}_trace5.bench.json" 2> "$OUT/${TAG}_trace5.err"
DB=$(find "$D" -name '*.db' | head -1)
python3 "$ROOT/tools/rocpd_summary.py" kernels "$DB" "$OUT/${TAG}_kernel_stats_5steps.csv"
python3 "$ROOT/tools/launch_census.py" "$OUT/${TAG}_kernel_stats_5steps.csv" 5 "$OUT/${TAG}_kernel_stats.csv" 10 > "$OUT/${TAG}_launch_census.txt"
python3 "$ROOT/tools/pmc_combine.py" "$OUT/${TAG}" "$OUT/${TAG}_pmc.json"
