#!/bin/bash
# The kernels of ONE training step in start order with queue, start (us from the step's first kernel) and duration, under the real
# multi-stream schedule (rocprofv3 --kernel-trace; the profiler makes the HOST slower, so gaps at the start of the step are its, but a
# kernel's duration IS what it took beside the other streams' kernels): gpurun_out/step_timeline.txt.  DESIGN.md lessons 35-36 read the
# backward's block boundaries off this.      tools/step_timeline.sh   (GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-split-math --no-roofline --no-other-configs --steps 4 --warmup 3 ${BENCH_EXTRA:-} > "$ROOT/gpurun_out/step_timeline.bench.json" 2> "$ROOT/gpurun_out/step_timeline.err"
F=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python3 - "$F" > "$ROOT/gpurun_out/step_timeline.txt" <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seg = rows[idx[-2] + 1: idx[-1] + 1]                      # between the last two optimizer launches: one whole step
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("glowtts::", "").replace("(anonymous namespace)::", "")
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  q{r['Queue_Id']:>2}  {n[:80]}  grid={r['Grid_Size_X']}")
PY
