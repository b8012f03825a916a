#!/bin/bash
# Developer tool: per-pixel stores against block stores of the canvas (rtc_scene_set_canvas_store) on one GPU:
# kernel time and memory-side write requests (TCC_EA0_WRREQ) of bench.py --workload <w>.
# usage (on the GPU box, from the repo root): bash scripts/store_modes.sh <workload>
set -u
W=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/store_modes_$W
mkdir -p $OUT
for mode in pixels blocks; do
  timeout -k 10 200 python3 $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --canvas-store $mode > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err || { tail -5 $OUT/bench_$mode.err; exit 1; }
  python3 -c "import json,sys; j=json.load(open('$OUT/bench_$mode.json')); print('$mode', 'frame_ms', round(j['ms_per_step'],4), 'kernel_ms', round(j['roofline']['kernel_ms'],4))"
  ( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 180 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT -o wr_$mode -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $W --canvas-store $mode > $OUT/wr_$mode.log 2>&1 ) || { echo "pmc pass failed"; tail -3 $OUT/wr_$mode.log; exit 1; }
  python3 - <<PY
import csv, glob
res = {}
for f in glob.glob("$OUT/**/wr_${mode}_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "render_kernel<" in row["Kernel_Name"] and "weighted" not in row["Kernel_Name"]:
            res.setdefault(row["Counter_Name"], {})[int(row["Dispatch_Id"])] = float(row["Counter_Value"])
c = {k: v[max(v)] for k, v in res.items()}
wr, wr64 = c.get("TCC_EA0_WRREQ_sum", 0), c.get("TCC_EA0_WRREQ_64B_sum", 0)
print("$mode write requests %.3f M (64-byte: %.3f M) = %.1f MB" % (wr / 1e6, wr64 / 1e6, (wr64 * 64 + (wr - wr64) * 32) / 1e6))
PY
done
