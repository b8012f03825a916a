#!/bin/bash
# Developer tool: the numbers quoted in the design notes, in one GPU call.
# usage (on the GPU box, repo root): bash scripts/final_numbers.sh <tag>
set -u
O=gpurun_out/final_$1; mkdir -p $O
run() { name=$1; shift; timeout -k 10 280 python bench.py "$@" 2>$O/$name.err | grep '"metric"' > $O/$name.json || { echo "FAILED $name"; tail -5 $O/$name.err; exit 1; }; echo "done $name"; }
run c3_default
run c4_accel --workload c4 --accel --no-cpu-baseline --steps 5
for d in 0 1 2 3; do run c3_depth$d --depth $d --no-cpu-baseline; done
run c2 --workload c2
run c5 --workload c5 --no-cpu-baseline
run c4 --workload c4 --steps 3
run c3_multi4 --no-cpu-baseline --multi-ranks 4
for w in c3 c5 c2; do timeout -k 10 120 python scripts/partition_timing.py $w > $O/partition_$w.txt 2>&1 || exit 1; done
timeout -k 10 120 python scripts/first_frame_timing.py > $O/first_frame.txt 2>&1
timeout -k 10 120 python scripts/host_path_timing.py > $O/host_path.txt 2>&1
echo finished
