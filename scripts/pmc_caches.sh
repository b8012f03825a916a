#!/bin/bash
# Developer tool: hit rates of the instruction cache and of the scalar data cache for the frame kernel, one --pmc pass
# per run (the SQC_ counters are counters of the SQ block: at most eight per pass).
# usage (on the GPU box, from the repo root): bash scripts/pmc_caches.sh <tag> <workload> <lib.so>
set -u
TAG=$1; WL=$2; LIB=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RTC_HIP_LIB=$R/$LIB
pass() {
  p=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --workload $WL > $OUT/$p.log 2>&1 || { echo "pass $p failed"; tail -5 $OUT/$p.log; exit 1; }
}
pass I SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES
pass D SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQC_TC_STALL
python3 $R/scripts/pmc_summary.py $OUT "render_kernel<" | tee $OUT/summary.txt
