#!/bin/bash
# Kernel-trace statistics and hardware counters of bench.py for a list of workloads, condensed into
# gpurun_out/profiles_<tag>/r3_<workload>_{kernel_stats,pmc}_<tag>.txt (copy what is to be kept into profiles/).
# usage (on the GPU box, from the repo root): bash scripts/profile_all.sh <tag> c2 c3 c4 c5
set -u
TAG=$1; shift
PREFIX=${RTC_PROFILE_PREFIX:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
DST=$R/gpurun_out/profiles_$TAG
mkdir -p $DST
for W in "$@"; do
  STEPS=20; [ "$W" = c4 ] && STEPS=3
  P=$R/gpurun_out/prof_${TAG}_$W
  mkdir -p $P
  ( cd /tmp && export TMPDIR=/tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $P -o $W -- python3 $R/bench.py --workload $W --steps $STEPS --warmup 3 --no-cpu-baseline --no-stream-leg > $P/bench.log 2>&1 )
  rc=$?; echo "kernel-trace $W rc=$rc"; [ $rc -ne 0 ] && exit 1
  DB=$(ls $P/*_results.db $P/*/*_results.db 2>/dev/null | head -1)
  python3 $R/scripts/rocprof_summary.py "$DB" | sed "s#$R/##" > $DST/${PREFIX}_${W}_kernel_stats_$TAG.txt
  grep "\"metric\"" $P/bench.log | tail -1 > $DST/${PREFIX}_${W}_bench_$TAG.json
  bash $R/scripts/pmc.sh ${TAG}_$W --workload $W || exit 1
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$W | sed "s#$R/##" > $DST/${PREFIX}_${W}_pmc_$TAG.txt
  echo "done $W"
done
ls $DST
