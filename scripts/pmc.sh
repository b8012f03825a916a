#!/bin/bash
# Collect hardware counters of the render kernel, one rocprofv3 --pmc pass per run.
# usage (on the GPU box, from the repo root): bash scripts/pmc.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $name -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg $BENCH_ARGS > $OUT/$name.log 2>&1
  rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
BENCH_ARGS="$*"
pass A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE || exit 1
pass B SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES || exit 1
pass C SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 || exit 1
pass D TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum || exit 1
pass E TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum || exit 1
ls $OUT
