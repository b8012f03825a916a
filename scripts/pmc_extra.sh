#!/bin/bash
# Developer tool: latency-side counters of the frame kernel (what the waves wait for), one --pmc pass per run.
# usage (on the GPU box, from the repo root): bash scripts/pmc_extra.sh <tag> <workload> <lib.so>
set -u
TAG=$1; WL=$2; LIB=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RTC_HIP_LIB=$R/$LIB
pass() {
  p=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $OUT -o $p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --workload $WL > $OUT/$p.log 2>&1 || { echo "pass $p failed"; tail -5 $OUT/$p.log; exit 1; }
}
pass L SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT
pass M SQ_INSTS_VALU_CVT SQ_INSTS_VALU_IOPS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH_LEVEL SQ_INSTS_SMEM
pass N SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_LEVEL_WAVES SQ_CYCLES
python3 $R/scripts/pmc_summary.py $OUT "render_kernel<" | tee $OUT/summary.txt
