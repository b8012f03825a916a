#!/bin/bash
# Developer tool: hardware-counter comparison of several builds of librtc_hip.so on one workload.
# usage (on the GPU box, from the repo root): bash scripts/pmc_ab.sh <tag> <workload> lib1.so[:NAME=value] lib2.so ...
# Three rocprofv3 --pmc passes per build (one pass per run, the program directly after `--`), condensed by
# scripts/pmc_ab_line.py into one line per build: kernel cycles, VALU / SALU / SMEM / VMEM wave-instructions, active
# lanes per VALU instruction, VALU busy, share of f64 add/mul/fma among the VALU instructions, dependency-wait share.
set -u
TAG=$1; WL=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}
  name=$(basename $lib .so)
  rest=${spec#"$lib"}
  [ -n "$rest" ] && name="$name$(echo $rest | tr ':=' '__')"
  export RTC_HIP_LIB=$R/$lib
  # environment settings carried by the spec (lib.so:NAME=value:NAME=value)
  IFS=':' read -ra kv <<< "${rest#:}"
  for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
  d=$OUT/$name
  mkdir -p $d
  pass() {
    p=$1; shift
    timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $d -o $p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stream-leg --workload $WL > $d/$p.log 2>&1 || { echo "pass $p of $name failed"; tail -5 $d/$p.log; exit 1; }
  }
  pass A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
  pass B SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES
  pass C SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_LDS
  python3 $R/scripts/pmc_ab_line.py $d $name | tee -a $OUT/summary.txt
  for e in "${kv[@]}"; do [ -n "$e" ] && unset "${e%%=*}"; done
done
