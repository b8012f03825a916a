#!/bin/bash
# Developer tool: quick parity gate + interleaved A/B timing of library builds on the GPU box.
# usage: bash scripts/ab_gpu.sh <tag> <workloads> lib1.so lib2.so ...   (the LAST lib is the candidate the parity gate runs on)
set -u
TAG=$1; WL=$2; shift 2
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
CAND="${@: -1}"
RTC_HIP_LIB=$PWD/$CAND timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread -k "not 7680 and not full_size and not cpp_" > $OUT/parity.log 2>&1 || { tail -20 $OUT/parity.log; exit 1; }
tail -1 $OUT/parity.log
RTC_HIP_LIB=$PWD/$CAND timeout -k 10 100 python scripts/fuzz_parity.py 25 > $OUT/fuzz.log 2>&1 || { tail -5 $OUT/fuzz.log; exit 1; }
tail -2 $OUT/fuzz.log
timeout -k 10 400 python scripts/ab.py --rounds 3 --iters 6 --workloads $WL "$@" 2>&1 | tee $OUT/ab.log
