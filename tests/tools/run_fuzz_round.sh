#!/bin/bash
# The randomised differential runs against the CPU oracle, one after the other, each under its
# own time limit; logs under gpurun_out/fuzz_<tag>/.  usage: tests/tools/run_fuzz_round.sh <tag> <seed> [seconds each]
set -e
TAG=${1:-r} ; SEED=${2:-11} ; SEC=${3:-150}
OUT=gpurun_out/fuzz_$TAG ; mkdir -p $OUT
LIM=$((SEC + 90))
timeout -k 10 $LIM python -u tests/tools/fuzz_pnp.py $SEC $SEED > $OUT/pnp.log 2>&1 && tail -1 $OUT/pnp.log
timeout -k 10 $LIM python -u tests/tools/fuzz_counts.py $SEC $SEED > $OUT/counts.log 2>&1 && tail -1 $OUT/counts.log
timeout -k 10 $LIM python -u tests/tools/fuzz_matcher.py $SEC $SEED > $OUT/matcher.log 2>&1 && tail -1 $OUT/matcher.log
POSEHIP_KNN_PATH=matrix timeout -k 10 $LIM python -u tests/tools/fuzz_matcher.py $SEC $SEED > $OUT/matcher_matrix.log 2>&1 && tail -1 $OUT/matcher_matrix.log
timeout -k 10 $LIM python -u tests/tools/fuzz_matcher_big.py $SEC $SEED > $OUT/matcher_big.log 2>&1 && tail -1 $OUT/matcher_big.log
timeout -k 10 $LIM python -u tests/tools/fuzz_matcher_dev.py $SEC $((SEED + 1)) > $OUT/matcher_dev.log 2>&1 && tail -1 $OUT/matcher_dev.log
echo all fuzz runs ok
